// Backward correlation  C[dz,dx,dy] = sum_{b,z,x,y} delta[b,z,x,y] * x[b, z+dz-pz, x+dx-px, y+dy-py]   on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32).  (reference: the conv3d weight gradient autograd computes for
// SceneNet.forward, core/models/SCENE_Net.py:322-339 -- by linearity one kernel-shaped tensor, see backward.hip.)
//
// For one input plane z' and one input row r the sum over y is a small GEMM:
//     D_dz[dy][dx] += sum_y  A[dy][y] * Bm[y][dx],   A[dy][y] = x[z'][r][y + dy - py]        (Toeplitz of the row)
//                                                    Bm[y][dx] = delta[z' - dz + pz][r - dx + px][y]
// M = dy (ky <= 16), N = dx (kx <= 16), K = y in steps of 4; the accumulator tile D_dz stays in registers over every
// row, plane, tile and batch element a workgroup visits (one tile per dz).  Each MFMA takes one LDS dword per lane
// for A (19 consecutive floats, broadcast) and one for Bm (16 rows x 4 columns; the delta row stride is = 4 mod 64
// floats, so the 64 addresses fall into 64 different banks).
//
// Workgroup job = (b, delta plane z, tile of TXR input rows): delta plane (+ kx-1 halo rows, with the relu(tanh)
// derivative fused) staged once, then the kz input planes z+dz-pz streamed through LDS.  Staging issues all of a
// thread's global loads (4 elements each) before the first LDS store, so a plane costs one memory latency, not one
// per element.  Binary occupancy is sparse: while a plane is staged every non-zero marks the K steps whose Toeplitz
// window contains it, and a wave only issues the MFMAs of marked steps.  Persistent workgroups; per-workgroup partial sums are reduced in a fixed
// order by corr_reduce_kernel (bit-reproducible).
//
// Bound: MFMA (fp32).  Algorithmic flops 2*V*kz*kx*ky per tile; the 16x16 tiles execute 2*V*kz*16*16.
#include "common.h"

namespace sn {
int corr_mfma_supported(int kz, int kx, int ky);
int corr_mfma_rows(int B, int Z, int X, int Y, int kz, int kx, int ky);
int corr_mfma_launch(const void* x, int x_dtype, const void* gout, const void* out, int g_dtype, int B, int Z, int X,
                     int Y, int kz, int kx, int ky, float* partial_ws, float* C, hipStream_t s);
}  // namespace sn

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct CorrShape {
    int B, Z, X, Y, kz, kx, ky;
    int pz, px, py;
    int TXR, nxt, njobs;  // input rows per job, row tiles per plane, jobs
    int YK;               // K extent: Y rounded up to 32 (zero filled)
    int DR, DS;           // delta tile: rows, row stride (floats)
    int XS;               // input tile row stride (floats)
    int vec;              // Y % 4 == 0 and 16-byte (byte input: 4-byte) aligned pointers: 4-element loads
};

template <typename T>
__device__ __forceinline__ float to_f32(T v) { return (float)v; }

// four consecutive elements as floats
__device__ __forceinline__ void load_quad(const uint8_t* p, float (&v)[4]) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    v[0] = (float)(u & 255u); v[1] = (float)((u >> 8) & 255u); v[2] = (float)((u >> 16) & 255u); v[3] = (float)(u >> 24);
}
__device__ __forceinline__ void load_quad(const float* p, float (&v)[4]) {
    const float4 f = *reinterpret_cast<const float4*>(p);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}
__device__ __forceinline__ void load_quad(const __bf16* p, float (&v)[4]) {   // bf16 storage: 8 bytes, widened to fp32
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ void load_quad(const double* p, float (&v)[4]) {
    const double2 a = reinterpret_cast<const double2*>(p)[0], b = reinterpret_cast<const double2*>(p)[1];
    v[0] = (float)a.x; v[1] = (float)a.y; v[2] = (float)b.x; v[3] = (float)b.y;
}

template <typename XT, int KZMAX, typename DT>
__global__ __launch_bounds__(kThreads, KZMAX <= 9 ? 6 : 4) void corr_mfma_kernel(const XT* __restrict__ x, const DT* __restrict__ gout,
                                                             const DT* __restrict__ out, CorrShape s,
                                                             float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dl = lds;                                 // [DR][DS]   delta rows q0 .. q0+DR-1
    float* xl = dl + s.DR * s.DS;                    // [TXR][XS]  input rows x0 .. x0+TXR-1, columns y - py
    unsigned* flags = reinterpret_cast<unsigned*>(xl + s.TXR * s.XS + 16);  // [2][TXR] K steps (of 4 y) to run
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, lk = lane >> 4;
    const size_t plane = (size_t)s.X * s.Y;

    f32x4 acc[KZMAX];
#pragma unroll
    for (int i = 0; i < KZMAX; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int i = tid; i < 2 * s.TXR; i += kThreads) flags[i] = 0u;
    const int nsteps = s.YK >> 2;   // <= 32: tracked one by one, else every step of a non-empty row runs
    const int Y4 = s.Y >> 2;
    const bool xl_vec4 = (s.py & 3) == 0 && ((s.DR * s.DS) & 3) == 0;   // 16-byte aligned input-tile stores
    if (s.vec)                      // the vector staging only writes columns that hold data: zero the padding once
        for (int i = tid; i < s.DR * s.DS + s.TXR * s.XS; i += kThreads) lds[i] = 0.f;
    // K steps whose window [4 ks, 4 ks + ky + 2] (LDS columns) contains column c
    auto steps_of = [&](int c) -> unsigned {
        if (nsteps > 32) return 0xffffffffu;
        int lo_s = (c - s.ky - 2 + 3) >> 2, hi_s = c >> 2;
        lo_s = lo_s < 0 ? 0 : lo_s;
        hi_s = hi_s > nsteps - 1 ? nsteps - 1 : hi_s;
        return (hi_s >= lo_s) ? ((0xffffffffu >> (31 - (hi_s - lo_s))) << lo_s) : 0u;
    };
    const int ncol = lm < s.kx ? lm : s.kx - 1;      // columns dx >= kx are discarded; keep their reads in range
    int parity = 0;

    for (int job = blockIdx.x; job < s.njobs; job += gridDim.x) {
        // z-major job order: the planes that hold most of the set voxels (ground returns) are a few z; with b fastest
        // a workgroup's jobs (job, job + grid, ...) land on different z instead of the same heavy one
        // ([measured] C2, LiDAR-shaped occupancy: 204 -> 103 us)
        int j = job;
        const int xt = j % s.nxt; j /= s.nxt;
        const int b = j % s.B;
        const int z = j / s.B;
        const int x0 = xt * s.TXR;
        const int q0 = x0 - (s.kx - 1 - s.px);
        __syncthreads();  // previous job's readers of dl / xl are done
        // ---- delta tile (with d relu(tanh(s))/ds when the forward output is given)
        const size_t dbase = ((size_t)b * s.Z + z) * plane;
        if (s.vec) {
            const int nunits = s.DR * Y4;
            for (int base = 0; base < nunits; base += 3 * kThreads) {
                float g[3][4], o[3][4];
                int at[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {   // all loads first
                    const int unit = base + u * kThreads + tid;
                    const int rr = unit / Y4, c4 = unit - rr * Y4, q = q0 + rr;
                    at[u] = (unit < nunits) ? rr * s.DS + 4 * c4 : -1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[u][j] = 0.f, o[u][j] = 1.f;
                    if (unit < nunits && q >= 0 && q < s.X) {
                        const size_t idx = dbase + (size_t)q * s.Y + 4 * c4;
                        load_quad(gout + idx, g[u]);
                        if (out) load_quad(out + idx, o[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (at[u] < 0) continue;
                    float dv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float d = g[u][j];
                        if (out) d = (o[u][j] > 0.f) ? d * (1.f - o[u][j] * o[u][j]) : 0.f;
                        dv[j] = d;
                    }
                    // one 16-byte store per lane (DS % 4 == 0): four ds_write_b32 at stride 4 were 4-way conflicted
                    *reinterpret_cast<float4*>(dl + at[u]) = make_float4(dv[0], dv[1], dv[2], dv[3]);
                }
            }
        } else {
            for (int i = tid; i < s.DR * s.DS; i += kThreads) {
                const int c = i % s.DS, rr = i / s.DS, q = q0 + rr;
                float d = 0.f;
                if (c < s.Y && q >= 0 && q < s.X) {
                    const size_t idx = dbase + (size_t)q * s.Y + c;
                    d = (float)gout[idx];
                    if (out) {
                        const float o = (float)out[idx];
                        d = (o > 0.f) ? d * (1.f - o * o) : 0.f;
                    }
                }
                dl[i] = d;
            }
        }
#pragma unroll
        for (int dz = 0; dz < KZMAX; ++dz) {
            const int zp = z + dz - s.pz;
            if (dz >= s.kz || zp < 0 || zp >= s.Z) continue;  // block-uniform
            unsigned* fl = flags + parity * s.TXR;
            unsigned* fl_next = flags + (parity ^ 1) * s.TXR;
            // ---- input plane rows -> LDS (zero halo in y), K-step marks per row
            const size_t xbase = ((size_t)b * s.Z + zp) * plane;
            if (s.vec) {
                const int nunits = s.TXR * Y4;
                for (int base = 0; base < nunits; base += 2 * kThreads) {
                    float v[2][4];
                    int at[2], row[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {   // all loads first
                        const int unit = base + u * kThreads + tid;
                        const int rr = unit / Y4, c4 = unit - rr * Y4, r = x0 + rr;
                        at[u] = (unit < nunits) ? rr * s.XS + s.py + 4 * c4 : -1;
                        row[u] = rr;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[u][j] = 0.f;
                        if (unit < nunits && r < s.X) load_quad(x + xbase + (size_t)r * s.Y + 4 * c4, v[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (at[u] < 0) continue;
                        unsigned bits = 0u;
                        if (xl_vec4) {
                            *reinterpret_cast<float4*>(xl + at[u]) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) xl[at[u] + j] = v[u][j];
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (v[u][j] != 0.f) bits |= steps_of(at[u] - row[u] * s.XS + j);
                        if (bits) atomicOr(&fl[row[u]], bits);
                    }
                }
            } else {
                for (int i = tid; i < s.TXR * s.XS; i += kThreads) {
                    const int c = i % s.XS, rr = i / s.XS, r = x0 + rr, y = c - s.py;
                    float v = 0.f;
                    if (r < s.X && y >= 0 && y < s.Y) v = to_f32(x[xbase + (size_t)r * s.Y + y]);
                    xl[i] = v;
                    if (v != 0.f) atomicOr(&fl[rr], steps_of(c));
                }
            }
            for (int i = tid; i < s.TXR; i += kThreads) fl_next[i] = 0u;
            __syncthreads();
            // ---- rows of this wave
            for (int rr = wave; rr < s.TXR; rr += kWaves) {
                unsigned m = __builtin_amdgcn_readfirstlane(fl[rr]);  // wave-uniform
                if (!m) continue;
                const float* ap = xl + rr * s.XS + lk + lm;
                const float* bp = dl + (rr + s.kx - 1 - ncol) * s.DS + lk;
                if (nsteps <= 32) {
                    while (m) {  // two marked steps per trip: the loads of both are in flight before the MFMAs
                        const int k0 = __builtin_ctz(m) << 2;
                        m &= m - 1;
                        const int k1 = m ? (__builtin_ctz(m) << 2) : -1;
                        if (m) m &= m - 1;
                        const float a0 = ap[k0], b0 = bp[k0];
                        float a1 = 0.f, b1 = 0.f;
                        if (k1 >= 0) a1 = ap[k1], b1 = bp[k1];
                        acc[dz] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[dz], 0, 0, 0);
                        if (k1 >= 0) acc[dz] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[dz], 0, 0, 0);
                    }
                } else {
                    for (int k0 = 0; k0 < s.YK; k0 += 32) {
                        float a[8], bb[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            a[u] = ap[k0 + 4 * u];
                            bb[u] = bp[k0 + 4 * u];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            acc[dz] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bb[u], acc[dz], 0, 0, 0);
                    }
                }
            }
            __syncthreads();  // xl is rewritten by the next plane
            parity ^= 1;
        }
    }

    // ---- waves add their tiles in order into LDS, then the workgroup writes its partial row
    __syncthreads();
    float* red = lds;  // [kz][256]
    for (int w = 0; w < kWaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int dz = 0; dz < KZMAX; ++dz) {
                if (dz < s.kz) {
                    float* r4 = red + dz * 256 + lane * 4;
                    if (w == 0) {
                        r4[0] = acc[dz][0]; r4[1] = acc[dz][1]; r4[2] = acc[dz][2]; r4[3] = acc[dz][3];
                    } else {
                        r4[0] += acc[dz][0]; r4[1] += acc[dz][1]; r4[2] += acc[dz][2]; r4[3] += acc[dz][3];
                    }
                }
            }
        }
        __syncthreads();
    }
    const int ntaps = s.kz * s.kx * s.ky;
    float* prow = partial + (size_t)blockIdx.x * ntaps;
    for (int t = tid; t < ntaps; t += kThreads) {
        const int dy = t % s.ky, dx = (t / s.ky) % s.kx, dz = t / (s.ky * s.kx);
        // D[m = dy][n = dx] lives in lane (dy/4)*16 + dx, register dy%4
        prow[t] = red[dz * 256 + ((dy >> 2) * 16 + dx) * 4 + (dy & 3)];
    }
}

// C[t] = sum_k partial[k][t]: one wave per tap, lanes stride the rows, xor-tree at the end (fixed order)
__global__ __launch_bounds__(256) void corr_rows_reduce_kernel(const float* __restrict__ partial, int nrows, int ntaps,
                                                               float* __restrict__ C) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= ntaps) return;
    float a = 0.f;
    for (int k = lane; k < nrows; k += 64) a += partial[(size_t)k * ntaps + t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) C[t] = a;
}

bool plan(int B, int Z, int X, int Y, int kz, int kx, int ky, CorrShape* s, size_t* lds_bytes, int* grid) {
    if (kz > 16 || kx > 16 || ky > 16) return false;
    s->B = B; s->Z = Z; s->X = X; s->Y = Y; s->kz = kz; s->kx = kx; s->ky = ky;
    s->pz = (kz - 1) / 2; s->px = (kx - 1) / 2; s->py = (ky - 1) / 2;
    s->YK = (Y + 31) / 32 * 32;
    s->DS = s->YK + 4;
    s->XS = s->YK + 16;
    const size_t budget = 150 * 1024;
    int txr = (X + 7) / 8 * 8;
    for (;; txr -= 8) {
        if (txr < 8) return false;
        const size_t need = ((size_t)(txr + kx - 1) * s->DS + (size_t)txr * s->XS + 16 + 2 * txr) * 4;
        const size_t red = (size_t)kz * 256 * 4;
        if ((need > red ? need : red) <= budget) {
            *lds_bytes = need > red ? need : red;
            break;
        }
    }
    s->TXR = txr;
    s->DR = txr + kx - 1;
    s->nxt = (X + txr - 1) / txr;
    s->njobs = B * Z * s->nxt;
    const int per_cu = (int)((160 * 1024) / *lds_bytes);
    const int cap = 256 * (per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu));  // <= 768 rows (sn_conv_corr_blocks)
    *grid = s->njobs < cap ? s->njobs : cap;
    return true;
}

}  // namespace

int sn::corr_mfma_supported(int kz, int kx, int ky) { return kz <= 16 && kx <= 16 && ky <= 16; }

int sn::corr_mfma_rows(int B, int Z, int X, int Y, int kz, int kx, int ky) {
    CorrShape s;
    size_t lds;
    int grid;
    return plan(B, Z, X, Y, kz, kx, ky, &s, &lds, &grid) ? grid : 0;
}

int sn::corr_mfma_launch(const void* x, int x_dtype, const void* gout, const void* out, int g_dtype, int B, int Z, int X,
                         int Y, int kz, int kx, int ky, float* partial_ws, float* C, hipStream_t stream) {
    CorrShape s;
    size_t lds;
    int grid;
    if (!plan(B, Z, X, Y, kz, kx, ky, &s, &lds, &grid))
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_corr: shape outside the MFMA correlation kernel");
    const size_t xa = (x_dtype == SN_U8 || x_dtype == SN_OCC8) ? 4 : 16;
    const size_t ga = g_dtype == SN_BF16 ? 8 : 16;
    s.vec = (Y % 4 == 0) && ((uintptr_t)x % xa == 0) && ((uintptr_t)gout % ga == 0) &&
            (!out || (uintptr_t)out % ga == 0);
#define SN_CORR_LAUNCH(XT, KZMAX)                                                                                 \
    do {                                                                                                          \
        if (g_dtype == SN_BF16) {                                                                                 \
            auto kern = corr_mfma_kernel<XT, KZMAX, __bf16>;                                                      \
            if (sn::ensure_dynamic_lds((const void*)kern, (int)lds) != hipSuccess)                                \
                return sn::check_launch("sn_conv_corr(hipFuncSetAttribute)");                                     \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, (const XT*)x, (const __bf16*)gout,  \
                               (const __bf16*)out, s, partial_ws);                                                \
        } else {                                                                                                  \
            auto kern = corr_mfma_kernel<XT, KZMAX, float>;                                                       \
            if (sn::ensure_dynamic_lds((const void*)kern, (int)lds) != hipSuccess)                                \
                return sn::check_launch("sn_conv_corr(hipFuncSetAttribute)");                                     \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, (const XT*)x, (const float*)gout,   \
                               (const float*)out, s, partial_ws);                                                 \
        }                                                                                                         \
    } while (0)
#define SN_CORR_KZ(XT)                                                                                            \
    do {                                                                                                          \
        if (kz <= 9) SN_CORR_LAUNCH(XT, 9);                                                                       \
        else SN_CORR_LAUNCH(XT, 16);                                                                              \
    } while (0)
    switch (x_dtype) {
        case SN_F32: SN_CORR_KZ(float); break;
        case SN_F64: SN_CORR_KZ(double); break;
        case SN_U8:
        case SN_OCC8: SN_CORR_KZ(uint8_t); break;
        default: return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_corr: x_dtype %d", x_dtype);
    }
#undef SN_CORR_KZ
#undef SN_CORR_LAUNCH
    if (int rc = sn::check_launch("sn_conv_corr(mfma)")) return rc;
    const int ntaps = kz * kx * ky;
    hipLaunchKernelGGL(corr_rows_reduce_kernel, dim3((ntaps + 3) / 4), dim3(256), 0, stream, partial_ws, grid, ntaps, C);
    return sn::check_launch("sn_conv_corr(reduce)");
}
