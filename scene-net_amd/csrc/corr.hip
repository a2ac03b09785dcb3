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
size_t corr_sparse_ws_bytes(int x_dtype, int B, int Z, int X, int Y, int kz, int kx, int ky);
int corr_sparse_launch(const void* x, const void* gout, const void* out, int g_dtype, int B, int Z, int X, int Y, int kz,
                       int kx, int ky, void* ws, float* C, hipStream_t s);
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

// C[t] = sum_k partial[k][t]: one wave per tap, lanes stride the rows, xor-tree at the end (fixed order).  ([measured] 5.6 us
// for 768 x 729; a coalesced form -- 32 taps per workgroup, 16 row chunks per tap, 23 workgroups -- took 14.5 us: too few
// waves to cover the latency of 48 dependent-free but serial loads each.)
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


// ------------------------------------------------------------------------------------------------- K4s: sparse gather
// The same correlation for BINARY OCCUPANCY (SN_OCC8), read the other way round:
//     C[dz,dx,dy] = sum over the SET voxels (b, z', x', y') of  delta[b, z'-dz+pz, x'-dx+px, y'-dy+py]
// A LiDAR tile sets 1-4 % of its voxels ([measured] the synthetic C2 tiles: 9.2 k of 262 k), so the sum has
// nnz x 729 terms where the GEMM above executes V x 9 x 256 products and then skips the K steps that are all zero.
// No matrix core is involved: the work is LDS reads and fp32 adds.
//
// Two launches.  corr_lists_kernel turns every (b, z', tile of TXR input rows; TXR Y <= 2048) into a LIST of its set
// voxels -- 16-bit byte offsets of the voxel's cell in the gather's delta tile, in memory order (a wave scan + wave
// totals, no atomics: the order, hence every fp32 sum, is the same in every run), padded with entries that read zeros.
// corr_gather_kernel's job = (b, delta plane z, tile): the delta rows the tile's taps can reach are staged once, with the
// relu(tanh) derivative fused, into a zero-haloed LDS tile of row pitch P = ky (mod 64) -- the 64 taps a wave reads for
// one voxel fall into 64 different banks -- and the kz lists of the planes z + dz - pz are copied beside it.  Thread
// (group g, tap (dx, dy)) adds delta[cell - (dx P + dy)] over every (groups)-th quad of list dz into its accumulator for
// dz; 512 / (kx ky) groups walk a list side by side.  The next job's global loads (delta, forward output, list chunk) are
// issued before the gather and consumed after it.  Accumulators persist over a persistent workgroup's jobs; groups,
// then workgroups, are summed in a fixed order.
// ([measured] C2, 32 tiles: lists built inside the gather kernel -- every plane compacted by the nine jobs that read it,
// ~2000 instructions per wave and job around a gather of ~400 -- took 102-106 us in two variants, no better than the
// GEMM form's 96; bound by instruction issue, not by LDS or HBM.)
constexpr int kSpThreads = 512;
constexpr int kSpStage = 8;        // delta elements per thread and job: 6 single loads (DR Y <= 6 x 512) or 2 quads
constexpr int kSpListThreads = 256;   // corr_lists_kernel: 8 tile bytes per thread

struct SparseShape {
    int B, Z, X, Y, kz, kx, ky, pz, px, py;
    int TXR, nxt, njobs;
    int DR;            // delta rows staged per job: TXR + kx - 1
    int P;             // row pitch of the delta tile (floats)
    int dl_floats;     // delta tile + a zero tail of (kx - 1) P + ky floats (what a padding entry's taps read), rounded to 4
    int T, groups;     // taps per plane, voxel groups
    int capP;          // entries per list (TXR Y + padding; a multiple of 8)
    int vec;           // list kernel: a thread's 8 input bytes are one aligned load and lie in one row
    int vec4;          // gather: the delta tile is staged by four-element loads (Y % 4 == 0, aligned pointers)
    unsigned y_magic;  // e / Y == mulhi(e, y_magic) for every e the kernels form (checked on the host)
    unsigned nxt_magic, b_magic;   // likewise job / nxt and (job / nxt) / B for every job index
    int dbg;           // wrong-result timing switch (SN_CONV_DEBUG builds only: SN_K4S_SKIP): 1 no job does anything, 2 no gather
};

// list index of (b, z', tile xt); the gather reads the lists of z' = z + dz - pz
__device__ __forceinline__ int list_index(const SparseShape& s, int b, int z, int xt) { return (b * s.Z + z) * s.nxt + xt; }

__global__ __launch_bounds__(kSpListThreads) void corr_lists_kernel(const uint8_t* __restrict__ x, SparseShape s,
                                                                     uint16_t* __restrict__ lists,
                                                                     int* __restrict__ counts) {
    __shared__ int wtot[kSpListThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = blockIdx.x;
    const int xt = li % s.nxt, bz = li / s.nxt;
    const int x0 = xt * s.TXR;
    const int nrows = (s.X - x0 < s.TXR) ? s.X - x0 : s.TXR;
    const int nbytes = nrows * s.Y;
    const uint8_t* src = x + (size_t)bz * s.X * s.Y + (size_t)x0 * s.Y + 8 * tid;
    uint32_t w0 = 0u, w1 = 0u;
    if (s.vec) {
        if (8 * tid < nbytes) {
            const uint2 v = *reinterpret_cast<const uint2*>(src);
            w0 = v.x; w1 = v.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (8 * tid + k < nbytes) w0 |= (uint32_t)src[k] << (8 * k);
            if (8 * tid + 4 + k < nbytes) w1 |= (uint32_t)src[4 + k] << (8 * k);
        }
    }
    auto nibble = [](uint32_t w) -> uint32_t {   // bit k: byte k is non-zero
        uint32_t v = w | (w >> 4);
        v |= v >> 2;
        v |= v >> 1;
        v &= 0x01010101u;
        return ((v * 0x01020408u) >> 24) & 0xfu;   // bits 0, 8, 16, 24 -> 0, 1, 2, 3
    };
    const uint32_t mask = nibble(w0) | (nibble(w1) << 4);
    const int c = __builtin_popcount(mask);
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kSpListThreads / 64; ++k) {
        const int t = wtot[k];
        if (k < wave) base += t;
        total += t;
    }
    uint16_t* list = lists + (size_t)li * s.capP;
    if (mask) {
        int pos = base + incl - c;
        int row = (int)__umulhi((unsigned)(8 * tid), s.y_magic), col = 8 * tid - row * s.Y;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((mask >> k) & 1u) list[pos++] = (uint16_t)(((row + s.kx - 1) * s.P + col + s.ky - 1) * 4);
            if (++col >= s.Y) { col = 0; ++row; }
        }
    }
    // padding: the last quad of every group's walk is complete; its entries read the zero tail of the tile
    if (tid < 4 * s.groups) list[total + tid] = (uint16_t)((s.DR * s.P + (s.kx - 1) * s.P + (s.ky - 1)) * 4);
    // what the gather needs to know: the list's length in ROUNDS (quads per group) -- no division on its side
    if (tid == 0) counts[li] = (total + 4 * s.groups - 1) / (4 * s.groups);
}

template <typename DT, int KZMAX>
__global__ __launch_bounds__(kSpThreads, 6) void corr_gather_kernel(const DT* __restrict__ gout, const DT* __restrict__ out,
                                                                     const uint16_t* __restrict__ lists,
                                                                     const int* __restrict__ counts, SparseShape s,
                                                                     float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dl = lds;
    uint16_t* ll = reinterpret_cast<uint16_t*>(dl + s.dl_floats);   // the job's kz lists, back to back (padded lengths)
    const int tid = threadIdx.x;
    const size_t plane = (size_t)s.X * s.Y;
    const int grp = tid / s.T, tap = tid - grp * s.T;
    const bool gathers = grp < s.groups;   // (the threads past the last group walk group 0's quads; their sums are dropped)
    const int tdx = tap / s.ky, tdy = tap - tdx * s.ky;
    // a voxel's cell is (row rr + kx - 1, column y + ky - 1) of the tile; tap (dx, dy) reads dx rows and dy columns before it
    const char* const tap_base = reinterpret_cast<const char*>(dl) - (tdx * s.P + tdy) * 4;
    const int quad = 4 * s.groups;   // list lengths are padded to this: every group runs the same number of rounds
    const uint16_t* const my_quads = ll + 4 * (gathers ? grp : 0);

    float acc[KZMAX];
#pragma unroll
    for (int i = 0; i < KZMAX; ++i) acc[i] = 0.f;
    for (int i = tid; i < s.dl_floats; i += kSpThreads) dl[i] = 0.f;   // halo columns and the tail stay zero for good

    struct Job { int b, z, xt, e_lo; };   // delta element e of the tile is grid element e_lo + e of plane (b, z)
    auto decode = [&](int job) -> Job {
        // z-major, batch fastest: a persistent workgroup's jobs fall on different z (the ground planes hold most set voxels)
        Job r;
        const int j = s.nxt_magic ? (int)__umulhi((unsigned)job, s.nxt_magic) : job;   // (magic 0: divisor 1)
        r.xt = job - j * s.nxt;
        r.z = s.b_magic ? (int)__umulhi((unsigned)j, s.b_magic) : j;
        r.b = j - r.z * s.B;
        r.e_lo = (r.xt * s.TXR - (s.kx - 1 - s.px)) * s.Y;   // rows are whole: (q0 + rr) Y + y = q0 Y + e
        return r;
    };
    // a job's list lengths in rounds (wave-uniform: scalar loads); list dz is list_index(b, z - pz, xt) + dz nxt
    auto list_counts = [&](const Job& jb, int (&n)[KZMAX]) {
        const int i0 = list_index(s, jb.b, jb.z - s.pz, jb.xt);
#pragma unroll
        for (int dz = 0; dz < KZMAX; ++dz) {
            const int zp = jb.z + dz - s.pz;
            n[dz] = (dz < s.kz && zp >= 0 && zp < s.Z) ? counts[i0 + dz * s.nxt] : 0;
        }
    };
    // chunk c (four entries) of the job's lists, from global memory; chunks past the end read as nothing.  Which list a
    // chunk belongs to is found without branches (the lengths are wave-uniform, the chunk index is not).
    auto load_chunk = [&](const Job& jb, const int (&n)[KZMAX], int c) -> uint2 {
        int dz = 0, rel = c, end = 0;
#pragma unroll
        for (int k = 0; k < KZMAX; ++k) {
            end += n[k] * s.groups;              // chunks of lists 0 .. k
            const bool past = c >= end;
            dz += past ? 1 : 0;
            rel = past ? c - end : rel;
        }
        uint2 v = make_uint2(0u, 0u);
        if (c < end) {
            const int idx = list_index(s, jb.b, jb.z - s.pz, jb.xt) + dz * s.nxt;
            v = *reinterpret_cast<const uint2*>(lists + (size_t)idx * s.capP + 4 * rel);
        }
        return v;
    };
    // the delta tile's elements of this thread: loads are unconditional (an element outside the grid reads element 0 and
    // is zeroed when the tile is written: no divergent control flow around twelve loads)
    const int nel = s.DR * s.Y;
    const unsigned e_hi = (unsigned)(s.X * s.Y);
    auto issue_delta = [&](const Job& jb, float (&g)[kSpStage], float (&o)[kSpStage]) {
        const size_t dbase = ((size_t)jb.b * s.Z + jb.z) * plane;
        if (s.vec4) {
            // four elements per load (rows are whole and Y % 4 == 0: a quad lies in one row, inside or outside the grid as a
            // whole): two loads per thread and array instead of five ([measured] the scalar form: a latency-bound stream,
            // 22 of the kernel's 42 us with every job's work switched off; bf16 storage was no faster than fp32)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float gq[4] = {0.f, 0.f, 0.f, 0.f}, oq[4] = {1.f, 1.f, 1.f, 1.f};
                if (4 * u * kSpThreads < nel) {   // (uniform)
                    const int ge = jb.e_lo + 4 * (u * kSpThreads + tid);
                    const unsigned at = (unsigned)ge < e_hi ? (unsigned)ge : 0u;
                    load_quad(gout + dbase + at, gq);
                    if (out) load_quad(out + dbase + at, oq);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) g[4 * u + k] = gq[k], o[4 * u + k] = oq[k];
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            g[u] = 0.f; o[u] = 1.f;
            if (u * kSpThreads < nel) {   // (uniform)
                const int ge = jb.e_lo + u * kSpThreads + tid;
                const unsigned at = (unsigned)ge < e_hi ? (unsigned)ge : 0u;
                g[u] = (float)gout[dbase + at];
                if (out) o[u] = (float)out[dbase + at];
            }
        }
    };

    // Software pipeline over the workgroup's jobs: the list lengths are fetched two jobs ahead (the chunk addresses depend
    // on them), delta / output / list chunk one job ahead -- issued before a gather, consumed after it.
    float g[kSpStage], o[kSpStage];
    int n[KZMAX], nn[KZMAX];
    uint2 chunk = make_uint2(0u, 0u);
    int job = blockIdx.x;
    Job cur = decode(job < s.njobs ? job : 0), nxt = cur;
#pragma unroll
    for (int dz = 0; dz < KZMAX; ++dz) n[dz] = nn[dz] = 0;
    if (job < s.njobs) {
        list_counts(cur, n);
        issue_delta(cur, g, o);
        chunk = load_chunk(cur, n, tid);
        if (job + (int)gridDim.x < s.njobs) {
            nxt = decode(job + gridDim.x);
            list_counts(nxt, nn);
        }
    }
    for (; job < s.njobs; job += gridDim.x) {
        int rounds[KZMAX], total = 0;
#pragma unroll
        for (int dz = 0; dz < KZMAX; ++dz) rounds[dz] = n[dz], total += n[dz];
        const bool work = total > 0 && !SN_DBG(s, 1);   // no set voxel in reach of this delta tile: nothing to add
        if (work) {
            __syncthreads();   // the previous job's gathers are done with the delta tile and the lists
            // ---- delta tile: rows q0 .. q0 + DR - 1 (zero outside the grid), columns ky - 1 - py + y
            if (s.vec4) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = 4 * (u * kSpThreads + tid);
                    if (e < nel) {
                        const int rr = (int)__umulhi((unsigned)e, s.y_magic), y = e - rr * s.Y;
                        const bool outside = (unsigned)(cur.e_lo + e) >= e_hi;   // (the quad's row)
                        float* dst = dl + rr * s.P + (s.ky - 1 - s.py) + y;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float d = g[4 * u + k];
                            if (out) d = (o[4 * u + k] > 0.f) ? d * (1.f - o[4 * u + k] * o[4 * u + k]) : 0.f;
                            dst[k] = outside ? 0.f : d;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int e = u * kSpThreads + tid;
                    if (e < nel) {
                        const int rr = (int)__umulhi((unsigned)e, s.y_magic), y = e - rr * s.Y;
                        float d = g[u];
                        if (out) d = (o[u] > 0.f) ? d * (1.f - o[u] * o[u]) : 0.f;
                        if ((unsigned)(cur.e_lo + e) >= e_hi) d = 0.f;   // a row outside the grid
                        dl[rr * s.P + (s.ky - 1 - s.py) + y] = d;
                    }
                }
            }
            // ---- the lists: chunk tid came with the prefetch; a job with more than 512 chunks fetches the rest now
            const int nchunks = total * s.groups;
            if (tid < nchunks) *reinterpret_cast<uint2*>(ll + 4 * tid) = chunk;
            for (int c = tid + kSpThreads; c < nchunks; c += kSpThreads)
                *reinterpret_cast<uint2*>(ll + 4 * c) = load_chunk(cur, n, c);
            __syncthreads();
        }
        // ---- the next job's loads fly during the gather
        const int next = job + gridDim.x;
        if (next < s.njobs) {
            cur = nxt;
            int any = 0;
#pragma unroll
            for (int dz = 0; dz < KZMAX; ++dz) n[dz] = nn[dz], any |= nn[dz];
            if (any) {   // (uniform; a job with no set voxel in reach reads nothing)
                issue_delta(cur, g, o);
                chunk = load_chunk(cur, n, tid);
            }
            if (next + (int)gridDim.x < s.njobs) {
                nxt = decode(next + gridDim.x);
                list_counts(nxt, nn);
            }
        }
        if (work && !SN_DBG(s, 2)) {
            // every thread of a wave runs the same rounds (uniform trip counts: scalar loop control); the entries of the
            // next round are requested before this round's four reads
            const uint16_t* q = my_quads;
#pragma unroll
            for (int dz = 0; dz < KZMAX; ++dz) {
                if (dz >= s.kz) continue;
                const int nr = rounds[dz];
                if (nr == 0) continue;
                float a = acc[dz];
                auto visit4 = [&](const uint2& e) {
                    const float v0 = *reinterpret_cast<const float*>(tap_base + (e.x & 0xffffu));
                    const float v1 = *reinterpret_cast<const float*>(tap_base + (e.x >> 16));
                    const float v2 = *reinterpret_cast<const float*>(tap_base + (e.y & 0xffffu));
                    const float v3 = *reinterpret_cast<const float*>(tap_base + (e.y >> 16));
                    a += v0; a += v1; a += v2; a += v3;
                };
                // two rounds per trip, the entries of each requested one round ahead into their own registers (hipcc turns
                // a single rotating pair into "request, wait, copy" at the top of the round: two LDS round trips per round)
                uint2 ea = *reinterpret_cast<const uint2*>(q);
                int r = 0;
                for (; r + 1 < nr; r += 2) {
                    const uint2 eb = *reinterpret_cast<const uint2*>(q + quad);
                    __builtin_amdgcn_sched_barrier(0);
                    visit4(ea);
                    q += 2 * quad;
                    ea = *reinterpret_cast<const uint2*>(q);   // (one quad past the last list: slack in `ll`)
                    __builtin_amdgcn_sched_barrier(0);
                    visit4(eb);
                }
                if (r < nr) {
                    visit4(ea);
                    q += quad;
                }
                acc[dz] = a;
            }
        }
    }

    // ---- groups add up in order, then the workgroup writes its partial row
    __syncthreads();
    float* red = lds;   // [groups][kz][T]
    if (gathers) {
#pragma unroll
        for (int dz = 0; dz < KZMAX; ++dz)
            if (dz < s.kz) red[(grp * s.kz + dz) * s.T + tap] = acc[dz];
    }
    __syncthreads();
    const int ntaps = s.kz * s.T;
    float* prow = partial + (size_t)blockIdx.x * ntaps;
    for (int t = tid; t < ntaps; t += kSpThreads) {
        float a = red[t];
        for (int gq = 1; gq < s.groups; ++gq) a += red[gq * ntaps + t];
        prow[t] = a;
    }
}

// u / d == mulhi(u, magic) for all 0 <= u <= umax?  magic = floor(2^32 / d) + 1, checked exhaustively (umax is at most a
// few million); d = 1 has no such multiplier: magic 0 stands for "the quotient is u" (div_magic below).
bool division_magic_u(unsigned d, unsigned umax, unsigned& magic) {
    if (d == 1) {
        magic = 0u;
        return true;
    }
    magic = (unsigned)((((unsigned long long)1 << 32) / d) + 1ull);
    for (unsigned long long u = 0; u <= umax; ++u)
        if ((unsigned)((u * magic) >> 32) != (unsigned)(u / d)) return false;
    return true;
}

struct SparsePlan {
    SparseShape s;
    size_t lds_bytes;
    int grid;          // gather workgroups = partial rows
    int nlists;        // B Z nxt
    size_t off_counts, off_lists, ws_bytes;   // workspace: [grid x ntaps floats | nlists ints | nlists x capP uint16]
};

bool plan_sparse(int B, int Z, int X, int Y, int kz, int kx, int ky, int tile_bytes, SparsePlan* p) {
    SparseShape* s = &p->s;
    const int KZMAX = kz <= 9 ? 9 : 16;
    if (kz > 16 || kx * ky > kSpThreads || Y > 8 * kSpListThreads) return false;
    s->B = B; s->Z = Z; s->X = X; s->Y = Y; s->kz = kz; s->kx = kx; s->ky = ky;
    s->pz = (kz - 1) / 2; s->px = (kx - 1) / 2; s->py = (ky - 1) / 2;
    s->T = kx * ky;
    s->groups = kSpThreads / s->T;
    // input rows per job: 8 bytes per thread of the list kernel at most, and the delta rows they reach must be kSpStage
    // elements per thread of the gather
    int cap_bytes = tile_bytes > 0 ? tile_bytes : 8 * kSpListThreads;
    if (cap_bytes > 8 * kSpListThreads) cap_bytes = 8 * kSpListThreads;
    int txr = cap_bytes / Y;
    if (txr < 1) txr = 1;
    if (txr > X) txr = X;
    while (txr > 1 && (txr + kx - 1) * Y > 6 * kSpThreads) --txr;
    if ((txr + kx - 1) * Y > 6 * kSpThreads) return false;
    s->TXR = txr;
    s->nxt = (X + txr - 1) / txr;
    s->DR = txr + kx - 1;
    int P = Y + ky - 1;
    while (P % 64 != ky % 64) ++P;
    s->P = P;
    s->dl_floats = (s->DR * P + (kx - 1) * P + ky + 3) / 4 * 4;
    if ((size_t)s->dl_floats * 4 > 65535) return false;   // list entries are 16-bit byte offsets into the tile
    const int quad = 4 * s->groups;
    s->capP = ((txr * Y + quad - 1) / quad * quad + quad + 7) / 8 * 8;
    const long long njobs = (long long)B * Z * s->nxt;
    if (njobs > 0x7fffffff / 2 || (long long)B * Z * X * Y > 0x7fffffff) return false;
    s->njobs = (int)njobs;
    size_t body = (size_t)s->dl_floats * 4 + (size_t)KZMAX * s->capP * 2 + (size_t)quad * 2 + 16;   // (+ the quad the last round requests ahead)
    const size_t red = (size_t)s->groups * kz * s->T * 4;   // the groups' sums at the end reuse the tile and the lists
    if (red > body) body = red;
    const size_t need = (body + 15) / 16 * 16;
    if (need > 150 * 1024) return false;
    p->lds_bytes = need;
    // e / Y by multiplication for every e the kernels divide: e < kSpStage x 512 (staging), 8 x 256 (lists)
    const unsigned emax = (unsigned)(kSpStage * kSpThreads + 8 * kSpListThreads);
    s->y_magic = (unsigned)((((unsigned long long)1 << 32) / (unsigned)Y) + 1ull);
    for (unsigned e = 0; e <= emax; ++e)
        if ((unsigned)(((unsigned long long)e * s->y_magic) >> 32) != e / (unsigned)Y) return false;
    unsigned um;
    if (!division_magic_u((unsigned)s->nxt, (unsigned)s->njobs, um)) return false;
    s->nxt_magic = um;
    if (!division_magic_u((unsigned)B, (unsigned)(s->njobs / s->nxt + 1), um)) return false;
    s->b_magic = um;
    const int per_cu = (int)((160 * 1024) / need);
    const int cap = 256 * (per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu));   // <= 768 rows
    p->grid = s->njobs < cap ? s->njobs : cap;
    p->nlists = s->njobs;
    const size_t ntaps = (size_t)kz * kx * ky;
    p->off_counts = ((size_t)p->grid * ntaps * 4 + 255) / 256 * 256;
    p->off_lists = (p->off_counts + (size_t)p->nlists * 4 + 255) / 256 * 256;
    p->ws_bytes = p->off_lists + (size_t)p->nlists * s->capP * 2;
    s->dbg = 0;
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

size_t sn::corr_sparse_ws_bytes(int x_dtype, int B, int Z, int X, int Y, int kz, int kx, int ky) {
    if (x_dtype != SN_OCC8 || sn::option_extra(sn::kOptCorrDense)) return 0;
    SparsePlan p;
    return plan_sparse(B, Z, X, Y, kz, kx, ky, sn::option_corr_sparse_tile_bytes(), &p) ? p.ws_bytes : 0;
}

int sn::corr_sparse_launch(const void* x, const void* gout, const void* out, int g_dtype, int B, int Z, int X, int Y, int kz,
                           int kx, int ky, void* ws, float* C, hipStream_t stream) {
    SparsePlan p;
    if (!plan_sparse(B, Z, X, Y, kz, kx, ky, sn::option_corr_sparse_tile_bytes(), &p))
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_corr_ws: shape outside the sparse correlation kernels");
    SparseShape& s = p.s;
    s.vec = (Y % 8 == 0) && ((uintptr_t)x % 8 == 0);
    {   // quads of the delta tile: 2 x 4 x 512 elements at most, whole rows of a multiple of four, aligned pointers
        const size_t ga = g_dtype == SN_BF16 ? 8 : 16;
        s.vec4 = (Y % 4 == 0) && (s.DR * Y <= 8 * kSpThreads) && ((uintptr_t)gout % ga == 0) && (!out || (uintptr_t)out % ga == 0);
    }
    s.dbg = sn::debug_env_int("SN_K4S_SKIP");
    float* partial = reinterpret_cast<float*>(ws);
    int* counts = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + p.off_counts);
    uint16_t* lists = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(ws) + p.off_lists);
    hipLaunchKernelGGL(corr_lists_kernel, dim3(p.nlists), dim3(kSpListThreads), 0, stream, (const uint8_t*)x, s, lists,
                       counts);
    if (int rc = sn::check_launch("sn_conv_corr_ws(lists)")) return rc;
#define SN_CORR_SPARSE(DT, KZMAX)                                                                                 \
    do {                                                                                                          \
        auto kern = corr_gather_kernel<DT, KZMAX>;                                                                \
        if (sn::ensure_dynamic_lds((const void*)kern, (int)p.lds_bytes) != hipSuccess)                            \
            return sn::check_launch("sn_conv_corr_ws(gather: hipFuncSetAttribute)");                              \
        hipLaunchKernelGGL(kern, dim3(p.grid), dim3(kSpThreads), p.lds_bytes, stream, (const DT*)gout,            \
                           (const DT*)out, (const uint16_t*)lists, (const int*)counts, s, partial);               \
    } while (0)
    if (g_dtype == SN_BF16) {
        if (kz <= 9) SN_CORR_SPARSE(__bf16, 9);
        else SN_CORR_SPARSE(__bf16, 16);
    } else {
        if (kz <= 9) SN_CORR_SPARSE(float, 9);
        else SN_CORR_SPARSE(float, 16);
    }
#undef SN_CORR_SPARSE
    if (int rc = sn::check_launch("sn_conv_corr_ws(gather)")) return rc;
    const int ntaps = kz * kx * ky;
    hipLaunchKernelGGL(corr_rows_reduce_kernel, dim3((ntaps + 3) / 4), dim3(256), 0, stream, partial, p.grid, ntaps, C);
    return sn::check_launch("sn_conv_corr_ws(reduce)");
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
