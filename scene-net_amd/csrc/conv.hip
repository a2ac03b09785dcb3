// K3 -- GENEO bank convolution as an LDS-tiled implicit GEMM on the fp32 matrix cores, with the
// convex-combination head (sum_i lambda_i conv_i -> relu(tanh)) fused into the epilogue.
//
// Follows SceneNet.forward, core/models/SCENE_Net.py:322-339: F.conv3d(x, kernels, padding='same')
// is a cross-correlation with zero padding (k-1)/2 left, k/2 right per axis.
//
// GEMM view, per 16-voxel strip along y:   D[g][n] += sum_t  W[g][t] * x[pos(n) + off(t)]
//   M = 16 GENEO kernels (rows, A operand = weights), N = 16 voxels (cols, B operand = im2col
//   of the LDS halo tile), K = kz*kx*ky taps in steps of 4 (v_mfma_f32_16x16x4_f32).
//   Operand lane maps (lane l): A[row = l&15][k = l>>4], B[k = l>>4][col = l&15],
//   D[row = 4*(l>>4) + r][col = l&15], r = 0..3.
//
// Persistent workgroups (one per CU, 512 threads = 8 waves, 2 per SIMD) walk the TZ x TX x 64 output
// tiles of the whole batch:
//   LDS = weight table [T4][64] (tap-step major, lane minor: one conflict-free ds_read_b32 per step)
//         and tap offset table [T4][4], staged ONCE per workgroup, plus the zero-padded fp32 halo
//         tile [TZ+kz-1][TX+kx-1][YP] (YP = 88: row stride chosen so that the two tap rows a 32-lane
//         LDS group can touch at a ky=9 row crossing fall on disjoint banks).
//   kDouble: two halo buffers -- the next tile's halo is fetched into registers before the MFMA loop of
//         the current tile starts and written to the other buffer after it, so global latency never
//         stalls the matrix pipe (one workgroup per CU: nothing else would hide it).
//   A wave "round" = 8 accumulator tiles (2 x-rows x 4 y-strips): per tap step one weight VGPR feeds
//   8 MFMAs, each with its own ds_read_b32 of the halo tile; hand software-pipelined (ping-pong).
#include "common.h"
#include <cstdlib>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// torch.relu keeps NaN (fmaxf(NaN, 0) would return 0)
__device__ __forceinline__ float relu_nan(float v) { return (v > 0.0f || v != v) ? v : 0.0f; }

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int TY = 64;   // y extent of a workgroup tile
constexpr int YP = 88;   // LDS row stride (floats): TY + up to 24 halo columns
constexpr int NV = 8;    // accumulator tiles per wave round: 2 x-rows x 4 y-strips
constexpr int kMaxLds = 160 * 1024;
constexpr int kTablePad = 2;    // extra all-zero tap steps the software pipeline may prefetch
constexpr int kStageRows = 18;  // halo rows a wave stages in registers (kDouble): ZP*XP <= 8*18 = 144

struct ConvShape {
    int B, Z, X, Y, G;
    int kz, kx, ky;
    int TZ, TX;          // workgroup tile (z, x); y is TY
    int nzt, nxt, nyt;   // tiles per axis
    int ntiles;          // B * nzt * nxt * nyt
    int T4;              // tap steps of 4, rounded up to even (ping-pong unroll)
    int Gtot, g0;        // this launch handles kernels g0 .. g0+G-1 of a bank of Gtot (act channel stride)
    int head;            // bit 0: add the partial sum already in `out`; bit 1: apply relu(tanh) (else store raw)
    sn::Gate gate;       // run only if every condition holds (common.h: Gate)
};

template <typename T>
__device__ __forceinline__ float load_as_float(const T* p, size_t i) { return (float)p[i]; }

struct TileCoord {
    int b, z0, x0, y0;
};

__device__ __forceinline__ TileCoord tile_coord(const ConvShape& s, int tile) {
    TileCoord c;
    c.y0 = (tile % s.nyt) * TY; tile /= s.nyt;
    c.x0 = (tile % s.nxt) * s.TX; tile /= s.nxt;
    c.z0 = (tile % s.nzt) * s.TZ; tile /= s.nzt;
    c.b = tile;
    return c;
}

// A zero page in device memory: out-of-grid halo elements are LOADED from here (address select, then an
// unconditional load) rather than skipped -- a conditional load makes hipcc branch around it and wait
// vmcnt(0) per element, which would serialise the staging.
__device__ double g_zero_page[2] = {0.0, 0.0};

// Halo staging, row based: halo row r = zz*XP + xx (ZP*XP rows of YL = TY+ky-1 valid columns).  Wave w owns
// rows w, w+8, ...; lanes run along y: column `lane` and column 64+lane (the latter only for lane < YL-64).
// All row arithmetic is wave-uniform and incremental (no integer division in the loop).
template <typename XT>
struct HaloStage {
    XT a[kStageRows], b[kStageRows];

    __device__ __forceinline__ void load(const XT* __restrict__ x, const ConvShape& s, const TileCoord& c, int wave,
                                         int lane, int XP, int rows) {
        const XT* zero = reinterpret_cast<const XT*>(g_zero_page);
        const int gy0 = c.y0 - (s.ky - 1) / 2 + lane, gy1 = gy0 + 64;
        const bool ok0 = (gy0 >= 0 && gy0 < s.Y);
        const bool ok1 = (lane < s.ky - 1 && gy1 >= 0 && gy1 < s.Y);
        int zz = wave / XP, xx = wave - zz * XP;
#pragma unroll
        for (int j = 0; j < kStageRows; ++j) {
            const int r = wave + j * kWaves;
            const int gz = c.z0 - (s.kz - 1) / 2 + zz, gx = c.x0 - (s.kx - 1) / 2 + xx;
            const bool okr = (r < rows && gz >= 0 && gz < s.Z && gx >= 0 && gx < s.X);
            const XT* row = x + (((size_t)c.b * s.Z + gz) * s.X + gx) * s.Y;
            const XT* p0 = (okr && ok0) ? row + gy0 : zero;
            const XT* p1 = (okr && ok1) ? row + gy1 : zero;
            a[j] = *p0;
            b[j] = *p1;
            xx += kWaves;
            while (xx >= XP) { xx -= XP; ++zz; }
        }
    }
    // makes every staged value "defined here": the loads' vmcnt wait lands at this point, not earlier
    __device__ __forceinline__ void fence() {
#pragma unroll
        for (int j = 0; j < kStageRows; ++j) {
            if constexpr (sizeof(XT) == 8) {
                asm volatile("" : "+v"(a[j]), "+v"(b[j]));
            } else {
                unsigned ua = (unsigned)a[j], ub = (unsigned)b[j];
                if constexpr (sizeof(XT) == 4) { ua = __float_as_uint((float)a[j]); ub = __float_as_uint((float)b[j]); }
                asm volatile("" : "+v"(ua), "+v"(ub));
                if constexpr (sizeof(XT) == 4) { a[j] = (XT)__uint_as_float(ua); b[j] = (XT)__uint_as_float(ub); }
                else { a[j] = (XT)ua; b[j] = (XT)ub; }
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ xs, const ConvShape& s, int wave, int lane,
                                          int rows) const {
#pragma unroll
        for (int j = 0; j < kStageRows; ++j) {
            const int r = wave + j * kWaves;
            if (r < rows) {
                xs[r * YP + lane] = (float)a[j];
                if (lane < s.ky - 1) xs[r * YP + 64 + lane] = (float)b[j];
            }
        }
    }
};

// generic (any number of rows): load kStageRows*kWaves rows at a time, store, repeat
template <typename XT>
__device__ __forceinline__ void halo_fill(float* __restrict__ xs, const XT* __restrict__ x, const ConvShape& s,
                                          const TileCoord& c, int wave, int lane, int XP, int rows) {
    const XT* zero = reinterpret_cast<const XT*>(g_zero_page);
    const int gy0 = c.y0 - (s.ky - 1) / 2 + lane, gy1 = gy0 + 64;
    const bool ok0 = (gy0 >= 0 && gy0 < s.Y);
    const bool ok1 = (lane < s.ky - 1 && gy1 >= 0 && gy1 < s.Y);
    constexpr int kBatch = 8;
    for (int r0 = wave; r0 < rows; r0 += kWaves * kBatch) {
        XT a[kBatch], b[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int r = r0 + u * kWaves;
            const int zz = r / XP, xx = r - zz * XP;
            const int gz = c.z0 - (s.kz - 1) / 2 + zz, gx = c.x0 - (s.kx - 1) / 2 + xx;
            const bool okr = (r < rows && gz >= 0 && gz < s.Z && gx >= 0 && gx < s.X);
            const XT* row = x + (((size_t)c.b * s.Z + gz) * s.X + gx) * s.Y;
            a[u] = *((okr && ok0) ? row + gy0 : zero);
            b[u] = *((okr && ok1) ? row + gy1 : zero);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int r = r0 + u * kWaves;
            if (r < rows) {
                xs[r * YP + lane] = (float)a[u];
                if (lane < s.ky - 1) xs[r * YP + 64 + lane] = (float)b[u];
            }
        }
    }
}

template <typename XT, typename OT, bool kDouble>
__global__ __launch_bounds__(kThreads) void conv_bank_kernel(const XT* __restrict__ x,
                                                             const float* __restrict__ bank,
                                                             const float* __restrict__ lambdas, ConvShape s,
                                                             OT* __restrict__ act, OT* __restrict__ out) {
    if (!s.gate.pass()) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;

    const int ZP = s.TZ + s.kz - 1, XP = s.TX + s.kx - 1;
    const int rows = ZP * XP;
    const int ntaps = s.kz * s.kx * s.ky;
    const int TT = s.T4 + kTablePad;                         // table rows incl. prefetch overrun (zero weights)
    float* Wt = lds;                                         // [TT][64]
    int* offt = reinterpret_cast<int*>(lds + TT * 64);       // [TT][4]
    float* xs0 = lds + TT * 64 + TT * 4;                     // [ZP][XP][YP]  (x2 when kDouble)
    const int halo_floats = ZP * XP * YP;

    // ---- once per workgroup: weights  Wt[t][l] = bank[g = l&15][tap = 4t + (l>>4)]  (loads batched)
    constexpr int kStage = 8;
    for (int base = tid; base < TT * 64; base += kThreads * kStage) {
        float v[kStage];
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
            const int i = base + u * kThreads;
            const int t = i >> 6, l = i & 63;
            const int g = l & 15, tap = 4 * t + (l >> 4);
            v[u] = (i < TT * 64 && g < s.G && tap < ntaps) ? bank[(size_t)g * ntaps + tap] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
            const int i = base + u * kThreads;
            if (i < TT * 64) Wt[i] = v[u];
        }
    }
    for (int i = tid; i < TT * 4; i += kThreads) {
        int o = 0;
        if (i < ntaps) {
            const int dy = i % s.ky, dx = (i / s.ky) % s.kx, dz = i / (s.ky * s.kx);
            o = (dz * XP + dx) * YP + dy;
        }
        offt[i] = o;
    }
    // columns YL..YP-1 of every halo row are never read with a non-zero weight but are read by the padded tap
    // steps (offset 0 .. + strip offsets < YL), so they never need initialising.

    float lam[4] = {0.f, 0.f, 0.f, 0.f};
    if (out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g = 4 * q + r;
            lam[r] = (g < s.G) ? lambdas[g] : 0.0f;
        }
    }

    const int half_tx = s.TX >> 1;
    const int nrounds = s.TZ * half_tx;
    const size_t V = (size_t)s.Z * s.X * s.Y;

    int tile = blockIdx.x;
    if (tile >= s.ntiles) return;
    // ---- first halo tile straight into buffer 0
    halo_fill(xs0, x, s, tile_coord(s, tile), wave, lane, XP, rows);
    __syncthreads();

    int cur = 0;
    for (; tile < s.ntiles; tile += gridDim.x) {
        const TileCoord c = tile_coord(s, tile);
        const int next = tile + gridDim.x;
        const bool has_next = next < s.ntiles;
        float* xs = xs0 + (kDouble ? cur * halo_floats : 0);

        // ---- (kDouble) next tile's halo: global loads issued now, consumed after the MFMA loop
        HaloStage<XT> stage;
        if (kDouble && has_next) stage.load(x, s, tile_coord(s, next), wave, lane, XP, rows);

        for (int round = wave; round < nrounds; round += kWaves) {
            const int lz = round / half_tx, lx = (round - lz * half_tx) * 2;
            const float* xrow = xs + (lz * XP + lx) * YP + n;

            f32x4 acc[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};

            // Software pipeline, ping-pong unrolled by 2: while the 8 MFMAs of tap step t issue, the weight and
            // the 8 halo-tile operands of step t+1 (and the tap offset of step t+2) are already in flight.
            const float* wp = Wt + lane;
            const int* op = offt + q;
            float wa = wp[0], wb;
            int oa = op[4], ob;  // offset of the NEXT step to load
            float xa[NV], xb[NV];
            {
                const float* xp = xrow + op[0];
#pragma unroll
                for (int v = 0; v < NV; ++v) xa[v] = xp[(v >> 2) * YP + (v & 3) * 16];
            }
            for (int t = 0; t < s.T4; t += 2) {
                {   // prefetch step t+1 into (wb, xb); offset for step t+2
                    wb = wp[(t + 1) * 64];
                    ob = op[(t + 2) * 4];
                    const float* xp = xrow + oa;
#pragma unroll
                    for (int v = 0; v < NV; ++v) xb[v] = xp[(v >> 2) * YP + (v & 3) * 16];
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the loads of t+1 ahead of the MFMAs of t
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xa[v], acc[v], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                {   // prefetch step t+2 into (wa, xa); offset for step t+3
                    wa = wp[(t + 2) * 64];
                    oa = op[(t + 3) * 4];
                    const float* xp = xrow + ob;
#pragma unroll
                    for (int v = 0; v < NV; ++v) xa[v] = xp[(v >> 2) * YP + (v & 3) * 16];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb, xb[v], acc[v], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }

            // ---- epilogue
            const int gz = c.z0 + lz;
            if (gz >= s.Z) continue;
            if (act) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int gx = c.x0 + lx + (v >> 2), gy = c.y0 + (v & 3) * 16 + n;
                    if (gx < s.X && gy < s.Y) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int g = 4 * q + r;
                            if (g < s.G)
                                act[((size_t)c.b * s.Gtot + s.g0 + g) * V + ((size_t)gz * s.X + gx) * s.Y + gy] =
                                    (OT)acc[v][r];
                        }
                    }
                }
            }
            if (out) {
                float sums[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float p = lam[0] * acc[v][0];
                    p = fmaf(lam[1], acc[v][1], p);
                    p = fmaf(lam[2], acc[v][2], p);
                    p = fmaf(lam[3], acc[v][3], p);
                    p += __shfl_xor(p, 16, 64);
                    p += __shfl_xor(p, 32, 64);
                    sums[v] = p;
                }
                // lane (q, n) writes y-strip q of both x-rows: 64 lanes = 256 contiguous bytes per row
#pragma unroll
                for (int xr = 0; xr < 2; ++xr) {
                    const float a0 = sums[4 * xr + 0], a1 = sums[4 * xr + 1], a2 = sums[4 * xr + 2],
                                a3 = sums[4 * xr + 3];
                    const float sv = (q == 0) ? a0 : (q == 1) ? a1 : (q == 2) ? a2 : a3;
                    const int gx = c.x0 + lx + xr, gy = c.y0 + q * 16 + n;
                    if (gx < s.X && gy < s.Y) {
                        OT* o = out + (size_t)c.b * V + ((size_t)gz * s.X + gx) * s.Y + gy;
                        float t = sv;
                        if (s.head & 1) t += (float)*o;  // kernels of earlier 16-groups (G > 16)
                        *o = (OT)((s.head & 2) ? relu_nan(tanhf(t)) : t);
                    }
                }
            }
        }

        // ---- hand the next halo tile over
        if (kDouble) {
            if (has_next) {
                stage.fence();
                stage.store(xs0 + (cur ^ 1) * halo_floats, s, wave, lane, rows);
            }
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();  // every wave is done reading the single buffer
            if (has_next) halo_fill(xs0, x, s, tile_coord(s, next), wave, lane, XP, rows);
            __syncthreads();
        }
    }
}

size_t lds_bytes(const ConvShape& s, bool dbl) {
    const size_t ZP = s.TZ + s.kz - 1, XP = s.TX + s.kx - 1;
    const size_t TT = s.T4 + kTablePad;
    return (TT * 64 + TT * 4 + (dbl ? 2 : 1) * ZP * XP * YP) * sizeof(float);
}

int num_cus() {
    static thread_local int cached = 0;
    if (cached) return cached;
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 256;
    }
    cached = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    return cached;
}

template <typename XT, typename OT, bool kDouble>
int launch(const void* x, const float* bank, const float* lambdas, const ConvShape& s, void* act, void* out,
           hipStream_t stream) {
    auto kern = conv_bank_kernel<XT, OT, kDouble>;
    const size_t lds = lds_bytes(s, kDouble);
    if (sn::ensure_dynamic_lds((const void*)kern, kMaxLds) != hipSuccess)
        return sn::check_launch("sn_conv_bank(hipFuncSetAttribute)");
    int grid = num_cus();  // persistent: one workgroup per CU (LDS admits exactly one)
    if (grid > s.ntiles) grid = s.ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, (const XT*)x, bank, lambdas, s, (OT*)act,
                       (OT*)out);
    return sn::check_launch("sn_conv_bank");
}

template <typename XT, typename OT>
int launch2(bool dbl, const void* x, const float* bank, const float* lambdas, const ConvShape& s, void* act,
            void* out, hipStream_t st) {
    return dbl ? launch<XT, OT, true>(x, bank, lambdas, s, act, out, st)
               : launch<XT, OT, false>(x, bank, lambdas, s, act, out, st);
}

void set_tile(ConvShape& s, int tz, int tx) {
    s.TZ = tz; s.TX = tx;
    s.nzt = (s.Z + tz - 1) / tz; s.nxt = (s.X + tx - 1) / tx;
    s.ntiles = s.B * s.nzt * s.nxt * s.nyt;
}

}  // namespace

namespace sn {
int conv_occ_i8(const uint8_t* x, const float* bank, const float* lambdas, int B, int Z, int X, int Y, int G, int Gtot,
                int g0, int head, int kz, int kx, int ky, void* act, void* out, int out_dtype,
                hipStream_t stream);  // conv_i8.hip
int conv_occ_i8s(const uint8_t* x, const float* bank, const float* lambdas, int B, int Z, int X, int Y, int G, int Gtot,
                 int g0, int head, int kz, int kx, int ky, void* act, void* out, int out_dtype,
                 hipStream_t stream);  // conv_i8s.hip
int conv_bank_group(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X, int Y,
                    int G, int Gtot, int g0, int head, int kz, int kx, int ky, void* act, void* out, int out_dtype,
                    sn_stream_t stream);
}

extern "C" int sn_conv_bank(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X,
                            int Y, int G, int kz, int kx, int ky, void* act, void* out, int out_dtype,
                            sn_stream_t stream) {
    if (!x || !bank) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: null x or bank");
    if (!act && !out) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: both act and out are null");
    if (out && !lambdas) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: out needs lambdas");
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || G <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: non-positive extent");
    if (G > 16) {
        // one MFMA row block holds 16 kernels: run the bank in groups of 16; `out` carries the raw partial sum
        // between the launches and the last one applies relu(tanh)
        const size_t ntaps = (size_t)kz * kx * ky;
        for (int g0 = 0; g0 < G; g0 += 16) {
            const int gc = (G - g0 < 16) ? G - g0 : 16;
            const int head = (g0 > 0 ? 1 : 0) | (g0 + gc >= G ? 2 : 0);
            const int rc = sn::conv_bank_group(x, x_dtype, bank + g0 * ntaps, lambdas ? lambdas + g0 : nullptr, B, Z, X,
                                               Y, gc, G, g0, head, kz, kx, ky, act, out, out_dtype, stream);
            if (rc != SN_OK) return rc;
        }
        return SN_OK;
    }
    return sn::conv_bank_group(x, x_dtype, bank, lambdas, B, Z, X, Y, G, G, 0, 2, kz, kx, ky, act, out, out_dtype,
                               stream);
}

int sn::conv_bank_group(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X,
                        int Y, int G, int Gtot, int g0, int head, int kz, int kx, int ky, void* act, void* out,
                        int out_dtype, sn_stream_t stream) {
    if (ky - 1 > YP - TY) return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_bank: ky=%d > %d", ky, YP - TY + 1);
    if (out_dtype != SN_F32 && out_dtype != SN_F64)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: out_dtype %d", out_dtype);
    if ((size_t)B * Z * X * Y > (size_t)1 << 40) return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_bank: grid too large");

    if (x_dtype == SN_OCC8) {
        // binary occupancy bytes: int8 matrix cores (exact integer accumulation of 24-bit fixed-point weights)
        if (!sn::option_extra(sn::kOptConvNoI8)) {
            // ky = 9: the stride-4 kernel (one halo copy, 12 MFMA steps); everything else: the four-copy kernel
            if (!sn::option_conv_i8_legacy()) {
                const int rs = sn::conv_occ_i8s((const uint8_t*)x, bank, lambdas, B, Z, X, Y, G, Gtot, g0, head, kz, kx, ky,
                                                act, out, out_dtype, sn::as_stream(stream));
                if (rs <= 0) return rs;
            }
            const int rc = sn::conv_occ_i8((const uint8_t*)x, bank, lambdas, B, Z, X, Y, G, Gtot, g0, head, kz, kx, ky,
                                           act, out, out_dtype, sn::as_stream(stream));
            if (rc <= 0) return rc;
        }
        x_dtype = SN_U8;  // shape not served by the int8 kernel: same bytes through the fp32 kernel
    }

    ConvShape s;
    s.B = B; s.Z = Z; s.X = X; s.Y = Y; s.G = G; s.kz = kz; s.kx = kx; s.ky = ky;
    s.Gtot = Gtot; s.g0 = g0; s.head = head;
    s.gate = sn::current_gate();
    s.T4 = (((kz * kx * ky + 3) / 4) + 1) & ~1;
    s.nyt = (Y + TY - 1) / TY;
    const int cus = num_cus();
    // Preferred: double-buffered halo (global latency hidden behind the MFMA loop).  Needs two halo buffers in
    // LDS and at most kStageRows halo rows per wave (staged in registers); take the largest such tile that still
    // leaves every CU a few tiles.  Otherwise: single buffer, largest tile that fits.
    static const int cand[][2] = {{8, 8}, {4, 8}, {4, 4}, {2, 4}, {1, 4}, {1, 2}};
    bool dbl = false, found = false;
    // [measured, C2] the single-buffer 8x8x64 tile (4 rounds per wave between barriers, waves drift apart so one
    // wave's epilogue overlaps its SIMD partner's MFMAs) runs 127.7 TF; the double-buffered 4x4x64 tile (1 round
    // per wave per barrier: epilogues and halo hand-over of all waves coincide) 118.5 TF.  Double buffering is
    // therefore opt-in (SN_CONV_DOUBLE_BUFFER=1) until its tile can hold more rounds.
    if (sn::option_extra(sn::kOptConvDoubleBuffer)) {
        for (const auto& c : cand) {
            set_tile(s, c[0], c[1]);
            const int rows = (s.TZ + kz - 1) * (s.TX + kx - 1);
            if (lds_bytes(s, true) > (size_t)kMaxLds || rows > kStageRows * kWaves) continue;
            dbl = found = true;
            if (s.ntiles >= 4 * cus || (c[0] == 1 && c[1] == 2)) break;
        }
    }
    if (!found) {
        for (const auto& c : cand) {
            set_tile(s, c[0], c[1]);
            if (lds_bytes(s, false) > (size_t)kMaxLds) continue;
            found = true;
            if (s.ntiles >= 4 * cus) break;
        }
    }
    if (!found)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_bank: kernel %dx%dx%d does not fit the 160 KiB LDS tile", kz, kx,
                        ky);
    hipStream_t st = sn::as_stream(stream);
#define SN_DISPATCH(XT)                                                                        \
    return (out_dtype == SN_F32) ? launch2<XT, float>(dbl, x, bank, lambdas, s, act, out, st)  \
                                 : launch2<XT, double>(dbl, x, bank, lambdas, s, act, out, st)
    switch (x_dtype) {
        case SN_F32: SN_DISPATCH(float);
        case SN_F64: SN_DISPATCH(double);
        case SN_U8: SN_DISPATCH(uint8_t);
        default: return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: x_dtype %d", x_dtype);
    }
#undef SN_DISPATCH
}
