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
// Workgroup = 512 threads (8 waves, 2 per SIMD), one TZ x TX x 64 output tile:
//   LDS = weight table [T4][64] (tap-step major, lane minor: one conflict-free ds_read_b32 per
//         step), tap offset table [T4][4], and the zero-padded input halo tile
//         [TZ+kz-1][TX+kx-1][YP] in fp32 (YP = 88: row stride chosen so that the two tap rows a
//         32-lane LDS group can touch at a ky=9 row crossing fall on disjoint banks).
//   A wave "round" = 8 accumulator tiles (2 x-rows x 4 y-strips): per tap step one weight
//   VGPR feeds 8 MFMAs, each with its own ds_read_b32 of the halo tile.
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int TY = 64;   // y extent of a workgroup tile
constexpr int YP = 88;   // LDS row stride (floats): TY + up to 24 halo columns
constexpr int NV = 8;    // accumulator tiles per wave round: 2 x-rows x 4 y-strips
constexpr int kMaxLds = 160 * 1024;
constexpr int kTablePad = 2;  // extra all-zero tap steps the software pipeline may prefetch

struct ConvShape {
    int B, Z, X, Y, G;
    int kz, kx, ky;
    int TZ, TX;          // workgroup tile (z, x); y is TY
    int nzt, nxt, nyt;   // tiles per axis
    int T4;              // tap steps of 4, rounded up to even (ping-pong unroll)
};

template <typename T>
__device__ __forceinline__ float load_as_float(const T* p, size_t i) { return (float)p[i]; }

template <typename XT, typename OT>
__global__ __launch_bounds__(kThreads) void conv_bank_kernel(const XT* __restrict__ x,
                                                             const float* __restrict__ bank,
                                                             const float* __restrict__ lambdas, ConvShape s,
                                                             OT* __restrict__ act, OT* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;

    const int ZP = s.TZ + s.kz - 1, XP = s.TX + s.kx - 1;
    const int ntaps = s.kz * s.kx * s.ky;
    const int TT = s.T4 + kTablePad;                         // table rows incl. prefetch overrun (zero weights)
    float* Wt = lds;                                         // [TT][64]
    int* offt = reinterpret_cast<int*>(lds + TT * 64);       // [TT][4]
    float* xs = lds + TT * 64 + TT * 4;                      // [ZP][XP][YP]

    // ---- which tile
    int bid = blockIdx.x;
    const int yt = bid % s.nyt; bid /= s.nyt;
    const int xt = bid % s.nxt; bid /= s.nxt;
    const int zt = bid % s.nzt; bid /= s.nzt;
    const int b = bid;
    const int z0 = zt * s.TZ, x0 = xt * s.TX, y0 = yt * TY;
    const int pz = (s.kz - 1) / 2, px = (s.kx - 1) / 2, py = (s.ky - 1) / 2;

    // ---- stage weights: Wt[t][l] = bank[g = l&15][tap = 4t + (l>>4)].  Loads are issued kStage at a time
    // so their L2 latencies overlap instead of adding up (one workgroup per CU: nothing else hides them).
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
    // ---- stage the zero-padded halo tile (fp32): a wave takes whole rows (row index math is wave-uniform),
    // lanes run along y (coalesced), kStage rows in flight per wave.
    {
        const XT* xb = x + (size_t)b * s.Z * s.X * s.Y;
        const int YL = TY + s.ky - 1;
        const int rows = ZP * XP;
        const int gy0 = y0 - py + lane, gy1 = gy0 + 64;
        const bool ok0 = (gy0 >= 0 && gy0 < s.Y);                 // lane < YL always (YL >= 64)
        const bool ok1 = (lane + 64 < YL && gy1 >= 0 && gy1 < s.Y);
        for (int r0 = wave; r0 < rows; r0 += kWaves * kStage) {
            float v0[kStage], v1[kStage];
#pragma unroll
            for (int u = 0; u < kStage; ++u) {
                const int r = r0 + u * kWaves;
                const int zz = r / XP, xx = r - zz * XP;
                const int gz = z0 - pz + zz, gx = x0 - px + xx;
                const bool okr = (r < rows && gz >= 0 && gz < s.Z && gx >= 0 && gx < s.X);
                const size_t rowbase = ((size_t)gz * s.X + gx) * s.Y;
                v0[u] = (okr && ok0) ? load_as_float(xb, rowbase + gy0) : 0.0f;
                v1[u] = (okr && ok1) ? load_as_float(xb, rowbase + gy1) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kStage; ++u) {
                const int r = r0 + u * kWaves;
                if (r < rows) {
                    xs[r * YP + lane] = v0[u];
                    if (lane + 64 < YP) xs[r * YP + lane + 64] = v1[u];
                }
            }
        }
    }
    __syncthreads();

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
            for (int v = 0; v < NV; ++v) acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xa[v], acc[v], 0, 0, 0);
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
            for (int v = 0; v < NV; ++v) acc[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb, xb[v], acc[v], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue
        const int gz = z0 + lz;
        if (gz >= s.Z) continue;
        if (act) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int gx = x0 + lx + (v >> 2), gy = y0 + (v & 3) * 16 + n;
                if (gx < s.X && gy < s.Y) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int g = 4 * q + r;
                        if (g < s.G)
                            act[((size_t)b * s.G + g) * V + ((size_t)gz * s.X + gx) * s.Y + gy] = (OT)acc[v][r];
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
                const int gx = x0 + lx + xr, gy = y0 + q * 16 + n;
                if (gx < s.X && gy < s.Y)
                    out[(size_t)b * V + ((size_t)gz * s.X + gx) * s.Y + gy] = (OT)fmaxf(tanhf(sv), 0.0f);
            }
        }
    }
}

size_t lds_bytes(const ConvShape& s) {
    const size_t ZP = s.TZ + s.kz - 1, XP = s.TX + s.kx - 1;
    const size_t TT = s.T4 + kTablePad;
    return (TT * 64 + TT * 4 + ZP * XP * YP) * sizeof(float);
}

template <typename XT, typename OT>
int launch(const void* x, const float* bank, const float* lambdas, const ConvShape& s, void* act, void* out,
           hipStream_t stream) {
    auto kern = conv_bank_kernel<XT, OT>;
    const size_t lds = lds_bytes(s);
    static thread_local const void* configured = nullptr;
    if (configured != (const void*)kern) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess)
            return sn::check_launch("sn_conv_bank(hipFuncSetAttribute)");
        configured = (const void*)kern;
    }
    const unsigned grid = (unsigned)((size_t)s.B * s.nzt * s.nxt * s.nyt);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, (const XT*)x, bank, lambdas, s, (OT*)act,
                       (OT*)out);
    return sn::check_launch("sn_conv_bank");
}

}  // namespace

extern "C" int sn_conv_bank(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X,
                            int Y, int G, int kz, int kx, int ky, void* act, void* out, int out_dtype,
                            sn_stream_t stream) {
    if (!x || !bank) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: null x or bank");
    if (!act && !out) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: both act and out are null");
    if (out && !lambdas) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: out needs lambdas");
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || G <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: non-positive extent");
    if (G > 16) return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_bank: G=%d > 16 (one MFMA row block per call)", G);
    if (ky - 1 > YP - TY) return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_bank: ky=%d > %d", ky, YP - TY + 1);
    if (out_dtype != SN_F32 && out_dtype != SN_F64)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: out_dtype %d", out_dtype);

    ConvShape s;
    s.B = B; s.Z = Z; s.X = X; s.Y = Y; s.G = G; s.kz = kz; s.kx = kx; s.ky = ky;
    s.T4 = (((kz * kx * ky + 3) / 4) + 1) & ~1;
    s.nyt = (Y + TY - 1) / TY;
    // largest (TZ, TX) that fits LDS and still gives the 256 CUs a few workgroups each
    static const int cand[][2] = {{8, 8}, {4, 8}, {4, 4}, {2, 4}, {1, 4}, {1, 2}};
    bool found = false;
    for (const auto& c : cand) {
        s.TZ = c[0]; s.TX = c[1];
        s.nzt = (Z + s.TZ - 1) / s.TZ; s.nxt = (X + s.TX - 1) / s.TX;
        if (lds_bytes(s) > (size_t)kMaxLds) continue;
        found = true;
        if ((size_t)B * s.nzt * s.nxt * s.nyt >= 1024) break;
    }
    if (!found)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_bank: kernel %dx%dx%d does not fit the 160 KiB LDS tile", kz, kx,
                        ky);
    hipStream_t st = sn::as_stream(stream);
#define SN_DISPATCH(XT)                                                                   \
    return (out_dtype == SN_F32) ? launch<XT, float>(x, bank, lambdas, s, act, out, st)   \
                                 : launch<XT, double>(x, bank, lambdas, s, act, out, st)
    switch (x_dtype) {
        case SN_F32: SN_DISPATCH(float);
        case SN_F64: SN_DISPATCH(double);
        case SN_U8: SN_DISPATCH(uint8_t);
        default: return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank: x_dtype %d", x_dtype);
    }
#undef SN_DISPATCH
}
