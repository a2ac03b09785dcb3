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

#include "conv_fp32.inc"
using namespace fp32k;

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
    bool dbl = false;
    // Preferred when asked for (sn_set_option "conv_double_buffer"): double-buffered halo.  [measured, C2] the single-buffer
    // 8x8x64 tile (4 rounds per wave between barriers, waves drift apart so one wave's epilogue overlaps its SIMD partner's
    // MFMAs) runs 127.7 TF; the double-buffered 4x4x64 tile (1 round per wave per barrier) 118.5 TF: opt-in.
    if (!plan_fp32(s, B, Z, X, Y, G, Gtot, g0, head, kz, kx, ky, num_cus(), sn::option_extra(sn::kOptConvDoubleBuffer) != 0, dbl))
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
