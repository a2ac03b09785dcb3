// K2 -- GENEO bank builder: the parameterised geometric kernels (cylinder, cone/arrow,
// negative sphere) are built on the fly from their few scalars, one workgroup per GENEO,
// values staged in LDS for the zero-sum / neg-factor mean subtraction.
//
// Follows (closed form of) cylinderv2.compute_kernel  core/models/geneos/cylinder.py:152-176,
//                           arrow.compute_kernel       core/models/geneos/arrow.py:214-252,
//                           negSpherev2.compute_kernel core/models/geneos/neg_sphere.py:166-199,
// including the reference's flat-order re-view ("scramble") for non-square / non-cubic sizes.
#include "common.h"

namespace {

#include "conv_prep.h"   // prep_one_kernel: the int8 contraction's per-bank preparation, run as this kernel's tail

#include "bank_body.inc"

constexpr int kThreads = kBankThreads;

template <bool kPrep>
__global__ __launch_bounds__(kThreads) void geneo_bank_kernel(const float* __restrict__ params,
                                                              const int32_t* __restrict__ kinds, int kz, int kx,
                                                              int ky, float* __restrict__ bank,
                                                              int32_t* __restrict__ status, int G,
                                                              float* __restrict__ lambdas,
                                                              const int32_t* __restrict__ order, int last,
                                                              float* __restrict__ lambdas_out,
                                                              uint8_t* __restrict__ prep) {
    extern __shared__ float lds[];
    geneo_bank_body<kPrep>(lds, blockIdx.x, threadIdx.x, params, kinds, kz, kx, ky, bank, status, G, lambdas, order, last,
                           lambdas_out, prep);
}

__global__ void effective_lambdas_kernel(float* __restrict__ lambdas, const int32_t* __restrict__ order, int G,
                                         int last, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    effective_lambdas_thread(lambdas, order, G, last, out);
}

}  // namespace

extern "C" int sn_effective_lambdas(float* lambdas, const int32_t* order, int G, int last, float* out,
                                    sn_stream_t stream) {
    if (!lambdas || !order || !out) return sn::fail(SN_ERR_INVALID_ARG, "sn_effective_lambdas: null pointer");
    if (G <= 0 || last < 0 || last >= G) return sn::fail(SN_ERR_INVALID_ARG, "sn_effective_lambdas: bad G / last");
    hipLaunchKernelGGL(effective_lambdas_kernel, dim3(1), dim3(64), 0, sn::as_stream(stream), lambdas, order, G, last,
                       out);
    return sn::check_launch("sn_effective_lambdas");
}

extern "C" int sn_geneo_bank(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky, float* bank,
                             int32_t* status, sn_stream_t stream) {
    if (!params || !kinds || !bank) return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank: null pointer");
    if (G <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank: non-positive extent (G=%d k=%d,%d,%d)", G, kz, kx, ky);
    const long vol = (long)kz * kx * ky;
    if (vol > 12000) return sn::fail(SN_ERR_UNSUPPORTED, "sn_geneo_bank: kernel volume %ld > 12000", vol);
    size_t lds = (size_t)(vol + kz + 1) * sizeof(float);
    hipLaunchKernelGGL(geneo_bank_kernel<false>, dim3(G), dim3(kThreads), lds, sn::as_stream(stream), params, kinds, kz, kx,
                       ky, bank, status, G, (float*)nullptr, (const int32_t*)nullptr, 0, (float*)nullptr, (uint8_t*)nullptr);
    return sn::check_launch("sn_geneo_bank");
}

extern "C" int sn_geneo_bank_lambdas(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky,
                                     float* bank, int32_t* status, float* lambdas, const int32_t* order, int last,
                                     float* lambdas_out, sn_stream_t stream) {
    if (!params || !kinds || !bank || !lambdas || !order || !lambdas_out)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank_lambdas: null pointer");
    if (G <= 0 || kz <= 0 || kx <= 0 || ky <= 0 || last < 0 || last >= G)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank_lambdas: bad extent / last");
    const long vol = (long)kz * kx * ky;
    if (vol > 12000) return sn::fail(SN_ERR_UNSUPPORTED, "sn_geneo_bank_lambdas: kernel volume %ld > 12000", vol);
    size_t lds = (size_t)(vol + kz + 1) * sizeof(float);
    hipLaunchKernelGGL(geneo_bank_kernel<false>, dim3(G + 1), dim3(kThreads), lds, sn::as_stream(stream), params, kinds, kz,
                       kx, ky, bank, status, G, lambdas, order, last, lambdas_out, (uint8_t*)nullptr);
    return sn::check_launch("sn_geneo_bank_lambdas");
}

extern "C" int sn_geneo_bank_prep(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky, float* bank,
                                  int32_t* status, float* lambdas, const int32_t* order, int last, float* lambdas_out,
                                  void* prep, sn_stream_t stream) {
    if (!params || !kinds || !bank || !prep) return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank_prep: null pointer");
    if (lambdas && (!order || !lambdas_out)) return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank_prep: lambdas without order / out");
    if (G <= 0 || (lambdas && (last < 0 || last >= G))) return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank_prep: bad G / last");
    if (kz != 9 || kx != 9 || ky != 9)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_geneo_bank_prep: the prepared contraction serves 9 x 9 x 9 kernels (got %d,%d,%d)", kz,
                        kx, ky);
    if (reinterpret_cast<uintptr_t>(prep) & 15) return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank_prep: prep must be 16-byte aligned");
    const size_t lds = (size_t)(729 + kz + 1 + 8) * sizeof(float);
    const int ngeneo = 16 * ((G + 15) / 16);
    hipLaunchKernelGGL(geneo_bank_kernel<true>, dim3(ngeneo + (lambdas ? 1 : 0)), dim3(kThreads), lds, sn::as_stream(stream),
                       params, kinds, kz, kx, ky, bank, status, G, lambdas, order, last, lambdas_out,
                       static_cast<uint8_t*>(prep));
    return sn::check_launch("sn_geneo_bank_prep");
}
