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

constexpr int kThreads = 256;
constexpr float kEps = 1e-8f;        // default `epsilon` of the v2 gaussians
constexpr float kPi = 3.14159274f;   // torch.pi rounded to fp32 by the fp32 multiply

// sig * exp(r^4 * (-1 / (2 (rad + eps)^2)))
__device__ __forceinline__ float geneo_gauss(float r2, float rad, float sig) {
    float re = rad + kEps;
    float c = -1.0f / (2.0f * (re * re));
    return sig * expf((r2 * r2) * c);
}

// v1 generators: exp((r^2 - rad^2)^2 * (-1 / (2 sig^2)))   (cylinder.py:72-79, arrow.py:157-168, neg_sphere.py:123-131)
__device__ __forceinline__ float geneo_ring(float r2, float rad, float sig) {
    float cx = r2 - rad * rad;
    float c = -1.0f / (2.0f * (sig * sig));
    return expf((cx * cx) * c);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// SCENE_Net.py:329-335: the frozen coefficient is 1 - sum(all, in ParameterDict order) + itself, a sequential fp32
// sum; one thread reproduces it bit for bit.  Also stores it back (the reference re-creates that parameter).
__device__ void effective_lambdas_thread(float* __restrict__ lambdas, const int32_t* __restrict__ order, int G,
                                         int last, float* __restrict__ out) {
    float total = 0.f;
    for (int i = 0; i < G; ++i) total = __fadd_rn(total, lambdas[order[i]]);
    const float eff = __fadd_rn(__fsub_rn(1.0f, total), lambdas[last]);
    for (int g = 0; g < G; ++g) out[g] = (g == last) ? eff : lambdas[g];
    lambdas[last] = eff;
}

// grid G (+ 1 when the effective coefficients ride along: sn_geneo_bank_lambdas; that extra workgroup does them).
// kPrep (sn_geneo_bank_prep, 9 x 9 x 9 kernels): grid 16 ceil(G / 16) (+ 1); every workgroup g < 16 ceil(G / 16) also
// prepares kernel g for the int8 contraction (conv_prep.h) from the weights it has just built, still in LDS --
// workgroups G .. write the zero entries of the last group of 16.
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
    const int g = blockIdx.x;
    const int tid = threadIdx.x;
    const int ngeneo = kPrep ? 16 * ((G + 15) / 16) : G;
    if (g >= ngeneo) {
        if (tid == 0 && lambdas) effective_lambdas_thread(lambdas, order, G, last, lambdas_out);
        return;
    }
    if constexpr (kPrep) {
        if (g >= G) {   // a pad entry of the last group of 16: all-zero kernel
            for (int i = tid; i < 729; i += kThreads) lds[i] = 0.0f;
            prep_one_kernel(lds, reinterpret_cast<int*>(lds + 732), false, g & 15, prep + (size_t)(g >> 4) * SN_CONV_PREP_BYTES,
                            tid);
            return;
        }
    }
    const int nfloor = kx * ky;
    const int vol = kz * nfloor;
    float* vals = lds;          // [vol]
    float* seg_sum = lds + vol; // [kz] (cy / cone) or [1] (neg)

    const float* p = params + (size_t)g * SN_NPARAM;
    const int kind = kinds[g];
    const float radius = p[SN_P_RADIUS];
    const float sigma = p[SN_P_SIGMA];
    const float cx = (kx - 1) * 0.5f, cy = (ky - 1) * 0.5f, cz = (kz - 1) * 0.5f;

    const bool is_cone = (kind == SN_GENEO_CONE || kind == SN_GENEO_CONE_V1);
    const bool is_neg = (kind == SN_GENEO_NEG || kind == SN_GENEO_NEG_V1);
    int hc = 0;
    float cone_radius = 0.f, tan_inc = 0.f, neg = 0.f, inc_pi = 0.f;
    if (is_cone) {
        hc = (int)p[SN_P_APEX];  // truncation, arrow.py:235
        int bad = (hc < 0 || hc > kz);
        hc = min(max(hc, 0), kz);
        if (status && tid == 0) status[g] = bad;
        cone_radius = p[SN_P_CONE_RADIUS];
        float inc = fminf(fmaxf(p[SN_P_CONE_INC], 0.0f), 0.499f);  // arrow.py:244 (v2 only)
        tan_inc = tanf(inc * kPi);
        inc_pi = p[SN_P_CONE_INC] * kPi;  // v1: not clamped, arrow.py:189
    } else {
        if (status && tid == 0) status[g] = 0;
        if (is_neg) neg = p[SN_P_NEG_FACTOR];
    }

    for (int idx = tid; idx < vol; idx += kThreads) {
        float v;
        if (is_neg) {
            // output element idx of the row-major [kz,kx,ky] view takes flat-column row idx, whose
            // index triple is (k_z, i_x, j_y) = (idx % kz, (idx / kz) % kx, idx / (kz*kx))
            float dz = (float)(idx % kz) - cz;
            float dx = (float)((idx / kz) % kx) - cx;
            float dy = (float)(idx / (kz * kx)) - cy;
            const float r2 = dx * dx + dy * dy + dz * dz;
            v = (kind == SN_GENEO_NEG) ? (-neg) * geneo_gauss(r2, radius, sigma) : geneo_ring(r2, radius, sigma);
        } else {
            int z = idx / nfloor, n = idx - z * nfloor;
            // element n of the row-major [kx,ky] view takes flat-column row n = (i_x, j_y) = (n % kx, n / kx)
            float dx = (float)(n % kx) - cx;
            float dy = (float)(n / kx) - cy;
            const float r2 = dx * dx + dy * dy;
            if (kind == SN_GENEO_CY || kind == SN_GENEO_CONE) {
                float rad = radius;
                if (kind == SN_GENEO_CONE && z < kz - hc) rad = cone_radius * (float)z * tan_inc;
                v = geneo_gauss(r2, rad, sigma);
            } else {  // v1: ring gaussian; cone slices use sigma_h, h = 0 nearest the cylinder (prepend order)
                float sig = sigma;
                if (kind == SN_GENEO_CONE_V1 && z < kz - hc)
                    sig = cone_radius * sinf(inc_pi / (float)(2 + (kz - hc - 1 - z)));
                v = geneo_ring(r2, radius, sig);
            }
        }
        vals[idx] = v;
    }
    __syncthreads();

    const int nseg = is_neg ? 1 : kz;
    const int seg_len = is_neg ? vol : nfloor;
    const int wave = tid >> 6, lane = tid & 63;
    for (int s = wave; s < nseg; s += kThreads / 64) {
        float acc = 0.f;
        for (int i = lane; i < seg_len; i += 64) acc += vals[s * seg_len + i];
        acc = wave_sum(acc);
        if (lane == 0) seg_sum[s] = acc;
    }
    __syncthreads();

    float* out = bank + (size_t)g * vol;
    for (int idx = tid; idx < vol; idx += kThreads) {
        float mean;
        if (kind == SN_GENEO_NEG)
            mean = (seg_sum[0] + neg) / (float)vol;  // sum_negfactor, neg_sphere.py:181-182
        else if (kind == SN_GENEO_NEG_V1)
            mean = seg_sum[0] / (float)vol + neg;  // sum_zero(.) - neg_factor, neg_sphere.py:151
        else
            mean = seg_sum[idx / nfloor] / (float)nfloor;  // sum_zero, cylinder.py:81-82
        const float wv = vals[idx] - mean;
        out[idx] = wv;
        if constexpr (kPrep) vals[idx] = wv;   // (each thread rewrites only what it read: the means are in seg_sum)
    }
    if constexpr (kPrep)   // vol == 729 (checked by the host); seg_sum is dead after the loop's barrier inside
        prep_one_kernel(vals, reinterpret_cast<int*>(seg_sum + kz), true, g & 15, prep + (size_t)(g >> 4) * SN_CONV_PREP_BYTES, tid);
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
