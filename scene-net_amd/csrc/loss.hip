// K5 -- training criterion on the prediction grid: one streaming pass over (pred, gt) gives every sum
// the reference's WeightedMSE / FocalTversky / BinaryDice need; a one-block kernel turns them into the loss and the
// coefficients of its gradient; a second streaming pass writes dL/dpred.
//
// Reference (what is restated, not how): core/criterions/w_mse.py:114-151, tversky_loss.py:81-95,
// dice_loss.py:33-51, geneo_loss.py:72-81, :131-161.
//
// Bound: HBM.  Algorithmic bytes per element: forward sizeof(pred) + sizeof(gt); backward the same + sizeof(pred)
// written.  All accumulation is fp64 in a fixed order (per-thread strided, wave shuffle tree, waves in order, parts
// in order), so a loss value is bit-reproducible.
#include "common.h"
#include <type_traits>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBins = SN_LOSS_MAX_BINS;

using bf16 = __bf16;   // SN_BF16: storage only -- every value is widened to fp32 before it is used
template <typename T> struct ComputeOf { using type = T; };
template <> struct ComputeOf<bf16> { using type = float; };

template <typename T>
struct Vec4 {
    T v[4];
};

template <typename T>
__device__ __forceinline__ Vec4<T> load4(const T* p) {
    Vec4<T> r;
    if constexpr (sizeof(T) == 2) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        __builtin_memcpy(&r, &u, 8);
    } else if constexpr (sizeof(T) == 4) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        __builtin_memcpy(&r, &u, 16);
    } else if constexpr (sizeof(T) == 8) {
        const uint4 u0 = reinterpret_cast<const uint4*>(p)[0], u1 = reinterpret_cast<const uint4*>(p)[1];
        __builtin_memcpy(&r.v[0], &u0, 16);
        __builtin_memcpy(&r.v[2], &u1, 16);
    } else {
        const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
        __builtin_memcpy(&r, &u, 4);
    }
    return r;
}

template <typename T>
__device__ __forceinline__ void store4(T* p, const Vec4<T>& r) {
    if constexpr (sizeof(T) == 2) {
        uint2 u;
        __builtin_memcpy(&u, &r, 8);
        *reinterpret_cast<uint2*>(p) = u;
    } else if constexpr (sizeof(T) == 4) {
        uint4 u;
        __builtin_memcpy(&u, &r, 16);
        *reinterpret_cast<uint4*>(p) = u;
    } else {
        uint4 u0, u1;
        __builtin_memcpy(&u0, &r.v[0], 16);
        __builtin_memcpy(&u1, &r.v[2], 16);
        reinterpret_cast<uint4*>(p)[0] = u0;
        reinterpret_cast<uint4*>(p)[1] = u1;
    }
}

// max(log(v), -100) as torch's binary_cross_entropy clamps it, in pred's dtype
template <typename T>
__device__ __forceinline__ T bce_log(T v) {
    if constexpr (sizeof(T) == 8) return fmax(log(v), -100.0);
    else return fmaxf(logf(v), -100.0f);
}

// w_mse.py:122 -- argmin_k |y - ranges[k]|, first minimum; arithmetic in gt's dtype promoted with the fp32 ranges
// (f64 gt: fp64; f32 gt: fp32; byte gt, our extension: fp32).
template <typename GT>
struct BinOf {
    using C = typename std::conditional<std::is_same<GT, double>::value, double, float>::type;
    C r[kMaxBins];
    int H;
    __device__ void init(const float* ranges, int h) {
        H = h;
#pragma unroll
        for (int k = 0; k < kMaxBins; ++k) r[k] = (C)ranges[k < h ? k : h - 1];
    }
    __device__ __forceinline__ int operator()(GT y) const {
        const C yy = (C)y;
        C best = fabs(yy - r[0]);
        int idx = 0;
#pragma unroll
        for (int k = 1; k < kMaxBins; ++k) {
            const C d = fabs(yy - r[k]);
            const bool lt = (k < H) && (d < best);
            best = lt ? d : best;
            idx = lt ? k : idx;
        }
        return idx;
    }
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---------------------------------------------------------------- pass 1: partial statistics
// grid = (parts, B); block `part` owns elements [part*span, (part+1)*span) of sample b (span a multiple of 4).
// kBinary (gt is SN_OCC8, values in {0,1}): two classes, everything in registers (~10 VALU per element).
// Otherwise: per-thread per-bin accumulators live in LDS ([bin][thread], conflict free) -- register accumulators
// would cost a 16-way select per element and make the pass VALU-bound; byte targets find their bin in a 256-entry
// LDS table, float targets by the first-minimum search of the reference.
template <typename GT>
__device__ __forceinline__ int bin_lookup(const BinOf<GT>& bin, const int* lut, GT tv) {
    if constexpr (sizeof(GT) == 1) return lut[tv];
    else return bin(tv);
}

template <typename PT, typename GT, bool kBinary>
__global__ __launch_bounds__(kThreads) void loss_stats_kernel(const PT* __restrict__ pred, const GT* __restrict__ gt,
                                                              long n_per, long span, const float* __restrict__ ranges,
                                                              int H, int terms, double* __restrict__ parts) {
    __shared__ double red[kThreads / 64][3 * kMaxBins + 5];
    __shared__ double sq_l[kBinary ? 1 : kMaxBins * kThreads];
    __shared__ int cnt_l[kBinary ? 1 : kMaxBins * kThreads];
    using CT = typename ComputeOf<PT>::type;   // arithmetic type of pred-precision terms (bf16 storage: fp32)
    using BT = CT;  // per-thread per-bin BCE partials in pred's precision (a few hundred terms each)
    __shared__ BT bce_l[kBinary ? 1 : kMaxBins * kThreads];
    const bool want_bce = (terms & SN_LOSS_WBCE) != 0;
    __shared__ int lut[256];
    const int part = blockIdx.x, b = blockIdx.y, nparts = gridDim.x, tid = threadIdx.x;
    const long lo = (long)part * span, hi = (lo + span < n_per) ? lo + span : n_per;
    const PT* p = pred + (size_t)b * n_per;
    const GT* t = gt + (size_t)b * n_per;
    BinOf<GT> bin;
    bin.init(ranges, H);
    const int lane = tid & 63, wave = tid >> 6;
    const int nstat = 3 * H + 5;
    double s_pt = 0, s_p = 0, s_t = 0, s_pp = 0, s_tt = 0;
    const bool vec = (n_per % 4 == 0);  // the sample base is then 4-element aligned (host checks the pointers)

    if constexpr (kBinary) {
        double sq_all = 0, sq1 = 0, bce_all = 0, bce1 = 0;
        int n_all = 0, n1 = 0;
        auto take = [&](PT pv_, GT tv) {
            const CT pv = (CT)pv_;
            const double pd = (double)pv;
            const bool one = tv != 0;
            const double e = (one ? 1.0 : 0.0) - pd, e2 = e * e;
            n_all += 1;
            n1 += one ? 1 : 0;
            sq_all += e2;
            sq1 += one ? e2 : 0.0;
            s_pt += one ? pd : 0.0;
            s_p += pd;
            s_pp += pd * pd;
            if (want_bce) {  // torch BCELoss: -(t max(log p, -100) + (1 - t) max(log(1 - p), -100)), in pred's dtype
                const CT l = bce_log<CT>(one ? pv : (CT)1 - pv);
                bce_all -= (double)l;
                bce1 -= one ? (double)l : 0.0;
            }
        };
        if (vec) {
            for (long i = lo + 4 * (long)tid; i + 3 < hi; i += 4 * kThreads) {
                const Vec4<PT> pv = load4(p + i);
                const Vec4<GT> tv = load4(t + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) take(pv.v[j], tv.v[j]);
            }
        } else {
            for (long i = lo + tid; i < hi; i += kThreads) take(p[i], t[i]);
        }
        const int b0 = bin((GT)0), b1 = bin((GT)1);
        const double c1 = wave_sum((double)n1), c0 = wave_sum((double)(n_all - n1));
        const double q1 = wave_sum(sq1), q0 = wave_sum(sq_all - sq1);
        const double a = wave_sum(s_pt), c = wave_sum(s_p), e = wave_sum(s_pp);
        const double g1 = wave_sum(bce1), g0 = wave_sum(bce_all - bce1);
        if (lane == 0) {
            for (int j = 0; j < nstat; ++j) red[wave][j] = 0.0;
            red[wave][b0] += c0; red[wave][b1] += c1;
            red[wave][H + b0] += q0; red[wave][H + b1] += q1;
            red[wave][2 * H + 5 + b0] += g0; red[wave][2 * H + 5 + b1] += g1;
            red[wave][2 * H + 0] = a; red[wave][2 * H + 1] = c; red[wave][2 * H + 2] = c1;
            red[wave][2 * H + 3] = e; red[wave][2 * H + 4] = c1;
        }
    } else {
        if constexpr (sizeof(GT) == 1) lut[tid] = bin((GT)tid);
        for (int k = 0; k < H; ++k) sq_l[k * kThreads + tid] = 0.0, cnt_l[k * kThreads + tid] = 0, bce_l[k * kThreads + tid] = (BT)0;
        __syncthreads();
        auto take = [&](PT pv_, GT tv) {
            const CT pv = (CT)pv_;
            const double pd = (double)pv, td = (double)tv;
            const double e = td - pd;
            const int k = bin_lookup(bin, lut, tv) * kThreads + tid;
            sq_l[k] += e * e;
            cnt_l[k] += 1;
            if (want_bce) {
                const CT tt = (CT)tv;
                bce_l[k] -= tt * bce_log<CT>(pv) + ((CT)1 - tt) * bce_log<CT>((CT)1 - pv);
            }
            s_pt += pd * td;
            s_p += pd;
            s_t += td;
            s_pp += pd * pd;
            s_tt += td * td;
        };
        if (vec) {
            for (long i = lo + 4 * (long)tid; i + 3 < hi; i += 4 * kThreads) {
                const Vec4<PT> pv = load4(p + i);
                const Vec4<GT> tv = load4(t + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) take(pv.v[j], tv.v[j]);
            }
        } else {
            for (long i = lo + tid; i < hi; i += kThreads) take(p[i], t[i]);
        }
        for (int j = 0; j < H; ++j) {
            const double a = wave_sum((double)cnt_l[j * kThreads + tid]), c = wave_sum(sq_l[j * kThreads + tid]);
            const double g = wave_sum((double)bce_l[j * kThreads + tid]);
            if (lane == 0) red[wave][j] = a, red[wave][H + j] = c, red[wave][2 * H + 5 + j] = g;
        }
        const double a = wave_sum(s_pt), c = wave_sum(s_p), d = wave_sum(s_t), e = wave_sum(s_pp), f = wave_sum(s_tt);
        if (lane == 0) {
            red[wave][2 * H + 0] = a; red[wave][2 * H + 1] = c; red[wave][2 * H + 2] = d;
            red[wave][2 * H + 3] = e; red[wave][2 * H + 4] = f;
        }
    }
    __syncthreads();
    if (tid < nstat) {
        double s = 0;
        for (int w = 0; w < kThreads / 64; ++w) s += red[w][tid];
        parts[((size_t)b * nparts + part) * nstat + tid] = s;
    }
}

// ---------------------------------------------------------------- penalties over the ~50 scalars
// geneo_loss.py:36-70: value = w * ( sum_{mask>=1} relu(-v) + relu(-(1 - sum_{mask==2} v)) ) and its gradient.  fp32 sums in
// a fixed order (strided partials, then a tree).  A device function: its own launch (param_penalty_kernel) or the opening of
// the criterion's combine launch (sn_criterion_forward).  Called by every thread of the workgroup (barriers inside); the
// first 256 threads work; pl: 2 N floats of LDS.
struct PenaltyArgs {
    const float* P;
    const int8_t* mask;
    int N;
    float w;
    int with_sum;
    float* value;     // [1]
    float* grad;      // [N]
    float* total32;   // [1]: (float)(dense loss) + value -- the criterion's scalar
};

__device__ void param_penalty_body(const float* __restrict__ P, const int8_t* __restrict__ mask, int N, float w, int with_sum,
                                   float* __restrict__ value, float* __restrict__ grad, float* pl) {
    int* ml = reinterpret_cast<int*>(pl + N);
    __shared__ int last_neg_s;
    __shared__ float part_pen[256], part_sum[256];
    const int tid = threadIdx.x;
    if (tid < 256)
        for (int i = tid; i < N; i += 256) pl[i] = P[i], ml[i] = mask[i];
    __syncthreads();
    // per-thread strided partial sums, then thread 0 adds the 256 partials in order (deterministic)
    if (tid < 256) {
        float pen = 0.f, free_sum = 0.f;
        for (int i = tid; i < N; i += 256) {
            if (ml[i] >= 1) pen += fmaxf(-pl[i], 0.f);
            if (ml[i] == 2) free_sum += pl[i];
        }
        part_pen[tid] = pen;
        part_sum[tid] = free_sum;
    }
    __syncthreads();
    if (tid < 64) {                // 64 lanes x 4 partials, xor-tree: fixed order
        float pen = 0.f, free_sum = 0.f;
        for (int k = 0; k < 4; ++k) pen += part_pen[tid * 4 + k], free_sum += part_sum[tid * 4 + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pen += __shfl_xor(pen, o, 64), free_sum += __shfl_xor(free_sum, o, 64);
        if (tid == 0) {
            const float last = 1.f - free_sum; // the frozen coefficient, 1 - sum(others)
            const int last_neg = with_sum && (last < 0.f);
            if (last_neg) pen += -last;
            last_neg_s = last_neg;
            value[0] = w * pen;
        }
    }
    __syncthreads();
    const bool last_neg = last_neg_s != 0;
    if (tid < 256)
        for (int i = tid; i < N; i += 256) {
            float g = 0.f;
            if (ml[i] >= 1 && pl[i] < 0.f) g -= 1.f;  // d relu(-v)/dv
            if (ml[i] == 2 && last_neg) g += 1.f;     // d relu(-(1 - sum))/dv
            grad[i] = w * g;
        }
}

__global__ __launch_bounds__(256) void param_penalty_kernel(const float* __restrict__ P,
                                                            const int8_t* __restrict__ mask, int N, float w,
                                                            int with_sum, float* __restrict__ value,
                                                            float* __restrict__ grad) {
    extern __shared__ float pl[];          // [N] values, then [N] masks
    param_penalty_body(P, mask, N, w, with_sum, value, grad, pl);
}

// ---------------------------------------------------------------- one block: parts -> stats -> loss + coefficients
struct LossCfg {
    int terms;
    double mse_weight, tv_alpha, tv_beta, gamma, tv_smooth, dice_smooth;
};

constexpr int kCombineThreads = 1024;
__global__ __launch_bounds__(kCombineThreads) void loss_combine_kernel(const double* __restrict__ parts, int B, int nparts,
                                                                long n_per, int H, const float* __restrict__ bin_w,
                                                                LossCfg cfg, double* __restrict__ stats,
                                                                double* __restrict__ loss, double* __restrict__ coef,
                                                                float* __restrict__ loss32, PenaltyArgs pen) {
    if (pen.P) {   // (the criterion's penalties open the launch: sn_criterion_forward)
        extern __shared__ float pen_lds[];
        param_penalty_body(pen.P, pen.mask, pen.N, pen.w, pen.with_sum, pen.value, pen.grad, pen_lds);
    }
    __shared__ double tot[3 * kMaxBins + 5];
    __shared__ double dice_part[kCombineThreads];
    const int nstat = 3 * H + 5;
    for (int i = threadIdx.x; i < B * nstat; i += kCombineThreads) {
        const int b = i / nstat, j = i % nstat;
        const double* src = parts + (size_t)b * nparts * nstat + j;
        double s = 0;
        int q = 0;
        for (; q + 32 <= nparts; q += 32) {   // 32 independent loads in flight, summed in order
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = src[(size_t)(q + u) * nstat];
#pragma unroll
            for (int u = 0; u < 32; ++u) s += v[u];
        }
        for (; q + 8 <= nparts; q += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(q + u) * nstat];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; q < nparts; ++q) s += src[(size_t)q * nstat];
        stats[i] = s;
    }
    __syncthreads();  // stats[] written by this block is visible to it after the barrier
    if (threadIdx.x < nstat) {
        double s = 0;
        int b = 0;
        for (; b + 8 <= B; b += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = stats[(size_t)(b + u) * nstat + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < B; ++b) s += stats[(size_t)b * nstat + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __shared__ double sh_c[kMaxBins], sh_e[kMaxBins], sh_t[2], sh_w[kMaxBins];
    if (threadIdx.x < kMaxBins) sh_w[threadIdx.x] = threadIdx.x < H ? (double)bin_w[threadIdx.x] : 0.0;
    // dice, per sample (dice_loss.py:38-41), mean over the batch; gradient coefficients per sample (kept in registers
    // until the Tversky coefficients are known: every thread then writes its samples' coefficients once)
    double dsum = 0;
    for (int b = threadIdx.x; b < B; b += kCombineThreads)
        if (cfg.terms & SN_LOSS_DICE) {
            const double* s = stats + (size_t)b * nstat + 2 * H;
            dsum += 1.0 - (s[0] + cfg.dice_smooth) / (s[3] + s[4] + cfg.dice_smooth);
        }
    dice_part[threadIdx.x] = dsum;
    __syncthreads();
    for (int o = kCombineThreads / 2; o > 0; o >>= 1) {   // fixed tree
        if ((int)threadIdx.x < o) dice_part[threadIdx.x] += dice_part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)B * (double)n_per;
        double wmse = 0, focal = 0, wbce = 0;
        double mean_w = 0;
        for (int k = 0; k < H; ++k) mean_w += tot[k] * sh_w[k];
        mean_w /= n;
        for (int k = 0; k < kMaxBins; ++k) {
            double c = 0, e = 0;
            if (k < H) {
                const double wk = sh_w[k] / mean_w;                // w_mse.py:144
                if (cfg.terms & SN_LOSS_WMSE) {
                    wmse += wk * tot[H + k];
                    c = 2.0 * cfg.mse_weight * wk / n;             // d/dp of mean(mse_weight w (t - p)^2)
                }
                if (cfg.terms & SN_LOSS_WBCE) {                    // mean(w * bce), dice_loss.py:77-80
                    wbce += wk * tot[2 * H + 5 + k];
                    e = wk / n;
                }
            }
            sh_c[k] = c;
            sh_e[k] = e;
        }
        wbce = (cfg.terms & SN_LOSS_WBCE) ? wbce / n : 0.0;
        wmse = (cfg.terms & SN_LOSS_WMSE) ? cfg.mse_weight * wmse / n : 0.0;
        double tA = 0, tB = 0;
        if (cfg.terms & SN_LOSS_FOCAL_TVERSKY) {                   // tversky_loss.py:86-93
            const double TP = tot[2 * H], FP = tot[2 * H + 1] - TP, FN = tot[2 * H + 2] - TP;
            const double a = cfg.tv_alpha, be = cfg.tv_beta, sm = cfg.tv_smooth, g = cfg.gamma;
            const double N = TP + sm, D = TP + a * FP + be * FN + sm;
            const double T = N / D, u = 1.0 - T;
            focal = pow(u, g);
            const double dF = -g * pow(u, g - 1.0);                // dFocal/dT
            tA = dF * (D - N * (1.0 - a - be)) / (D * D);          // dT/dp_i = (t_i D - N (t_i (1-a-b) + a)) / D^2
            tB = dF * (-N * a) / (D * D);
        }
        sh_t[0] = tA;
        sh_t[1] = tB;
        const double dice = (cfg.terms & SN_LOSS_DICE) ? dice_part[0] / B : 0.0;
        loss[0] = wmse + focal + dice + wbce;
        loss[1] = wmse;
        loss[2] = focal;
        loss[3] = dice;
        loss[4] = wbce;
        if (loss32) {   // (the same five numbers rounded once: what a float32 criterion returns, without a cast launch)
#pragma unroll
            for (int k = 0; k < 5; ++k) loss32[k] = (float)loss[k];
        }
        // (thread 0 wrote pen.value itself: the float32 sum GENEO_Loss.forward formed with a launch of its own)
        if (pen.P) pen.total32[0] = (float)loss[0] + pen.value[0];
    }
    __syncthreads();
    if (threadIdx.x < kMaxBins) {
        coef[threadIdx.x] = sh_c[threadIdx.x];
        coef[kMaxBins + threadIdx.x] = sh_e[threadIdx.x];
    }
    for (int b = threadIdx.x; b < B; b += kCombineThreads) {
        double A = 0, C = 0;
        if (cfg.terms & SN_LOSS_DICE) {
            const double* s = stats + (size_t)b * nstat + 2 * H;
            const double num = s[0] + cfg.dice_smooth, den = s[3] + s[4] + cfg.dice_smooth;
            A = -1.0 / den / B;               // d(1 - num/den)/dp_i = -(t_i den - num 2 p_i)/den^2
            C = 2.0 * num / (den * den) / B;
        }
        coef[2 * kMaxBins + 3 * b + 0] = A + sh_t[0];
        coef[2 * kMaxBins + 3 * b + 1] = sh_t[1];
        coef[2 * kMaxBins + 3 * b + 2] = C;
    }
}

// ---------------------------------------------------------------- pass 2: dL/dpred
template <typename PT, typename GT, bool kBinary>
__global__ __launch_bounds__(kThreads) void loss_grad_kernel(const PT* __restrict__ pred, const GT* __restrict__ gt,
                                                             long n_per, long span, const float* __restrict__ ranges,
                                                             int H, const double* __restrict__ coef,
                                                             const double* __restrict__ upstream,
                                                             const float* __restrict__ upstream32,
                                                             PT* __restrict__ grad, int B, const float* __restrict__ pen_grad,
                                                             int pen_n, float* __restrict__ pen_out) {
    if ((int)blockIdx.y >= B) {   // the rider row (sn_criterion_backward): the penalties' gradient times the upstream scalar
        if (blockIdx.x == 0) {
            const float upf = upstream ? (float)*upstream : (upstream32 ? *upstream32 : 1.0f);
            for (int i = threadIdx.x; i < pen_n; i += kThreads) pen_out[i] = pen_grad[i] * upf;
        }
        return;
    }
    using C = typename ComputeOf<PT>::type;  // gradient arithmetic in pred's dtype (bf16 storage: fp32)
    __shared__ C ck[kMaxBins], ek[kMaxBins];
    __shared__ int lut[256];
    const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const long lo = (long)part * span, hi = (lo + span < n_per) ? lo + span : n_per;
    const PT* p = pred + (size_t)b * n_per;
    const GT* t = gt + (size_t)b * n_per;
    PT* g = grad + (size_t)b * n_per;
    BinOf<GT> bin;
    bin.init(ranges, H);
    const double up = upstream ? *upstream : (upstream32 ? (double)*upstream32 : 1.0);
    const C A = (C)(coef[2 * kMaxBins + 3 * b] * up), Bc = (C)(coef[2 * kMaxBins + 3 * b + 1] * up),
            Cc = (C)(coef[2 * kMaxBins + 3 * b + 2] * up);
    C c0 = 0, c1 = 0, e0 = 0, e1 = 0;
    bool any_e = false;   // weighted-BCE term present (block-uniform)
    for (int k = 0; k < kMaxBins; ++k) any_e |= coef[kMaxBins + k] != 0.0;
    if constexpr (kBinary) {
        c0 = (C)(coef[bin((GT)0)] * up);
        c1 = (C)(coef[bin((GT)1)] * up);
        e0 = (C)(coef[kMaxBins + bin((GT)0)] * up);
        e1 = (C)(coef[kMaxBins + bin((GT)1)] * up);
    } else {
        if (tid < kMaxBins) ck[tid] = (C)(coef[tid] * up), ek[tid] = (C)(coef[kMaxBins + tid] * up);
        if constexpr (sizeof(GT) == 1) lut[tid] = bin((GT)tid);
        __syncthreads();
    }
    auto one = [&](PT pv, GT tv) -> PT {
        const C pc = (C)pv;
        // d bce / dp as torch computes it: (p - t) / max((1 - p) p, EPSILON), EPSILON = float(1e-12) in ATen
        const C kBceEps = (C)1e-12f;
        if constexpr (kBinary) {
            // t in {0,1}:  c (p - t) + A t + B + C p
            C g = (tv != 0) ? (c1 * (pc - (C)1) + A + Bc + Cc * pc) : (c0 * pc + Bc + Cc * pc);
            if (any_e) {
                const C den = ((C)1 - pc) * pc;
                g += ((tv != 0) ? e1 * (pc - (C)1) : e0 * pc) / (den > kBceEps ? den : kBceEps);
            }
            return (PT)g;
        } else {
            const int k = bin_lookup(bin, lut, tv);
            const C c = ck[k], tc = (C)tv;
            C g = c * (pc - tc) + A * tc + Bc + Cc * pc;
            if (any_e) {
                const C den = ((C)1 - pc) * pc;
                g += ek[k] * (pc - tc) / (den > kBceEps ? den : kBceEps);
            }
            return (PT)g;
        }
    };
    if (n_per % 4 == 0) {
        for (long i = lo + 4 * (long)tid; i + 3 < hi; i += 4 * kThreads) {
            const Vec4<PT> pv = load4(p + i);
            const Vec4<GT> tv = load4(t + i);
            Vec4<PT> r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r.v[j] = one(pv.v[j], tv.v[j]);
            store4(g + i, r);
        }
    } else {
        for (long i = lo + tid; i < hi; i += kThreads) g[i] = one(p[i], t[i]);
    }
}

int check_common(const char* fn, const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                 const float* ranges, int H) {
    if (!pred || !gt || !ranges) return sn::fail(SN_ERR_INVALID_ARG, "%s: null pointer", fn);
    if (B <= 0 || n_per <= 0) return sn::fail(SN_ERR_INVALID_ARG, "%s: B and n_per must be positive", fn);
    if (H < 1 || H > kMaxBins) return sn::fail(SN_ERR_UNSUPPORTED, "%s: 1 <= H <= %d bins", fn, kMaxBins);
    if (pred_dtype != SN_F32 && pred_dtype != SN_F64 && pred_dtype != SN_BF16)
        return sn::fail(SN_ERR_INVALID_ARG, "%s: pred must be SN_F32, SN_F64 or SN_BF16", fn);
    if (pred_dtype == SN_BF16 && gt_dtype == SN_F64)
        return sn::fail(SN_ERR_UNSUPPORTED, "%s: bf16 predictions take SN_F32 / SN_U8 / SN_OCC8 targets", fn);
    if (gt_dtype != SN_F32 && gt_dtype != SN_F64 && gt_dtype != SN_U8 && gt_dtype != SN_OCC8)
        return sn::fail(SN_ERR_INVALID_ARG, "%s: bad gt dtype %d", fn, gt_dtype);
    const size_t pa = pred_dtype == SN_BF16 ? 8 : 16, ga = (gt_dtype == SN_F64 || gt_dtype == SN_F32) ? 16 : 4;
    if (n_per % 4 == 0 && (((uintptr_t)pred % pa) || ((uintptr_t)gt % ga)))
        return sn::fail(SN_ERR_INVALID_ARG, "%s: pred / gt must be 16-byte (bf16 pred: 8-byte, byte gt: 4-byte) aligned", fn);
    return SN_OK;
}

// span per part: a multiple of 4 so that the vector loop of every part starts aligned
long span_of(int64_t n_per, int nparts) {
    long s = (long)((n_per + nparts - 1) / nparts);
    return (s + 3) / 4 * 4;
}

}  // namespace

#define SN_LOSS_DISPATCH(KERNEL)                                                                                  \
    do {                                                                                                          \
        if (pred_dtype == SN_BF16) {                                                                              \
            if (gt_dtype == SN_F32) KERNEL(bf16, float, false);                                                   \
            else if (gt_dtype == SN_OCC8) KERNEL(bf16, uint8_t, true);                                            \
            else KERNEL(bf16, uint8_t, false);                                                                    \
        } else if (pred_dtype == SN_F32) {                                                                        \
            if (gt_dtype == SN_F32) KERNEL(float, float, false);                                                  \
            else if (gt_dtype == SN_F64) KERNEL(float, double, false);                                            \
            else if (gt_dtype == SN_OCC8) KERNEL(float, uint8_t, true);                                           \
            else KERNEL(float, uint8_t, false);                                                                   \
        } else {                                                                                                  \
            if (gt_dtype == SN_F32) KERNEL(double, float, false);                                                 \
            else if (gt_dtype == SN_F64) KERNEL(double, double, false);                                           \
            else if (gt_dtype == SN_OCC8) KERNEL(double, uint8_t, true);                                          \
            else KERNEL(double, uint8_t, false);                                                                  \
        }                                                                                                         \
    } while (0)

static int loss_forward_impl(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                             const float* ranges, const float* bin_w, int H, int terms, double mse_weight,
                             double tversky_alpha, double tversky_beta, double focal_gamma, double tversky_smooth,
                             double dice_smooth, double* parts_ws, double* stats, double* loss, float* loss_f32,
                             double* coef, const PenaltyArgs& pen, sn_stream_t stream) {
    if (int rc = check_common("sn_loss_forward", pred, pred_dtype, gt, gt_dtype, B, n_per, ranges, H)) return rc;
    if (!bin_w || !parts_ws || !stats || !loss || !coef)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_loss_forward: null pointer");
    if (terms <= 0 || (terms & ~(SN_LOSS_WMSE | SN_LOSS_FOCAL_TVERSKY | SN_LOSS_DICE | SN_LOSS_WBCE)))
        return sn::fail(SN_ERR_INVALID_ARG, "sn_loss_forward: bad terms mask %d", terms);
    if (B > 65535) return sn::fail(SN_ERR_UNSUPPORTED, "sn_loss_forward: B <= 65535");
    hipStream_t s = sn::as_stream(stream);
    const int nparts = SN_LOSS_PARTS(n_per);
    const long span = span_of(n_per, nparts);
#define SN_STATS(PT, GT, BIN)                                                                                     \
    hipLaunchKernelGGL((loss_stats_kernel<PT, GT, BIN>), dim3(nparts, B), dim3(kThreads), 0, s, (const PT*)pred,  \
                       (const GT*)gt, (long)n_per, span, ranges, H, terms, parts_ws)
    SN_LOSS_DISPATCH(SN_STATS);
#undef SN_STATS
    if (int rc = sn::check_launch("sn_loss_forward(stats)")) return rc;
    LossCfg cfg{terms, mse_weight, tversky_alpha, tversky_beta, focal_gamma, tversky_smooth, dice_smooth};
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(kCombineThreads), pen.P ? (size_t)pen.N * 8 : 0, s, parts_ws, B,
                       nparts, (long)n_per, H, bin_w, cfg, stats, loss, coef, loss_f32, pen);
    return sn::check_launch("sn_loss_forward(combine)");
}

extern "C" int sn_loss_forward_m(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                                 const float* ranges, const float* bin_w, int H, int terms, double mse_weight,
                                 double tversky_alpha, double tversky_beta, double focal_gamma, double tversky_smooth,
                                 double dice_smooth, double* parts_ws, double* stats, double* loss, float* loss_f32,
                                 double* coef, sn_stream_t stream) {
    return loss_forward_impl(pred, pred_dtype, gt, gt_dtype, B, n_per, ranges, bin_w, H, terms, mse_weight, tversky_alpha,
                             tversky_beta, focal_gamma, tversky_smooth, dice_smooth, parts_ws, stats, loss, loss_f32, coef,
                             PenaltyArgs{nullptr, nullptr, 0, 0.f, 0, nullptr, nullptr, nullptr}, stream);
}

// The whole criterion of a GENEO_Loss family member in the two launches of sn_loss_forward: the penalties over the packed
// parameters (sn_param_penalty's arithmetic) open the combine launch, which also writes total_f32 = (float)loss[0] + penalty
// -- the float32 sum the reference forms at geneo_loss.py:86-91 / 155-161 -- so a replayed training step has no penalty
// launch and no one-element add.
extern "C" int sn_criterion_forward(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                                    const float* ranges, const float* bin_w, int H, int terms, double mse_weight,
                                    double tversky_alpha, double tversky_beta, double focal_gamma, double tversky_smooth,
                                    double dice_smooth, double* parts_ws, double* stats, double* loss, float* loss_f32,
                                    double* coef, const float* P, const int8_t* mask, int N, float weight, int with_sum,
                                    float* pen_value, float* pen_grad, float* total_f32, sn_stream_t stream) {
    if (!P || !mask || !pen_value || !pen_grad || !total_f32)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_criterion_forward: null pointer");
    if (N <= 0 || N > 8192) return sn::fail(SN_ERR_UNSUPPORTED, "sn_criterion_forward: 1 <= N <= 8192");
    return loss_forward_impl(pred, pred_dtype, gt, gt_dtype, B, n_per, ranges, bin_w, H, terms, mse_weight, tversky_alpha,
                             tversky_beta, focal_gamma, tversky_smooth, dice_smooth, parts_ws, stats, loss, loss_f32, coef,
                             PenaltyArgs{P, mask, N, weight, with_sum, pen_value, pen_grad, total_f32}, stream);
}

extern "C" int sn_loss_forward(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                               const float* ranges, const float* bin_w, int H, int terms, double mse_weight,
                               double tversky_alpha, double tversky_beta, double focal_gamma, double tversky_smooth,
                               double dice_smooth, double* parts_ws, double* stats, double* loss, double* coef,
                               sn_stream_t stream) {
    return sn_loss_forward_m(pred, pred_dtype, gt, gt_dtype, B, n_per, ranges, bin_w, H, terms, mse_weight, tversky_alpha,
                             tversky_beta, focal_gamma, tversky_smooth, dice_smooth, parts_ws, stats, loss, nullptr, coef,
                             stream);
}

extern "C" int sn_param_penalty(const float* P, const int8_t* mask, int N, float weight, int with_sum, float* value,
                                float* grad, sn_stream_t stream) {
    if (!P || !mask || !value || !grad) return sn::fail(SN_ERR_INVALID_ARG, "sn_param_penalty: null pointer");
    if (N <= 0) return sn::fail(SN_ERR_INVALID_ARG, "sn_param_penalty: N must be positive");
    if (N > 8192) return sn::fail(SN_ERR_UNSUPPORTED, "sn_param_penalty: N <= 8192");
    hipLaunchKernelGGL(param_penalty_kernel, dim3(1), dim3(256), (size_t)N * 8, sn::as_stream(stream), P, mask, N,
                       weight, with_sum, value, grad);
    return sn::check_launch("sn_param_penalty");
}

static int loss_backward_impl(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                              const float* ranges, int H, const double* coef, const void* upstream_v, int up_dtype,
                              void* grad_pred, const float* pen_grad, int pen_n, float* pen_out, sn_stream_t stream) {
    if (int rc = check_common("sn_loss_backward", pred, pred_dtype, gt, gt_dtype, B, n_per, ranges, H)) return rc;
    if (upstream_v && up_dtype != SN_F64 && up_dtype != SN_F32)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_loss_backward_u: up_dtype %d (SN_F64 | SN_F32)", up_dtype);
    const double* upstream = (upstream_v && up_dtype == SN_F64) ? static_cast<const double*>(upstream_v) : nullptr;
    const float* upstream32 = (upstream_v && up_dtype == SN_F32) ? static_cast<const float*>(upstream_v) : nullptr;
    if (!coef || !grad_pred) return sn::fail(SN_ERR_INVALID_ARG, "sn_loss_backward: null pointer");
    if (n_per % 4 == 0 && ((uintptr_t)grad_pred % (pred_dtype == SN_BF16 ? 8 : 16)))
        return sn::fail(SN_ERR_INVALID_ARG, "sn_loss_backward: grad_pred must be 16-byte (bf16: 8-byte) aligned");
    if (B > 65535) return sn::fail(SN_ERR_UNSUPPORTED, "sn_loss_backward: B <= 65535");
    hipStream_t s = sn::as_stream(stream);
    // the gradient pass has no reduction tail: 8192-element spans ([measured] 13.8 us; 14.6 us at the statistics
    // pass's 16384, which is the better size there: 28.4 -> 25.2 us for statistics + combine)
    const int nparts = n_per <= 8192 ? 1 : (n_per >= 8192L * 512 ? 512 : (int)((n_per + 8191) / 8192));
    const long span = span_of(n_per, nparts);
#define SN_GRAD(PT, GT, BIN)                                                                                      \
    hipLaunchKernelGGL((loss_grad_kernel<PT, GT, BIN>), dim3(nparts, B + (pen_grad ? 1 : 0)), dim3(kThreads), 0, s,  \
                       (const PT*)pred, (const GT*)gt, (long)n_per, span, ranges, H, coef, upstream, upstream32,     \
                       (PT*)grad_pred, B, pen_grad, pen_n, pen_out)
    SN_LOSS_DISPATCH(SN_GRAD);
#undef SN_GRAD
    return sn::check_launch("sn_loss_backward");
}

extern "C" int sn_loss_backward_u(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                                  const float* ranges, int H, const double* coef, const void* upstream_v, int up_dtype,
                                  void* grad_pred, sn_stream_t stream) {
    return loss_backward_impl(pred, pred_dtype, gt, gt_dtype, B, n_per, ranges, H, coef, upstream_v, up_dtype, grad_pred,
                              nullptr, 0, nullptr, stream);
}

// sn_loss_backward_u with the penalties' gradient riding in the same launch: pen_out [N] = pen_grad [N] (of
// sn_criterion_forward) times the upstream scalar -- what autograd did with a multiply launch of its own.
extern "C" int sn_criterion_backward(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                                     const float* ranges, int H, const double* coef, const void* upstream_v, int up_dtype,
                                     void* grad_pred, const float* pen_grad, int N, float* pen_out, sn_stream_t stream) {
    if (!pen_grad || !pen_out || N <= 0) return sn::fail(SN_ERR_INVALID_ARG, "sn_criterion_backward: null pointer / N");
    return loss_backward_impl(pred, pred_dtype, gt, gt_dtype, B, n_per, ranges, H, coef, upstream_v, up_dtype, grad_pred,
                              pen_grad, N, pen_out, stream);
}

extern "C" int sn_loss_backward(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                                const float* ranges, int H, const double* coef, const double* upstream,
                                void* grad_pred, sn_stream_t stream) {
    return sn_loss_backward_u(pred, pred_dtype, gt, gt_dtype, B, n_per, ranges, H, coef, upstream, SN_F64, grad_pred, stream);
}
