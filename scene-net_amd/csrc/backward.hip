// Backward of the GENEO forward path (reference: autograd through SceneNet.forward,
// core/models/SCENE_Net.py:322-339, and the generator graphs of core/models/geneos/*.py).
//
// With s = sum_g lambda_g conv3d(x, K_g) and out = relu(tanh(s)), linearity collapses the whole backward onto ONE
// correlation:   C[t] = sum_{b,v} delta[b,v] * x[b, v + t - p],   delta = dL/dout * (out > 0) * (1 - out^2)
//   dL/dK_g[t]   = lambda_g * C[t]
//   dL/dlambda_g = <K_g, C>
// (729 numbers instead of a [729 x B*V] x [B*V x 16] weight-gradient GEMM), and the generator Jacobians are
// analytic: every kernel is W = P(f(theta)) (+ const), P = subtraction of a slice / global mean, which is a
// symmetric projection, so dL/dtheta = < P(dW), df/dtheta >.
#include "common.h"

namespace sn {  // corr.hip
int corr_mfma_supported(int kz, int kx, int ky);
int corr_mfma_rows(int B, int Z, int X, int Y, int kz, int kx, int ky);
size_t corr_sparse_ws_bytes(int x_dtype, int B, int Z, int X, int Y, int kz, int kx, int ky);
int corr_sparse_launch(const void* x, const void* gout, const void* out, int g_dtype, int B, int Z, int X, int Y, int kz,
                       int kx, int ky, void* ws, float* C, hipStream_t s);
int corr_mfma_launch(const void* x, int x_dtype, const void* gout, const void* out, int g_dtype, int B, int Z, int X,
                     int Y, int kz, int kx, int ky, float* partial_ws, float* C, hipStream_t s);
}  // namespace sn

namespace {

constexpr int kThreads = 512;
constexpr int TZ = 8, TX = 8, TY = 64;   // delta tile per workgroup
constexpr float kEps = 1e-8f;
constexpr float kPi = 3.14159274f;

template <typename T>
__device__ __forceinline__ float as_float(const T* p, size_t i) { return (float)p[i]; }

// ---------------------------------------------------------------- C[t] partials
// grid = tiles; LDS: x halo [TZ+kz-1][TX+kx-1][YP] f32 and delta [TZ][TX][TY] f32.  Thread -> tap(s); loop over the
// tile's voxels, skipping delta == 0 (block-uniform branch: delta is read by every thread at the same address).
template <typename XT>
__global__ __launch_bounds__(kThreads) void corr_partial_kernel(const XT* __restrict__ x,
                                                                const float* __restrict__ gout,
                                                                const float* __restrict__ out, int B, int Z, int X,
                                                                int Y, int kz, int kx, int ky, int nzt, int nxt,
                                                                int nyt, float* __restrict__ partial) {
    extern __shared__ float lds[];
    const int ZP = TZ + kz - 1, XP = TX + kx - 1, YP = TY + ky - 1;
    float* xs = lds;                         // [ZP][XP][YP]
    float* ds = lds + ZP * XP * YP;          // [TZ][TX][TY]
    const int ntaps = kz * kx * ky;
    int bid = blockIdx.x;
    const int y0 = (bid % nyt) * TY; bid /= nyt;
    const int x0 = (bid % nxt) * TX; bid /= nxt;
    const int z0 = (bid % nzt) * TZ; bid /= nzt;
    const int b = bid;
    const int pz = (kz - 1) / 2, px = (kx - 1) / 2, py = (ky - 1) / 2;
    const size_t V = (size_t)Z * X * Y;
    for (int i = threadIdx.x; i < ZP * XP * YP; i += kThreads) {
        const int c = i % YP, r = i / YP, xx = r % XP, zz = r / XP;
        const int gz = z0 - pz + zz, gx = x0 - px + xx, gy = y0 - py + c;
        float v = 0.f;
        if (gz >= 0 && gz < Z && gx >= 0 && gx < X && gy >= 0 && gy < Y)
            v = as_float(x, (size_t)b * V + ((size_t)gz * X + gx) * Y + gy);
        xs[i] = v;
    }
    int any = 0;
    for (int i = threadIdx.x; i < TZ * TX * TY; i += kThreads) {
        const int c = i % TY, r = i / TY, xx = r % TX, zz = r / TX;
        const int gz = z0 + zz, gx = x0 + xx, gy = y0 + c;
        float d = 0.f;
        if (gz < Z && gx < X && gy < Y) {
            const size_t idx = (size_t)b * V + ((size_t)gz * X + gx) * Y + gy;
            d = gout[idx];
            if (out) {
                const float o = out[idx];
                d = (o > 0.f) ? d * (1.f - o * o) : 0.f;  // d relu(tanh(s)) / ds
            }
        }
        ds[i] = d;
        any |= (d != 0.f);
    }
    __syncthreads();
    float* prow = partial + (size_t)blockIdx.x * ntaps;
    if (!__syncthreads_or(any)) {
        for (int t = threadIdx.x; t < ntaps; t += kThreads) prow[t] = 0.f;
        return;
    }
    for (int t = threadIdx.x; t < ntaps; t += kThreads) {
        const int dy = t % ky, dx = (t / ky) % kx, dz = t / (ky * kx);
        const float* xt = xs + (dz * XP + dx) * YP + dy;
        float acc = 0.f;
        for (int zz = 0; zz < TZ; ++zz)
            for (int xx = 0; xx < TX; ++xx) {
                const float* dr = ds + (zz * TX + xx) * TY;
                const float* xr = xt + (zz * XP + xx) * YP;
#pragma unroll 8
                for (int c = 0; c < TY; ++c) {
                    const float d = dr[c];
                    if (d != 0.f) acc = fmaf(d, xr[c], acc);
                }
            }
        prow[t] = acc;
    }
}

// C[t] = sum over blocks of partial[blk][t], fixed order (bit-reproducible)
__global__ void corr_reduce_kernel(const float* __restrict__ partial, int nblk, int ntaps, float* __restrict__ C) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntaps) return;
    float acc = 0.f;
    for (int k = 0; k < nblk; ++k) acc += partial[(size_t)k * ntaps + t];
    C[t] = acc;
}

// ---------------------------------------------------------------- generator Jacobians
// a * b rounded to fp32 before anything is added to it (HIP's __fmul_rn is a plain product that hipcc contracts into an
// fma with a following add)
__device__ __forceinline__ float mul_rounded(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < blockDim.x / 64; ++w) s += red[w];
    return s;
}

// one workgroup per GENEO: dparams[g][slot] = < P(dW_g), df/dslot >  (+ the neg_factor constant terms)
// Fused form (sn_geneo_backward): dW_g = lambda_g C is formed on the fly from the correlation C, and the workgroup
// also writes dL/dlambda_g - dL/dlambda_last, dL/dlambda_g = <K_g, C> (the frozen coefficient 1 - sum(others) hands
// its gradient, negated, to every other one: SCENE_Net.py:331).
__global__ __launch_bounds__(256) void geneo_bank_bwd_kernel(const float* __restrict__ params,
                                                             const int32_t* __restrict__ kinds, int kz, int kx,
                                                             int ky, const float* __restrict__ dW,
                                                             float* __restrict__ dparams,
                                                             const float* __restrict__ corr,
                                                             const float* __restrict__ lambdas,
                                                             const float* __restrict__ bank, int last,
                                                             float* __restrict__ dlam) {
    extern __shared__ float lds[];
    __shared__ float red[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int nfloor = kx * ky, vol = kz * nfloor;
    float* dwc = lds;            // [vol]  P(dW)
    float* seg = lds + vol;      // [kz]
    const float* p = params + (size_t)g * SN_NPARAM;
    const int kind = kinds[g];
    const bool is_neg = (kind == SN_GENEO_NEG || kind == SN_GENEO_NEG_V1);
    const bool is_cone = (kind == SN_GENEO_CONE || kind == SN_GENEO_CONE_V1);
    const bool v1 = (kind >= SN_GENEO_CY_V1);
    const float radius = p[SN_P_RADIUS], sigma = p[SN_P_SIGMA];
    const float cx = (kx - 1) * 0.5f, cy = (ky - 1) * 0.5f, cz = (kz - 1) * 0.5f;
    const float* dw = corr ? corr : dW + (size_t)g * vol;
    const float dws = corr ? lambdas[g] : 1.0f;   // fused: dW_g[i] = lambda_g * C[i] (the product torch formed before)

    // P(dW): subtract the slice mean (cy / cone) or the global mean (neg)
    const int nseg = is_neg ? 1 : kz, seg_len = is_neg ? vol : nfloor;
    for (int s = tid >> 6; s < nseg; s += 4) {
        float a = 0.f;
        for (int i = tid & 63; i < seg_len; i += 64) a += mul_rounded(dws, dw[s * seg_len + i]);   // the rounded product, as torch's lambda * C was
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((tid & 63) == 0) seg[s] = a;
    }
    __syncthreads();
    for (int i = tid; i < vol; i += 256) dwc[i] = mul_rounded(dws, dw[i]) - seg[is_neg ? 0 : i / nfloor] / (float)seg_len;
    __syncthreads();
    const float sum_dw = is_neg ? seg[0] : 0.f;

    int hc = 0;
    float cr = 0.f, inc = 0.f, T = 0.f, nf = 0.f;
    bool inc_open = false;
    if (is_cone) {
        hc = min(max((int)p[SN_P_APEX], 0), kz);
        cr = p[SN_P_CONE_RADIUS];
        inc = p[SN_P_CONE_INC];
        inc_open = (inc >= 0.0f && inc <= 0.499f);  // torch.clamp passes the gradient inside [min, max]
        T = tanf(fminf(fmaxf(inc, 0.0f), 0.499f) * kPi);
    }
    if (is_neg) nf = p[SN_P_NEG_FACTOR];

    float g_radius = 0.f, g_sigma = 0.f, g_cr = 0.f, g_inc = 0.f, g_nf = 0.f;
    for (int idx = tid; idx < vol; idx += 256) {
        const float w = dwc[idx];
        float r2;
        int z = 0;
        if (is_neg) {
            const float dz = (float)(idx % kz) - cz, dx = (float)((idx / kz) % kx) - cx,
                        dy = (float)(idx / (kz * kx)) - cy;
            r2 = dx * dx + dy * dy + dz * dz;
        } else {
            z = idx / nfloor;
            const int n = idx - z * nfloor;
            const float dx = (float)(n % kx) - cx, dy = (float)(n / kx) - cy;
            r2 = dx * dx + dy * dy;
        }
        const bool cone_slice = is_cone && (z < kz - hc);
        if (!v1) {
            // f = sig * E, E = exp(r^4 c), c = -1 / (2 a^2), a = rad + eps:  df/dsig = E,  df/drad = sig E r^4 / a^3
            float rad = radius;
            if (kind == SN_GENEO_CONE && cone_slice) rad = cr * (float)z * T;
            const float a = rad + kEps, r4 = r2 * r2;
            const float E = expf(r4 * (-1.0f / (2.0f * a * a)));
            const float dfdrad = sigma * E * r4 / (a * a * a);
            if (kind == SN_GENEO_NEG) {  // f = -nf * sig * E
                g_sigma += w * (-nf * E);
                g_radius += w * (-nf * dfdrad);
                g_nf += w * (-sigma * E);
            } else {
                g_sigma += w * E;
                if (cone_slice) {
                    g_cr += w * dfdrad * ((float)z * T);
                    if (inc_open) g_inc += w * dfdrad * (cr * (float)z * kPi * (1.0f + T * T));
                } else {
                    g_radius += w * dfdrad;
                }
            }
        } else {
            // f = exp(q^2 c), q = r^2 - rad^2, c = -1 / (2 sig^2):  df/drad = -4 rad q c f,  df/dsig = f q^2 / sig^3
            float sig = sigma, dsig_dcr = 0.f, dsig_dinc = 0.f;
            if (kind == SN_GENEO_CONE_V1 && cone_slice) {
                const float h = (float)(2 + (kz - hc - 1 - z));
                const float arg = inc * kPi / h;
                sig = cr * sinf(arg);
                dsig_dcr = sinf(arg);
                dsig_dinc = cr * cosf(arg) * kPi / h;
            }
            const float q = r2 - radius * radius, c = -1.0f / (2.0f * sig * sig);
            const float f = expf(q * q * c);
            const float dfdrad = -4.0f * radius * q * c * f;
            const float dfdsig = f * q * q / (sig * sig * sig);
            g_radius += w * dfdrad;
            if (kind == SN_GENEO_CONE_V1 && cone_slice) {
                g_cr += w * dfdsig * dsig_dcr;
                g_inc += w * dfdsig * dsig_dinc;
            } else {
                g_sigma += w * dfdsig;
            }
        }
    }
    g_radius = block_sum(g_radius, red);
    g_sigma = block_sum(g_sigma, red);
    g_cr = block_sum(g_cr, red);
    g_inc = block_sum(g_inc, red);
    g_nf = block_sum(g_nf, red);
    if (tid == 0) {
        float* o = dparams + (size_t)g * SN_NPARAM;
        for (int k = 0; k < SN_NPARAM; ++k) o[k] = 0.f;
        o[SN_P_RADIUS] = g_radius;
        o[SN_P_SIGMA] = g_sigma;
        if (is_cone) {
            o[SN_P_CONE_RADIUS] = g_cr;
            o[SN_P_CONE_INC] = g_inc;  // apex: truncated to an index, no gradient (non-trainable, arrow.py:134)
        }
        if (kind == SN_GENEO_NEG) o[SN_P_NEG_FACTOR] = g_nf - sum_dw / (float)vol;  // W = P(f) - nf / vol
        if (kind == SN_GENEO_NEG_V1) o[SN_P_NEG_FACTOR] = -sum_dw;                  // W = P(f) - nf
    }
    if (corr && dlam) {
        float a = 0.f, b = 0.f;
        const float* kg = bank + (size_t)g * vol;
        const float* kl = bank + (size_t)last * vol;
        for (int i = tid; i < vol; i += 256) {
            a = fmaf(kg[i], corr[i], a);
            b = fmaf(kl[i], corr[i], b);
        }
        a = block_sum(a, red);
        b = block_sum(b, red);
        if (tid == 0) dlam[g] = a - b;   // exactly 0 for g == last (same sums)
    }
}

}  // namespace

extern "C" int sn_conv_corr_t(const void* x, int x_dtype, const void* gout_v, const void* out_v, int g_dtype, int B,
                              int Z, int X, int Y, int kz, int kx, int ky, float* partial_ws, float* C,
                              sn_stream_t stream) {
    if (!x || !gout_v || !partial_ws || !C) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_corr: null pointer");
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_corr: non-positive extent");
    if (g_dtype != SN_F32 && g_dtype != SN_BF16)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_corr_t: g_dtype %d (SN_F32 | SN_BF16)", g_dtype);
    // matrix-core kernel (corr.hip) for every kernel extent <= 16; the thread-per-tap kernel below serves the rest
    if (sn::corr_mfma_supported(kz, kx, ky) && sn::corr_mfma_rows(B, Z, X, Y, kz, kx, ky) > 0)
        return sn::corr_mfma_launch(x, x_dtype, gout_v, out_v, g_dtype, B, Z, X, Y, kz, kx, ky, partial_ws, C,
                                    sn::as_stream(stream));
    if (g_dtype != SN_F32)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_corr_t: bf16 gradients need a kernel extent <= 16 per axis");
    const float* gout = (const float*)gout_v;
    const float* out = (const float*)out_v;
    const int nzt = (Z + TZ - 1) / TZ, nxt = (X + TX - 1) / TX, nyt = (Y + TY - 1) / TY;
    const size_t lds = ((size_t)(TZ + kz - 1) * (TX + kx - 1) * (TY + ky - 1) + (size_t)TZ * TX * TY) * sizeof(float);
    if (lds > 150 * 1024) return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_corr: kernel %dx%dx%d too large", kz, kx, ky);
    const int nblk = B * nzt * nxt * nyt, ntaps = kz * kx * ky;
    hipStream_t s = sn::as_stream(stream);
#define SN_CORR(XT)                                                                                               \
    do {                                                                                                          \
        auto kern = corr_partial_kernel<XT>;                                                                      \
        if (sn::ensure_dynamic_lds((const void*)kern, (int)lds) != hipSuccess)                                    \
            return sn::check_launch("sn_conv_corr(hipFuncSetAttribute)");                                         \
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(kThreads), lds, s, (const XT*)x, gout, out, B, Z, X, Y, kz, kx, \
                           ky, nzt, nxt, nyt, partial_ws);                                                        \
    } while (0)
    switch (x_dtype) {
        case SN_F32: SN_CORR(float); break;
        case SN_F64: SN_CORR(double); break;
        case SN_U8:
        case SN_OCC8: SN_CORR(uint8_t); break;
        default: return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_corr: x_dtype %d", x_dtype);
    }
#undef SN_CORR
    hipLaunchKernelGGL(corr_reduce_kernel, dim3((ntaps + 255) / 256), dim3(256), 0, s, partial_ws, nblk, ntaps, C);
    return sn::check_launch("sn_conv_corr");
}

extern "C" int sn_conv_corr(const void* x, int x_dtype, const float* gout, const float* out, int B, int Z, int X,
                            int Y, int kz, int kx, int ky, float* partial_ws, float* C, sn_stream_t stream) {
    return sn_conv_corr_t(x, x_dtype, gout, out, SN_F32, B, Z, X, Y, kz, kx, ky, partial_ws, C, stream);
}

extern "C" size_t sn_conv_corr_ws_bytes(int x_dtype, int B, int Z, int X, int Y, int kz, int kx, int ky) {
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || kz <= 0 || kx <= 0 || ky <= 0) return 0;
    const size_t rows = (size_t)sn_conv_corr_blocks(B, Z, X, Y) * kz * kx * ky * sizeof(float);
    const size_t sparse = sn::corr_sparse_ws_bytes(x_dtype, B, Z, X, Y, kz, kx, ky);
    return sparse > rows ? sparse : rows;
}

extern "C" int sn_conv_corr_ws(const void* x, int x_dtype, const void* gout, const void* out, int g_dtype, int B, int Z,
                               int X, int Y, int kz, int kx, int ky, void* ws, size_t ws_bytes, float* C,
                               sn_stream_t stream) {
    if (!x || !gout || !ws || !C) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_corr_ws: null pointer");
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_corr_ws: non-positive extent");
    if (g_dtype != SN_F32 && g_dtype != SN_BF16)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_corr_ws: g_dtype %d (SN_F32 | SN_BF16)", g_dtype);
    const size_t rows = (size_t)sn_conv_corr_blocks(B, Z, X, Y) * kz * kx * ky * sizeof(float);
    if (ws_bytes < rows)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_corr_ws: workspace of %zu bytes, %zu needed (sn_conv_corr_ws_bytes)",
                        ws_bytes, rows);
    // binary occupancy: the gather over the set voxels (corr.hip, K4s) unless "corr_dense" asks for the GEMM form or the
    // caller's workspace has no room for the lists
    const size_t sparse = sn::corr_sparse_ws_bytes(x_dtype, B, Z, X, Y, kz, kx, ky);
    if (sparse && ws_bytes >= sparse)
        return sn::corr_sparse_launch(x, gout, out, g_dtype, B, Z, X, Y, kz, kx, ky, ws, C, sn::as_stream(stream));
    return sn_conv_corr_t(x, x_dtype, gout, out, g_dtype, B, Z, X, Y, kz, kx, ky, (float*)ws, C, stream);
}

extern "C" int sn_conv_corr_blocks(int B, int Z, int X, int Y) {
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0) return 0;
    const long tiles = (long)B * ((Z + TZ - 1) / TZ) * ((X + TX - 1) / TX) * ((Y + TY - 1) / TY);
    return (int)(tiles > 768 ? tiles : 768);  // the matrix-core kernel writes at most 768 partial rows
}

extern "C" int sn_geneo_bank_bwd(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky,
                                 const float* dW, float* dparams, sn_stream_t stream) {
    if (!params || !kinds || !dW || !dparams) return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank_bwd: null pointer");
    if (G <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_bank_bwd: non-positive extent");
    const long vol = (long)kz * kx * ky;
    if (vol > 12000) return sn::fail(SN_ERR_UNSUPPORTED, "sn_geneo_bank_bwd: kernel volume %ld > 12000", vol);
    hipLaunchKernelGGL(geneo_bank_bwd_kernel, dim3(G), dim3(256), (size_t)(vol + kz + 1) * sizeof(float),
                       sn::as_stream(stream), params, kinds, kz, kx, ky, dW, dparams, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr);
    return sn::check_launch("sn_geneo_bank_bwd");
}

extern "C" int sn_geneo_backward(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky,
                                 const float* bank, const float* lambdas, const float* corr, int last,
                                 float* dparams, float* dlambdas, sn_stream_t stream) {
    if (!params || !kinds || !bank || !lambdas || !corr || !dparams || !dlambdas)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_backward: null pointer");
    if (G <= 0 || kz <= 0 || kx <= 0 || ky <= 0 || last < 0 || last >= G)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_geneo_backward: bad extent / last");
    const long vol = (long)kz * kx * ky;
    if (vol > 12000) return sn::fail(SN_ERR_UNSUPPORTED, "sn_geneo_backward: kernel volume %ld > 12000", vol);
    hipLaunchKernelGGL(geneo_bank_bwd_kernel, dim3(G), dim3(256), (size_t)(vol + kz + 1) * sizeof(float),
                       sn::as_stream(stream), params, kinds, kz, kx, ky, (const float*)nullptr, dparams, corr, lambdas,
                       bank, last, dlambdas);
    return sn::check_launch("sn_geneo_backward");
}
