// What rate does v_mfma_i32_16x16x64_i8 sustain on an MI355X with nothing else in the loop?
// 256 workgroups x 512 threads (2 waves per SIMD, as the conv kernels run), each wave NACC independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_i8_peak.hip -o /tmp/mfma_i8_peak && /tmp/mfma_i8_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using i32x4 = __attribute__((ext_vector_type(4))) int;

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// kRandom: every MFMA of a trip reads different, random operand registers (what a real kernel's data looks like to
// the power management); otherwise one constant pair
template <int NACC, bool kRandom>
__global__ __launch_bounds__(512) void peak_kernel(int iters, int* out) {
    i32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = i32x4{0, 0, 0, 0};
    constexpr int NOP = kRandom ? 4 : 1;
    i32x4 a[NOP], b[NOP];
#pragma unroll
    for (int i = 0; i < NOP; ++i) {
        const unsigned s = mix(threadIdx.x * 977u + blockIdx.x * 131u + i * 7919u + 1u);
        a[i] = kRandom ? i32x4{(int)mix(s), (int)mix(s + 1), (int)mix(s + 2), (int)mix(s + 3)} : i32x4{(int)threadIdx.x, 1, 2, 3};
        b[i] = kRandom ? i32x4{(int)mix(s + 4), (int)mix(s + 5), (int)mix(s + 6), (int)mix(s + 7)} : i32x4{4, 5, (int)blockIdx.x, 7};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i % NOP], b[(i / NOP) % NOP], acc[i], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 0x7fffffff) out[0] = s;
}

template <int NACC, bool kRandom>
void run(int waves_per_simd, int spin) {
    int* d = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess) return;
    const int iters = 4000, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // spin-up: the clocks settle only after ~100 ms of sustained load (a cold first launch reads ~15 % low)
    for (int i = 0; i < spin; ++i) peak_kernel<NACC, kRandom><<<256, threads>>>(iters, d);
    (void)hipDeviceSynchronize();
    const int reps = 10;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) peak_kernel<NACC, kRandom><<<256, threads>>>(iters, d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double mfma = 256.0 * (threads / 64) * iters * NACC;
    const double tops = mfma * 2.0 * 16 * 16 * 64 / (ms * 1e-3) / 1e12;
    printf("%s operands  waves/SIMD %d  accumulators %2d: %8.3f ms  %7.1f TOP/s  (%.2f cycles per MFMA per SIMD at 2.4 GHz)\n",
           kRandom ? "random  " : "constant", waves_per_simd, NACC, ms, tops, (ms * 1e-3 * 2.4e9) / (mfma / 1024.0));
    (void)hipFree(d);
}

int main() {
    for (int spin : {1, 150}) {   // cold-ish, then after ~0.2 s of load per configuration
        printf("-- %d spin-up launches\n", spin);
        for (int w : {1, 2}) {
            run<12, false>(w, spin);
            run<24, false>(w, spin);
            run<12, true>(w, spin);
            run<24, true>(w, spin);
        }
    }
    return 0;
}
