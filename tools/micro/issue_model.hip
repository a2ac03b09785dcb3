// How do vector and matrix instructions of SEVERAL waves share one SIMD?  The question behind K3' (csrc/conv_i8z.inc:
// 5.0 VALU per v_mfma_i32_16x16x64_i8, three waves per SIMD, matrix pipe 42 % busy): is the launch bound by the SIMD's
// issue (then only fewer instructions help) or by how the waves' phases line up (then the schedule helps)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/issue_model.hip -o /tmp/issue_model && /tmp/issue_model
// One workgroup per CU (100 KB of LDS), W waves per SIMD.  A wave's stream per trip: 48 MFMAs on 12 independent accumulators
// and 48 * NV v_alignbyte on 8 independent chains, laid out
//   mode 0  fine:   [1 MFMA, NV VALU] x 48
//   mode 1  groups: [12 MFMA, 12 NV VALU] x 4            (K3's step: operands, then the step's MFMAs)
//   mode 2  phases: [48 MFMA] [48 NV VALU]               (a round and its epilogue)
// Reported: time per trip per SIMD relative to the MFMA-only stream of the same W (= 16 cycles per MFMA when the pipe is
// full), i.e. cycles per MFMA; the bounds to compare with: 16 (matrix pipe), 8 + 2 NV (issue, VALU at 2 cycles),
// 8 + 4 NV (issue, VALU at 4 cycles), 16 + c NV (no overlap at all).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <algorithm>
#include <vector>
using i32x4 = __attribute__((ext_vector_type(4))) int;

__device__ __forceinline__ unsigned mixu(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int NV>
__device__ __forceinline__ void valu_block(unsigned (&x)[8], const unsigned (&y)[8], int count, int& rot) {
#pragma unroll
    for (int k = 0; k < count; ++k) {
        const int j = (rot + k) & 7;
        x[j] = __builtin_amdgcn_alignbyte(x[j], y[j], 1);
    }
    rot = (rot + count) & 7;
}

template <int W, int NV, int MODE, bool kMfma>
__global__ __launch_bounds__(W * 256) void issue_kernel(int iters, int* out, long long* stamps) {
    extern __shared__ int lds[];
    i32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = i32x4{0, 0, 0, 0};
    const unsigned s = mixu(threadIdx.x * 977u + blockIdx.x * 131u + 1u);
    i32x4 a[3], b[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = i32x4{(int)mixu(s + i), (int)mixu(s + 11 + i), (int)mixu(s + 22 + i), (int)mixu(s + 33 + i)};
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = i32x4{(int)mixu(s + 4 + i), (int)mixu(s + 55 + i), (int)mixu(s + 66 + i), (int)mixu(s + 77 + i)};
    unsigned x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = mixu(s + 100 + i); y[i] = mixu(s + 200 + i); }
    if (threadIdx.x == 0) lds[0] = (int)s;
    __syncthreads();
    // waves of one SIMD start out of phase (as the kernel's waves are): wave w idles w/W of a trip's MFMA time first
    const int wave = threadIdx.x >> 6;
    const int delay = ((wave >> 2) * 48 * 16) / W;
    for (int i = 0; i < delay / 64; ++i) __builtin_amdgcn_s_sleep(1);
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int rot = 0;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 48; ++m) {
                if constexpr (kMfma) acc[m % 12] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m % 3], b[(m / 3) & 3], acc[m % 12], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                valu_block<NV>(x, y, NV, rot);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                valu_block<NV>(x, y, 12 * NV, rot);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kMfma) {
#pragma unroll
                    for (int m = 0; m < 12; ++m) acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m % 3], b[(m / 3) & 3], acc[m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if constexpr (kMfma) {
#pragma unroll
                for (int m = 0; m < 48; ++m) acc[m % 12] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m % 3], b[(m / 3) & 3], acc[m % 12], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            valu_block<NV>(x, y, 48 * NV, rot);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    int t = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) t += (int)x[i];
    if (t == 0x7fffffff) out[0] = t;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (W * 4) + wave] = t1 - t0;
}

// ---- (2) LDS reads beside the MFMAs (K3L's loop, csrc/conv_lin.hip: per step 12 MFMAs and 3 ds_read_b128 + 9 ds_read_b64 of
// operands for the NEXT step, two waves per SIMD): per trip 48 MFMAs in groups of two, ND reads of kBytes behind each group;
// the reads' results are only waited for at the end of the trip (s_waitcnt lgkmcnt(0)): what is measured is what a read costs
// the instruction stream, not its latency.  Addresses: lane l reads at 16 l (+ 1 KB per read): conflict-free.
template <int W, int ND, int kBytes, bool kMfma>
__global__ __launch_bounds__(W * 256) void lds_kernel(int iters, int* out, long long* stamps) {
    extern __shared__ int lds[];
    for (int i = threadIdx.x; i < 24 * 1024 / 4; i += W * 256) lds[i] = (int)mixu(i);
    i32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = i32x4{0, 0, 0, 0};
    const unsigned s = mixu(threadIdx.x * 977u + blockIdx.x * 131u + 1u);
    i32x4 a[3], b[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = i32x4{(int)mixu(s + i), (int)mixu(s + 11 + i), (int)mixu(s + 22 + i), (int)mixu(s + 33 + i)};
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = i32x4{(int)mixu(s + 4 + i), (int)mixu(s + 55 + i), (int)mixu(s + 66 + i), (int)mixu(s + 77 + i)};
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)lds + (threadIdx.x & 63) * 16;
    const long long t0 = (long long)__builtin_readcyclecounter();
    unsigned sink = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            if constexpr (kMfma) {
                acc[(2 * g) % 12] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(2 * g) % 3], b[g & 3], acc[(2 * g) % 12], 0, 0, 0);
                acc[(2 * g + 1) % 12] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(2 * g + 1) % 3], b[g & 3], acc[(2 * g + 1) % 12], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < ND; ++r) {
                if constexpr (kBytes == 16) {
                    i32x4 v;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "i"(((r * 5) % 20) * 1024));
                    asm volatile("" ::"v"(v));
                } else {
                    unsigned long long v;
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "i"(((r * 5) % 20) * 1024));
                    asm volatile("" ::"v"(v));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    int t = (int)sink;
#pragma unroll
    for (int i = 0; i < 12; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 0x7fffffff) out[0] = t;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (W * 4) + wave] = t1 - t0;
}

template <int W, int ND, int kBytes, bool kMfma = true>
void run_lds(const char* what) {
    int* d = nullptr;
    long long* st = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess || hipMalloc(&st, 256 * W * 4 * 8) != hipSuccess) return;
    const int iters = 400;
    auto k = lds_kernel<W, ND, kBytes, kMfma>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<<<256, W * 256, 100 * 1024>>>(iters, d, st);
    (void)hipDeviceSynchronize();
    const int reps = 5;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k<<<256, W * 256, 100 * 1024>>>(iters, d, st);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double ns_per_slot = ms * 1e6 / (iters * 48.0 * W);   // per MFMA slot of the SIMD
    printf("%-44s W=%d: %7.3f ms/launch  %6.2f ns per MFMA slot of the SIMD (%.2f reads per MFMA)\n", what, W, ms, ns_per_slot,
           ND / 2.0);
    fflush(stdout);
    (void)hipFree(d);
    (void)hipFree(st);
}

template <int W>
void sweep_lds() {
    printf("---- LDS reads beside MFMAs, %d wave(s) per SIMD (MFMA only = 16 cycles per slot)\n", W);
    run_lds<W, 0, 16>("MFMA only");
    run_lds<W, 1, 16>("[2 MFMA, 1 ds_read_b128]");
    run_lds<W, 2, 16>("[2 MFMA, 2 ds_read_b128]");
    run_lds<W, 1, 8>("[2 MFMA, 1 ds_read_b64]");
    run_lds<W, 2, 8>("[2 MFMA, 2 ds_read_b64]");
    run_lds<W, 3, 8>("[2 MFMA, 3 ds_read_b64]");
    run_lds<W, 4, 8>("[2 MFMA, 4 ds_read_b64]");
    run_lds<W, 2, 16, false>("[2 ds_read_b128] alone");
    run_lds<W, 2, 8, false>("[2 ds_read_b64] alone");
    run_lds<W, 4, 8, false>("[4 ds_read_b64] alone");
}

static double g_base[4];   // ms per trip of the MFMA-only stream, by W

template <int W, int NV, int MODE, bool kMfma = true>
double run(const char* what) {
    int* d = nullptr;
    long long* st = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess || hipMalloc(&st, 256 * W * 4 * 8) != hipSuccess) return 0;
    const int iters = 400;
    auto k = issue_kernel<W, NV, MODE, kMfma>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<<<256, W * 256, 100 * 1024>>>(iters, d, st);
    (void)hipDeviceSynchronize();
    const int reps = 5;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k<<<256, W * 256, 100 * 1024>>>(iters, d, st);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<long long> h(256 * W * 4);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double per_mfma_wave = med / (iters * 48.0);          // counter ticks per MFMA slot of ONE wave
    const double per_mfma_simd = per_mfma_wave / W;             // ... of the SIMD (W waves run side by side)
    printf("%-34s W=%d NV=%d: %7.3f ms/launch  wave: %7.2f ticks per MFMA slot  SIMD: %6.2f\n", what, W, NV, ms, per_mfma_wave,
           per_mfma_simd);
    fflush(stdout);
    (void)hipFree(d);
    (void)hipFree(st);
    return ms;
}

template <int W>
void sweep() {
    printf("---- %d wave(s) per SIMD (ticks are of the shader's cycle counter; the MFMA-only line calibrates: 16 cycles per MFMA)\n", W);
    run<W, 0, 2>("MFMA only");
    run<W, 4, 2, false>("VALU only (4 per slot)");
    run<W, 2, 0>("fine  [1 MFMA, 2 VALU]");
    run<W, 4, 0>("fine  [1 MFMA, 4 VALU]");
    run<W, 5, 0>("fine  [1 MFMA, 5 VALU]");
    run<W, 6, 0>("fine  [1 MFMA, 6 VALU]");
    run<W, 8, 0>("fine  [1 MFMA, 8 VALU]");
    run<W, 2, 1>("group [12 MFMA, 24 VALU]");
    run<W, 4, 1>("group [12 MFMA, 48 VALU]");
    run<W, 5, 1>("group [12 MFMA, 60 VALU]");
    run<W, 8, 1>("group [12 MFMA, 96 VALU]");
    run<W, 2, 2>("phase [48 MFMA][96 VALU]");
    run<W, 4, 2>("phase [48 MFMA][192 VALU]");
    run<W, 5, 2>("phase [48 MFMA][240 VALU]");
    run<W, 8, 2>("phase [48 MFMA][384 VALU]");
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'l') {
        sweep_lds<1>();
        sweep_lds<2>();
        sweep_lds<3>();
        return 0;
    }
    sweep<1>();
    sweep<2>();
    sweep<3>();
    return 0;
}
