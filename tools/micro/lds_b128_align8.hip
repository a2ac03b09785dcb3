// Does ds_read_b128 work at 8-byte (not 16-byte) alignment on gfx950, and what does it cost at a 24-byte lane stride?
// (K3L's 24-byte row cells: a lane's 16 operand bytes start at 24 row + 8 t.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_b128_align8.hip -o /tmp/lds_b128_align8 && /tmp/lds_b128_align8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__global__ void check_kernel(int* bad) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = (uint8_t)(i * 37 + (i >> 8) * 11 + 5);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int t = 0; t < 3; ++t) {
        const uint8_t* p = buf + 24 * lane + 8 * t + 64 * (threadIdx.x >> 6);
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
        for (int j = 0; j < 4; ++j) {
            uint32_t e = 0;
            for (int b = 0; b < 4; ++b) e |= (uint32_t)p[4 * j + b] << (8 * b);
            if (v[j] != e) atomicAdd(bad, 1);
        }
    }
}

// mode 0: ds_read_b128, 16-byte lane stride, aligned (today's B reads of the 32-byte packing)
// mode 1: ds_read_b128, 24-byte lane stride, 8-byte aligned
// mode 2: two ds_read_b64, 16-byte lane stride (today's 24-byte packing)
// mode 3: ds_read2_b64 offset1 = offset0 + 1, 24-byte lane stride
template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(int iters, int* out) {
    extern __shared__ uint8_t lds[];
    for (int i = threadIdx.x; i < 40 * 1024 / 4; i += 512) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint32_t a16 = base + n * 16 + q * 1040;                 // rows of a lane group, groups some rows apart
    const uint32_t a24 = base + n * 24 + q * (24 * 43) + 8 * (q & 1) + 8 * (q >> 1);
    unsigned sum = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (MODE == 0) {
                u32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a16), "i"(k * 4096));
                asm volatile("" ::"v"(v));
            } else if constexpr (MODE == 1) {
                u32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a24), "i"(k * 4104));
                asm volatile("" ::"v"(v));
            } else if constexpr (MODE == 2) {
                unsigned long long v, w;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a16), "i"(k * 4096));
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w) : "v"(a16), "i"(k * 4096 + 264));
                asm volatile("" ::"v"(v), "v"(w));
            } else {
                u32x4 v;
                asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a24 + k * 4104), "i"(0), "i"(1));
                asm volatile("" ::"v"(v));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (sum == 0x12345678u) out[0] = (int)sum;
}

template <int MODE>
void run(const char* what) {
    int* d = nullptr;
    (void)hipMalloc(&d, 4);
    auto k = rate_kernel<MODE>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int iters = 4000;
    for (int i = 0; i < 3; ++i) k<<<256, 512, 100 * 1024>>>(iters, d);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<<<256, 512, 100 * 1024>>>(iters, d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-62s %7.3f ms  %6.2f ns per 16 operand bytes per lane, per SIMD (2 waves)\n", what, ms, ms * 1e6 / (iters * 8.0 * 2));
    (void)hipFree(d);
}

int main() {
    int* bad = nullptr;
    (void)hipMalloc(&bad, 4);
    (void)hipMemset(bad, 0, 4);
    check_kernel<<<1, 256>>>(bad);
    int h = -1;
    (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("ds_read_b128 at 8-byte alignment, 24-byte lane stride: %d wrong dwords of 3072\n", h);
    run<0>("ds_read_b128, 16-byte stride, aligned");
    run<1>("ds_read_b128, 24-byte stride, 8-byte aligned");
    run<2>("2 x ds_read_b64, 16-byte stride");
    run<3>("ds_read2_b64 (offset1 = offset0 + 1), 24-byte stride");
    return h == 0 ? 0 : 1;
}
