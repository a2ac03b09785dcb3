// What does a VALU instruction cost next to a saturated int8 matrix pipe?  And what does an LDS read at a byte offset
// that is not a multiple of four cost?  Two questions behind K3' (csrc/conv_i8s.hip), answered on the part itself.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_mix.hip -o /tmp/mfma_valu_mix && /tmp/mfma_valu_mix
// (1) 256 workgroups x 512 threads (2 waves per SIMD), each wave loops over 24 independent MFMAs (random operands) with
//     KV v_alignbyte between them (results feed the next trip's B operands, so they cannot be dropped).
// (2) every lane reads 8 bytes from LDS at byte offset 4 * lane + r + row * 96 (r = 0..3), 16 reads per trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
using i32x4 = __attribute__((ext_vector_type(4))) int;

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// kOp: which VALU instruction rides between the MFMAs: 0 v_alignbyte_b32, 1 v_perm_b32, 2 v_add_u32, 3 v_cvt_f32_i32,
// 4 v_lshl_add_u64 (counted as ONE instruction per two dwords)
template <int kOp>
__device__ __forceinline__ int valu_op(int hi, int lo) {
    if constexpr (kOp == 0) return (int)__builtin_amdgcn_alignbyte((unsigned)hi, (unsigned)lo, 1);
    else if constexpr (kOp == 1) return (int)__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x02030405u);
    else if constexpr (kOp == 2) return hi + lo;
    else if constexpr (kOp == 3) return __float_as_int((float)lo) ^ hi;   // cvt + xor: two instructions
    else return 0;
}

template <int KV, int kOp = 0>
__global__ __launch_bounds__(512) void mix_kernel(int iters, int* out) {
    constexpr int NACC = 24;
    i32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = i32x4{0, 0, 0, 0};
    i32x4 a[3], b[8];
    const unsigned s = mix(threadIdx.x * 977u + blockIdx.x * 131u + 1u);
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = i32x4{(int)mix(s + i), (int)mix(s + 11 + i), (int)mix(s + 22 + i), (int)mix(s + 33 + i)};
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = i32x4{(int)mix(s + 4 + i), (int)mix(s + 55 + i), (int)mix(s + 66 + i), (int)mix(s + 77 + i)};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            // KV / 8 VALU per group of three MFMAs
#pragma unroll
            for (int k = 0; k < KV / 8; ++k) {
                const int j = (v + 1 + k) & 7;
                if constexpr (kOp == 4) {
                    unsigned long long u = ((unsigned long long)(unsigned)b[j][1] << 32) | (unsigned)b[j][0];
                    u += ((unsigned long long)(unsigned)b[j][3] << 32) | (unsigned)b[j][2];
                    b[j][0] = (int)(unsigned)u;
                    b[j][1] = (int)(unsigned)(u >> 32);
                } else {
                    b[j][k & 3] = valu_op<kOp>(b[j][(k + 1) & 3], b[j][k & 3]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < 3; ++d) acc[3 * v + d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[d], b[v], acc[3 * v + d], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int t = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 0x7fffffff) out[0] = t;
}

template <int KV, int kOp = 0>
void run_mix(int spin) {
    int* d = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess) return;
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < spin; ++i) mix_kernel<KV, kOp><<<256, 512>>>(iters, d);
    (void)hipDeviceSynchronize();
    const int reps = 10;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) mix_kernel<KV, kOp><<<256, 512>>>(iters, d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double trips_per_simd = 2.0 * iters;   // two waves per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / trips_per_simd;
    static const char* names[] = {"v_alignbyte", "v_perm", "v_add_u32", "v_cvt_f32_i32 + v_xor", "v_lshl_add_u64"};
    printf("24 MFMA + %2d %-22s per trip: %7.3f ms  %7.1f cycles per trip per SIMD at 2.4 GHz (24 MFMA = 384)\n", KV, names[kOp], ms, cyc);
    (void)hipFree(d);
}

// ---- (2) LDS reads at byte offsets
template <int kBytes>
__global__ __launch_bounds__(512) void lds_kernel(int iters, int r, int* out) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[96 * 256 + 64];
    for (int i = threadIdx.x; i < (96 * 256 + 64) / 4; i += 512) reinterpret_cast<uint32_t*>(buf)[i] = mix(i);
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    unsigned sum = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t v8[16];
        uint32_t v4[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {   // 16 reads in flight, one wait
            const uint8_t* p = buf + ((it + k * 7 + q * 2) & 127) * 96 + 4 * n + r;
            if (kBytes == 8) asm volatile("ds_read_b64 %0, %1\n" : "=v"(v8[k]) : "v"((uint32_t)(uintptr_t)p));
            else asm volatile("ds_read_b32 %0, %1\n" : "=v"(v4[k]) : "v"((uint32_t)(uintptr_t)p));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += kBytes == 8 ? ((unsigned)v8[k] ^ (unsigned)(v8[k] >> 32)) : v4[k];
    }
    if (sum == 0x12345678u) out[0] = (int)sum;
}

// correctness of an unaligned read: compare with bytes
__global__ void lds_check_kernel(int* bad) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = (uint8_t)(i * 37 + 11);
    __syncthreads();
    const int lane = threadIdx.x;
    for (int r = 0; r < 4; ++r) {
        const uint8_t* p = buf + 4 * lane + r;
        uint64_t v;
        asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
        uint64_t e = 0;
        for (int b = 0; b < 8; ++b) e |= (uint64_t)p[b] << (8 * b);
        if (v != e) atomicAdd(bad, 1);
    }
}

template <int kBytes>
void run_lds(int r) {
    int* d = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess) return;
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    lds_kernel<kBytes><<<256, 512>>>(iters, r, d);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) lds_kernel<kBytes><<<256, 512>>>(iters, r, d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double reads_per_cu = 8.0 * iters * 16;   // wave-instructions per CU
    printf("ds_read_b%d at byte offset r = %d: %7.3f ms  %6.1f cycles per wave-instruction per CU  (%5.1f B/cycle/CU)\n",
           kBytes * 8, r, ms, ms * 1e-3 * 2.4e9 / reads_per_cu, 64.0 * kBytes / (ms * 1e-3 * 2.4e9 / reads_per_cu));
    (void)hipFree(d);
}

int main() {
    int* bad = nullptr;
    (void)hipMalloc(&bad, 4);
    (void)hipMemset(bad, 0, 4);
    lds_check_kernel<<<1, 64>>>(bad);
    int hb = -1;
    (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("unaligned ds_read_b64: %d mismatching reads of 256\n", hb);
    for (int r = 0; r < 4; ++r) run_lds<8>(r);
    for (int r = 0; r < 4; r += 2) run_lds<4>(r);
    run_mix<0>(150);
    run_mix<8>(20);
    run_mix<16>(20);
    run_mix<24>(20);
    run_mix<48>(20);
    run_mix<96>(20);
    run_mix<48, 1>(20);
    run_mix<96, 1>(20);
    run_mix<48, 2>(20);
    run_mix<96, 2>(20);
    run_mix<48, 3>(20);
    run_mix<48, 4>(20);
    run_mix<96, 4>(20);
    return 0;
}
