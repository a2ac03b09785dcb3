// Hardware-semantics probe behind the z-walk's hand-over protocol (csrc/conv_i8z.inc): is an LDS-DMA's data visible to
// ANOTHER wave's ds_read when the only ordering is  [issuer] s_waitcnt vmcnt(0) -> ds_add_u32 counter   /
// [reader] ds_read_b32 counter (spin) -> ds_read data  -- i.e. a counter instead of the vmcnt + s_barrier the guides
// prescribe?  And the two side questions: do DS instructions of one wave reach the LDS in order (ds_write_b128 then
// ds_add_u32), and does restoring M0 right behind a global_load_lds disturb its destination?
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ldsdma_handover.hip -o /tmp/ldsdma_handover && /tmp/ldsdma_handover
//
// 256 workgroups x 12 waves.  A workgroup streams `items` of 384 bytes (24 lanes x 16 B: one pass of the z-walk) through a
// ring of 16 LDS slots.  Waves claim tickets from an LDS counter; ticket t PRODUCES item t + D (waits until the slot's
// previous item has been consumed, moves the item into the slot, publishes landed[slot]) and CONSUMES item t (spins on
// landed[slot], reads the 96 dwords, compares them with the item's hash, publishes done[slot]).  With D = 1..2 the consumer
// of an item is already spinning when its landed signal arrives: the hottest hand-over the protocol can have.
// A stale read returns the slot's previous item (16 items earlier), whose hash differs: counted as a mismatch.
//
// modes:  0  global_load_lds_dwordx4 (M0 saved, set, restored to a CANARY address) -> vmcnt(0) -> ds_add_u32; spin by ds_read
//         1  as 0, the spin through a generic volatile pointer (FLAT load, the z-walk's first form)
//         2  register staged: global_load_dwordx4 -> ds_write_b128 -> ds_add_u32 (DS order only; no LDS-DMA)
//         3  NEGATIVE CONTROL: as 0 without the vmcnt(0) -- must show mismatches, or the probe cannot see the race
//         4  as 0, plus 4 more waves per workgroup that stream loads and stores (memory latency under load)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

constexpr int kWaves = 12;
constexpr int kSlotBytes = 384;
constexpr int kRing = 16;
constexpr int kSpinMax = 1 << 22;
constexpr unsigned kCanary = 0xC0FFEE11u;

__host__ __device__ inline unsigned mixh(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline unsigned item_word(int wg, int item, int w) { return mixh((unsigned)wg * 0x9e3779b1u + (unsigned)item * 96u + (unsigned)w + 1u); }

__global__ void fill_kernel(unsigned* src, int nsrc) {
    const int wg = blockIdx.x;
    for (int i = threadIdx.x; i < nsrc * 96; i += blockDim.x) src[(size_t)wg * nsrc * 96 + i] = item_word(wg, i / 96, i % 96);
}

struct Result {
    unsigned long long mismatches, timeouts, canary_hits, consumed;
    unsigned first[8];   // wg, item, word, got, expected, ticket, -, -
};

template <int kMode>
__global__ __launch_bounds__(64 * (kWaves + 4)) void probe_kernel(const unsigned* __restrict__ src, int nsrc, int items, int D,
                                                                  unsigned* __restrict__ sink, Result* res) {
    __shared__ __attribute__((aligned(16))) unsigned ring[kRing * 96];
    __shared__ unsigned canary[256];
    __shared__ int landed[kRing], done[kRing], ticket_ctr, stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const unsigned* mysrc = src + (size_t)wg * nsrc * 96;
    for (int i = tid; i < 256; i += blockDim.x) canary[i] = kCanary;
    for (int i = tid; i < kRing * 96; i += blockDim.x) ring[i] = 0xDEADBEEFu;
    if (tid < kRing) { landed[tid] = 0; done[tid] = 0; }
    if (tid == 0) { ticket_ctr = 0; stop = 0; }
    __syncthreads();

    auto lds_addr = [](const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; };
    auto add_lane0 = [&](int* ctr) {
        if (lane == 0) {
            const uint32_t a = lds_addr(ctr);
            const int one = 1;
            asm volatile("ds_add_u32 %0, %1" ::"v"(a), "v"(one) : "memory");
        }
    };
    auto read_ctr = [&](int* ctr) -> int {
        if constexpr (kMode == 1) return *reinterpret_cast<const volatile int*>(ctr);
        else return *reinterpret_cast<const volatile __attribute__((address_space(3))) int*>((__attribute__((address_space(3))) int*)ctr);
    };
    unsigned long long mism = 0, touts = 0, cons = 0;
    bool reported = false;
    auto spin = [&](int* ctr, int expect) {
        int spins = 0;
        while (true) {
            const int seen = read_ctr(ctr);
            if (__builtin_amdgcn_readfirstlane(seen) >= expect) break;
            if (++spins > kSpinMax) { ++touts; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
    };
    auto produce = [&](int u) {
        const int slot = u & 15;
        spin(&done[slot], u >> 4);                       // the slot's previous items (u - 16, ...) have been consumed
        const unsigned* g = mysrc + (size_t)(u % nsrc) * 96 + 4 * lane;
        unsigned* dst = ring + slot * 96;
        if constexpr (kMode == 2) {
            if (lane < 24) {
                const uint4 v = *reinterpret_cast<const uint4*>(g);
                *reinterpret_cast<uint4*>(dst + 4 * lane) = v;
            }
            add_lane0(&landed[slot]);                    // DS order only: the stores, then the add
        } else {
            const uint32_t lds_uni = __builtin_amdgcn_readfirstlane(lds_addr(dst));
            const uint32_t bogus = __builtin_amdgcn_readfirstlane(lds_addr(canary));
            if (lane < 24) {
                uint32_t m0_saved;
                // (M0 is left pointing at the canary block right behind the load: a load that sampled M0 late would land there)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                             "s_mov_b32 m0, %3\n\ts_nop 4\n\ts_mov_b32 m0, %0"
                             : "=&s"(m0_saved) : "s"(lds_uni), "v"(g), "s"(bogus) : "memory");
            }
            if constexpr (kMode != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            add_lane0(&landed[slot]);
            if constexpr (kMode == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto consume = [&](int t) {
        const int slot = t & 15;
        spin(&landed[slot], (t >> 4) + 1);
        const unsigned* p = ring + slot * 96;
        const unsigned a = p[lane];
        const unsigned b = lane < 32 ? p[64 + lane] : 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        add_lane0(&done[slot]);
        const unsigned ea = item_word(wg, t % nsrc, lane), eb = item_word(wg, t % nsrc, 64 + lane);
        const bool bad_a = a != ea, bad_b = lane < 32 && b != eb;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(bad_a) | __builtin_amdgcn_ballot_w64(bad_b);
        if (m) {
            mism += __builtin_popcountll(__builtin_amdgcn_ballot_w64(bad_a)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(bad_b));
            if (!reported && (bad_a || bad_b) && atomicAdd(&res->first[7], 1u) == 0u) {
                res->first[0] = wg; res->first[1] = t; res->first[2] = bad_a ? lane : 64 + lane;
                res->first[3] = bad_a ? a : b; res->first[4] = bad_a ? ea : eb;
                // which item does the stale value belong to?
                int owner = -1;
                for (int back = 1; back <= 4; ++back) {
                    const int it = t - 16 * back;
                    if (it >= 0 && item_word(wg, it % nsrc, bad_a ? lane : 64 + lane) == (bad_a ? a : b)) { owner = it; break; }
                }
                res->first[5] = (unsigned)owner;
            }
            reported = true;
        }
        ++cons;
    };

    // prologue: items 0 .. D-1 by wave 0, behind a barrier
    if (wave == 0)
        for (int u = 0; u < D && u < items; ++u) produce(u);   // (load-generator waves, if any, only meet the barrier)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave >= kWaves) {
        // load generators (mode 4 only; behind the prologue's barrier, which counts every wave that has not ended): stream
        // the source and write a sink until the workers are through
        if constexpr (kMode == 4) {
            unsigned acc = 0;
            size_t i = (size_t)(wave - kWaves) * 64 + lane;
            const size_t n = (size_t)nsrc * 96;
            // (bounded: a probe must not be able to hang the GPU)
            for (int trips = 0; trips < (1 << 20) && *reinterpret_cast<volatile __attribute__((address_space(3))) int*>((__attribute__((address_space(3))) int*)&stop) == 0; ++trips) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { acc += mysrc[i]; i += 256; if (i >= n) i -= n; }
                sink[(size_t)wg * 256 + (wave - kWaves) * 64 + lane] = acc;
            }
        }
        return;
    }

    while (true) {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&ticket_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= items) break;
        if (t + D < items) produce(t + D);
        consume(t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        atomicAdd(&res->mismatches, mism);
        atomicAdd(&res->timeouts, touts);
        atomicAdd(&res->consumed, cons);
    }
    // the last wave out checks the canary block and stops the load generators: every wave draws one ticket >= items when it
    // leaves the loop and one more here, so the draw numbered items + 2 kWaves - 1 is behind all of them
    int mine = 0;
    if (lane == 0) mine = __hip_atomic_fetch_add(&ticket_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    mine = __builtin_amdgcn_readfirstlane(mine);
    if (mine == items + 2 * kWaves - 1) {
        unsigned long long hits = 0;
        for (int i = lane; i < 256; i += 64) hits += __builtin_popcountll(__builtin_amdgcn_ballot_w64(canary[i] != kCanary));
        if (lane == 0) {
            atomicAdd(&res->canary_hits, hits);
            stop = 1;
        }
    }
}

template <int kMode>
static void run(const char* name, const unsigned* src, int nsrc, int items, int D, unsigned* sink, Result* dres) {
    hipMemset(dres, 0, sizeof(Result));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe_kernel<kMode>, dim3(256), dim3(64 * (kWaves + (kMode == 4 ? 4 : 0))), 0, 0, src, nsrc, items, D, sink, dres);
    hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    Result r;
    hipMemcpy(&r, dres, sizeof(r), hipMemcpyDeviceToHost);
    printf("%-44s D=%d items/wg=%d  consumed %llu  mismatched words %llu  spin give-ups %llu  canary words hit %llu  %.1f ms (%s)\n",
           name, D, items, r.consumed, r.mismatches, r.timeouts, r.canary_hits, ms, hipGetErrorString(err));
    if (r.mismatches)
        printf("    first: wg %u item %u word %u got %08x expected %08x -- the value belongs to item %d\n", r.first[0], r.first[1],
               r.first[2], r.first[3], r.first[4], (int)r.first[5]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 400000;
    const int nsrc = 2048;   // distinct items per workgroup (0.75 MB per workgroup, 192 MB in all: L2 misses and hits mixed)
    unsigned *src, *sink;
    Result* dres;
    hipMalloc(&src, (size_t)256 * nsrc * 96 * 4);
    hipMalloc(&sink, (size_t)256 * 256 * 4);
    hipMalloc(&dres, sizeof(Result));
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, src, nsrc);
    hipDeviceSynchronize();
    for (int D = 1; D <= 4; D *= 2) {
        run<0>("LDS-DMA, vmcnt(0) -> ds_add; ds_read spin", src, nsrc, items, D, sink, dres);
        run<1>("LDS-DMA, vmcnt(0) -> ds_add; FLAT spin", src, nsrc, items, D, sink, dres);
        run<2>("register staged, ds_write -> ds_add", src, nsrc, items, D, sink, dres);
        run<4>("LDS-DMA, vmcnt(0) -> ds_add; under load", src, nsrc, items, D, sink, dres);
    }
    run<3>("NEGATIVE CONTROL: ds_add before vmcnt(0)", src, nsrc, items / 8, 1, sink, dres);
    run<3>("NEGATIVE CONTROL: ds_add before vmcnt(0)", src, nsrc, items / 8, 4, sink, dres);
    return 0;
}
