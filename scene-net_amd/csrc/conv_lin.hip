// K3L -- SceneNet.forward through LINEARITY, for binary occupancy on the int8 matrix cores.
//
// relu(tanh(sum_g lambda_g conv3d(x, K_g))) == relu(tanh(conv3d(x, sum_g lambda_g K_g)))  (equal to
// 5e-16 in the reference's own fp64; core/models/SCENE_Net.py:322-339).  When the per-kernel bank activations are not
// asked for, the 16-kernel contraction collapses to ONE kernel K* = sum_g lambda_g K_g, and a single-kernel 3-D
// correlation maps onto MFMA through the Toeplitz structure along y:
//
//     out[z][x][y0+m] = sum_{dz,dx} sum_{y'} T_{dz,dx}[m][y'] * X[z+dz][x+dx][y0+y'],   T[m][y'] = K*[dz][dx][y'-m]
//
// M = 16 consecutive output y (m), N = 16 output rows (x), K = the 32-byte window y' of a kernel row (dz,dx), two
// kernel rows per v_mfma_i32_16x16x64_i8.  The B operand is then 16 CONSECUTIVE, 16-byte aligned halo bytes per
// lane (one ds_read_b128; no byte-shifted copies), the A operand a precomputed banded Toeplitz table of the three
// balanced base-256 digits of the 24-bit fixed-point K* (exact int32 accumulation, as in conv_i8.hip).
// Executed matrix work per output voxel: ceil(kz*kx/2) * 3 MFMAs per 256 outputs = 0.48 MFMA/voxel at 9^3 -- 0.375 with
// the rows packed at 24 K-bytes (kW24 below, ky <= 9) -- against 3 MFMA/voxel for the 16-kernel contraction.
//
// Bound: MFMA (int8) with LDS well below its limit: per step a wave reads 3 + 4 ds_read_b128 for 12 MFMAs.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace sn {
int conv_bank_group(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X, int Y,
                    int G, int Gtot, int g0, int head, int kz, int kx, int ky, void* act, void* out, int out_dtype,
                    sn_stream_t stream);   // conv.hip
int conv_fused_lin(const uint8_t* x, const float* bank, const float* lambdas, int B, int Z, int X, int Y, int G, int kz,
                   int kx, int ky, void* out, int out_dtype, hipStream_t stream, int32_t* verdict = nullptr, bool assume_served = false,
                   const void* prep = nullptr);
int conv_fused_prep_launch(const float* bank, const float* lambdas, int G, int kz, int kx, int ky, void* blob,
                           hipStream_t stream);
}

namespace {

using i32x4 = __attribute__((ext_vector_type(4))) int;

__device__ __forceinline__ float relu_nan(float v) { return (v > 0.0f || v != v) ? v : 0.0f; }

// relu(tanh(v)): 0 for v <= 0, else 1 - 2 / (exp(2v) + 1) on the hardware exp / rcp (abs. error ~2e-7, inside the
// 1e-4 bar; the 16-kernel kernels call tanhf).  NaN stays NaN like torch.relu(torch.tanh(.)); +inf -> 1.
__device__ __forceinline__ float relu_tanh(float v) {
    // branch-free, v_exp_f32 + v_rcp_f32 (1 ulp; the correctly rounded reciprocal was ten instructions, twice per lane and round)
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f);
    const float r = (v > 0.0f) ? t : 0.0f;
    return (v != v) ? v : r;
}

#ifdef SN_CONV_TIMING   // make -B EXTRA=-DSN_CONV_TIMING; read by tools/lin_timing.py
__device__ unsigned long long g_lin_t[1024 * 16];   // per workgroup: 0 start, 1 tables done, 2 end, 3 tiles, 4..6 table phases, 8.. per wave
#define SN_LTT(k) do { if (threadIdx.x == 0) g_lin_t[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define SN_LTT(k) do {} while (0)
#endif
#include "conv_lin_tables.inc"   // LinShape, lin_plan, lin_tables: the per-bank work (the kernel itself and conv_lin_prep_kernel)

#ifndef SN_LIN_SETS
#define SN_LIN_SETS 2   // [measured, round 4] 3 (two steps of look-ahead, 253 VGPRs): 43.9 us against 43.4 -- the waves do not wait for latency
#endif
constexpr int kThreads = kLinThreads;
constexpr int TZ = kLinTZ, TX = kLinTX, TY = kLinTY, YB = kLinYB;
constexpr int kMaxLds = kLinMaxLds;

// kW24: kernel rows packed at 24 bytes of K instead of 32 (ky <= 9: a 16-y strip's window is 16 + ky - 1 <= 24 bytes), 2.67
// rows per MFMA step instead of 2: 24 % fewer steps.  K byte kappa = 24 p + o belongs to kernel row p, window byte o; a
// lane's 16 K bytes are two 8-byte pieces pi = 8 st + 2 q + e (row pi / 3, third pi % 3), so the B operand is two
// ds_read_b64 (halo offsets from a small LDS table, fetched two steps ahead) and the A table is built per piece.
template <typename OT, bool kW24>
__global__ __launch_bounds__(kThreads) void conv_lin_i8_kernel(const uint8_t* __restrict__ x,
                                                               const float* __restrict__ bank,
                                                               const float* __restrict__ lambdas, LinShape s,
                                                               OT* __restrict__ out) {
#ifdef SN_CONV_TIMING
    if (threadIdx.x == 0) g_lin_t[blockIdx.x * 16 + 0] = wall_clock64();
#endif
    if (!s.gate.pass()) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint4* At = reinterpret_cast<uint4*>(lds);                                  // [nsteps + 1][3][64], last step zero
    float* misc = reinterpret_cast<float*>(At + (size_t)(s.nsteps + 1) * 3 * 64);                // [64]: scale, per-wave maxima, error sums
    int* offtab = reinterpret_cast<int*>(misc + 64);                            // kW24: [(nsteps + 3) * 8] piece -> halo offset
    uint8_t* halo = reinterpret_cast<uint8_t*>(offtab + (kW24 ? (s.nsteps + 3) * 8 : 0));   // [NC][NRP] x 16 bytes
    float* kstar = reinterpret_cast<float*>(halo);                              // [ntaps] (prologue only; aliases halo)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const size_t V = (size_t)s.Z * s.X * s.Y;
    // halo rows (z0 - pz .., x0 - px ..), columns y0 - PYA .. + YB, read as aligned global dwords into registers
    // (kHaloRegs per thread; larger halos take a synchronous remainder pass)
    constexpr int DW = YB / 4;
    constexpr int kHaloRegs = 16;
    uint32_t hreg[kHaloRegs];
    auto tile_origin = [&](int tl, int& b, int& z0, int& x0, int& y0) {
        y0 = (tl % s.nyt) * TY; tl /= s.nyt;
        x0 = (tl % s.nxt) * TX; tl /= s.nxt;
        z0 = (tl % s.nzt) * TZ; tl /= s.nzt;
        b = tl;
    };
    // Thread -> (dword column i, first halo row): kThreads / DW = 25 rows per pass (the last 12 threads idle), so a
    // pass advances every thread by the same (dz, dx) and the row split r -> (zz, xx) is incremental -- one add and
    // one wrap per load instead of divisions (every thread runs this 16 times per tile, next to the MFMA loop).
    constexpr int RPP = kThreads / DW;
    const int h_i = tid % DW, h_r0 = tid / DW;
    const bool h_thread = tid < RPP * DW;
    const int h_zz0 = h_r0 / s.XP, h_xx0 = h_r0 - h_zz0 * s.XP;
    const int h_dzz = RPP / s.XP, h_dxx = RPP - h_dzz * s.XP;
    const int h_lds0 = ((h_i >> 2) * s.NRP + h_r0) * 4 + (h_i & 3);   // dword index of (row h_r0, column i) in the halo
    int hb_b = 0, hb_z0 = 0, hb_x0 = 0, hb_y0 = 0;   // tile the registers belong to
    // passes [u0, u1) of the tile (hb_*): f(u, value)
    auto halo_rows = [&](int u0, int u1, auto&& f) {
        const uint8_t* xb = x + (size_t)hb_b * V;
        const int gy = hb_y0 - s.PYA + 4 * h_i;
        const bool oky = h_thread && (unsigned)gy < (unsigned)s.Y;
        int zz = h_zz0, xx = h_xx0, r = h_r0;
        for (int u = 0; u < u0; ++u) {   // (u0 > 0 only for halos beyond the registers)
            r += RPP; zz += h_dzz; xx += h_dxx;
            if (xx >= s.XP) { xx -= s.XP; ++zz; }
        }
#pragma unroll
        for (int u = u0; u < u1; ++u) {
            const int gz = hb_z0 - s.pz + zz, gx = hb_x0 - s.px + xx;
            uint32_t v = 0u;
            if (oky && r < s.rows && (unsigned)gz < (unsigned)s.Z && (unsigned)gx < (unsigned)s.X)
                v = *reinterpret_cast<const uint32_t*>(xb + ((gz * s.X + gx) * s.Y + gy));
            f(u, v);
            r += RPP; zz += h_dzz; xx += h_dxx;
            const bool w = xx >= s.XP;
            xx -= w ? s.XP : 0;
            zz += w;
        }
    };
    auto halo_store = [&](int u, uint32_t v) {
        if (h_thread && h_r0 + u * RPP < s.rows) reinterpret_cast<uint32_t*>(halo)[h_lds0 + u * RPP * 4] = v;
    };
    auto halo_issue = [&](int tl) {
        tile_origin(tl, hb_b, hb_z0, hb_x0, hb_y0);
        halo_rows(0, kHaloRegs, [&](int u, uint32_t v) { hreg[u] = v; });
    };
    const int h_passes = (s.rows + RPP - 1) / RPP;
    auto halo_commit = [&]() {
#pragma unroll
        for (int u = 0; u < kHaloRegs; ++u) halo_store(u, hreg[u]);
        for (int u = kHaloRegs; u < h_passes; ++u)   // halos beyond the registers: synchronous
            halo_rows(u, u + 1, [&](int uu, uint32_t v) { halo_store(uu, v); });
    };

#ifdef SN_CONV_TIMING
#define SN_LT(k) do { if (threadIdx.x == 0) g_lin_t[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define SN_LT(k) do {} while (0)
#endif
    SN_LT(4);
    // ---- prologue: the tables -- copied from a prepared blob (sn_conv_fused_prep / a rider of the voxelisation), or built here
    // the first tile's halo is requested first and travels meanwhile ([measured] the bank loads ahead of it instead: the
    // prologue got 0.6 us longer -- the halo's 16 loads then queue behind 48 others)
    if ((int)blockIdx.x < s.ntiles) halo_issue(blockIdx.x);
    // The guard declined and NOBODY is enqueued behind this launch (the caller assumed "served" from a verdict that does not
    // belong to these tables -- a stale cache): every workgroup takes the same decision, so together they fill `out` with
    // NaN, and the sticky status is latched.  Loud, not quiet (round 4; the z-walk does the same, conv_i8z.inc).
    auto declined_unserved = [&]() {
        const OT nan = (OT)__int_as_float(0x7fc00000);
        const size_t n = (size_t)s.B * V;
        for (size_t i = (size_t)blockIdx.x * kThreads + tid; i < n; i += (size_t)gridDim.x * kThreads) out[i] = nan;
        if (tid == 0 && blockIdx.x == 0) sn::sticky_latch(s.sticky, 2, 0, 1);
    };
    // The prepared table ((nsteps + 1) * 3 pieces of 1 KB: 98 KB at 9^3) arrives by LDS-DMA, every piece in flight at once.
    // [measured, round 4, tools/lin_timing.py] the register copy it replaces -- load, wait, ds_write: 12 dependent trips to
    // L2 per thread -- made the prologue 6.0 of the kernel's 46 us; now 3.8, of which the first tile's halo (16 loads per
    // thread from HBM, requested first) is ~3: STREAMING the table in behind the first tile's MFMAs (eight steps per group,
    // s_waitcnt vmcnt(0) + s_barrier gates in the loop) brought the prologue to 3.5 and cost 2 us in the loop -- 44.9 us
    // against 43.3, removed again.
    auto table_dma = [&]() {
        const int pieces = (s.nsteps + 1) * 3;
        const uint32_t at_uni = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds);
        for (int pc = wave; pc < pieces; pc += kThreads / 64) {
            const uint8_t* src = s.prep + (size_t)pc * 1024 + lane * 16;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(at_uni + pc * 1024);   // wave-uniform; lane l's 16 bytes land at + 16 l
            uint32_t m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "s"(dst), "v"(src) : "memory");
        }
    };
    if (s.prep) {
        const LinBlob lb = lin_blob_layout(s.nsteps, kW24);
        const int32_t verdict = *reinterpret_cast<const int32_t*>(s.prep + lb.tail_off + 4);
        if (verdict != 0 && s.tol > 0.0f) {   // the bound was over the tolerance: the gated fp32 launches behind take over
            if (s.served) declined_unserved();
            return;
        }
#ifdef SN_LIN_PROLOGUE_REGS   // (the round-3 copy, for A/B runs)
        const uint4* src4 = reinterpret_cast<const uint4*>(s.prep);
        for (int i = tid; i < (s.nsteps + 1) * 3 * 64; i += kThreads) At[i] = src4[i];
#else
        table_dma();
#endif
        if constexpr (kW24) {
            const int* ts = reinterpret_cast<const int*>(s.prep + lb.tab_off);
            for (int i = tid; i < (s.nsteps + 3) * 8; i += kThreads) offtab[i] = ts[i];
        }
        if (tid == 0) misc[0] = *reinterpret_cast<const float*>(s.prep + lb.tail_off);
    } else {
        if (lin_tables<kW24>(bank, lambdas, s, At, offtab, misc, misc, kstar, s.route, blockIdx.x == 0)) {
            if (s.served) declined_unserved();
            return;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed (the barrier publishes them)
    __syncthreads();   // kstar (aliasing the halo) is dead, tables are complete
    const float scale = misc[0];
    if (SN_DBG(s, 1)) return;

    // Deferred epilogue: a tile's 16 outputs per lane stay in registers as pre-activations and are finished
    // (exp / rcp / 16-byte stores) BETWEEN the MFMAs of the next tile's first 16 steps, instead of in a phase where
    // the matrix pipe idles.  Needs >= 16 steps; smaller kernels finish each tile at once.
    const bool defer = s.nsteps >= 16 && !SN_DBG(s, 16);
    float pv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) pv[k] = 0.f;
    OT* pout = out;      // address of strip 0's first output of the pending tile
    int pgy = 0;         // its y
    bool pend = false;   // a tile is pending and this lane's row is inside the grid
    float rr[4] = {0.f, 0.f, 0.f, 0.f};
    auto store4 = [&](OT* o, int gy, const float (&r)[4]) {
        if (gy + 3 < s.Y) {
            if constexpr (sizeof(OT) == 2) {   // bf16 storage: four values, one 8-byte store
                const OT h[4] = {(OT)r[0], (OT)r[1], (OT)r[2], (OT)r[3]};
                uint2 u;
                __builtin_memcpy(&u, h, 8);
                *reinterpret_cast<uint2*>(o) = u;
            } else if constexpr (sizeof(OT) == 4) {
                *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
            } else {
                reinterpret_cast<double2*>(o)[0] = make_double2((double)r[0], (double)r[1]);
                reinterpret_cast<double2*>(o)[1] = make_double2((double)r[2], (double)r[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (gy + i < s.Y) o[i] = (OT)r[i];
        }
    };
    auto epi_item = [&](auto K) {   // output k = 4 v + i of the pending tile
        constexpr int k = decltype(K)::value;
        rr[k & 3] = relu_tanh(pv[k]);
        if constexpr ((k & 3) == 3) {
            if (pend) store4(pout + 16 * (k >> 2), pgy + 16 * (k >> 2), rr);
        }
    };
#define SN_EPI2(K) epi_item(std::integral_constant<int, (K)>{}); epi_item(std::integral_constant<int, (K) + 1>{});
#ifdef SN_CONV_TIMING
    if (tid == 0) { g_lin_t[blockIdx.x * 16 + 1] = wall_clock64(); g_lin_t[blockIdx.x * 16 + 3] = 0; g_lin_t[blockIdx.x * 16 + 7] = 0; }
    if (lane == 0) {   // (bits 40..: HW_ID -- which SIMD the wave sits on; the busy time accumulates below)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_lin_t[blockIdx.x * 16 + 8 + wave] = (unsigned long long)(hwid & 0xffffu) << 40;
    }
#endif
    for (int tile = blockIdx.x; tile < s.ntiles; tile += gridDim.x) {
#ifdef SN_CONV_TIMING
        const unsigned long long t_tile = wall_clock64();
        if (tid == 0) g_lin_t[blockIdx.x * 16 + 3] += 1;
#endif
        int b, z0, x0, y0;
        tile_origin(tile, b, z0, x0, y0);
        // ---- halo tile: committed from registers requested during the previous tile's MFMA loop
        halo_commit();
        __syncthreads();
#ifdef SN_CONV_TIMING
        if (tid == 0) g_lin_t[blockIdx.x * 16 + 7] += wall_clock64() - t_tile;   // commit + barrier
#endif
        if (tile + (int)gridDim.x < s.ntiles) halo_issue(tile + gridDim.x);   // lands while the MFMAs below run

        // ---- this wave's z plane: 4 strips x 3 digit planes of 16 y x 16 rows
        const int lz = wave;
        i32x4 acc[3][4];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[d][v] = i32x4{0, 0, 0, 0};
        const uint8_t* hb = halo + (lz * s.XP + n) * 16;
        // software pipeline, two register sets: the operands of step st + 1 are requested before the 12 MFMAs of
        // step st issue; the halo offset of a step comes from registers (dz, dx advance by two kernel rows per step)
        const int nst = SN_DBG(s, 2) ? 0 : s.nsteps;   // pairs of steps, then a lone one if odd
        int pdz = 0, pdx = (q >> 1);                 // kernel row p = 2 st + (q >> 1) of this lane group
        if (pdx >= s.kx) { pdx -= s.kx; ++pdz; }   // kx == 1
        const int hoff = (q & 1) * s.NRP * 16;   // the window's second chunk is one chunk plane further
        const int cstride = s.NRP * 16;
        auto row_off = [&]() -> int {
            const int dzc = pdz < s.kz ? pdz : s.kz - 1, dxc = pdz < s.kz ? pdx : s.kx - 1;  // past the end: zero weights
            return (dzc * s.XP + dxc) * 16 + hoff;
        };
        auto advance = [&]() {   // two kernel rows on, branch-free (2 <= 2 kx: at most two wraps)
            pdx += 2;
            int w = pdx >= s.kx;
            pdx -= w ? s.kx : 0;
            pdz += w;
            w = pdx >= s.kx;
            pdx -= w ? s.kx : 0;
            pdz += w;
        };
        uint4 aA[3], xA[4], aB[3], xB[4];
        auto load_step = [&](int st, uint4 (&a)[3], uint4 (&xv)[4]) {
            const uint8_t* bp = hb + row_off();
#pragma unroll
            for (int d = 0; d < 3; ++d) a[d] = At[(st * 3 + d) * 64 + lane];
#pragma unroll
            for (int v = 0; v < 4; ++v) xv[v] = *reinterpret_cast<const uint4*>(bp + v * cstride);
            advance();
        };
        int2 offA = make_int2(0, 0), offB = make_int2(0, 0);   // kW24: halo offsets of this lane's two pieces
        if constexpr (kW24) {
            offA = *reinterpret_cast<const int2*>(offtab + 2 * q);
            offB = *reinterpret_cast<const int2*>(offtab + 8 + 2 * q);
        }
        // kW24, -DSN_LIN_SETS=3: three register sets -- the operands of step st + 2 are requested before the MFMAs of step
        // st issue.  [measured, round 4] no faster than two (43.9 against 43.4 us at C2, 253 VGPRs): the third of their time
        // the waves of a SIMD pair are parked on the next step's reads is LDS throughput (eight waves x 13 reads a step),
        // not latency that a longer look-ahead would cover.  Kept as a build option.
        constexpr int kSets = kW24 ? SN_LIN_SETS : 2;
        auto load_step24 = [&](int st, uint4 (&a)[3], uint4 (&xv)[4], int2& off) {
            const uint8_t* b0 = hb + off.x;
            const uint8_t* b1 = hb + off.y;
            // (floor experiments, -DSN_CONV_DEBUG builds only, tools/debug/lin_floor.py: 32 = the A table is read for the
            // first two steps only, 64 = the halo operands likewise -- the MFMAs then run on stale registers)
            if (!(SN_DBG(s, 32) && st > 1)) {
                const int stc = st < s.nsteps ? st : s.nsteps;   // (three sets request up to two steps past the end: the zero step)
#pragma unroll
                for (int d = 0; d < 3; ++d) a[d] = At[(stc * 3 + d) * 64 + lane];
            }
            if (!(SN_DBG(s, 64) && st > 1)) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(b0 + v * cstride);
                    const uint2 hi = *reinterpret_cast<const uint2*>(b1 + v * cstride);
                    xv[v] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            // for this set's next use (kSets steps on; past the table: the zero steps' offsets, which are 0)
            const int nx = st + kSets < s.nsteps + 2 ? st + kSets : s.nsteps + 2;
            off = *reinterpret_cast<const int2*>(offtab + 8 * nx + 2 * q);
        };
        auto mma_step = [&](const uint4 (&a)[3], const uint4 (&xv)[4]) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    acc[d][v] = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const i32x4*)&a[d], *(const i32x4*)&xv[v],
                                                                      acc[d][v], 0, 0, 0);
        };
        // the table holds an even number of steps plus one all-zero step, so the loop body is branch-free: a join
        // between "prefetched" and "did not prefetch" paths would make hipcc wait for the NEW loads (lgkmcnt(3))
        if constexpr (kW24) load_step24(0, aA, xA, offA);
        else load_step(0, aA, xA);
        __builtin_amdgcn_sched_barrier(0);
        // per step: the 7 LDS requests of the next step are spread between this step's 12 MFMAs (2 MFMAs, 1 read, ...)
#define SN_LIN_G(NR)                                                                                               \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#define SN_LIN_INTERLEAVE()                                                                                        \
    if constexpr (kW24) { SN_LIN_G(2) SN_LIN_G(2) SN_LIN_G(2) SN_LIN_G(2) SN_LIN_G(2) SN_LIN_G(2) }   /* 12 reads */     \
    else { SN_LIN_G(2) SN_LIN_G(1) SN_LIN_G(1) SN_LIN_G(1) SN_LIN_G(1) SN_LIN_G(1) }                  /* 7 reads */
#define SN_LIN_LOAD(ST, A, X, OFF)                                                                                 \
    if constexpr (kW24) load_step24((ST), A, X, OFF);                                                             \
    else load_step((ST), A, X);
#define SN_LIN_PAIR(ST)                                                                                            \
    SN_LIN_LOAD((ST) + 1, aB, xB, offB)                                                                           \
    mma_step(aA, xA);                                                                                             \
    SN_LIN_INTERLEAVE()                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    SN_LIN_LOAD((ST) + 2, aA, xA, offA)                                                                           \
    mma_step(aB, xB);                                                                                             \
    SN_LIN_INTERLEAVE()                                                                                           \
    __builtin_amdgcn_sched_barrier(0);
        int st0 = 0;
        if constexpr (kSets == 3) {
            uint4 aC[3], xC[4];
            int2 offC = *reinterpret_cast<const int2*>(offtab + 16 + 2 * q);
            load_step24(1, aB, xB, offB);
            __builtin_amdgcn_sched_barrier(0);
#define SN_LIN_TRIPLE(ST)                                                                                          \
    load_step24((ST) + 2, aC, xC, offC);                                                                          \
    mma_step(aA, xA);                                                                                             \
    SN_LIN_INTERLEAVE()                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    load_step24((ST) + 3, aA, xA, offA);                                                                          \
    mma_step(aB, xB);                                                                                             \
    SN_LIN_INTERLEAVE()                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    load_step24((ST) + 4, aB, xB, offB);                                                                          \
    mma_step(aC, xC);                                                                                             \
    SN_LIN_INTERLEAVE()                                                                                           \
    __builtin_amdgcn_sched_barrier(0);
#define SN_EPI1(K) epi_item(std::integral_constant<int, (K)>{});
            if (defer && nst >= 15) {   // block-uniform; the pending tile's outputs ride along the first 15 steps
                SN_LIN_TRIPLE(0)  SN_EPI1(0) SN_EPI1(1) SN_EPI1(2)
                SN_LIN_TRIPLE(3)  SN_EPI1(3) SN_EPI1(4) SN_EPI1(5)
                SN_LIN_TRIPLE(6)  SN_EPI1(6) SN_EPI1(7) SN_EPI1(8)
                SN_LIN_TRIPLE(9)  SN_EPI1(9) SN_EPI1(10) SN_EPI1(11)
                SN_LIN_TRIPLE(12) SN_EPI1(12) SN_EPI1(13) SN_EPI1(14) SN_EPI1(15)
                st0 = 15;
            }
            int st = st0;
            for (; st + 2 < nst; st += 3) {
                SN_LIN_TRIPLE(st)
            }
            if (st < nst) mma_step(aA, xA);       // one or two steps left: requested by the last triple (block-uniform)
            if (st + 1 < nst) mma_step(aB, xB);
#undef SN_LIN_TRIPLE
#undef SN_EPI1
        } else {
        if (defer && nst >= 16) {   // block-uniform; the pending tile's outputs ride along the first 16 steps
            SN_LIN_PAIR(0)  SN_EPI2(0)
            SN_LIN_PAIR(2)  SN_EPI2(2)
            SN_LIN_PAIR(4)  SN_EPI2(4)
            SN_LIN_PAIR(6)  SN_EPI2(6)
            SN_LIN_PAIR(8)  SN_EPI2(8)
            SN_LIN_PAIR(10) SN_EPI2(10)
            SN_LIN_PAIR(12) SN_EPI2(12)
            SN_LIN_PAIR(14) SN_EPI2(14)
            st0 = 16;
        }
        for (int st = st0; st + 1 < nst; st += 2) {
            SN_LIN_PAIR(st)
        }
        if (nst & 1) mma_step(aA, xA);   // odd step count: the last step was loaded by the last pair (block-uniform)
        }
#undef SN_LIN_PAIR
#undef SN_LIN_LOAD
#undef SN_LIN_INTERLEAVE
#undef SN_LIN_G
        // ---- epilogue: D[m = 4 q + i][n]: lane holds 4 consecutive y of row x0 + n
        const int gz = z0 + lz, gx = x0 + n;
        const bool inside = gz < s.Z && gx < s.X && !SN_DBG(s, 4);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int low = acc[1][v][i] * 256 + acc[0][v][i];
                pv[4 * v + i] = fmaf((float)acc[2][v][i], 65536.0f, (float)low) * scale;
            }
        pend = inside;
        pgy = y0 + 4 * q;
        pout = out + (size_t)b * V + ((size_t)(inside ? gz : 0) * s.X + (inside ? gx : 0)) * s.Y + pgy;
        if (!defer) {
            SN_EPI2(0) SN_EPI2(2) SN_EPI2(4) SN_EPI2(6) SN_EPI2(8) SN_EPI2(10) SN_EPI2(12) SN_EPI2(14)
            pend = false;
        }
#ifdef SN_CONV_TIMING
        if (lane == 0) g_lin_t[blockIdx.x * 16 + 8 + wave] += wall_clock64() - t_tile;
#endif
        __syncthreads();   // every wave is done with the halo before the next tile overwrites it
    }
    if (defer) {   // the last tile's outputs
        SN_EPI2(0) SN_EPI2(2) SN_EPI2(4) SN_EPI2(6) SN_EPI2(8) SN_EPI2(10) SN_EPI2(12) SN_EPI2(14)
    }
#undef SN_EPI2
#ifdef SN_CONV_TIMING
    if (tid == 0) g_lin_t[blockIdx.x * 16 + 2] = wall_clock64();
#endif
}

// The tables into a blob, by one workgroup: what every workgroup of conv_lin_i8_kernel otherwise builds for itself
// ([measured] C2: 8.5 of the kernel's ~50 us; tools/debug/lin_floor.py).
template <bool kW24>
__global__ __launch_bounds__(kThreads) void conv_lin_prep_kernel(const float* __restrict__ bank,
                                                                 const float* __restrict__ lambdas, LinShape s,
                                                                 uint8_t* __restrict__ blob) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* misc = reinterpret_cast<float*>(lds);          // [64]
    float* kstar = misc + 64;
    const LinBlob lb = lin_blob_layout(s.nsteps, kW24);
    lin_tables<kW24>(bank, lambdas, s, reinterpret_cast<uint4*>(blob), reinterpret_cast<int*>(blob + lb.tab_off),
                     reinterpret_cast<float*>(blob + lb.tail_off), misc, kstar,
                     reinterpret_cast<int32_t*>(blob + lb.tail_off + 4), true);
}

int num_cus() {
    static thread_local int cached = 0;
    if (!cached) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cached = p.multiProcessorCount;
        if (cached <= 0) cached = 256;
    }
    return cached;
}

}  // namespace

namespace {
// The shape plan, shared by the launch and by sn_conv_fused_supported: false when the shape is outside this kernel.
}  // namespace

extern "C" int sn_conv_fused_supported(int B, int Z, int X, int Y, int kz, int kx, int ky) {
    LinShape s;
    size_t lds = 0;
    bool w24 = false;
    return lin_plan(B, Z, X, Y, 1, kz, kx, ky, s, lds, w24) ? 1 : 0;
}

// returns SN_ERR_UNSUPPORTED (without touching the error text) when the shape is outside this kernel
int sn::conv_fused_lin(const uint8_t* x, const float* bank, const float* lambdas, int B, int Z, int X, int Y, int G,
                       int kz, int kx, int ky, void* out, int out_dtype, hipStream_t stream, int32_t* verdict,
                       bool assume_served, const void* prep) {
    if (((uintptr_t)x % 4) || ((uintptr_t)out % (out_dtype == SN_BF16 ? 8 : 16))) return SN_ERR_UNSUPPORTED;
    LinShape s;
    size_t lds = 0;
    bool w24 = false;
    if (!lin_plan(B, Z, X, Y, G, kz, kx, ky, s, lds, w24)) return SN_ERR_UNSUPPORTED;
    s.gate = sn::current_gate();
    int grid = num_cus();
    if (grid > s.ntiles) grid = s.ntiles;
    s.dbg = sn::debug_env_int("SN_CONV_LIN_DBG");   // (0 in the product: common.h)
    s.tol = sn::option_conv_i8_tolerance();
    if (out_dtype == SN_BF16) s.tol = 0.0f;   // bf16 storage rounds at 2^-9: the 24-bit fixed point is not what limits it
    s.route = nullptr;
    s.prep = static_cast<const uint8_t*>(prep);
    s.served = assume_served ? 1 : 0;
    s.sticky = sn::sticky_device_ptr(stream);
    if (prep) {
        // the verdict was written with the tables (at the tolerance in force then): the gated launches read it there
        s.route = s.tol > 0.0f ? reinterpret_cast<int32_t*>(const_cast<uint8_t*>(s.prep) + lin_blob_layout(s.nsteps, w24).tail_off + 4)
                               : nullptr;
    } else if (s.tol > 0.0f) {
        s.route = verdict ? verdict : sn::device_flag_slot(stream);   // (a caller-owned word can be read back: sn_conv_fused_v)
        if (!s.route) s.tol = 0.0f;   // no flag memory: run unguarded rather than fail
    }
#define SN_LAUNCH_LIN(OT)                                                                                         \
    do {                                                                                                          \
        auto kern = w24 ? conv_lin_i8_kernel<OT, true> : conv_lin_i8_kernel<OT, false>;                           \
        if (sn::ensure_dynamic_lds((const void*)kern, kMaxLds) != hipSuccess)                                     \
            return sn::check_launch("sn_conv_fused(hipFuncSetAttribute)");                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, x, bank, lambdas, s, (OT*)out);         \
    } while (0)
    if (out_dtype == SN_F32) SN_LAUNCH_LIN(float);
    else if (out_dtype == SN_F64) SN_LAUNCH_LIN(double);
    else if (out_dtype == SN_BF16) SN_LAUNCH_LIN(__bf16);
    else return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused: out_dtype %d", out_dtype);
#undef SN_LAUNCH_LIN
    if (int rc = sn::check_launch("sn_conv_fused")) return rc;
    // assume_served: the caller has READ this verdict (0) for these very weights, coefficients and tolerance -- it depends on
    // nothing else -- so the gated launches would do nothing and are left out (the kernel still writes its verdict)
    if (s.route && !assume_served) {   // the 16-kernel contraction on the fp32 matrix pipe, enqueued behind: runs only if the guard sent it there
        sn::GateScope guard(s.route, 1);
        const size_t ntaps = (size_t)kz * kx * ky;
        for (int g0 = 0; g0 < G; g0 += 16) {
            const int gc = (G - g0 < 16) ? G - g0 : 16;
            const int head = (g0 > 0 ? 1 : 0) | (g0 + gc >= G ? 2 : 0);
            const int rc = sn::conv_bank_group(x, SN_U8, bank + g0 * ntaps, lambdas + g0, B, Z, X, Y, gc, G, g0, head, kz, kx,
                                               ky, nullptr, out, out_dtype, reinterpret_cast<sn_stream_t>(stream));
            if (rc != SN_OK) return rc;
        }
    }
    return SN_OK;
}

// float grid -> bytes (x != 0) + "not binary" flag; 4 elements per thread iteration
template <typename T>
__global__ __launch_bounds__(256) void binarize_kernel(const T* __restrict__ x, size_t n, uint8_t* __restrict__ occ,
                                                       int32_t* __restrict__ not_binary) {
    bool bad = false;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        T v[4];
        if constexpr (sizeof(T) == 4) {
            const float4 f = reinterpret_cast<const float4*>(x)[i];
            v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        } else {
            const double2 a = reinterpret_cast<const double2*>(x)[2 * i], b = reinterpret_cast<const double2*>(x)[2 * i + 1];
            v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
        }
        uint32_t o = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o |= (v[j] != (T)0 ? 1u : 0u) << (8 * j);
            bad |= !(v[j] == (T)0 || v[j] == (T)1);   // NaN is "bad" too
        }
        reinterpret_cast<uint32_t*>(occ)[i] = o;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            occ[i] = x[i] != (T)0;
            bad |= !(x[i] == (T)0 || x[i] == (T)1);
        }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(not_binary, 1);
}

extern "C" int sn_forward_auto(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X,
                               int Y, int G, int kz, int kx, int ky, uint8_t* occ_ws, int32_t* not_binary, void* out,
                               int out_dtype, sn_stream_t stream) {
    if (!x || !bank || !lambdas || !occ_ws || !not_binary || !out)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_forward_auto: null pointer");
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || G <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_forward_auto: non-positive extent");
    if (x_dtype != SN_F32 && x_dtype != SN_F64)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_forward_auto: x must be SN_F32 or SN_F64 (byte grids: sn_conv_fused)");
    if ((uintptr_t)x % 16 || (uintptr_t)occ_ws % 4)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_forward_auto: x must be 16-byte, occ_ws 4-byte aligned");
    hipStream_t s = sn::as_stream(stream);
    const size_t n = (size_t)B * Z * X * Y;
    if (hipMemsetAsync(not_binary, 0, sizeof(int32_t), s) != hipSuccess) return sn::check_launch("sn_forward_auto(memset)");
    const int blocks = (int)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 + 1 : 2048);
    if (x_dtype == SN_F32)
        hipLaunchKernelGGL(binarize_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, n, occ_ws, not_binary);
    else
        hipLaunchKernelGGL(binarize_kernel<double>, dim3(blocks), dim3(256), 0, s, (const double*)x, n, occ_ws, not_binary);
    if (int rc = sn::check_launch("sn_forward_auto(binarize)")) return rc;
    {   // binary grid: the int8 path on the bytes (through linearity where the shape allows, else the contraction)
        sn::GateScope gate(not_binary, 0);
        int rc = sn::conv_fused_lin(occ_ws, bank, lambdas, B, Z, X, Y, G, kz, kx, ky, out, out_dtype, s);
        if (rc == SN_ERR_UNSUPPORTED) {
            const size_t ntaps = (size_t)kz * kx * ky;
            for (int g0 = 0; g0 < G; g0 += 16) {
                const int gc = (G - g0 < 16) ? G - g0 : 16;
                const int head = (g0 > 0 ? 1 : 0) | (g0 + gc >= G ? 2 : 0);
                rc = sn::conv_bank_group(occ_ws, SN_OCC8, bank + g0 * ntaps, lambdas + g0, B, Z, X, Y, gc, G, g0, head, kz,
                                         kx, ky, nullptr, out, out_dtype, stream);
                if (rc != SN_OK) break;
            }
        }
        if (rc != SN_OK) return rc;
    }
    {   // anything else: the fp32 contraction on x itself
        sn::GateScope gate(not_binary, 1);
        const size_t ntaps = (size_t)kz * kx * ky;
        for (int g0 = 0; g0 < G; g0 += 16) {
            const int gc = (G - g0 < 16) ? G - g0 : 16;
            const int head = (g0 > 0 ? 1 : 0) | (g0 + gc >= G ? 2 : 0);
            const int rc = sn::conv_bank_group(x, x_dtype, bank + g0 * ntaps, lambdas + g0, B, Z, X, Y, gc, G, g0, head, kz,
                                               kx, ky, nullptr, out, out_dtype, stream);
            if (rc != SN_OK) return rc;
        }
    }
    return SN_OK;
}

extern "C" int sn_conv_fused(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X,
                             int Y, int G, int kz, int kx, int ky, void* out, int out_dtype, sn_stream_t stream) {
    if (!x || !bank || !lambdas || !out) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused: null pointer");
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || G <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused: non-positive extent");
    if (x_dtype != SN_OCC8)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_fused: binary occupancy (SN_OCC8) input only; use sn_conv_bank");
    const int rc = sn::conv_fused_lin((const uint8_t*)x, bank, lambdas, B, Z, X, Y, G, kz, kx, ky, out, out_dtype,
                                      sn::as_stream(stream));
    if (rc == SN_ERR_UNSUPPORTED)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_fused: shape outside the kernel (Y %% 4, ky window, LDS); use "
                                            "sn_conv_bank");
    return rc;
}

extern "C" size_t sn_conv_fused_prep_bytes(int kz, int kx, int ky) {
    LinShape s;
    size_t lds = 0;
    bool w24 = false;
    if (!lin_plan(1, TZ, TX, TY, 1, kz, kx, ky, s, lds, w24)) return 0;
    return lin_blob_layout(s.nsteps, w24).bytes;
}

// the tables of (bank, lambdas) into `blob` (sn_conv_fused_prep_bytes, 16-byte aligned), with the guard's verdict at the
// tolerance in force now: one workgroup
int sn::conv_fused_prep_launch(const float* bank, const float* lambdas, int G, int kz, int kx, int ky, void* blob,
                               hipStream_t stream) {
    LinShape s;
    size_t lds = 0;
    bool w24 = false;
    if (!lin_plan(1, TZ, TX, TY, G, kz, kx, ky, s, lds, w24)) return SN_ERR_UNSUPPORTED;
    s.gate = sn::current_gate();
    s.tol = sn::option_conv_i8_tolerance();
    s.route = nullptr;
    s.prep = nullptr;
    s.dbg = 0;
    const size_t scratch = 256 + (((size_t)kz * kx * ky + 3) & ~(size_t)3) * sizeof(float) + (size_t)kz * kx * 3 * 17 * 4 + 16;
    if (w24) hipLaunchKernelGGL(conv_lin_prep_kernel<true>, dim3(1), dim3(kThreads), scratch, stream, bank, lambdas, s, (uint8_t*)blob);
    else hipLaunchKernelGGL(conv_lin_prep_kernel<false>, dim3(1), dim3(kThreads), scratch, stream, bank, lambdas, s, (uint8_t*)blob);
    return sn::check_launch("sn_conv_fused_prep");
}

extern "C" int sn_conv_fused_prep(const float* bank, const float* lambdas, int G, int kz, int kx, int ky, void* blob,
                                  sn_stream_t stream) {
    if (!bank || !lambdas || !blob) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused_prep: null pointer");
    if (G <= 0 || kz <= 0 || kx <= 0 || ky <= 0) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused_prep: non-positive extent");
    if ((uintptr_t)blob % 16) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused_prep: blob must be 16-byte aligned");
    const int rc = sn::conv_fused_prep_launch(bank, lambdas, G, kz, kx, ky, blob, sn::as_stream(stream));
    if (rc == SN_ERR_UNSUPPORTED)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_fused_prep: kernel %d x %d x %d outside the combined kernel", kz, kx, ky);
    return rc;
}

extern "C" int sn_conv_fused_prepared(const void* x, int x_dtype, const float* bank, const float* lambdas, const void* blob,
                                      int B, int Z, int X, int Y, int G, int kz, int kx, int ky, void* out, int out_dtype,
                                      int assume_served, sn_stream_t stream) {
    if (!x || !bank || !lambdas || !out || !blob) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused_prepared: null pointer");
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || G <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused_prepared: non-positive extent");
    if (x_dtype != SN_OCC8)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_fused_prepared: binary occupancy (SN_OCC8) input only; use sn_conv_bank");
    if ((uintptr_t)blob % 16) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused_prepared: blob must be 16-byte aligned");
    const int rc = sn::conv_fused_lin((const uint8_t*)x, bank, lambdas, B, Z, X, Y, G, kz, kx, ky, out, out_dtype,
                                      sn::as_stream(stream), nullptr, assume_served != 0, blob);
    if (rc == SN_ERR_UNSUPPORTED)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_fused_prepared: shape outside the kernel (Y %% 4, ky window, LDS); use "
                                            "sn_conv_bank");
    return rc;
}

extern "C" int sn_conv_fused_v(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X,
                               int Y, int G, int kz, int kx, int ky, void* out, int out_dtype, int32_t* verdict,
                               int assume_served, sn_stream_t stream) {
    if (!x || !bank || !lambdas || !out || !verdict) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused_v: null pointer");
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || G <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_fused_v: non-positive extent");
    if (x_dtype != SN_OCC8)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_fused_v: binary occupancy (SN_OCC8) input only; use sn_conv_bank");
    const int rc = sn::conv_fused_lin((const uint8_t*)x, bank, lambdas, B, Z, X, Y, G, kz, kx, ky, out, out_dtype,
                                      sn::as_stream(stream), verdict, assume_served != 0);
    if (rc == SN_ERR_UNSUPPORTED)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_conv_fused_v: shape outside the kernel (Y %% 4, ky window, LDS); use "
                                            "sn_conv_bank");
    return rc;
}

#ifdef SN_CONV_TIMING
extern "C" void sn_debug_lin_times(unsigned long long* host) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lin_t), sizeof(unsigned long long) * 1024 * 16);
}
#endif
