// K3'' -- GENEO bank convolution for BINARY OCCUPANCY on the int8 matrix cores, ky = 9: stride-4 voxel map.
//
// Same contraction and the same fixed-point arithmetic as csrc/conv_i8.hip (SceneNet.forward,
// core/models/SCENE_Net.py:322-339 on ToFullDense(Voxelization(points)), torch_transforms.py:33-34): per kernel
// 24-bit fixed-point weights in three balanced base-256 digits, three v_mfma_i32_16x16x64_i8 per 64 K-slots, exact
// int32 sums, fp32 recombination and head.  What differs is how the B operand (the im2col of the occupancy bytes)
// is fed, which decides both the number of MFMAs and everything around them:
//
//   * lane (q, n) of accumulator tile (h, r) is voxel  y = y0 + 4 n + r  of x-row h  (r = 0..3: four "residue"
//     tiles cover 64 consecutive y).  All 16 lanes of a tile then have the SAME byte alignment, and the 9-tap
//     window of a kernel row (dz, dx) is bytes r .. r+8 of the three aligned halo dwords D0 D1 D2 = dwords n, n+1,
//     n+2 of the halo row -- shared by the four residues.  The operand dword for taps 4c..4c+3 is
//     v_alignbyte(D[c+1], D[c], r): ONE halo copy in LDS (1 byte per voxel) instead of four byte-shifted ones, and
//     a quarter of the LDS reads;
//   * tap 8 of a row is byte r of D2.  The tap-8 bytes of FOUR kernel rows are packed into one K dword by a 4x4
//     byte transpose of their D2's (8 v_perm_b32 give the dwords of all four residues), so a kernel row costs
//     2 1/4 K-dwords, not 3:  81 rows -> 183 dwords -> 12 MFMA steps of 16 dwords instead of 16 (-25 % MFMAs);
//   * the 65 KiB the shifted copies took hold a RING of three halo buffers filled by LDS-DMA two tiles ahead.
//     A wave moves from tile to tile on two LDS counters per buffer (landed / done) -- no workgroup barrier in the
//     tile loop --, and rounds are claimed from an LDS ticket counter, so the two waves of a SIMD drift apart by
//     themselves and one's epilogue meets the other's MFMAs (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).
//
// Cost model ([measured], tools/micro/mfma_valu_mix.hip and DESIGN.md section 4): a launch takes about 19.5 cycles per
// MFMA plus 2.5 cycles per VALU instruction, summed over both waves of a SIMD -- VALU work is NOT hidden behind the
// partner's MFMAs.  Hence the instruction-count discipline below: operand shifts issued four at a time behind running
// MFMAs, tail offsets read in batches, the head mixed straight from the integer sums with packed FMAs, the wave index
// kept scalar, the head's tanh on v_exp / v_rcp.
//
// Quantisation (per kernel g): S_g = 8355711 / max|W_g| (8355711 = 127 * 65793, the largest magnitude three
// balanced digits hold), Q = rint(W * S_g) in fp64, act = S * (max|W_g| / 8355711).  The prologue also computes the
// exact worst case of the quantisation error over all binary inputs, max(sum of positive errors, sum of negative
// errors), per kernel and lambda-weighted; a bank whose bound exceeds the tolerance is not run here: every
// workgroup returns and *route = 1 sends the launch to the fp32 kernel enqueued behind this one (conv.hip).
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace {

#include "conv_prep.h"
#include "conv_fp32.inc"   // fp32k::conv_bank_body: the fp32 form, for the combined fallback launch (conv_i8_fallback_kernel)

using i32x4 = __attribute__((ext_vector_type(4))) int;

// relu(tanh(v)): 0 for v <= 0, else 1 - 2 / (exp(2v) + 1) on the hardware exp / rcp (abs. error ~2e-7, inside the 1e-4
// bar; same form as conv_lin.hip).  NaN stays NaN like torch.relu(torch.tanh(.)); +inf -> 1.
__device__ __forceinline__ float relu_tanh(float v) {
    // branch-free, v_exp_f32 + v_rcp_f32 (1 ulp; the correctly rounded reciprocal was ten instructions, twice per lane and round)
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f);
    const float r = (v > 0.0f) ? t : 0.0f;
    return (v != v) ? v : r;
}

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int TY = 64;          // y extent of a workgroup tile: 16 lanes x 4 residues
constexpr int NV = 8;           // accumulator tiles per wave round: 2 x-rows x 4 residues
constexpr int DW = 24;          // halo row stride in dwords (96 bytes = 16 + 64 + 16): rows 2 (mod 4) apart differ by 16 banks
constexpr int PYA = 16;         // halo origin y0 - 16: every 16-byte piece of a row is wholly inside or outside the grid
constexpr int D0 = PYA / 4 - 1; // dword of a row that holds tap 0 of lane 0, residue 0 (py = 4)
constexpr int kRowPieces = DW / 4;             // 16-byte pieces per halo row
constexpr int kDmaRows = 64 / kRowPieces;      // halo rows one LDS-DMA wave-instruction moves (10; 4 lanes idle)
constexpr int kNB = 3;          // halo ring
constexpr int kMaxRQ = 24;      // kernel rows per lane group (kz * kx <= 96)
constexpr int kMaxLds = 160 * 1024;
constexpr int kSpinMax = 1 << 22;

// which kernel rows (dz, dx) lane group q carries, in what order (host-built: plan_rows)
struct RowPlan {
    short halo[4][kMaxRQ];   // halo row index dz * XP + dx (pad entries: any valid row)
    short tap[4][kMaxRQ];    // kernel row index dz * kx + dx, -1 = pad (zero weights)
};

// The folded kernel (conv_occ_i8f_kernel): kernel rows (dz, dx') of a bank that is symmetric in x and y, dx' = 0..4.
// kFoldSteps MFMA steps of kFoldRows folded rows per lane group; slot (st, j) is SINGLE (dx' = 4: one halo row) for j = 2 of
// steps 0..2 and DOUBLE (dx' < 4: halo rows dx' and 8 - dx' summed) elsewhere -- the same for all four lane groups, so the
// step code has no per-lane case.
#ifndef SN_I8F_AHEAD
#define SN_I8F_AHEAD 2
#endif
// (kFoldSteps, kFoldRows, kFoldSlots, fold_slot_*: conv_prep.h)
struct FoldPlan {
    short h1[4][kFoldSlots];    // halo row index dz * XP + dx'
    short h2[4][kFoldSlots];    // halo row index dz * XP + 8 - dx'   (single slots: unused)
    short tap[4][kFoldSlots];   // kernel row dz * 9 + dx', -1 = pad (zero weights)
};

struct Shape {
    int B, Z, X, Y, G;
    int kz, kx;
    int TZ, TX, nzt, nxt, nyt, ntiles;
    int XP, ZP;            // halo rows per z plane (incl. pad), halo planes
    int RQ, NP, NT, ODD;   // rows per lane group; pair steps; tap-8 quads; RQ odd (last row's chunks in the tail)
    int NTS, KS;           // tail steps, all steps
    int Gtot, g0, head;    // kernel group of a larger bank (see conv.hip)
    sn::Gate gate;
    int32_t* route;        // out: 1 = quantisation bound exceeded, fp32 kernel takes the launch; 0 = done here
    float tol;             // bound on the worst-case activation error allowed on the int8 path (<= 0: no check)
    int stagger;           // s_sleep units (64 clocks) waves 4-7 wait once before their first round
    int dynamic;           // rounds of a tile are claimed from an LDS ticket counter (else dealt: wave, wave + 8, ...)
    int dbg;
    int32_t* sticky;       // the device's sticky status words (common.h): a spin that gives up latches code 3
    RowPlan plan;
    FoldPlan fplan;        // conv_occ_i8f_kernel only
};

struct TileCoord {
    int b, z0, x0, y0;
};

__device__ __forceinline__ TileCoord tile_coord(const Shape& s, int tile) {
    TileCoord c;
    c.y0 = (tile % s.nyt) * TY; tile /= s.nyt;
    c.x0 = (tile % s.nxt) * s.TX; tile /= s.nxt;
    c.z0 = (tile % s.nzt) * s.TZ; tile /= s.nzt;
    c.b = tile;
    return c;
}

__device__ __attribute__((aligned(16))) uint32_t g_zero_word_s[4] = {0u, 0u, 0u, 0u};
// diagnostics (sn_conv_i8_path_counts): launches the folded kernel served, declined (bank not symmetric), sent to fp32
__device__ unsigned long long g_fold_counts[4] = {0ull, 0ull, 0ull, 0ull};

__device__ __forceinline__ float load_now(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ double load_now(const double* p) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// workgroup barrier that orders LDS traffic only: LDS-DMA in flight stays in flight
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS-DMA of one tile's raw halo rows into a ring buffer, 16 bytes per lane: one wave-instruction moves TEN whole halo
// rows (60 lanes; lane -> (row lane / 6, piece lane % 6) is the same for every instruction), instructions are dealt
// round-robin over the waves, and from one to the next every lane advances by 80 rows: the row split r -> (zz, xx) is
// one add and a conditional wrap, no division.  [measured] item = linear dword index, 4 bytes per lane: 12 instructions
// and 1.9 us per wave and tile; this form: 4 instructions.  Pieces outside the grid are fetched from a zero block (with
// Y % 16 == 0 and the origin at y0 - 16 a piece never straddles the grid's edge).  Inline asm: the builtin's LDS write
// would make hipcc wait vmcnt(0) before every later ds_read (see conv_i8.hip); arrival is tracked by this wave's own
// vmcnt(0) + the `landed` counter.
struct DmaLane {
    int rl, i;        // row within an instruction (kDmaRows = idle lane), 16-byte piece within the row
    int zz0, xx0;     // (zz, xx) of this lane's row in the wave's first instruction
    int dzz, dxx;     // advance per instruction: kDmaRows * kWaves rows = dzz planes + dxx rows
};
__device__ __forceinline__ DmaLane dma_lane(const Shape& s, int wave, int lane) {
    DmaLane d;
    d.rl = lane / kRowPieces;
    d.i = lane - d.rl * kRowPieces;
    const int r0 = kDmaRows * wave + d.rl;
    d.zz0 = r0 / s.XP;
    d.xx0 = r0 - d.zz0 * s.XP;
    d.dzz = (kDmaRows * kWaves) / s.XP;
    d.dxx = kDmaRows * kWaves - d.dzz * s.XP;
    return d;
}
__device__ __forceinline__ void halo_dma_issue(uint32_t* __restrict__ buf, const uint8_t* __restrict__ x,
                                               const Shape& s, const TileCoord& c, const DmaLane& d, int wave) {
    const int hrows = s.ZP * s.XP;
    const uint8_t* tb = x + (size_t)c.b * s.Z * s.X * s.Y;
    const int oz = c.z0 - (s.kz - 1) / 2, ox = c.x0 - (s.kx - 1) / 2;
    const int gy = c.y0 - PYA + 16 * d.i;
    const bool oky = d.rl < kDmaRows && (unsigned)gy < (unsigned)s.Y;
    int zz = d.zz0, xx = d.xx0;
    for (int r0 = kDmaRows * wave; r0 < hrows; r0 += kDmaRows * kWaves) {
        const int gz = oz + zz, gx = ox + xx;
        const bool ok = oky && (unsigned)gz < (unsigned)s.Z && (unsigned)gx < (unsigned)s.X;
        const void* src = ok ? static_cast<const void*>(tb + ((gz * s.X + gx) * s.Y + gy))
                             : static_cast<const void*>(g_zero_word_s);
        const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(buf + r0 * DW);
        const uint32_t lds_uni = __builtin_amdgcn_readfirstlane(lds_base);
        // idle lanes and rows past the end of the buffer stay out (EXEC): their piece would land in the next ring slot
        if (d.rl < kDmaRows && r0 + d.rl < hrows) {
            uint32_t m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "s"(lds_uni), "v"(src) : "memory");
        }
        zz += d.dzz;
        xx += d.dxx;
        const bool w = xx >= s.XP;   // dxx < XP: one wrap at most
        xx -= w ? s.XP : 0;
        zz += w;
    }
}

// LDS counters between the waves of the workgroup (monotonic; one add per wave per event)
__device__ __forceinline__ void wave_signal(int* c, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool wave_wait(const int* c, int expect) {
    int spins = 0;
    // (read in the LDS address space: a generic volatile pointer makes hipcc emit a FLAT load + s_waitcnt vmcnt(0))
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile __attribute__((address_space(3))) int*>(
               (const __attribute__((address_space(3))) int*)c)) < expect) {
        if (++spins > kSpinMax) return false;   // cannot happen (every wave reaches its signals); never hang the GPU
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
    return true;
}

// 4x4 byte transpose: o[r] = (a.byte r, b.byte r, c.byte r, d.byte r)
__device__ __forceinline__ void transpose4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t (&o)[4]) {
    const uint32_t p0 = __builtin_amdgcn_perm(b, a, 0x05010400u);   // a0 b0 a1 b1
    const uint32_t p1 = __builtin_amdgcn_perm(b, a, 0x07030602u);   // a2 b2 a3 b3
    const uint32_t q0 = __builtin_amdgcn_perm(d, c, 0x05010400u);   // c0 d0 c1 d1
    const uint32_t q1 = __builtin_amdgcn_perm(d, c, 0x07030602u);   // c2 d2 c3 d3
    o[0] = __builtin_amdgcn_perm(q0, p0, 0x05040100u);
    o[1] = __builtin_amdgcn_perm(q0, p0, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(q1, p1, 0x05040100u);
    o[3] = __builtin_amdgcn_perm(q1, p1, 0x07060302u);
}

// 3 x 4 byte transpose with a zero fourth row: o[r] = (a.byte r, b.byte r, c.byte r, 0) -- six v_perm_b32 (selector byte
// 0x0c = the constant 0) instead of the eight of transpose4(a, b, c, 0)
__device__ __forceinline__ void transpose3(uint32_t a, uint32_t b, uint32_t c, uint32_t (&o)[4]) {
    const uint32_t p0 = __builtin_amdgcn_perm(b, a, 0x05010400u);   // a0 b0 a1 b1
    const uint32_t p1 = __builtin_amdgcn_perm(b, a, 0x07030602u);   // a2 b2 a3 b3
    o[0] = __builtin_amdgcn_perm(c, p0, 0x0c040100u);               // a0 b0 c0 0
    o[1] = __builtin_amdgcn_perm(c, p0, 0x0c050302u);               // a1 b1 c1 0
    o[2] = __builtin_amdgcn_perm(c, p1, 0x0c060100u);               // a2 b2 c2 0
    o[3] = __builtin_amdgcn_perm(c, p1, 0x0c070302u);               // a3 b3 c3 0
}

template <int R>
__device__ __forceinline__ uint32_t window(uint32_t hi, uint32_t lo) {   // bytes R .. R+3 of (hi : lo)
    if constexpr (R == 0) return lo;
    else return __builtin_amdgcn_alignbyte(hi, lo, R);
}

// fp32 bank -> LDS, coalesced, one batch of loads; the first two tiles' halos are requested while those loads fly (the
// other way round the bank's loads queue behind 24 DMA pieces per wave: vmcnt retires in order) and travel while the
// tables are built
__device__ __forceinline__ void stage_bank_and_first_halos(const Shape& s, const float* __restrict__ bank, int ntaps,
                                                           float* bank_s, uint32_t* hbuf, int hdw,
                                                           const uint8_t* __restrict__ x, int my_tiles, int tid, int wave,
                                                           int lane) {
    const int nb = s.G * ntaps;
    constexpr int kB = 24;
    for (int base = tid; base < nb; base += kThreads * kB) {
        float v[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const int i = base + u * kThreads;
            v[u] = bank[i < nb ? i : 0];
        }
        if (base == tid) {
            const DmaLane dl = dma_lane(s, wave, lane);
            if (my_tiles > 0) halo_dma_issue(hbuf, x, s, tile_coord(s, blockIdx.x), dl, wave);
            if (my_tiles > 1) halo_dma_issue(hbuf + hdw, x, s, tile_coord(s, blockIdx.x + gridDim.x), dl, wave);
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const int i = base + u * kThreads;
            bank_s[i < nb ? i : nb] = v[u];
        }
    }
}

// Per kernel: max|W|, the fixed-point scale, the exact worst case of the quantisation error, and the fixed-point weight
// Q = rint(W * S) written over W in bank_s (as int bits).  All 16 kernels at once: half a wave each (two passes of whole
// waves took 4.7 us: two dependent reduction chains).
__device__ __forceinline__ void quantise_kernels(const Shape& s, int ntaps, float* bank_s, float* scale, double* bnd,
                                                 int wave, int lane) {
    static_assert(2 * kWaves == 16, "one half wave per kernel");
    const int g = 2 * wave + (lane >> 5), l32 = lane & 31;
    float m = 0.0f;
    if (g < s.G)
        for (int t = l32; t < ntaps; t += 32) {
            const float a = fabsf(bank_s[g * ntaps + t]);
            m = (a <= 3.0e38f) ? fmaxf(m, a) : __int_as_float(0x7fc00000);   // NaN / inf poisons the kernel
        }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        const float u = __shfl_xor(m, o, 64);
        m = (m != m || u != u) ? __int_as_float(0x7fc00000) : fmaxf(m, u);
    }
    const double S = (m > 0.0f) ? kQMax / (double)m : 0.0;   // NaN: comparisons false -> S = 0, scale = NaN below
    const double invS = (double)m / kQMax;   // Q * invS instead of Q / S: 2e-16 relative, nothing next to the errors summed
    double ep = 0.0, en = 0.0;
    if (g < s.G && m > 0.0f)
        for (int t = l32; t < ntaps; t += 32) {
            const double w = (double)bank_s[g * ntaps + t];
            const int Q = __double2int_rn(w * S);
            const double e = (double)Q * invS - w;
            ep += e > 0.0 ? e : 0.0;
            en += e < 0.0 ? -e : 0.0;
            bank_s[g * ntaps + t] = __int_as_float(Q);   // the table builds read the fixed-point weight, not W
        }
    else if (g < s.G)
        for (int t = l32; t < ntaps; t += 32) bank_s[g * ntaps + t] = 0.0f;   // all-zero or poisoned kernel: Q = 0 (scale carries a NaN)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        ep += __shfl_xor(ep, o, 64);
        en += __shfl_xor(en, o, 64);
    }
    if (l32 == 0) {
        scale[g] = (m != m) ? m : (float)((double)m / kQMax);
        bnd[g] = ep > en ? ep : en;
    }
}

// quantise_kernels for a bank that is symmetric in x and y (the folded kernel, after its symmetry check): only the 9 x 5 x 5
// unique taps are visited, each standing for 1, 2 or 4 equal weights -- the same maximum, the same error sums up to the
// order of the fp64 additions, and Q written where the folded table build reads it (dx <= 4, dy <= 4).
__device__ __forceinline__ void quantise_kernels_folded(const Shape& s, float* bank_s, float* scale, double* bnd, int wave,
                                                        int lane) {
    const int g = 2 * wave + (lane >> 5), l32 = lane & 31;
    float sc;
    double bd, qp, qn;
    quantise_folded_half(bank_s + (g < s.G ? g : 0) * 729, g < s.G, l32, sc, bd, qp, qn);
    if (l32 == 0) {
        scale[g] = sc;
        bnd[g] = bd;
    }
}

// the guard's decision: all workgroups take it from the same numbers
template <typename OT, typename SH>
__device__ __forceinline__ bool bound_exceeded(const SH& s, const double* bnd, const float* __restrict__ lambdas,
                                               const OT* act, const OT* out) {
    if (!(s.tol > 0.0f)) return false;
    double worst = 0.0, mixed = 0.0;
    for (int g = 0; g < s.G; ++g) {
        worst = bnd[g] > worst ? bnd[g] : worst;
        if (out) mixed += fabs((double)lambdas[g]) * bnd[g];   // tanh and relu are 1-Lipschitz
    }
    return (act && worst > (double)s.tol) || (out && mixed > (double)s.tol);
}

// One round's epilogue, shared by the kernels of this file: recombine the three digit sums, store the bank activations
// (if requested), mix the 16 kernels into the head.  acc[d][v]: digit plane d of accumulator tile v = 4 h + r.
template <typename OT, typename SH>
__device__ __forceinline__ void finish_round(const SH& s, const TileCoord& c, int lz, int lx, int n, int q,
                                             i32x4 (&acc)[3][NV], const float* scale, const float* lamsc,
                                             const float* lamhi, OT* __restrict__ act, OT* __restrict__ out, size_t V) {
    const int gz = c.z0 + lz;
    if (gz >= s.Z) return;
    if (SN_DBG(s, 1)) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("" ::"v"(acc[d][v]));
        return;
    }
    const int gy4 = c.y0 + 4 * n;   // this lane's four residues: y = gy4 .. gy4 + 3
    float pm[NV];                   // this lane group's share of the head's mix, per tile
    const float4 lam4 = *reinterpret_cast<const float4*>(lamsc + 4 * q);
    const float lam[4] = {lam4.x, lam4.y, lam4.z, lam4.w};
    if (act) {
        // bank activations requested: every (kernel, voxel) value is formed, stored, and mixed
        float val[NV][4];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int low = acc[1][v][r] * 256 + acc[0][v][r];   // |.| < 2^25: exact
                val[v][r] = fmaf((float)acc[2][v][r], 65536.0f, (float)low);
            }
        const float4 sc4 = *reinterpret_cast<const float4*>(scale + 4 * q);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
        if (gy4 < s.Y) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int gx = c.x0 + lx + h;
                if (gx < s.X) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int g = 4 * q + r;
                        if (g < s.G) {
                            OT* o = act + ((size_t)c.b * s.Gtot + s.g0 + g) * V + ((size_t)gz * s.X + gx) * s.Y + gy4;
                            const float v0 = val[4 * h + 0][r] * sc[r], v1 = val[4 * h + 1][r] * sc[r],
                                        v2 = val[4 * h + 2][r] * sc[r], v3 = val[4 * h + 3][r] * sc[r];
                            if constexpr (sizeof(OT) == 4) {
                                *reinterpret_cast<float4*>(o) = make_float4(v0, v1, v2, v3);
                            } else {
                                reinterpret_cast<double2*>(o)[0] = make_double2((double)v0, (double)v1);
                                reinterpret_cast<double2*>(o)[1] = make_double2((double)v2, (double)v3);
                            }
                        }
                    }
                }
            }
        }
    }
    {
        // the head's mix: sum_r lam_r (low_r + 65536 hi_r) as two packed FMAs per kernel and PAIR of tiles, straight
        // from the integer accumulators (the same bits with or without `act`) -- 128 VALU per round instead of 164 (a VALU instruction costs the SIMD about
        // 2.5 cycles next to a busy matrix pipe: tools/micro/mfma_valu_mix.hip)
        // ([measured] round 3: recombining the three digit sums in int32 -- one conversion and one FMA per kernel and voxel,
        // for banks whose weights cannot sum past int32 -- was built in all four kernels and is not faster: z-walk 108.4 vs
        // 107.3 us, folded 124.1 vs 121.6, stride-4 199.6 vs 194.8; two dependent shift-adds replace one conversion)
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        const float4 hi4 = *reinterpret_cast<const float4*>(lamhi + 4 * q);
        const float lhi[4] = {hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
        for (int vp = 0; vp < NV / 2; ++vp) {
            const int v0 = 2 * vp, v1 = 2 * vp + 1;
            f32x2 p = {0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int low0 = acc[1][v0][r] * 256 + acc[0][v0][r], low1 = acc[1][v1][r] * 256 + acc[0][v1][r];
                const f32x2 lo = {(float)low0, (float)low1};
                const f32x2 hi = {(float)acc[2][v0][r], (float)acc[2][v1][r]};
                p = __builtin_elementwise_fma(f32x2{lam[r], lam[r]}, lo, p);
                p = __builtin_elementwise_fma(f32x2{lhi[r], lhi[r]}, hi, p);
            }
            pm[v0] = p.x;
            pm[v1] = p.y;
        }
    }
    if (out) {
        // sum over the four lane groups (the 16 kernels) with half / row swaps: one v_permlane32_swap + add sums
        // TWO tiles over lanes (l, l + 32) -- tile a ends in the lower half, b in the upper --, one
        // v_permlane16_swap + add does the same inside the halves.  Lane group q ends up holding exactly what it
        // stores: x-row q >> 1, residues 2 (q & 1) and 2 (q & 1) + 1 (12 VALU; eight ds_bpermute pairs before).
        auto sum_halves = [&](float a, float b) -> float {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            return __uint_as_float(r[0]) + __uint_as_float(r[1]);
        };
        auto sum_rows = [&](float u, float w) -> float {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(w), false, false);
            return __uint_as_float(r[0]) + __uint_as_float(r[1]);
        };
        const float e0 = sum_rows(sum_halves(pm[0], pm[4]), sum_halves(pm[2], pm[6]));   // rows: tiles 0 2 4 6
        const float e1 = sum_rows(sum_halves(pm[1], pm[5]), sum_halves(pm[3], pm[7]));   // rows: tiles 1 3 5 7
        const int h = q >> 1;
        const int gx = c.x0 + lx + h, gy = gy4 + 2 * (q & 1);
        if (gx < s.X && gy < s.Y) {
            // wave-uniform row base (scalar arithmetic) + a 32-bit lane offset: no 64-bit vector address math per round
            OT* row = out + ((size_t)c.b * V + ((size_t)gz * s.X + (c.x0 + lx)) * s.Y + c.y0);
            OT* o = row + (unsigned)(h * s.Y + 4 * n + 2 * (q & 1));
            float t0 = e0, t1 = e1;
            if (s.head & 1) { t0 += (float)load_now(o); t1 += (float)load_now(o + 1); }
            if (s.head & 2) { t0 = relu_tanh(t0); t1 = relu_tanh(t1); }
            if constexpr (sizeof(OT) == 4) *reinterpret_cast<float2*>(o) = make_float2(t0, t1);
            else *reinterpret_cast<double2*>(o) = make_double2((double)t0, (double)t1);
        }
    }
}

#ifdef SN_CONV_TIMING   // make -B EXTRA=-DSN_CONV_TIMING OUT=build/timing OBJDIR=build/obj_timing; read by tools/i8s_timing.py
__device__ unsigned long long g_i8s_t[1024 * 16];
__device__ unsigned long long g_i8s_w[1024 * 8 * 8];   // [workgroup][wave][phase]: summed wall_clock64 ticks (10 ns)
#define SN_ST(k) do { if (threadIdx.x == 0) g_i8s_t[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define SN_WT0() const unsigned long long t_ph0 = wall_clock64()
#define SN_WT(k, t0) do { if (lane == 0) g_i8s_w[(blockIdx.x * 8 + wave) * 8 + (k)] += wall_clock64() - (t0); } while (0)
#define SN_WNOW() wall_clock64()
#else
#define SN_ST(k) do {} while (0)
#define SN_WT(k, t0) do { (void)(t0); } while (0)
#define SN_WNOW() 0ull
#endif

// kNT >= 0: the step structure (pair steps, tap-8 quads, odd row) is known at compile time -- 9 x 9 kernel rows:
// <10, 6, 1> --, which makes every tail slot's kind and the peeled end of the pair loop static (one code path: with the
// run-time form hipcc keeps a second copy of the 96 accumulator registers across the join and spills); kNT < 0: taken
// from the shape.
// returns true when the quantisation guard sent the launch to the fp32 form (nothing written; *s.route = 1 if s.route)
template <typename OT, int kNP, int kNT, int kODD>
__device__ __forceinline__ bool conv_occ_i8s_body(const uint8_t* __restrict__ x, const float* __restrict__ bank,
                                                  const float* __restrict__ lambdas, const Shape& s,
                                                  OT* __restrict__ act, OT* __restrict__ out) {
    if (!s.gate.pass()) return false;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: what derives from it (a round's rows,
    const int n = lane & 15, q = lane >> 4;                                        // addresses) is then scalar arithmetic, not VALU
    SN_ST(0);

    const int NP = kNT >= 0 ? kNP : s.NP;
    const int NT = kNT >= 0 ? kNT : s.NT;
    const int ODD = kNT >= 0 ? kODD : s.ODD;
    const int NTS = (NT + 2 * ODD + 3) / 4;
    const int ntaps = s.kz * s.kx * 9;
    const int hrows = s.ZP * s.XP;
    const int hdw = hrows * DW;   // dwords per halo buffer
    // LDS carve-up
    uint4* Wd = reinterpret_cast<uint4*>(lds);                                   // [KS][3][64] x 16 B
    int2* roff = reinterpret_cast<int2*>(Wd + (size_t)s.KS * 3 * 64);            // [NP][4]   byte offsets of rows A, B
    int4* toff = reinterpret_cast<int4*>(roff + s.NP * 4);                       // [NT][4]   byte offsets of a quad's rows
    int* oddoff = reinterpret_cast<int*>(toff + s.NT * 4);                       // [4]       byte offset of the odd row
    float* scale = reinterpret_cast<float*>(oddoff + 4);                         // [16]   max|W_g| / 8355711
    float* lamsc = scale + 16;                                                   // [16]   lambda_g * scale_g
    float* lamhi = lamsc + 16;                                                   // [16]   65536 * lambda_g * scale_g (+ 64 bytes spare)
    double* bnd = reinterpret_cast<double*>(lamhi + 32);                                                       // [16]   worst-case error per kernel
    int* landed = reinterpret_cast<int*>(bnd + 16);                              // [kNB]  waves whose DMA pieces are in
    int* done = landed + 4;                                                      // [kNB]  waves finished with the buffer
    int* flags = done + 4;                                                       // [4]    0: route, 1: a spin gave up
    int* rclaim = flags + 4;                                                     // [kNB]  round tickets drawn per ring slot
    short* plan_s = reinterpret_cast<short*>(rclaim + 4);                        // [2][4][kMaxRQ] the row plan (halo, tap): indexed per lane
    uint32_t* hbuf = reinterpret_cast<uint32_t*>(plan_s + 2 * 4 * kMaxRQ);       // [kNB][hdw] raw halo dwords
    float* bank_s = reinterpret_cast<float*>(hbuf + (size_t)2 * hdw);            // [G][ntaps] (+1 pad), prologue only: over ring slot 2
                                                                                 // (first filled during tile 0's rounds) and beyond

#ifdef SN_CONV_TIMING
    if (lane < 8) g_i8s_w[(blockIdx.x * 8 + wave) * 8 + lane] = 0;
#endif
    const int my_tiles = (int)blockIdx.x < s.ntiles ? (s.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (tid < 16) landed[tid] = 0;   // landed[0..3], done[0..3], flags[0..3], rclaim[0..3]
    // the row plan is indexed per lane below: from the kernel argument that is a vector load from memory per use
    // ([measured] tables 4.8 us); one coalesced copy into LDS rides with the bank's loads
    if (tid < 2 * 4 * kMaxRQ) plan_s[tid] = reinterpret_cast<const short*>(&s.plan)[tid];
    const short* plan_halo = plan_s;                  // [4][kMaxRQ]
    const short* plan_tap = plan_s + 4 * kMaxRQ;      // [4][kMaxRQ]
    stage_bank_and_first_halos(s, bank, ntaps, bank_s, hbuf, hdw, x, my_tiles, tid, wave, lane);
    lds_barrier();
    SN_ST(1);
    quantise_kernels(s, ntaps, bank_s, scale, bnd, wave, lane);
    lds_barrier();
    SN_ST(2);
    // ---- route: all workgroups take the same decision from the same numbers
    {
        const bool exceeded = bound_exceeded<OT>(s, bnd, lambdas, act, out);
        if (blockIdx.x == 0 && tid == 0 && s.route) *s.route = exceeded ? 1 : 0;
        if (exceeded) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // do not leave with LDS-DMA in flight
            return true;
        }
    }
    // ---- digit table Wd[st][d][l = (qq, g)]: 4 dwords j, slot (st, qq, j)
    //   pair step st < NP:   j = 2 e + c  ->  taps 4c .. 4c+3 of row 2 st + e of lane group qq
    //   tail step:           slot t = 4 (st - NP) + j:  t < NT: tap 8 of rows 4t .. 4t+3;  then (ODD) chunks 0, 1 of row RQ-1
    for (int i = tid; i < s.KS * 64; i += kThreads) {
        const int l = i & 63, st = i >> 6;
        const int g = l & 15, qq = l >> 4;
        uint32_t w0[4] = {0u, 0u, 0u, 0u}, w1[4] = {0u, 0u, 0u, 0u}, w2[4] = {0u, 0u, 0u, 0u};
        if (g < s.G) {
            auto put = [&](int j, int b, int krow, int dy) {   // branch-free (krow < 0: Q = 0) so that the 16 reads of an entry go out together
                int Q = __float_as_int(bank_s[g * ntaps + (krow < 0 ? 0 : krow) * 9 + dy]);
                Q = krow < 0 ? 0 : Q;
                const int d0 = ((Q + 128) & 255) - 128;
                Q = (Q - d0) >> 8;
                const int d1 = ((Q + 128) & 255) - 128;
                const int d2 = (Q - d1) >> 8;
                w0[j] |= (uint32_t)(d0 & 255) << (8 * b);
                w1[j] |= (uint32_t)(d1 & 255) << (8 * b);
                w2[j] |= (uint32_t)(d2 & 255) << (8 * b);
            };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (st < s.NP) {
                    const int krow = plan_tap[qq * kMaxRQ + 2 * st + (j >> 1)];
#pragma unroll
                    for (int b = 0; b < 4; ++b) put(j, b, krow, 4 * (j & 1) + b);
                } else {
                    const int t = 4 * (st - s.NP) + j;
                    if (t < s.NT) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int ri = 4 * t + b;
                            put(j, b, ri < s.RQ ? plan_tap[qq * kMaxRQ + ri] : -1, 8);
                        }
                    } else if (s.ODD && t - s.NT < 2) {
                        const int krow = plan_tap[qq * kMaxRQ + s.RQ - 1];
#pragma unroll
                        for (int b = 0; b < 4; ++b) put(j, b, krow, 4 * (t - s.NT) + b);
                    }
                }
            }
        }
        Wd[(st * 3 + 0) * 64 + l] = make_uint4(w0[0], w0[1], w0[2], w0[3]);
        Wd[(st * 3 + 1) * 64 + l] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
        Wd[(st * 3 + 2) * 64 + l] = make_uint4(w2[0], w2[1], w2[2], w2[3]);
    }
    for (int i = tid; i < s.NP * 4; i += kThreads) {
        const int qq = i & 3, st = i >> 2;
        roff[i] = make_int2(plan_halo[qq * kMaxRQ + 2 * st] * (DW * 4), plan_halo[qq * kMaxRQ + 2 * st + 1] * (DW * 4));
    }
    for (int i = tid; i < s.NT * 4; i += kThreads) {
        const int qq = i & 3, t = i >> 2;
        int o[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ri = 4 * t + b;
            o[b] = plan_halo[qq * kMaxRQ + (ri < s.RQ ? ri : s.RQ - 1)] * (DW * 4);
        }
        toff[i] = make_int4(o[0], o[1], o[2], o[3]);
    }
    if (tid < 4) oddoff[tid] = plan_halo[tid * kMaxRQ + s.RQ - 1] * (DW * 4);
    SN_ST(3);

    // lambda_g * scale_g next to the scales: the epilogue reads its four of each per round (16-byte LDS reads) instead
    // of holding eight registers through the MFMA loop
    if (tid < 16) {
        const float ls = (out && tid < s.G) ? lambdas[tid] * scale[tid] : 0.0f;
        lamsc[tid] = ls;
        lamhi[tid] = 65536.0f * ls;   // exact
    }
    if (my_tiles == 0 || SN_DBG(s, 8)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return false;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the first two halos are in
    __syncthreads();                                   // tables complete, every wave's pieces in: no counters needed yet
    SN_ST(4);
    if (wave >= kWaves / 2 && s.stagger > 0)           // once: the younger wave of every SIMD pair runs half a round behind
        for (int i = 0; i < s.stagger; i += 100) __builtin_amdgcn_s_sleep(100);

    const int half_tx = s.TX >> 1;
    const int hx_shift = 31 - __builtin_clz(half_tx);
    const int nrounds = s.TZ * half_tx;
    const size_t V = (size_t)s.Z * s.X * s.Y;
    bool healthy = true;

    for (int it = 0; it < my_tiles; ++it) {
        const int tile = blockIdx.x + it * gridDim.x;
        const TileCoord c = tile_coord(s, tile);
        const int bi = it % kNB;
        const uint8_t* hb = reinterpret_cast<const uint8_t*>(hbuf + (size_t)bi * hdw);
        // halo `it` is complete once every wave has seen its own DMA pieces land (tiles 0 and 1: the barrier above)
        const unsigned long long t_tile = SN_WNOW();
        if (it >= 2) healthy &= wave_wait(&landed[bi], kWaves * (it / kNB + 1 - (bi < 2 ? 1 : 0)));
        SN_WT(0, t_tile);
        const bool dma_next = it + 2 < my_tiles && !SN_DBG(s, 2);
        // this wave's pieces of halo it+2.  Its ring slot was last read by tile it-1: every wave must be past that
        // tile -- two rounds into tile `it` they are ([measured] one round in, waves 0-3 waited ~4 us per tile for the
        // younger wave of their SIMD); a wave with fewer rounds issues after them
        bool dma_pending = dma_next;
        auto dma_ahead = [&]() {
            const int bn = (it + 2) % kNB;
            if (it >= 1) healthy &= wave_wait(&done[bn], kWaves * ((it - 1) / kNB + 1));
            // (the lane constants are recomputed here, ~15 instructions per tile, rather than kept in registers through the rounds)
            halo_dma_issue(hbuf + (size_t)bn * hdw, x, s, tile_coord(s, tile + 2 * gridDim.x), dma_lane(s, wave, lane), wave);
            dma_pending = false;
        };
        // Rounds are CLAIMED, not dealt: the older wave of a SIMD pair wins every tie for the matrix pipe and the issue
        // port ([measured] dealt 4 + 4, waves 4-7 needed 10 % longer and waves 0-3 waited for them before every DMA
        // issue; s_setprio did not move that), so whoever is ahead takes the next round of the tile.  Ring slot `bi` has
        // seen it / kNB earlier tiles, each of which drew nrounds valid tickets and one failing ticket per wave; the
        // next ticket is drawn a round ahead of its use.
        const int ticket0 = (it / kNB) * (nrounds + kWaves);
        auto claim = [&]() -> int {
            int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(&rclaim[bi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return __builtin_amdgcn_readfirstlane(t) - ticket0;
        };
        const int k_dma = nrounds >= 3 * kWaves ? 2 : 1;
        int k = 0, next = 0;
        for (int round = s.dynamic ? claim() : wave; round < nrounds; round = next, ++k) {
            next = s.dynamic ? claim() : round + kWaves;
            const unsigned long long t_r0 = SN_WNOW();
            if (k == k_dma && dma_pending) dma_ahead();
            SN_WT(4, t_r0);
            const unsigned long long t_r1 = SN_WNOW();
            if (SN_DBG(s, 128) && wave >= kWaves / 2) continue;   // timing experiment: one wave per SIMD works
            const int lz = round >> hx_shift, lx = (round & (half_tx - 1)) * 2;   // TX / 2 is a power of two (cand[] below)
            const uint8_t* xb = hb + ((lz * s.XP + lx) * DW + n + D0) * 4;

            i32x4 acc[3][NV];   // not zeroed: the first step's MFMAs take the constant 0 as their C operand

            auto load_w = [&](int st, i32x4 (&w)[3]) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const uint4 u = Wd[(st * 3 + d) * 64 + lane];
                    w[d] = i32x4{(int)u.x, (int)u.y, (int)u.z, (int)u.w};
                }
            };
            // raw dwords D0 D1 D2 of kernel rows A, B for x-rows h = 0, 1: [h][e][c]
            auto load_raw = [&](const int2& ro, uint32_t (&d)[2][2][3]) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const uint32_t* p = reinterpret_cast<const uint32_t*>(xb + (e ? ro.y : ro.x) + h * (DW * 4));
                        d[h][e][0] = p[0];
                        d[h][e][1] = p[1];
                        d[h][e][2] = p[2];
                    }
            };
            auto mma_tile = [&](const i32x4 (&w)[3], const i32x4& xv, int v, bool first = false) {
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    acc[d][v] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[d], xv, first ? i32x4{0, 0, 0, 0} : acc[d][v], 0, 0, 0);
            };
            // One pair step: 24 MFMAs and the 24 v_alignbyte that shift the raw dwords for residues 1..3.  Issue order is
            // fixed by hand -- four v_alignbyte (an operand used two groups later), then three MFMAs, ... -- so that the
            // VALU work always sits behind a running MFMA and never in front of one that needs it: left to the scheduler it
            // came out in clumps of eight, which idle the matrix pipe for half their length whenever the SIMD's other wave
            // is not in its own steps ([measured] a lone wave drove the pipe at 63 % inside these steps).
            auto no_extra = [](int) {};
            auto pair_compute = [&](const i32x4 (&w)[3], const uint32_t (&d)[2][2][3], bool first, auto&& extra) {
                auto shifted = [&](int h, auto R) -> i32x4 {
                    constexpr int r = decltype(R)::value;
                    return i32x4{(int)window<r>(d[h][0][1], d[h][0][0]), (int)window<r>(d[h][0][2], d[h][0][1]),
                                 (int)window<r>(d[h][1][1], d[h][1][0]), (int)window<r>(d[h][1][2], d[h][1][1])};
                };
                constexpr std::integral_constant<int, 1> R1{};
                constexpr std::integral_constant<int, 2> R2{};
                constexpr std::integral_constant<int, 3> R3{};
                const i32x4 x00{(int)d[0][0][0], (int)d[0][0][1], (int)d[0][1][0], (int)d[0][1][1]};
                const i32x4 x10{(int)d[1][0][0], (int)d[1][0][1], (int)d[1][1][0], (int)d[1][1][1]};
#define SN_FENCE() __builtin_amdgcn_sched_barrier(0)
                const i32x4 x01 = shifted(0, R1); SN_FENCE();
                mma_tile(w, x00, 0, first);       SN_FENCE();
                const i32x4 x02 = shifted(0, R2); SN_FENCE();
                mma_tile(w, x10, 4, first);       SN_FENCE();
                const i32x4 x03 = shifted(0, R3); SN_FENCE();
                mma_tile(w, x01, 1, first);       SN_FENCE();
                const i32x4 x11 = shifted(1, R1); SN_FENCE();
                mma_tile(w, x02, 2, first);       SN_FENCE();
                const i32x4 x12 = shifted(1, R2); SN_FENCE();
                mma_tile(w, x03, 3, first);       SN_FENCE();
                const i32x4 x13 = shifted(1, R3); SN_FENCE();
                mma_tile(w, x11, 5, first);       SN_FENCE();
                extra(0);                         SN_FENCE();   // (the last pair step: the first tail transposes ride here)
                mma_tile(w, x12, 6, first);       SN_FENCE();
                extra(1);                         SN_FENCE();
                mma_tile(w, x13, 7, first);       SN_FENCE();
                extra(2);
#undef SN_FENCE
            };

            constexpr std::integral_constant<int, 0> I0{};
            constexpr std::integral_constant<int, 1> I1{};
            // raw dwords of a tail step for x-row h, [j][b]: a quad slot holds D2 of its four kernel rows; an odd-row chunk
            // slot holds D_c, D_c+1 in b = 0, 1
            // the byte offsets of a tail step's quad rows (the same for both x-rows): one batch of LDS reads, requested a
            // phase before the raw dwords that need them ([measured] read inside tail_load -- table read, wait, four raw
            // reads, four times over -- a tail load was eight dependent LDS round trips: 30 us of a lone wave's 157)
            auto tail_offsets = [&](auto TS, int4 (&to)[4]) {
                constexpr int ts = decltype(TS)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = 4 * ts + j;
                    if (t < NT) to[j] = toff[t * 4 + q];
                    else if (ODD && t - NT < 2) to[j] = make_int4(oddoff[q] + 4 * (t - NT), 0, 0, 0);
                    else to[j] = make_int4(0, 0, 0, 0);
                }
            };
            auto tail_load = [&](auto TS, auto H, const int4 (&to)[4], uint32_t (&raw)[4][4]) {
                constexpr int ts = decltype(TS)::value, h = decltype(H)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = 4 * ts + j;
                    if (t < NT) {
                        const uint8_t* p = xb + h * (DW * 4) + 8;
                        raw[j][0] = *reinterpret_cast<const uint32_t*>(p + to[j].x);
                        raw[j][1] = *reinterpret_cast<const uint32_t*>(p + to[j].y);
                        raw[j][2] = *reinterpret_cast<const uint32_t*>(p + to[j].z);
                        raw[j][3] = *reinterpret_cast<const uint32_t*>(p + to[j].w);
                    } else if (ODD && t - NT < 2) {
                        const uint32_t* p = reinterpret_cast<const uint32_t*>(xb + h * (DW * 4) + to[j].x);
                        raw[j][0] = p[0];
                        raw[j][1] = p[1];
                        raw[j][2] = 0u;
                        raw[j][3] = 0u;
                    } else {
#pragma unroll
                        for (int b = 0; b < 4; ++b) raw[j][b] = 0u;
                    }
                }
            };
            // slot j of tail step TS for one x-row: raw dwords -> the K dword of every residue, x[r][j]
            auto tail_slot = [&](auto TS, int j, const uint32_t (&raw)[4][4], uint32_t (&x)[4][4]) {
                constexpr int ts = decltype(TS)::value;
                const int t = 4 * ts + j;
                if (t < NT) {
                    uint32_t o[4];
                    transpose4(raw[j][0], raw[j][1], raw[j][2], raw[j][3], o);
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r][j] = o[r];
                } else {   // odd-row chunk (or an unused slot: all zero)
                    const uint32_t lo = raw[j][0], hi = raw[j][1];
                    x[0][j] = lo;
                    x[1][j] = window<1>(hi, lo);
                    x[2][j] = window<2>(hi, lo);
                    x[3][j] = window<3>(hi, lo);
                }
            };
            // the 12 MFMAs of one (tail step, x-row); work(r) is issued behind residue r's three -- the next (step, x-row)'s
            // transposes, eight v_perm at a time ([measured] with all 32 in front of their own MFMAs a lone wave spent
            // 0.8 us per round in the tail steps for 0.3 us of matrix work)
            auto tail_mma = [&](auto H, const uint32_t (&x)[4][4], const i32x4 (&w)[3], auto&& work) {
                constexpr int h = decltype(H)::value;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mma_tile(w, i32x4{(int)x[r][0], (int)x[r][1], (int)x[r][2], (int)x[r][3]}, 4 * h + r);
                    __builtin_amdgcn_sched_barrier(0);
                    work(r);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };

            // ---- pair steps, two register sets: step st+1's operands are requested before step st's 24 MFMAs issue; the
            // first tail step's raw dwords are requested ahead of the last pair step's MFMAs in the same way
            i32x4 wa[3], wb[3];
            uint32_t da[2][2][3], db[2][2][3];
            uint32_t ta[4][4], tb[4][4];   // tail raw dwords, one (step, x-row) each, ping-pong
            int2 ra = roff[q], rb;
            load_w(0, wa);
            load_raw(ra, da);
            rb = roff[(NP > 1 ? 4 : 0) + q];
            __builtin_amdgcn_sched_barrier(0);
            // steps in pairs while at least one more follows; the last one or two are peeled below so that the tail's
            // raw dwords are live only there (carried through the loop they cost 32 registers: the kernel spilled)
            const int np_loop = (NP - 1) & ~1;   // >= 2 (NP = 10)
            {   // steps 0 and 1; step 0 opens the accumulators
                load_w(1, wb);
                load_raw(rb, db);
                ra = roff[2 * 4 + q];
                __builtin_amdgcn_sched_barrier(0);
                pair_compute(wa, da, true, no_extra);
                __builtin_amdgcn_sched_barrier(0);
                load_w(2, wa);
                load_raw(ra, da);
                rb = roff[(3 < NP ? 3 : NP - 1) * 4 + q];
                __builtin_amdgcn_sched_barrier(0);
                pair_compute(wb, db, false, no_extra);
                __builtin_amdgcn_sched_barrier(0);
            }
            int st = 2;
            for (; st < np_loop; st += 2) {
                load_w(st + 1, wb);
                load_raw(rb, db);
                ra = roff[(st + 2) * 4 + q];
                __builtin_amdgcn_sched_barrier(0);
                pair_compute(wa, da, false, no_extra);
                __builtin_amdgcn_sched_barrier(0);
                load_w(st + 2, wa);
                load_raw(ra, da);
                rb = roff[(st + 3 < NP ? st + 3 : NP - 1) * 4 + q];
                __builtin_amdgcn_sched_barrier(0);
                pair_compute(wb, db, false, no_extra);
                __builtin_amdgcn_sched_barrier(0);
            }
            int4 to[4];   // row offsets of the current tail step
            uint32_t xa[4][4], xq[4][4];   // a (tail step, x-row)'s K dwords [residue][slot], ping-pong
            // the first tail (step, x-row)'s transposes ride behind the last MFMAs of the last pair step
            auto first_transposes = [&](int k) {
                if (k == 0) tail_slot(I0, 0, ta, xa);
                if (k == 1) tail_slot(I0, 1, ta, xa);
                if (k == 2) { tail_slot(I0, 2, ta, xa); tail_slot(I0, 3, ta, xa); }
            };
            if (NP - np_loop == 2) {
                load_w(np_loop + 1, wb);
                load_raw(rb, db);
                tail_offsets(I0, to);
                __builtin_amdgcn_sched_barrier(0);
                pair_compute(wa, da, false, no_extra);
                __builtin_amdgcn_sched_barrier(0);
                load_w(NP, wa);   // the first tail step's digits
                tail_load(I0, I0, to, ta);
                tail_load(I0, I1, to, tb);
                if (NTS > 1) tail_offsets(I1, to);   // (behind the reads that use the first step's)
                __builtin_amdgcn_sched_barrier(0);
                pair_compute(wb, db, false, first_transposes);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                load_w(NP, wb);
                tail_offsets(I0, to);
                tail_load(I0, I0, to, ta);
                tail_load(I0, I1, to, tb);
                if (NTS > 1) tail_offsets(I1, to);
                __builtin_amdgcn_sched_barrier(0);
                pair_compute(wa, da, false, first_transposes);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 3; ++d) wa[d] = wb[d];
            }
            SN_WT(1, t_r1);
            const unsigned long long t_r2 = SN_WNOW();
            // ---- tail steps (at most two; the first one's digits are in wa): every (step, x-row)'s 12 MFMAs carry the
            // next one's transposes, whose raw dwords were requested a phase earlier
            if (NTS > 1) {
                load_w(NP + 1, wb);
                tail_load(I1, I0, to, ta);   // ta's dwords are in xa
            }
            __builtin_amdgcn_sched_barrier(0);
            tail_mma(I0, xa, wa, [&](int r) { tail_slot(I0, r, tb, xq); });
            __builtin_amdgcn_sched_barrier(0);
            if (NTS > 1) {
                tail_load(I1, I1, to, tb);   // tb's dwords are in xq
                __builtin_amdgcn_sched_barrier(0);
                tail_mma(I1, xq, wa, [&](int r) { tail_slot(I1, r, ta, xa); });
                __builtin_amdgcn_sched_barrier(0);
                tail_mma(I0, xa, wb, [&](int r) { tail_slot(I1, r, tb, xq); });
                __builtin_amdgcn_sched_barrier(0);
                tail_mma(I1, xq, wb, no_extra);
            } else {
                tail_mma(I1, xq, wa, no_extra);
            }
            SN_WT(2, t_r2);
            const unsigned long long t_r3 = SN_WNOW();
            // ---- epilogue: recombine the digits, bank activations, head
            finish_round<OT>(s, c, lz, lx, n, q, acc, scale, lamsc, lamhi, act, out, V);
            SN_WT(3, t_r3);
        }
        // ---- this wave is through with tile `it`
        const unsigned long long t_e = SN_WNOW();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_signal(&done[bi], lane);
        if (dma_pending) dma_ahead();
        if (dma_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces (and its stores) have landed
            wave_signal(&landed[(it + 2) % kNB], lane);
        }
        SN_WT(5, t_e);
        SN_WT(6, t_tile);
    }
    if (!healthy && lane == 0) {   // a halo hand-over never arrived: this launch's output cannot be trusted -- loud
        atomicAdd(&g_fold_counts[3], 1ull);   // (sn_conv_i8_spin_timeouts)
        sn::sticky_latch(s.sticky, 3, (int)blockIdx.x, 0);
    }
    SN_ST(5);
    return false;
}

// the stride-4 kernel as a launch of its own (banks the folded kernel is not tried on)
template <typename OT, int kNP, int kNT, int kODD>
__global__ __launch_bounds__(kThreads) void conv_occ_i8s_kernel(const uint8_t* __restrict__ x,
                                                                const float* __restrict__ bank,
                                                                const float* __restrict__ lambdas, Shape s,
                                                                OT* __restrict__ act, OT* __restrict__ out) {
    (void)conv_occ_i8s_body<OT, kNP, kNT, kODD>(x, bank, lambdas, s, act, out);
}

// ================================================================================================ folded kernel
// GENEO kernels are radial in (x, y) (cylinder.py:152-176, arrow.py:214-252, neg_sphere.py:166-199): W[dz][dx][dy] =
// W[dz][8-dx][dy] = W[dz][dx][8-dy], bit for bit (the generators evaluate the same expression on (dx-4)^2 + (dy-4)^2).
// For such a bank
//     sum_{dx,dy} W[dz][dx][dy] x[.., x+dx, y+dy]  =  sum_{dx'<=4, dy'<=4} W[dz][dx'][dy'] F[dz][dx'][dy'],
//     F = the sum of x over the orbit {dx', 8-dx'} x {dy', 8-dy'}  (1, 2 or 4 voxels: 0..4, exact in int8),
// i.e. 9 x 5 x 5 = 225 taps instead of 729 with EXACTLY the same integer sums -- 4 MFMA steps per 16 x 16 outputs instead
// of 12.  The folding is done on the way from LDS to the B operand: the two halo rows of a folded row are added dword by
// dword (bytes <= 2, no carries), the y window of residue r folds as  v_alignbyte(R1, R0, r) + v_perm(R2:R1, bytes r+8 ..
// r+5)  (bytes <= 4), and the centre taps dy' = 4 (byte r of R1) of a step's three rows are packed by the 4 x 4 byte
// transpose the stride-4 kernel uses for tap 8.  A step = 3 folded rows + their centre dword per lane group; 45 folded rows
// (+ 3 pads) over 4 lane groups x 12 slots.  Everything else -- halo ring, LDS-DMA, claimed rounds, quantisation, guard,
// epilogue -- is the stride-4 kernel's.  The prologue CHECKS the symmetry on the fp32 weights (bitwise); a bank that is not
// symmetric takes the stride-4 kernel's body instead, in the same launch.
// Compile-time halo pitch kXP (17 / 13 / 11 for TX = 8 / 4 / 2): the halo byte offset of a regular slot is
//     q * (2 kXP 96) + (a kXP + dx') 96        (lane group q carries planes dz = 2 q + a, a = 0, 1),
// one per-lane register plus an immediate of the LDS read -- no offset table, no address arithmetic in the rounds.  The two
// irregular slots (8: the single of plane 8 and three pads; 11: plane 8's doubles) take their offsets from the plan.

template <typename OT, int kXP>
__global__ __launch_bounds__(kThreads) void conv_occ_i8f_kernel(const uint8_t* __restrict__ x,
                                                                const float* __restrict__ bank,
                                                                const float* __restrict__ lambdas, Shape s,
                                                                OT* __restrict__ act, OT* __restrict__ out) {
    if (!s.gate.pass()) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    constexpr int ntaps = 729;
    SN_ST(0);
    const int hrows = s.ZP * s.XP;
    const int hdw = hrows * DW;
    // LDS carve-up
    uint4* Wd = reinterpret_cast<uint4*>(lds);                                   // [kFoldSteps][3][64] x 16 B
    float* scale = reinterpret_cast<float*>(Wd + (size_t)kFoldSteps * 3 * 64);   // [16]
    float* lamsc = scale + 16;                                                   // [16]
    float* lamhi = lamsc + 16;                                                   // [16] (+ 64 bytes spare)
    double* bnd = reinterpret_cast<double*>(lamhi + 32);                         // [16]
    int* landed = reinterpret_cast<int*>(bnd + 16);                              // [kNB]
    int* done = landed + 4;                                                      // [kNB]
    int* flags = done + 4;                                                       // [4]  1: a spin gave up, 2: the bank is not symmetric
    int* rclaim = flags + 4;                                                     // [kNB]
    short* plan_s = reinterpret_cast<short*>(rclaim + 4);                        // FoldPlan copy
    uint32_t* hbuf = reinterpret_cast<uint32_t*>(plan_s + 3 * 4 * kFoldSlots);   // [kNB][hdw]
    float* bank_s = reinterpret_cast<float*>(hbuf + (size_t)2 * hdw);            // prologue only: over ring slot 2 and beyond

    const int my_tiles = (int)blockIdx.x < s.ntiles ? (s.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (tid < 16) landed[tid] = 0;
    if (tid < 3 * 4 * kFoldSlots) plan_s[tid] = reinterpret_cast<const short*>(&s.fplan)[tid];
    const short* plan_h1 = plan_s;
    const short* plan_h2 = plan_s + 4 * kFoldSlots;
    const short* plan_tap = plan_s + 2 * 4 * kFoldSlots;
    stage_bank_and_first_halos(s, bank, ntaps, bank_s, hbuf, hdw, x, my_tiles, tid, wave, lane);
    lds_barrier();
    SN_ST(1);
    // ---- is every kernel symmetric in x and in y?  (bitwise on the fp32 weights; a NaN pattern compares like any other)
    {   // one (kernel, dz, dx <= 4) row of nine weights per item: against its mirror row 8 - dx, and against its own reverse
        bool asym = false;
        const uint32_t* wb = reinterpret_cast<const uint32_t*>(bank_s);
        for (int i = tid; i < s.G * 9 * 5; i += kThreads) {
            const int dx = i % 5, pl = i / 5;           // pl = g * 9 + dz
            const uint32_t* ra = wb + (pl * 9 + dx) * 9;
            const uint32_t* rb = wb + (pl * 9 + 8 - dx) * 9;
            uint32_t a[9], b[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) { a[k] = ra[k]; b[k] = rb[k]; }
#pragma unroll
            for (int k = 0; k < 9; ++k) asym |= a[k] != b[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) asym |= (a[k] != a[8 - k]) | (b[k] != b[8 - k]);
        }
        if (asym) flags[2] = 1;   // benign race: every writer stores 1
    }
    lds_barrier();
    if (flags[2]) {
        // not symmetric: the stride-4 kernel's whole job, in this launch (a second, gated launch cost 2.8 us per call
        // even when it had nothing to do).  Its body starts from scratch -- own LDS layout, own prologue, and it writes the
        // route flag (0 / 1) itself; the halos requested above must have landed before their LDS is reused.
        if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_fold_counts[1], 1ull);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        (void)conv_occ_i8s_body<OT, 10, 6, 1>(x, bank, lambdas, s, act, out);
        return;
    }
    SN_ST(6);   // (symmetry checked)
    quantise_kernels_folded(s, bank_s, scale, bnd, wave, lane);
    lds_barrier();
    SN_ST(2);
    {
        const bool exceeded = bound_exceeded<OT>(s, bnd, lambdas, act, out);
        if (blockIdx.x == 0 && tid == 0 && s.route) *s.route = exceeded ? 1 : 0;
        if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_fold_counts[exceeded ? 2 : 0], 1ull);
        if (exceeded) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    }
    // ---- digit table Wd[st][d][l = (qq, g)]: dword j < 3: taps dy' = 0..3 of the lane group's row (st, j); dword 3: tap
    // dy' = 4 of rows (st, 0..2) in bytes 0..2
    for (int i = tid; i < kFoldSteps * 64; i += kThreads) {
        const int l = i & 63, st = i >> 6;
        const int g = l & 15, qq = l >> 4;
        uint32_t w0[4] = {0u, 0u, 0u, 0u}, w1[4] = {0u, 0u, 0u, 0u}, w2[4] = {0u, 0u, 0u, 0u};
        if (g < s.G) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int slot = st * kFoldRows + (j < 3 ? j : b);
                    const int krow = (j == 3 && b == 3) ? -1 : plan_tap[qq * kFoldSlots + (j == 3 && b == 3 ? 0 : slot)];
                    const int dy = j < 3 ? b : 4;
                    int Q = __float_as_int(bank_s[g * ntaps + (krow < 0 ? 0 : krow) * 9 + dy]);
                    Q = krow < 0 ? 0 : Q;
                    const int d0 = ((Q + 128) & 255) - 128;
                    Q = (Q - d0) >> 8;
                    const int d1 = ((Q + 128) & 255) - 128;
                    const int d2 = (Q - d1) >> 8;
                    w0[j] |= (uint32_t)(d0 & 255) << (8 * b);
                    w1[j] |= (uint32_t)(d1 & 255) << (8 * b);
                    w2[j] |= (uint32_t)(d2 & 255) << (8 * b);
                }
        }
        Wd[(st * 3 + 0) * 64 + l] = make_uint4(w0[0], w0[1], w0[2], w0[3]);
        Wd[(st * 3 + 1) * 64 + l] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
        Wd[(st * 3 + 2) * 64 + l] = make_uint4(w2[0], w2[1], w2[2], w2[3]);
    }
    SN_ST(3);
    if (tid < 16) {
        const float ls = (out && tid < s.G) ? lambdas[tid] * scale[tid] : 0.0f;
        lamsc[tid] = ls;
        lamhi[tid] = 65536.0f * ls;
    }
    if (my_tiles == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SN_ST(4);

    const int half_tx = s.TX >> 1;
    const int hx_shift = 31 - __builtin_clz(half_tx);
    const int nrounds = s.TZ * half_tx;
    const size_t V = (size_t)s.Z * s.X * s.Y;
    bool healthy = true;
    // per-lane halo offsets: the lane group's planes, and the two irregular slots
    const int qbase = q * (2 * kXP * DW * 4);
    const int ir8 = plan_h1[q * kFoldSlots + 8] * (DW * 4);
    const int ir11a = plan_h1[q * kFoldSlots + 11] * (DW * 4), ir11b = plan_h2[q * kFoldSlots + 11] * (DW * 4);

    for (int it = 0; it < my_tiles; ++it) {
        const int tile = blockIdx.x + it * gridDim.x;
        const TileCoord c = tile_coord(s, tile);
        const int bi = it % kNB;
        const uint8_t* hb = reinterpret_cast<const uint8_t*>(hbuf + (size_t)bi * hdw);
        if (it >= 2) healthy &= wave_wait(&landed[bi], kWaves * (it / kNB + 1 - (bi < 2 ? 1 : 0)));
        const bool dma_next = it + 2 < my_tiles;
        bool dma_pending = dma_next;
        auto dma_ahead = [&]() {
            const int bn = (it + 2) % kNB;
            if (it >= 1) healthy &= wave_wait(&done[bn], kWaves * ((it - 1) / kNB + 1));
            halo_dma_issue(hbuf + (size_t)bn * hdw, x, s, tile_coord(s, tile + 2 * gridDim.x), dma_lane(s, wave, lane), wave);
            dma_pending = false;
        };
        const int ticket0 = (it / kNB) * (nrounds + kWaves);
        auto claim = [&]() -> int {
            int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(&rclaim[bi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return __builtin_amdgcn_readfirstlane(t) - ticket0;
        };
        const int k_dma = nrounds >= 3 * kWaves ? 2 : 1;
        int k = 0, next = 0;
        // a round's lane base inside the halo buffer: row (lz, lx), this lane's dword, plus the lane group's planes
        auto round_base = [&](int round) -> const uint8_t* {
            const int lz = round >> hx_shift, lx = (round & (half_tx - 1)) * 2;
            return hb + ((lz * kXP + lx) * DW + n + D0) * 4;
        };
        // raw dwords of unit (slot, x-row h): halo row h1 in r[0..2], h2 in r[3..5] (dwords D0 D1 D2)
        auto load_unit = [&](const uint8_t* xb, auto SLOT, auto HC, uint32_t (&r)[6]) {
            constexpr int slot = decltype(SLOT)::value, h = decltype(HC)::value;
            constexpr bool single = fold_slot_single(slot / kFoldRows, slot % kFoldRows);
            const uint8_t *p1, *p2;
            if constexpr (fold_slot_irregular(slot)) {
                p1 = xb + (slot == 8 ? ir8 : ir11a) + h * (DW * 4);
                p2 = xb + ir11b + h * (DW * 4);
            } else {
                constexpr int c1 = (fold_slot_a(slot) * kXP + fold_slot_dx(slot)) * (DW * 4) + h * (DW * 4);
                constexpr int c2 = (fold_slot_a(slot) * kXP + 8 - fold_slot_dx(slot)) * (DW * 4) + h * (DW * 4);
                p1 = xb + qbase + c1;   // (xb + qbase is one add per round; c1, c2 are immediates of the reads)
                p2 = xb + qbase + c2;
            }
            const uint32_t* d1 = reinterpret_cast<const uint32_t*>(p1);
            r[0] = d1[0]; r[1] = d1[1]; r[2] = d1[2];
            if constexpr (!single) {
                const uint32_t* d2 = reinterpret_cast<const uint32_t*>(p2);
                r[3] = d2[0]; r[4] = d2[1]; r[5] = d2[2];
            }
        };
        // fold one unit into component J of the four residues' operand quads X[r] (a step's B operand for tile (h, r) is the
        // quad (row 0, row 1, row 2, centres): written in place, component by component); also returns the summed dword 1,
        // whose byte r is the centre tap of residue r
        auto fold_unit = [&](const uint32_t (&r)[6], auto SLOT, i32x4 (&X)[4], uint32_t& centre) {
            constexpr int slot = decltype(SLOT)::value, J = slot % kFoldRows;
            constexpr bool single = fold_slot_single(slot / kFoldRows, J);
            // (dwords 0 and 1 in one 64-bit add -- v_lshl_add_u64: bytes <= 2 per addend, nothing carries across)
            const uint64_t R01 = single ? ((uint64_t)r[1] << 32 | r[0])
                                        : ((uint64_t)r[1] << 32 | r[0]) + ((uint64_t)r[4] << 32 | r[3]);
            const uint32_t R0 = (uint32_t)R01, R1 = (uint32_t)(R01 >> 32);
            const uint32_t R2 = single ? r[2] : r[2] + r[5];
            X[0][J] = (int)(R0 + __builtin_amdgcn_perm(R2, R1, 0x01020304u));
            X[1][J] = (int)(__builtin_amdgcn_alignbyte(R1, R0, 1) + __builtin_amdgcn_perm(R2, R1, 0x02030405u));
            X[2][J] = (int)(__builtin_amdgcn_alignbyte(R1, R0, 2) + __builtin_amdgcn_perm(R2, R1, 0x03040506u));
            X[3][J] = (int)(__builtin_amdgcn_alignbyte(R1, R0, 3) + __builtin_amdgcn_perm(R2, R1, 0x04050607u));
            centre = R1;
        };
        auto centres = [&](uint32_t c0, uint32_t c1, uint32_t c2, i32x4 (&X)[4]) {
            uint32_t o[4];
            transpose3(c0, c1, c2, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) X[r][3] = (int)o[r];
        };
        constexpr std::integral_constant<int, 0> H0{};
        constexpr std::integral_constant<int, 1> H1{};
        // operands of step ST at lane base xb, un-pipelined (a tile's first round)
        // a round's first step, un-pipelined: its six units' raw dwords and their folding.  (Requesting even the first two
        // units a round early, behind the previous round's last MFMAs, makes hipcc spill around the epilogue: 76 bytes.)
        uint32_t fraw[6][6];
        auto first_load_a = [&](const uint8_t* xb) {
            constexpr std::integral_constant<int, 0> L0{};
            load_unit(xb, L0, H0, fraw[0]); load_unit(xb, L0, H1, fraw[1]);
        };
        auto first_load_b = [&](const uint8_t* xb) {
            constexpr std::integral_constant<int, 1> L1{};
            constexpr std::integral_constant<int, 2> L2{};
            load_unit(xb, L1, H0, fraw[2]); load_unit(xb, L1, H1, fraw[3]);
            load_unit(xb, L2, H0, fraw[4]); load_unit(xb, L2, H1, fraw[5]);
        };
        auto first_fold = [&](i32x4 (&X)[2][4]) {
            constexpr std::integral_constant<int, 0> L0{};
            constexpr std::integral_constant<int, 1> L1{};
            constexpr std::integral_constant<int, 2> L2{};
            uint32_t c00, c01, c10, c11, c20, c21;
            fold_unit(fraw[0], L0, X[0], c00); fold_unit(fraw[1], L0, X[1], c01);
            fold_unit(fraw[2], L1, X[0], c10); fold_unit(fraw[3], L1, X[1], c11);
            fold_unit(fraw[4], L2, X[0], c20); fold_unit(fraw[5], L2, X[1], c21);
            centres(c00, c10, c20, X[0]);
            centres(c01, c11, c21, X[1]);
        };
        // The next step's operands are built in eight pieces behind the eight MFMA groups of the current step: pieces 0..5
        // fold unit u = (j, h) = (piece >> 1, piece & 1), pieces 6, 7 transpose the centres of x-row 0, 1.  The raw dwords of
        // unit u are requested two pieces ahead: units 0, 1 during pieces 6, 7 of the step BEFORE (pipe_open for the first).
        uint32_t praw[6][6], pcen[3][2];
        auto pipe_load = [&](const uint8_t* xb, auto ST, auto UC) {
            constexpr int st = decltype(ST)::value, u = decltype(UC)::value;
            load_unit(xb, std::integral_constant<int, st * kFoldRows + (u >> 1)>{}, std::integral_constant<int, (u & 1)>{}, praw[u]);
        };
        constexpr int kAhead = SN_I8F_AHEAD;   // raw dwords are requested this many pieces before they are folded
        auto pipe_open = [&](const uint8_t* xb, auto ST) {
            pipe_load(xb, ST, std::integral_constant<int, 0>{});
            pipe_load(xb, ST, std::integral_constant<int, 1>{});
            if constexpr (kAhead >= 3) pipe_load(xb, ST, std::integral_constant<int, 2>{});
        };
        // piece PC of building step ST (lane base xb) into X; pieces 6, 7 also request units 0, 1 of step NST at base nxb
        // (NST = -1: nothing follows)
        auto pipe_piece = [&](const uint8_t* xb, auto ST, auto PC, i32x4 (&X)[2][4], const uint8_t* nxb, auto NST) {
            constexpr int st = decltype(ST)::value, piece = decltype(PC)::value, nst = decltype(NST)::value;
            if constexpr (piece < 6) {
                constexpr int j = piece >> 1, h = piece & 1;
                fold_unit(praw[piece], std::integral_constant<int, st * kFoldRows + j>{}, X[h], pcen[j][h]);
            } else {
                constexpr int h = piece - 6;
                centres(pcen[0][h], pcen[1][h], pcen[2][h], X[h]);
            }
            if constexpr (piece + kAhead < 6) pipe_load(xb, ST, std::integral_constant<int, piece + kAhead>{});
            else if constexpr (piece >= 8 - kAhead && nst >= 0)
                pipe_load(nxb, std::integral_constant<int, (nst < 0 ? 0 : nst)>{}, std::integral_constant<int, piece - (8 - kAhead)>{});
        };
        constexpr std::integral_constant<int, 1> S1{};
        constexpr std::integral_constant<int, 2> S2{};
        constexpr std::integral_constant<int, 3> S3{};
        constexpr std::integral_constant<int, -1> SNone{};
        i32x4 Xa[2][4], Xb[2][4];   // operand quads [h][r] of the current / the next step
        for (int round = claim(); round < nrounds; round = next, ++k) {
            next = claim();
            if (k == k_dma && dma_pending) dma_ahead();
            const int lz = round >> hx_shift, lx = (round & (half_tx - 1)) * 2;
            const uint8_t* xb = round_base(round);

            i32x4 acc[3][NV];
            auto load_w = [&](int st, i32x4 (&w)[3]) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const uint4 u = Wd[(st * 3 + d) * 64 + lane];
                    w[d] = i32x4{(int)u.x, (int)u.y, (int)u.z, (int)u.w};
                }
            };
            auto mma_tile = [&](const i32x4 (&w)[3], const i32x4& xv, int v, bool first) {
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    acc[d][v] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[d], xv, first ? i32x4{0, 0, 0, 0} : acc[d][v], 0, 0, 0);
            };
            auto mma_step = [&](const i32x4 (&w)[3], const i32x4 (&X)[2][4], bool first, auto&& work) {
                auto group = [&](auto VC) {   // (compile-time v: the pieces index registers, never memory)
                    constexpr int v = decltype(VC)::value, h = v >> 2, r = v & 3;
                    mma_tile(w, X[h][r], v, first);
                    __builtin_amdgcn_sched_barrier(0);
                    work(VC);
                    __builtin_amdgcn_sched_barrier(0);
                };
                group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{});
                group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
                group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
                group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
            };
            i32x4 wa[3], wb[3];
            load_w(0, wa);
            first_load_a(xb);
            first_load_b(xb);
            first_fold(Xa);
            pipe_open(xb, S1);
            load_w(1, wb);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(wa, Xa, true, [&](auto v) { pipe_piece(xb, S1, v, Xb, xb, S2); });
            load_w(2, wa);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(wb, Xb, false, [&](auto v) { pipe_piece(xb, S2, v, Xa, xb, S3); });
            load_w(3, wb);
            __builtin_amdgcn_sched_barrier(0);
            // the last step carries step 0 of this wave's NEXT round of the tile.  No branch: without a next round the same
            // work runs on this round's rows and is dropped (a second copy of the two steps cost more in registers than
            // the wasted folds of a tile's last round cost in time)
            // (carrying step 0 of the wave's next round across the epilogue was tried: slower, 0.142 vs 0.132 ms -- the carried
            // operands spill around the epilogue and a tile's last round folds for nothing)
            mma_step(wa, Xa, false, [&](auto v) { pipe_piece(xb, S3, v, Xb, xb, SNone); });
            __builtin_amdgcn_sched_barrier(0);
            mma_step(wb, Xb, false, [](auto) {});
            finish_round<OT>(s, c, lz, lx, n, q, acc, scale, lamsc, lamhi, act, out, V);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_signal(&done[bi], lane);
        if (dma_pending) dma_ahead();
        if (dma_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_signal(&landed[(it + 2) % kNB], lane);
        }
    }
    if (!healthy && lane == 0) {   // a halo hand-over never arrived: this launch's output cannot be trusted -- loud
        atomicAdd(&g_fold_counts[3], 1ull);   // (sn_conv_i8_spin_timeouts)
        sn::sticky_latch(s.sticky, 3, (int)blockIdx.x, 0);
    }
    SN_ST(5);
}

size_t lds_bytes_fold(const Shape& s) {
    const size_t hdw = (size_t)s.ZP * s.XP * DW;
    const size_t ring = kNB * hdw * 4, alias = 2 * hdw * 4 + ((size_t)s.G * 729 + 1) * sizeof(float);
    return (size_t)kFoldSteps * 3 * 64 * 16 + 64 + 64 + 128 + 128 + 64 + sizeof(FoldPlan) +
           (ring > alias ? ring : alias) + 16;
}

// 45 folded rows (dz, dx') -> 4 lane groups x 12 slots (see FoldPlan).  Lane groups (0, 1) and (2, 3) are served by one
// LDS cycle each: at every slot their halo rows differ by 2 (mod 4) -- dz and dz + 2 at the same dx', or dx' and dx' + 2
// at dz = 8 -- which is 16 banks with 24-dword rows, for h1 and for h2 alike.
void plan_fold(int XP, FoldPlan& p) {
    for (int qq = 0; qq < 4; ++qq) {
        auto put = [&](int slot, int dz, int dxp, bool pad) {
            p.h1[qq][slot] = (short)(dz * XP + dxp);
            p.h2[qq][slot] = (short)(dz * XP + 8 - dxp);
            p.tap[qq][slot] = pad ? (short)-1 : (short)(dz * 9 + dxp);
        };
        for (int slot = 0; slot < kFoldSlots; ++slot)   // the regular slots: the closed form the kernel's immediates use
            if (!fold_slot_irregular(slot)) put(slot, 2 * qq + fold_slot_a(slot), fold_slot_dx(slot), false);
        static const int last_dx[4] = {0, 2, 1, 3};
        put(11, 8, last_dx[qq], false);        // plane 8's doubles: dx' and dx' + 2 for the lane groups of one LDS cycle
        static const int pad_dz[4] = {8, 6, 0, 2};
        put(8, pad_dz[qq], 4, qq != 0);        // (8, 4), then pads on rows 2 (mod 4) apart pairwise
    }
}

size_t lds_bytes(const Shape& s) {
    const size_t hdw = (size_t)s.ZP * s.XP * DW;
    return (size_t)s.KS * 3 * 64 * 16 + (size_t)s.NP * 4 * 8 + (size_t)s.NT * 4 * 16 + 16 + 64 + 64 + 128 + 128 + 64 + sizeof(RowPlan) +
           (kNB * hdw * 4 > 2 * hdw * 4 + ((size_t)s.G * s.kz * s.kx * 9 + 1) * sizeof(float)
                ? kNB * hdw * 4 : 2 * hdw * 4 + ((size_t)s.G * s.kz * s.kx * 9 + 1) * sizeof(float)) + 16;
}

int num_cus() {
    static thread_local int cached = 0;
    if (cached) return cached;
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 256;
    }
    cached = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    return cached;
}

// Kernel rows -> lane groups.  The two lane groups an LDS cycle serves (q = 0,1 and q = 2,3) read the rows at the
// same list position: with 24-dword halo rows they fall on disjoint banks when the halo row indices differ by 2
// (mod 4).  Rows are bucketed by halo index mod 4 and class c is paired with class c + 2; what is left over pairs up
// among itself (a 2-way conflict on those reads) or with a pad.  Any assignment is correct; this one is fast.
bool plan_rows(int kz, int kx, int XP, RowPlan& p, int& RQ) {
    const int R = kz * kx;
    RQ = (R + 3) / 4;
    if (RQ > kMaxRQ || RQ < 1) return false;
    static_assert((2 * DW) % 32 == 16, "plan_rows pairs halo rows 2 (mod 4) apart: 2 * DW must be 16 (mod 32) dwords");
    std::vector<int> cls[4];
    for (int dz = 0; dz < kz; ++dz)
        for (int dx = 0; dx < kx; ++dx) cls[(dz * XP + dx) & 3].push_back(dz * kx + dx);
    std::vector<std::pair<int, int>> units;
    std::vector<int> singles;
    for (int c = 0; c < 2; ++c) {
        const auto &a = cls[c], &b = cls[c + 2];
        const size_t m = a.size() < b.size() ? a.size() : b.size();
        for (size_t i = 0; i < m; ++i) units.emplace_back(a[i], b[i]);
        for (size_t i = m; i < a.size(); ++i) singles.push_back(a[i]);
        for (size_t i = m; i < b.size(); ++i) singles.push_back(b[i]);
    }
    for (size_t i = 0; i + 1 < singles.size(); i += 2) units.emplace_back(singles[i], singles[i + 1]);
    if (singles.size() & 1) units.emplace_back(singles.back(), -1);
    if ((int)units.size() > 2 * RQ) return false;   // cannot happen: ceil(R / 2) <= 2 ceil(R / 4)
    auto halo_of = [&](int krow) { return (krow / kx) * XP + (krow % kx); };
    for (int qq = 0; qq < 4; ++qq)
        for (int i = 0; i < kMaxRQ; ++i) { p.halo[qq][i] = 0; p.tap[qq][i] = -1; }
    for (size_t u = 0; u < units.size(); ++u) {
        const int half = (int)(u & 1), pos = (int)(u >> 1);
        const int a = units[u].first, b = units[u].second;
        p.tap[2 * half][pos] = (short)a;
        p.halo[2 * half][pos] = (short)halo_of(a);
        p.tap[2 * half + 1][pos] = (short)b;
        p.halo[2 * half + 1][pos] = (short)(b >= 0 ? halo_of(b) : halo_of(a));   // pad: same address, a broadcast
    }
    return true;
}

// tile shape and row plan of the stride-4 kernel for a (B, Z, X, Y) grid with 9 x 9 kernel rows: false = not served
bool plan_stride4(Shape& s, int B, int Z, int X, int Y, int kz, int kx, int cus) {
    static const int cand[][2] = {{8, 8}, {4, 8}, {4, 4}, {2, 4}, {1, 4}, {1, 2}};
    bool found = false;
    for (const auto& c : cand) {
        s.TZ = c[0]; s.TX = c[1];   // (TX / 2 a power of two: the kernel splits a round index by shift and mask)
        s.nzt = (Z + s.TZ - 1) / s.TZ; s.nxt = (X + s.TX - 1) / s.TX;
        const long long nt = (long long)B * s.nzt * s.nxt * s.nyt;
        if (nt > 0x7fffffff) return false;
        s.ntiles = (int)nt;
        s.ZP = s.TZ + kz - 1;
        s.XP = (s.TX + kx - 1) | 1;   // odd: rows of different z planes can pair up 2 (mod 4) apart
        if (!plan_rows(kz, kx, s.XP, s.plan, s.RQ)) return false;
        s.ODD = s.RQ & 1;
        s.NP = s.RQ / 2;
        s.NT = (s.RQ + 3) / 4;
        s.NTS = (s.NT + 2 * s.ODD + 3) / 4;
        s.KS = s.NP + s.NTS;
        // one instantiation: 81 kernel rows (9 x 9) = 10 pair steps, 6 quads, an odd row.  (The run-time form of the step
        // structure compiles, and is correct, but spills: hipcc keeps two copies of the accumulators across its joins.)
        if (s.NP != 10 || s.NT != 6 || s.ODD != 1) return false;
        if (s.ZP * s.XP > 32767) continue;
        if (lds_bytes(s) > (size_t)kMaxLds) continue;
        found = true;
        if (s.ntiles >= 4 * cus) break;
    }
    return found;
}

#include "conv_i8z.inc"

}  // namespace

namespace sn {

int conv_bank_group(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X, int Y,
                    int G, int Gtot, int g0, int head, int kz, int kx, int ky, void* act, void* out, int out_dtype,
                    sn_stream_t stream);   // conv.hip

// returns SN_OK, an error, or 1 when this shape is not served here (caller tries conv_occ_i8, then fp32)
int conv_occ_i8s(const uint8_t* x, const float* bank, const float* lambdas, int B, int Z, int X, int Y, int G, int Gtot,
                 int g0, int head, int kz, int kx, int ky, void* act, void* out, int out_dtype, hipStream_t stream) {
    if (ky != 9 || kz * kx != 81 || Y % 16 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0 || G > 16) return 1;
    if (act && (reinterpret_cast<uintptr_t>(act) & 15)) return 1;
    if (out && (reinterpret_cast<uintptr_t>(out) & 15)) return 1;
    if (sn::option_conv_skip_empty_tiles()) return 1;   // data-dependent tile skipping lives in conv_i8.hip
    Shape s;
    memset(&s, 0, sizeof(s));
    s.B = B; s.Z = Z; s.X = X; s.Y = Y; s.G = G; s.kz = kz; s.kx = kx;
    s.Gtot = Gtot; s.g0 = g0; s.head = head;
    s.gate = sn::current_gate();
    s.sticky = sn::sticky_device_ptr();
    s.nyt = (Y + TY - 1) / TY;
    // wrong-result / timing switches: -DSN_CONV_DEBUG builds only (common.h); 0 in the product
    s.dbg = sn::debug_env_int("SN_CONV_I8_DBG");
    s.stagger = sn::debug_env_int("SN_CONV_I8S_STAGGER");   // x 64 clocks; [measured] 100 (about half a round) vs 0: 0.2045 vs 0.2039 ms -- rounds are claimed, the waves spread by themselves
    s.dynamic = !sn::debug_env_int("SN_CONV_I8S_STATIC");
    const int cus = num_cus();
    const bool found = plan_stride4(s, B, Z, X, Y, kz, kx, cus);
    if (!found) return 1;
    // quantisation guard (see the header): tolerance on the worst-case activation error of the int8 path
    s.tol = sn::option_conv_i8_tolerance();
    const int grid = cus < s.ntiles ? cus : s.ntiles;
    // 1. the folded kernel (banks symmetric in x and y: every GENEO bank) -- it checks the symmetry on the device, runs the
    //    stride-4 body itself for a bank that is not, and leaves *flag = 0 (served) or 1 (bound exceeded: fp32 kernel)
    int32_t* flag = sn::device_flag_slot(stream);
    const bool fold = kz == 9 && kx == 9 && flag && sn::option_conv_i8_fold() &&
                      s.dbg == 0;   // (the debug switches belong to the stride-4 kernel)
    bool folded = false;
    if (fold) {
        Shape sf = s;
        plan_fold(sf.XP, sf.fplan);
        sf.route = flag;
        const size_t ldsf = lds_bytes_fold(sf) > lds_bytes(s) ? lds_bytes_fold(sf) : lds_bytes(s);   // (it may run the stride-4 body)
        if (ldsf <= (size_t)kMaxLds && (sf.XP == 17 || sf.XP == 13 || sf.XP == 11)) {
#define SN_LAUNCH_I8F(OT, XPV)                                                                                   \
    do {                                                                                                         \
        auto kern = conv_occ_i8f_kernel<OT, XPV>;                                                                \
        if (sn::ensure_dynamic_lds((const void*)kern, kMaxLds) != hipSuccess)                                    \
            return check_launch("sn_conv_bank(i8f: hipFuncSetAttribute)");                                       \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), ldsf, stream, x, bank, lambdas, sf, (OT*)act,       \
                           (OT*)out);                                                                            \
    } while (0)
#define SN_LAUNCH_I8F_XP(XPV)                                                                                    \
    do {                                                                                                         \
        if (out_dtype == SN_F32) SN_LAUNCH_I8F(float, XPV);                                                      \
        else SN_LAUNCH_I8F(double, XPV);                                                                         \
    } while (0)
            if (sf.XP == 17) SN_LAUNCH_I8F_XP(17);        // TX = 8
            else if (sf.XP == 13) SN_LAUNCH_I8F_XP(13);   // TX = 4
            else SN_LAUNCH_I8F_XP(11);                    // TX = 2
#undef SN_LAUNCH_I8F_XP
#undef SN_LAUNCH_I8F
            if (int rc = check_launch("sn_conv_bank(i8f)")) return rc;
            folded = true;
        }
    }
    // 2. the stride-4 kernel as its own launch, when the folded kernel was not tried (it runs the stride-4 body itself for
    //    a bank it declines)
    s.route = folded ? flag : nullptr;
    if (!folded) {
        if (s.tol > 0.0f) {
            s.route = flag;
            if (!s.route) s.tol = 0.0f;   // no flag memory: run unguarded rather than fail
        }
        const size_t lds = lds_bytes(s);
#define SN_LAUNCH_I8S(OT)                                                                                        \
    do {                                                                                                         \
        auto kern = conv_occ_i8s_kernel<OT, 10, 6, 1>;                                                           \
        if (sn::ensure_dynamic_lds((const void*)kern, kMaxLds) != hipSuccess)                                    \
            return check_launch("sn_conv_bank(i8s: hipFuncSetAttribute)");                                       \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, x, bank, lambdas, s, (OT*)act,         \
                           (OT*)out);                                                                            \
    } while (0)
        if (out_dtype == SN_F32) SN_LAUNCH_I8S(float);
        else SN_LAUNCH_I8S(double);
#undef SN_LAUNCH_I8S
        if (int rc = check_launch("sn_conv_bank(i8s)")) return rc;
    }
    if (s.route && s.tol > 0.0f) {
        // 3. the same launch on the fp32 matrix pipe, enqueued behind: runs only if a guard above sent it here
        sn::GateScope guard(s.route, 1);
        return sn::conv_bank_group(x, SN_U8, bank, lambdas, B, Z, X, Y, G, Gtot, g0, head, kz, kx, ky, act, out,
                                   out_dtype, reinterpret_cast<sn_stream_t>(stream));
    }
    return SN_OK;
}

// The z-walk over pre-folded planes (conv_i8z.inc) for a bank prepared by sn_conv_bank_prep.  Returns SN_OK, an error, or
// 1 when this shape is not served here (caller: sn_conv_bank's own kernels).
int conv_occ_i8z(const uint8_t* x, const float* bank, const float* lambdas, uint8_t* prep, int B, int Z, int X, int Y,
                 int G, int Gtot, int g0, int head, int kz, int kx, int ky, void* act, void* out, int out_dtype,
                 hipStream_t stream, int assume_served) {
    if (ky != 9 || kz != 9 || kx != 9 || Y % 16 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0 || G > 16) return 1;
    if ((long long)Z * X * Y >= (1ll << 31)) return 1;   // offsets inside a tile are 32-bit in the walk
    if (act && (reinterpret_cast<uintptr_t>(act) & 15)) return 1;
    if (out && (reinterpret_cast<uintptr_t>(out) & 15)) return 1;
    if (!prep || (reinterpret_cast<uintptr_t>(prep) & 15)) return 1;
    if (sn::option_conv_skip_empty_tiles() || !sn::option_conv_i8_fold() || sn::option_conv_i8_legacy()) return 1;
    const int cus = num_cus();
    // the stride-4 kernel's plan: what the launch runs, in place, for a bank that is not symmetric
    Shape s4;
    memset(&s4, 0, sizeof(s4));
    s4.B = B; s4.Z = Z; s4.X = X; s4.Y = Y; s4.G = G; s4.kz = kz; s4.kx = kx;
    s4.Gtot = Gtot; s4.g0 = g0; s4.head = head;
    s4.gate = sn::current_gate();
    s4.nyt = (Y + TY - 1) / TY;
    s4.dynamic = 1;
    if (!plan_stride4(s4, B, Z, X, Y, kz, kx, cus)) return 1;
    s4.tol = sn::option_conv_i8_tolerance();
    s4.route = nullptr;   // (the fallback launch reads the walk's verdict; the body must not rewrite it under other workgroups)
    int32_t* const route = reinterpret_cast<int32_t*>(prep + kPrepRoute);
    fp32k::ConvShape cs;
    bool dbl = false;
    if (!fp32k::plan_fp32(cs, B, Z, X, Y, G, Gtot, g0, head, kz, kx, ky, cus, false, dbl)) return 1;
    ZShape z;
    memset(&z, 0, sizeof(z));
    z.B = B; z.Z = Z; z.X = X; z.Y = Y; z.G = G; z.Gtot = Gtot; z.g0 = g0; z.head = head;
    z.gate = s4.gate;
    z.tol = s4.tol;
    z.route = route;
    z.dbg = sn::option_conv_i8z_inject_fault() ? 1 : 0;
    z.served = assume_served ? 1 : 0;
    z.sticky = sn::sticky_device_ptr();
    s4.sticky = z.sticky;
    z.nxt = (X + kZTX - 1) / kZTX;
    z.nyt = (Y + TY - 1) / TY;
    const long long ncol = (long long)B * z.nxt * z.nyt;
    if (ncol > (long long)cus * kZMaxJobs) return 1;
    // z segments per column: whole columns when they fill the chip evenly, else the split that minimises the planes the
    // busiest workgroup walks (a segment re-fetches 8 halo planes)
    int best_seg = 1;
    long long best_cost = -1;
    const int max_seg = (Z + 7) / 8;
    for (int ns = 1; ns <= max_seg; ++ns) {
        const int lz = (Z + ns - 1) / ns;
        const int nseg = (Z + lz - 1) / lz;
        const long long jobs = ncol * nseg;
        const long long per_wg = (jobs + cus - 1) / cus;
        if (per_wg > kZMaxJobs) continue;
        const long long cost = per_wg * (lz + 8) + kZD;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_seg = ns; }
    }
    if (best_cost < 0) return 1;
    z.LZ = (Z + best_seg - 1) / best_seg;
    z.nseg = (Z + z.LZ - 1) / z.LZ;
    z.PL = z.LZ + 8;
    const long long njobs = ncol * z.nseg;
    if (njobs > 0x7fffffff) return 1;
    z.njobs = (int)njobs;
    const int grid = (int)(njobs < cus ? njobs : cus);
    const int per_wg = (z.njobs + grid - 1) / grid;
    if (per_wg > kZMaxJobs) return 1;
    if (!division_magic((unsigned)z.PL, (unsigned)(per_wg * z.PL + 64), z.pl_magic)) return 1;
    z.swizzle = (grid % 8 == 0) ? 1 : 0;
    const size_t lds = lds_bytes_zwalk();
    const size_t lds_4 = lds_bytes(s4), lds_f = fp32k::lds_bytes(cs, false);
    const size_t lds_fb = lds_4 > lds_f ? lds_4 : lds_f;
    if (lds > (size_t)kMaxLds || lds_fb > (size_t)kMaxLds) return 1;
#define SN_LAUNCH_I8Z(OT, KH, KR, KW)                                                                            \
    do {                                                                                                         \
        auto kern = act ? conv_occ_i8z_kernel<OT, KH, KR, KW, true> : conv_occ_i8z_kernel<OT, KH, KR, KW, false>; \
        if (sn::ensure_dynamic_lds((const void*)kern, kMaxLds) != hipSuccess)                                    \
            return check_launch("sn_conv_bank_prepared(i8z: hipFuncSetAttribute)");                              \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * KW), lds, stream, x, lambdas, (const uint8_t*)prep, z,       \
                           (OT*)act, (OT*)out);                                                                  \
    } while (0)
#define SN_LAUNCH_I8Z_V(KH, KR, KW)                                                                              \
    do {                                                                                                         \
        if (out_dtype == SN_F32) SN_LAUNCH_I8Z(float, KH, KR, KW);                                               \
        else SN_LAUNCH_I8Z(double, KH, KR, KW);                                                                  \
    } while (0)
    // the shape of a ticket: rounds of two x-rows on 8 waves (2 per SIMD, the tile kernels' round), or rounds of one
    // x-row -- half the accumulator registers -- on 12 waves (3 per SIMD), one or two of them per ticket
    switch (sn::option_conv_i8z_variant()) {
        case 0: SN_LAUNCH_I8Z_V(2, 1, 8); break;
        case 1: SN_LAUNCH_I8Z_V(1, 1, 12); break;
        default: SN_LAUNCH_I8Z_V(1, 2, 12); break;
    }
#undef SN_LAUNCH_I8Z_V
#undef SN_LAUNCH_I8Z
    if (int rc = check_launch("sn_conv_bank_prepared(i8z)")) return rc;
    // the combined fallback launch: exits at once unless the walk's verdict sent the job on (unfolded int8 body / fp32 form).
    // assume_served: the caller has READ this blob's verdict (0) for these very weights, tolerance and outputs -- the
    // verdict depends on nothing else -- so the launch would do nothing; it is left out (~3.3 us of an empty 256-workgroup
    // dispatch).  The walk still writes its verdict: a caller that assumed wrongly can tell (sn_conv_prep_verdict).
    if (!assume_served) {
        const int fgrid = cus;
#define SN_LAUNCH_FB(OT)                                                                                         \
    do {                                                                                                         \
        auto kern = conv_i8_fallback_kernel<OT>;                                                                 \
        if (sn::ensure_dynamic_lds((const void*)kern, kMaxLds) != hipSuccess)                                    \
            return check_launch("sn_conv_bank_prepared(fallback: hipFuncSetAttribute)");                         \
        hipLaunchKernelGGL(kern, dim3(fgrid), dim3(kThreads), lds_fb, stream, x, bank, lambdas, (const int32_t*)route, s4, \
                           cs, (OT*)act, (OT*)out);                                                              \
    } while (0)
        if (out_dtype == SN_F32) SN_LAUNCH_FB(float);
        else SN_LAUNCH_FB(double);
#undef SN_LAUNCH_FB
    }
    return check_launch("sn_conv_bank_prepared(fallback)");
}

}  // namespace sn

extern "C" int sn_conv_bank_prep(const float* bank, int G, int kz, int kx, int ky, void* prep, sn_stream_t stream) {
    if (!bank || !prep) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank_prep: null pointer");
    if (G <= 0 || kz <= 0 || kx <= 0 || ky <= 0) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank_prep: non-positive extent");
    if (reinterpret_cast<uintptr_t>(prep) & 15) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank_prep: prep must be 16-byte aligned");
    if (kz != 9 || kx != 9 || ky != 9) return SN_OK;   // other shapes: sn_conv_bank_prepared does not read the blob
    for (int g0 = 0; g0 < G; g0 += 16) {
        const int gc = (G - g0 < 16) ? G - g0 : 16;
        hipLaunchKernelGGL(conv_prep_kernel, dim3(16), dim3(256), 0, sn::as_stream(stream), bank + (size_t)g0 * 729, gc,
                           static_cast<uint8_t*>(prep) + (size_t)(g0 / 16) * SN_CONV_PREP_BYTES);
    }
    return sn::check_launch("sn_conv_bank_prep");
}

static int conv_bank_prepared_impl(const void* x, int x_dtype, const float* bank, const float* lambdas, void* prep,
                                   int B, int Z, int X, int Y, int G, int kz, int kx, int ky, void* act, void* out,
                                   int out_dtype, sn_stream_t stream, int assume_served) {
    if (!x || !bank) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank_prepared: null x or bank");
    if (!act && !out) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank_prepared: both act and out are null");
    if (out && !lambdas) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank_prepared: out needs lambdas");
    if (B <= 0 || Z <= 0 || X <= 0 || Y <= 0 || G <= 0 || kz <= 0 || kx <= 0 || ky <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank_prepared: non-positive extent");
    if (out_dtype != SN_F32 && out_dtype != SN_F64)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_bank_prepared: out_dtype %d", out_dtype);
    if (x_dtype != SN_OCC8 || !prep || kz != 9 || kx != 9 || ky != 9 || (size_t)B * Z * X * Y > (size_t)1 << 40)
        return sn_conv_bank(x, x_dtype, bank, lambdas, B, Z, X, Y, G, kz, kx, ky, act, out, out_dtype, stream);
    for (int g0 = 0; g0 < G; g0 += 16) {
        const int gc = (G - g0 < 16) ? G - g0 : 16;
        const int head = G > 16 ? ((g0 > 0 ? 1 : 0) | (g0 + gc >= G ? 2 : 0)) : 2;
        int rc = sn::conv_occ_i8z((const uint8_t*)x, bank + (size_t)g0 * 729, lambdas ? lambdas + g0 : nullptr,
                                  static_cast<uint8_t*>(prep) + (size_t)(g0 / 16) * SN_CONV_PREP_BYTES, B, Z, X, Y, gc, G,
                                  g0, head, kz, kx, ky, act, out, out_dtype, sn::as_stream(stream), assume_served);
        if (rc == 1)   // shape not served by the z-walk: this group through sn_conv_bank's own kernels
            rc = sn::conv_bank_group(x, x_dtype, bank + (size_t)g0 * 729, lambdas ? lambdas + g0 : nullptr, B, Z, X, Y, gc,
                                     G, g0, head, kz, kx, ky, act, out, out_dtype, stream);
        if (rc != SN_OK) return rc;
    }
    return SN_OK;
}

extern "C" int sn_conv_bank_prepared(const void* x, int x_dtype, const float* bank, const float* lambdas, void* prep,
                                     int B, int Z, int X, int Y, int G, int kz, int kx, int ky, void* act, void* out,
                                     int out_dtype, sn_stream_t stream) {
    return conv_bank_prepared_impl(x, x_dtype, bank, lambdas, prep, B, Z, X, Y, G, kz, kx, ky, act, out, out_dtype, stream, 0);
}

extern "C" int sn_conv_bank_prepared_served(const void* x, int x_dtype, const float* bank, const float* lambdas, void* prep,
                                            int B, int Z, int X, int Y, int G, int kz, int kx, int ky, void* act,
                                            void* out, int out_dtype, sn_stream_t stream) {
    return conv_bank_prepared_impl(x, x_dtype, bank, lambdas, prep, B, Z, X, Y, G, kz, kx, ky, act, out, out_dtype, stream, 1);
}

extern "C" int sn_conv_prep_verdict_offset(void) { return kPrepRoute; }

extern "C" int sn_conv_i8_path_counts(unsigned long long* counts3) {
    if (!counts3) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_i8_path_counts: null pointer");
    unsigned long long h[4] = {0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fold_counts), sizeof(h)) != hipSuccess)
        return sn::check_launch("sn_conv_i8_path_counts");
    counts3[0] = h[0]; counts3[1] = h[1]; counts3[2] = h[2];
    return SN_OK;
}

extern "C" int sn_conv_i8_spin_timeouts(unsigned long long* count) {
    if (!count) return sn::fail(SN_ERR_INVALID_ARG, "sn_conv_i8_spin_timeouts: null pointer");
    unsigned long long h[4] = {0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fold_counts), sizeof(h)) != hipSuccess)
        return sn::check_launch("sn_conv_i8_spin_timeouts");
    *count = h[3];
    return SN_OK;
}

#ifdef SN_CONV_TIMING
extern "C" void sn_debug_i8s_times(unsigned long long* host) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_i8s_t), sizeof(unsigned long long) * 1024 * 16);
    (void)hipMemcpyFromSymbol(host + 1024 * 16, HIP_SYMBOL(g_i8s_w), sizeof(unsigned long long) * 1024 * 64);
}
#endif
