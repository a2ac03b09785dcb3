// K3' -- GENEO bank convolution for BINARY OCCUPANCY input on the int8 matrix cores.
//
// Same contraction as csrc/conv.hip (SceneNet.forward, core/models/SCENE_Net.py:322-339) specialised to what
// the network is actually fed: ToFullDense(Voxelization(points)) is {0,1} (torch_transforms.py:33-34), exact
// in int8.  The fp32 GENEO weights are turned into 24-bit fixed point per kernel,
//     Q[g][t] = rint(W[g][t] * S_g) (in fp64),  S_g = 8355711 / max_t |W[g][t]|   (|Q| <= 8355711 = 127 * 65793,
// the whole range of three balanced base-256 digits)
// and split into the digits Q = d0 + 256 d1 + 65536 d2, d_i in [-128, 127].  Three
// v_mfma_i32_16x16x64_i8 per 64 taps accumulate S_i = sum_t d_i[g][t] x[v+t] EXACTLY in int32 (order
// independent, bit-reproducible); the epilogue recombines (S2*65536 + S1*256 + S0) * (max|W_g| / 8355711) in fp32.
// Error vs the fp64 reference: the weight quantisation only, <= 0.5 * max|W_g| / 8355711 = 6e-8 max|W_g| per tap,
// plus the fp32 roundings of the recombination.  The prologue computes the exact worst case over all binary inputs
// (max of the summed positive and the summed negative errors, per kernel and lambda-weighted); a bank whose bound
// exceeds the tolerance (sn_set_option "conv_i8_tolerance_ppb") is handed to the fp32 kernel: *route = 1.
//
// GEMM view per 16-voxel strip along y: M = 16 kernels (A = digit bytes), N = 16 voxels (B = occupancy
// bytes), K = 64 "slots" per MFMA.  A lane (q = l>>4, n = l&15) supplies 16 bytes = 4 dwords per MFMA; each
// dword is a CHUNK = 4 consecutive dy taps of one (dz,dx) kernel row at voxel n.  ky taps are padded to
// C = ceil(ky/4) chunks (zero digits), so K is R*C*4 slots, R = kz*kx  (9^3: 972 slots, 16 MFMA steps x 3
// digits = 48 MFMAs of 16 cycles per 16x16 outputs, against 184 fp32 MFMAs of 32 cycles).
//
// A chunk starts at an arbitrary byte (voxel y + 4c), but ds_read_b32 wants 4-byte alignment: the halo tile
// is kept in LDS as FOUR copies, copy k shifted by k bytes, so lane n reads copy (n+delta)&3 at the aligned
// address below its start.  Both operands use the same slot->tap table, so only the row/column lane maps of
// the MFMA matter (A row = l&15, B col = l&15, D row = 4*(l>>4)+r, col = l&15), not the k order inside it.
#include "common.h"
#include <mutex>
#include <cstdlib>

namespace {

using i32x4 = __attribute__((ext_vector_type(4))) int;

// relu(tanh(v)): 0 for v <= 0, else 1 - 2 / (exp(2v) + 1) on the hardware exp / rcp (abs. error ~2e-7, inside the 1e-4
// bar; same form as conv_lin.hip and conv_i8s.hip).  NaN stays NaN like torch.relu(torch.tanh(.)); +inf -> 1.
__device__ __forceinline__ float relu_tanh(float v) {
    // branch-free, v_exp_f32 + v_rcp_f32 (1 ulp; the correctly rounded reciprocal was ten instructions, twice per lane and round)
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f);
    const float r = (v > 0.0f) ? t : 0.0f;
    return (v != v) ? v : r;
}

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int TY = 64;        // y extent of a workgroup tile
constexpr int NV = 8;         // accumulator tiles per wave round: 2 x-rows x 4 y-strips
// halo row stride in bytes, two builds of the kernel:
//   96 (24 dwords: rows 2 apart differ by 16 banks) -- ky up to 24, halo refilled synchronously from global;
//   80 (20 dwords: rows 4 apart differ by 16 banks) -- ky up to 12; the 16 KiB saved hold a STAGING copy of the
//      next tile's raw rows, filled by LDS-DMA (global_load_lds, no VGPRs) while the MFMA rounds run.
constexpr int kMaxC = 6;      // chunks per kernel row: ky <= 24
constexpr int kCopyPad = 16;  // copy stride = tile bytes + 16: the 4 copies start 4 banks apart
constexpr int kMaxLds = 160 * 1024;
constexpr int kTablePad = 0;  // the pipeline's look-ahead past the last step re-reads the last step (never consumed)

struct Shape {
    int B, Z, X, Y, G;
    int kz, kx, ky;
    int TZ, TX, nzt, nxt, nyt, ntiles;
    int C, R, KS;   // chunks per row, kernel rows, MFMA steps (even)
    int PYA, delta; // halo origin = y0 - PYA (PYA = roundup(py, 4)), delta = PYA - py
    int CB;         // bytes between the shifted copies
    int Gtot, g0, head;  // kernel group of a larger bank: act channel stride/offset; head bits (see conv.hip)
    int XPAD;       // unused halo rows appended to every z plane (bank placement, see conv_occ_i8)
    sn::Gate gate;       // run only if every condition holds (common.h: Gate)
    int32_t* route; // out: 1 = quantisation bound exceeded (the gated fp32 launch behind takes over), 0 = done here
    float tol;      // bound on the worst-case activation error allowed here (<= 0: no check)
    int perm;       // tile order multiplier, coprime to ntiles
    int skip_empty; // opt-in: skip the MFMA steps of halo tiles without a set voxel (result is exactly 0)
    int dbg;        // timing experiments only (SN_CONV_I8_DBG): 1 = no epilogue, 2 = no halo refill, 4 = no barrier,
                    // 8 = prologue only, 16 = prologue + first halo fill only
};

struct TileCoord {
    int b, z0, x0, y0;
};

// Workgroup w handles virtual tiles w, w+grid, ...; virtual tile t is actual tile (t * perm) mod ntiles, perm coprime
// to ntiles.  With the plain order a workgroup's tiles all share (x-tile, z-tile) residues, i.e. the same kind of
// region of every sample -- fine for dense work, but with conv_skip_empty_tiles some workgroups would get only empty
// tiles and others none.
__device__ __forceinline__ TileCoord tile_coord(const Shape& s, int tile) {
    TileCoord c;
    tile = (int)(((long long)tile * s.perm) % s.ntiles);
    c.y0 = (tile % s.nyt) * TY; tile /= s.nyt;
    c.x0 = (tile % s.nxt) * s.TX; tile /= s.nxt;
    c.z0 = (tile % s.nzt) * s.TZ; tile /= s.nzt;
    c.b = tile;
    return c;
}

__device__ uint32_t g_zero_word[4] = {0u, 0u, 0u, 0u};

// A global load the compiler does not schedule or count (G > 16: later chunks add to the partial sum in `out`).
// As a plain load hipcc keeps it "possibly in flight" around the round loop and opens every round with
// s_waitcnt vmcnt(..0), which also waits for the halo DMA issued at the top of the tile.
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

// The tile ticket, likewise outside the compiler's bookkeeping (and waited for on the spot, by the one thread that
// draws it): as a plain atomicAdd whose result is used after the rounds, hipcc flushes vmcnt before entering the
// round loop -- on the dense path too, where no ticket is drawn but the halo DMA is in flight.
__device__ __forceinline__ int ticket_draw(int* ticket) {
    int v;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)"
                 : "=v"(v) : "v"(ticket), "v"(1) : "memory");
    return v;
}

// workgroup barrier that orders LDS traffic only: global loads (and LDS-DMA) in flight stay in flight
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// slot (step s, lane group q, dword j) -> chunk list index.  The list is c-major (all rows of chunk column
// 0, then column 1, ...), and the two lane groups that share an LDS cycle (q = 0,1 and q = 2,3) are placed 2
// (YPB = 96) or 4 (YPB = 80) list entries = halo rows = 16 banks apart.
template <int YPB>
__device__ __forceinline__ int slot_index(int s, int q, int j) {
    if (YPB == 96) {
        const int perm = ((q & 1) << 1) | (q >> 1);  // 0,2,1,3: lane groups q, q^1 are 2 rows apart
        return s * 16 + j * 4 + perm;
    }
    return s * 16 + q * 4 + j;  // YPB = 80: lane groups q, q^1 are 4 rows apart
}

// halo tile: copies[k][r][i] (bytes), copy k = tile shifted by k bytes.  Global rows are read as aligned
// dwords (Y % 4 == 0, origin y0 - PYA is a multiple of 4): out-of-grid dwords come from a zero word.
template <int YPB>
__device__ __forceinline__ uint32_t halo_fill(uint8_t* __restrict__ xs, const uint8_t* __restrict__ x,
                                              const Shape& s, const TileCoord& c, int tid, int XP, int rows) {
    uint32_t seen = 0u;  // OR of every dword this thread staged
    constexpr int DW = YPB / 4;
    const int total = rows * DW;
    constexpr int kBatch = 11;  // (16*16 rows x 24 dwords) / 512 threads = 10.5: one batch, one latency
    for (int base = tid; base < total; base += kThreads * kBatch) {
        uint32_t lo[kBatch], hi[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int idx = base + u * kThreads;
            const int r = idx / DW, i = idx - r * DW;
            const int zz = r / XP, xx = r - zz * XP;
            const int gz = c.z0 - (s.kz - 1) / 2 + zz, gx = c.x0 - (s.kx - 1) / 2 + xx;
            const int gy = c.y0 - s.PYA + 4 * i;
            const bool okr = (idx < total && gz >= 0 && gz < s.Z && gx >= 0 && gx < s.X);
            const uint8_t* row = x + (((size_t)c.b * s.Z + gz) * s.X + gx) * s.Y;
            const uint32_t* p0 = (okr && gy >= 0 && gy < s.Y) ? reinterpret_cast<const uint32_t*>(row + gy)
                                                             : g_zero_word;
            const uint32_t* p1 = (okr && gy + 4 >= 0 && gy + 4 < s.Y) ? reinterpret_cast<const uint32_t*>(row + gy + 4)
                                                                     : g_zero_word;
            lo[u] = *p0;
            hi[u] = *p1;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int idx = base + u * kThreads;
            if (idx < total) {
                seen |= lo[u];
                uint32_t* d = reinterpret_cast<uint32_t*>(xs) + idx;  // dword idx of copy 0
                d[0] = lo[u];
                d[(s.CB >> 2)] = __builtin_amdgcn_alignbyte(hi[u], lo[u], 1);
                d[2 * (s.CB >> 2)] = __builtin_amdgcn_alignbyte(hi[u], lo[u], 2);
                d[3 * (s.CB >> 2)] = __builtin_amdgcn_alignbyte(hi[u], lo[u], 3);
            }
        }
    }
    return seen;
}

// LDS-DMA of the next tile's raw rows: item idx = (row r, dword i) lands at stage[idx]; a wave-instruction
// writes 64 consecutive dwords (LDS address = wave-uniform base + lane*4), the global address is per lane
// (out-of-grid dwords are fetched from a zero word).  No VGPR destination, nothing to keep live.
template <int YPB>
__device__ __forceinline__ void halo_dma_issue(uint32_t* __restrict__ stage, const uint8_t* __restrict__ x,
                                               const Shape& s, const TileCoord& c, int wave, int lane, int XP,
                                               int rows) {
    constexpr int DW = YPB / 4;
    const int total = rows * DW;
    // Address arithmetic kept short -- every wave runs this once per tile next to the MFMA rounds: the row split
    // r -> (zz, xx) by a multiply-high (exact for r < 2^16), 32-bit offsets under one uniform 64-bit sample base.
    const uint32_t magic = 0xFFFFFFFFu / (uint32_t)XP + 1u;
    const uint8_t* tb = x + (size_t)c.b * s.Z * s.X * s.Y;
    const int oz = c.z0 - (s.kz - 1) / 2, ox = c.x0 - (s.kx - 1) / 2, oy = c.y0 - s.PYA;
    for (int base = wave * 64; base < total; base += kThreads) {
        const int idx = base + lane;
        const int r = idx / DW, i = idx - r * DW;
        const int zz = (int)__umulhi((uint32_t)r, magic), xx = r - zz * XP;
        const int gz = oz + zz, gx = ox + xx, gy = oy + 4 * i;
        const bool ok = idx < total && (unsigned)gz < (unsigned)s.Z && (unsigned)gx < (unsigned)s.X &&
                        (unsigned)gy < (unsigned)s.Y;
        const void* src = ok ? static_cast<const void*>(tb + ((gz * s.X + gx) * s.Y + gy))
                             : static_cast<const void*>(g_zero_word);
        // Inline asm on purpose: hipcc treats the builtin's LDS write as aliasing every later LDS read and puts
        // s_waitcnt vmcnt(0) in front of the MFMA rounds' first ds_read -- the DMA latency it is meant to hide.
        // The staging area is only read after the explicit vmcnt(0) + barrier at the end of the tile.  (vmcnt
        // retires in order, so the compiler's own counted waits, unaware of these loads, can only over-wait.)
        const uint32_t lds_base =
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(stage + base);
        uint32_t m0_saved;   // M0 carries the LDS base of the DMA; hand it back as found
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(src) : "memory");
    }
}

// staging -> the four byte-shifted copies (pure LDS traffic)
template <int YPB>
__device__ __forceinline__ uint32_t halo_expand(uint8_t* __restrict__ xs, const uint32_t* __restrict__ stage,
                                                const Shape& s, int tid, int rows) {
    constexpr int DW = YPB / 4;
    const int total = rows * DW;
    const int cs = s.CB >> 2;
    uint32_t seen = 0u;
    for (int idx = tid; idx < total; idx += kThreads) {
        const uint32_t lo = stage[idx];
        seen |= lo;
        const uint32_t hi = stage[idx + 1];  // last dword of a row: its shifted copies are never read back
        uint32_t* d = reinterpret_cast<uint32_t*>(xs) + idx;
        d[0] = lo;
        d[cs] = __builtin_amdgcn_alignbyte(hi, lo, 1);
        d[2 * cs] = __builtin_amdgcn_alignbyte(hi, lo, 2);
        d[3 * cs] = __builtin_amdgcn_alignbyte(hi, lo, 3);
    }
    return seen;
}

#ifdef SN_CONV_TIMING
__device__ unsigned long long g_conv_t[1024 * 16];
__device__ unsigned long long g_conv_w[1024 * 8];
#define SN_T(k) do { if (threadIdx.x == 0) g_conv_t[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define SN_TACC(k, t0) do { if (threadIdx.x == 0) g_conv_t[blockIdx.x * 16 + (k)] += wall_clock64() - (t0); } while (0)
#else
#define SN_T(k) do {} while (0)
#define SN_TACC(k, t0) do {} while (0)
#endif
template <typename OT, int YPB, bool kStage>
__global__ __launch_bounds__(kThreads) void conv_occ_i8_kernel(const uint8_t* __restrict__ x,
                                                               const float* __restrict__ bank,
                                                               const float* __restrict__ lambdas, Shape s,
                                                               int* __restrict__ ticket, OT* __restrict__ act,
                                                               OT* __restrict__ out) {
    if (!s.gate.pass()) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    SN_T(0);
#ifdef SN_CONV_TIMING
    if (tid == 0) { g_conv_t[blockIdx.x * 16 + 6] = 0; g_conv_t[blockIdx.x * 16 + 7] = 0; g_conv_t[blockIdx.x * 16 + 8] = 0; g_conv_t[blockIdx.x * 16 + 10] = 0; }
#endif

    const int ZP = s.TZ + s.kz - 1, XP = s.TX + s.kx - 1 + s.XPAD;
    const int rows = ZP * XP;
    const int ntaps = s.kz * s.kx * s.ky;
    const int KT = s.KS + kTablePad;
    // LDS carve-up
    uint4* Wd = reinterpret_cast<uint4*>(lds);                            // [KT][3][64] x 16 B
    int4* coff = reinterpret_cast<int4*>(lds + (size_t)KT * 3 * 64 * 16); // [KT][4] x 16 B
    float* scale = reinterpret_cast<float*>(coff + KT * 4);               // [16]   max|W_g| / 8355711
    double* Sq = reinterpret_cast<double*>(scale + 16);                   // [16]   8355711 / max|W_g|
    double* bnd = Sq + 16;                                                // [16]   worst-case quantisation error
    int* wseen = reinterpret_cast<int*>(bnd + 16);                        // [8]    per wave: halo tile has a set voxel
    int* tnext = wseen + 8;                                               // [1]    dynamic scheduling: next tile
    uint8_t* xs = reinterpret_cast<uint8_t*>(wseen + 16);                 // 4 copies x CB bytes
    uint32_t* stage = reinterpret_cast<uint32_t*>(xs + 4 * (size_t)s.CB); // kStage: [rows][YPB/4] (+1) raw dwords

    // ---- once per workgroup: the fp32 bank is staged in LDS (in the still unused halo area; coalesced,
    // batched loads), then the per-kernel fixed-point scale and the digit table are computed out of LDS.
    float* bank_s = reinterpret_cast<float*>(xs);  // [G][ntaps]
    {
        const int nb = s.G * ntaps;
        constexpr int kB = 24;   // 16 x 729 floats / 512 threads = 22.8: one batch of loads, one latency
        for (int base = tid; base < nb; base += kThreads * kB) {
            float v[kB];
#pragma unroll
            for (int u = 0; u < kB; ++u) {
                const int i = base + u * kThreads;
                v[u] = bank[i < nb ? i : 0];
            }
#pragma unroll
            for (int u = 0; u < kB; ++u) {   // branch-free: past-the-end items land in the padding slot bank_s[nb]
                const int i = base + u * kThreads;
                bank_s[i < nb ? i : nb] = v[u];
            }
        }
    }
    // the first tile's raw rows travel to the staging area (beyond the bank's alias) while the scales and the digit
    // table are worked out -- issued behind the bank's loads, whose wait would otherwise cover the DMA as well
    const bool dma_first = kStage && blockIdx.x < (unsigned)s.ntiles &&
                           (size_t)s.G * ntaps * sizeof(float) <= 4 * (size_t)s.CB;
    if (dma_first) halo_dma_issue<YPB>(stage, x, s, tile_coord(s, blockIdx.x), wave, lane, XP, rows);
    lds_barrier();   // not __syncthreads(): its fence would wait for the DMA just issued
    SN_T(1);
    for (int g = wave; g < 16; g += kWaves) {
        float m = 0.0f;
        if (g < s.G)
            for (int t = lane; t < ntaps; t += 64) {
                const float a = fabsf(bank_s[g * ntaps + t]);
                m = (a <= 3.0e38f) ? fmaxf(m, a) : __int_as_float(0x7fc00000);  // NaN / inf weight poisons the kernel
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float u = __shfl_xor(m, o, 64);
            m = (m != m || u != u) ? __int_as_float(0x7fc00000) : fmaxf(m, u);
        }
        const double S = (m > 0.0f) ? 8355711.0 / (double)m : 0.0;   // NaN: comparison false, scale = NaN below
        double ep = 0.0, en = 0.0;
        if (g < s.G && m > 0.0f)
            for (int t = lane; t < ntaps; t += 64) {
                const double w = (double)bank_s[g * ntaps + t];
                const double e = (double)__double2int_rn(w * S) * ((double)m / 8355711.0) - w;
                ep += e > 0.0 ? e : 0.0;
                en += e < 0.0 ? -e : 0.0;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ep += __shfl_xor(ep, o, 64);
            en += __shfl_xor(en, o, 64);
        }
        if (lane == 0) {
            Sq[g] = S;
            // fixed point cannot carry a NaN / inf weight: such a kernel's whole response is NaN, as it is in
            // conv3d (0 * NaN = NaN at every voxel)
            scale[g] = (m != m) ? m : (float)((double)m / 8355711.0);
            bnd[g] = ep > en ? ep : en;
        }
    }
    lds_barrier();
    SN_T(2);
    // route: every workgroup takes the same decision from the same numbers
    if (s.tol > 0.0f) {
        double worst = 0.0, mixed = 0.0;
        for (int g = 0; g < s.G; ++g) {
            worst = bnd[g] > worst ? bnd[g] : worst;
            if (out) mixed += fabs((double)lambdas[g]) * bnd[g];   // tanh and relu are 1-Lipschitz
        }
        const bool exceeded = (act && worst > (double)s.tol) || (out && mixed > (double)s.tol);
        if (blockIdx.x == 0 && tid == 0 && s.route) *s.route = exceeded ? 1 : 0;
        if (exceeded) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // do not leave with LDS-DMA in flight
            return;
        }
    } else if (blockIdx.x == 0 && tid == 0 && s.route) {
        *s.route = 0;
    }
    // ---- digit table Wd[s][d][l] (16 bytes: slot (q, p)) and chunk offset table coff[s][q] (4 dwords j)
    const int nchunks = s.R * s.C;
    for (int i = tid; i < KT * 64; i += kThreads) {  // one thread quantises 16 taps once and emits all 3 digit rows
        const int l = i & 63, st = i >> 6;
        const int g = l & 15, qq = l >> 4;
        uint32_t w0[4] = {0u, 0u, 0u, 0u}, w1[4] = {0u, 0u, 0u, 0u}, w2[4] = {0u, 0u, 0u, 0u};
        if (g < s.G && st < s.KS) {
            const double S = Sq[g];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int li = slot_index<YPB>(st, qq, j);
                if (li < nchunks) {
                    const int c = li / s.R, rho = li - c * s.R;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int dy = 4 * c + b;
                        if (dy < s.ky) {
                            int Q = __double2int_rn((double)bank_s[g * ntaps + rho * s.ky + dy] * S);
                            const int d0 = ((Q + 128) & 255) - 128;
                            Q = (Q - d0) >> 8;
                            const int d1 = ((Q + 128) & 255) - 128;
                            const int d2 = (Q - d1) >> 8;
                            w0[j] |= (uint32_t)(d0 & 255) << (8 * b);
                            w1[j] |= (uint32_t)(d1 & 255) << (8 * b);
                            w2[j] |= (uint32_t)(d2 & 255) << (8 * b);
                        }
                    }
                }
            }
        }
        Wd[(st * 3 + 0) * 64 + l] = make_uint4(w0[0], w0[1], w0[2], w0[3]);
        Wd[(st * 3 + 1) * 64 + l] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
        Wd[(st * 3 + 2) * 64 + l] = make_uint4(w2[0], w2[1], w2[2], w2[3]);
    }
    for (int i = tid; i < KT * 4; i += kThreads) {
        const int qq = i & 3, st = i >> 2;
        int o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int li = slot_index<YPB>(st, qq, j);
            if (st < s.KS && li < nchunks) {
                const int c = li / s.R, rho = li - c * s.R;
                const int dz = rho / s.kx, dx = rho - dz * s.kx;
                o[j] = (dz * XP + dx) * YPB + 4 * c;
            }
        }
        coff[i] = make_int4(o[0], o[1], o[2], o[3]);
    }

    SN_T(3);
    float lam[4] = {0.f, 0.f, 0.f, 0.f}, sc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int g = 4 * q + r;
        sc[r] = scale[g];
        if (out) lam[r] = (g < s.G) ? lambdas[g] * sc[r] : 0.0f;  // the rescale rides on lambda
    }

    const int half_tx = s.TX >> 1;
    const int nrounds = s.TZ * half_tx;
    const size_t V = (size_t)s.Z * s.X * s.Y;
    const int nd = n + s.delta;
    const int lanebase = (nd & 3) * s.CB + (nd & ~3);

    // Tile order: static (w, w+grid, ...) or, with a ticket counter (conv_skip_empty_tiles: tiles then cost very
    // different amounts), dynamic -- one thread draws the ticket of the tile after next while the rounds run.
    int tile = blockIdx.x;
    if (tile >= s.ntiles || SN_DBG(s, 8)) return;
    if (ticket && tid == 0) *tnext = gridDim.x + ticket_draw(ticket);
    if (dma_first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
    __syncthreads();  // the staged bank (aliasing the halo area) is dead from here on
    {
        const uint32_t seen = dma_first ? halo_expand<YPB>(xs, stage, s, tid, rows)
                                        : halo_fill<YPB>(xs, x, s, tile_coord(s, tile), tid, XP, rows);
        const bool w = __ballot(seen != 0u) != 0ull;
        if (lane == 0) wseen[wave] = w;
    }
    __syncthreads();
    if (SN_DBG(s, 16)) return;
    SN_T(4);

    while (tile < s.ntiles) {
#ifdef SN_CONV_TIMING
        const unsigned long long t_tile = wall_clock64();
        if (tid == 0) g_conv_t[blockIdx.x * 16 + 8] += 1;
#endif
        const TileCoord c = tile_coord(s, tile);
        const int next = ticket ? *tnext : tile + (int)gridDim.x;
        const bool has_next = (next < s.ntiles) && !SN_DBG(s, 2);
        if (kStage && has_next) halo_dma_issue<YPB>(stage, x, s, tile_coord(s, next), wave, lane, XP, rows);
        int after_next = 0;
        if (ticket && tid == 0) after_next = gridDim.x + ticket_draw(ticket);
        // opt-in (sn_set_option "conv_skip_empty_tiles"): a halo tile without a single set voxel convolves to
        // exactly 0 for every kernel -- its MFMA steps are skipped, the epilogue still writes the (zero) result
        int ks_run = s.KS;
        if (s.skip_empty) {
            int seen = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) seen |= wseen[w];
            if (!seen) ks_run = 0;
        }
        // The two waves of a SIMD do not advance evenly -- the older one wins the matrix pipe -- and a wave left alone
        // on its SIMD runs at about half rate [measured: 1.9x], so issue priority falls with the rounds a wave has
        // finished: whoever is behind its neighbour catches up.
        int rounds_done = 0;
        for (int round = wave; round < nrounds; round += kWaves) {
            if (rounds_done == 0) __builtin_amdgcn_s_setprio(3);
            else if (rounds_done == 1) __builtin_amdgcn_s_setprio(2);
            else if (rounds_done == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            ++rounds_done;
            // last round of the tile: ties go to the older wave, so the younger one leads for the first half
            const bool lead_half = (round + kWaves >= nrounds) && (wave >= kWaves / 2);
            if (lead_half) __builtin_amdgcn_s_setprio(1);
            const int lz = round / half_tx, lx = (round - lz * half_tx) * 2;
            const uint8_t* xb = xs + lanebase + (lz * XP + lx) * YPB;

            i32x4 acc[3][NV];
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[d][v] = i32x4{0, 0, 0, 0};

            // operands of accumulator tiles 2p, 2p+1 (two y-strips of one x-row): 4 ds_read2_b32
            auto gather_pair = [&](const int4& co, i32x4 (&xv)[NV], int p) {
                const uint8_t* p0 = xb + co.x;
                const uint8_t* p1 = xb + co.y;
                const uint8_t* p2 = xb + co.z;
                const uint8_t* p3 = xb + co.w;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int v = 2 * p + h;
                    const int to = (v >> 2) * YPB + (v & 3) * 16;
                    xv[v] = i32x4{*reinterpret_cast<const int*>(p0 + to), *reinterpret_cast<const int*>(p1 + to),
                                  *reinterpret_cast<const int*>(p2 + to), *reinterpret_cast<const int*>(p3 + to)};
                }
            };
            auto load_w = [&](int st, i32x4 (&w)[3]) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const uint4 u = Wd[(st * 3 + d) * 64 + lane];
                    w[d] = i32x4{(int)u.x, (int)u.y, (int)u.z, (int)u.w};
                }
            };
            auto mma_pair = [&](const i32x4 (&w)[3], const i32x4 (&xv)[NV], int p) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        acc[d][2 * p + h] =
                            __builtin_amdgcn_mfma_i32_16x16x64_i8(w[d], xv[2 * p + h], acc[d][2 * p + h], 0, 0, 0);
            };

            // Software pipeline at TILE-PAIR granularity.  lgkmcnt is a 4-bit counter: with a whole step's 20 LDS
            // reads in flight a counted wait cannot name "all but the last 20", so the prefetch distance is two
            // sub-blocks instead: sub-block (t, p) = {the 4 halo reads of pair p+2 of step t (p < 2) or of pair
            // p-2 of step t+1 (p >= 2); the 6 MFMAs of pair p of step t}.  One operand set X[8] serves every step
            // (a pair's registers are refilled two sub-blocks after their last use); the digit rows / chunk
            // offsets of the next step are requested at sub-block 0, a full step (or two sub-blocks) ahead.
            i32x4 wa[3], wb[3], X[NV];
            int4 ca = coff[q], cb;
            load_w(0, wa);
            gather_pair(ca, X, 0);
            gather_pair(ca, X, 1);
#define SN_SB(GC, GP, W, MP)                 \
    gather_pair(GC, X, GP);                  \
    mma_pair(W, X, MP);                      \
    __builtin_amdgcn_sched_barrier(0);
#define SN_TRIP                                                                                   \
    load_w(st + 1, wb);                                                                           \
    cb = coff[(st + 1) * 4 + q];                                                                  \
    SN_SB(ca, 2, wa, 0)                                                                           \
    SN_SB(ca, 3, wa, 1)                                                                           \
    SN_SB(cb, 0, wa, 2)                                                                           \
    SN_SB(cb, 1, wa, 3)                                                                           \
    const int s2 = (st + 2 < s.KS) ? st + 2 : s.KS - 1; /* look-ahead of the last trip: discarded */ \
    load_w(s2, wa);                                                                               \
    ca = coff[s2 * 4 + q];                                                                        \
    SN_SB(cb, 2, wb, 0)                                                                           \
    SN_SB(cb, 3, wb, 1)                                                                           \
    SN_SB(ca, 0, wb, 2)                                                                           \
    SN_SB(ca, 1, wb, 3)
            // two loops over one body: the priority changes between them, not behind a branch inside (a join in the
            // pipelined loop makes the compiler wait for every outstanding LDS read)
            const int ks_half = (ks_run >> 2) * 2;
            int st = 0;
            for (; st < ks_half; st += 2) { SN_TRIP }
            if (lead_half) __builtin_amdgcn_s_setprio(0);
            for (; st < ks_run - 2; st += 2) { SN_TRIP }
            if (st < ks_run) {
                // the round's last trip, peeled: no look-ahead (nothing follows), and no scheduling fences after its
                // first half, so the digit recombination of finished accumulator tiles can issue between the last MFMAs
                load_w(st + 1, wb);
                cb = coff[(st + 1) * 4 + q];
                SN_SB(ca, 2, wa, 0)
                SN_SB(ca, 3, wa, 1)
                gather_pair(cb, X, 0);
                mma_pair(wa, X, 2);
                gather_pair(cb, X, 1);
                mma_pair(wa, X, 3);
                gather_pair(cb, X, 2);
                mma_pair(wb, X, 0);
                gather_pair(cb, X, 3);
                mma_pair(wb, X, 1);
                mma_pair(wb, X, 2);
                mma_pair(wb, X, 3);
            }
#undef SN_TRIP
#undef SN_SB

            // ---- epilogue: recombine the digits, then the same head as the fp32 kernel
            const int gz = c.z0 + lz;
            if (gz >= s.Z) continue;
            if (SN_DBG(s, 1)) {  // timing experiment: keep the accumulators live, skip the epilogue
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int v = 0; v < NV; ++v) asm volatile("" ::"v"(acc[d][v]));
                continue;
            }
            // S = S2*65536 + (S1*256 + S0): the low pair fits int32 (|.| < 2^25), one fp32 rounding each
            if (act) {
                float val[NV][4];
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int low = acc[1][v][r] * 256 + acc[0][v][r];
                        val[v][r] = fmaf((float)acc[2][v][r], 65536.0f, (float)low);  // unscaled: x 2^-F_g below
                    }
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int gx = c.x0 + lx + (v >> 2), gy = c.y0 + (v & 3) * 16 + n;
                    if (gx < s.X && gy < s.Y) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int g = 4 * q + r;
                            if (g < s.G)
                                act[((size_t)c.b * s.Gtot + s.g0 + g) * V + ((size_t)gz * s.X + gx) * s.Y + gy] =
                                    (OT)(val[v][r] * sc[r]);
                        }
                    }
                }
            }
            if (out) {
                float sums[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    // sum_r lam_r (low_r + 65536 hi_r), two FMAs per kernel straight from the integer sums: the same
                    // operations in the same order as conv_i8s.hip's epilogue (bit-identical heads), with or without `act`
                    float p = 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int low = acc[1][v][r] * 256 + acc[0][v][r];
                        p = fmaf(lam[r], (float)low, p);
                        p = fmaf(65536.0f * lam[r], (float)acc[2][v][r], p);
                    }
                    p += __shfl_xor(p, 32, 64);   // (q, q+2) then (q, q+1): the association conv_i8s.hip's swaps give
                    p += __shfl_xor(p, 16, 64);
                    sums[v] = p;
                }
#pragma unroll
                for (int xr = 0; xr < 2; ++xr) {
                    const float a0 = sums[4 * xr + 0], a1 = sums[4 * xr + 1], a2 = sums[4 * xr + 2],
                                a3 = sums[4 * xr + 3];
                    const float sv = (q == 0) ? a0 : (q == 1) ? a1 : (q == 2) ? a2 : a3;
                    const int gx = c.x0 + lx + xr, gy = c.y0 + q * 16 + n;
                    if (gx < s.X && gy < s.Y) {
                        OT* o = out + (size_t)c.b * V + ((size_t)gz * s.X + gx) * s.Y + gy;
                        float t = sv;
                        if (s.head & 1) t += (float)load_now(o);
                        *o = (OT)((s.head & 2) ? relu_tanh(t) : t);
                    }
                }
            }
        }
        // [measured] prefetching the next halo into REGISTERS across the rounds (16 VGPRs) makes hipcc spill (the
        // kernel wants > 400 registers at 2 waves/SIMD) and a 1-wave/SIMD build runs 1.9x slower; LDS-DMA into a
        // staging area (kStage) needs no registers.
        SN_TACC(10, t_tile);   // wave 0's own rounds
#ifdef SN_CONV_TIMING
        if (lane == 0) g_conv_w[blockIdx.x * 8 + wave] += wall_clock64() - t_tile;
#endif
        if (kStage) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
        if (!SN_DBG(s, 4)) __syncthreads();  // every wave is done reading the halo tile (and every DMA landed)
        SN_TACC(6, t_tile);
#ifdef SN_CONV_TIMING
        const unsigned long long t_sw = wall_clock64();
#endif
        if (has_next) {
            const uint32_t seen = kStage ? halo_expand<YPB>(xs, stage, s, tid, rows)
                                         : halo_fill<YPB>(xs, x, s, tile_coord(s, next), tid, XP, rows);
            const bool w = __ballot(seen != 0u) != 0ull;
            if (lane == 0) wseen[wave] = w;
        }
        if (ticket && tid == 0) *tnext = after_next;  // everybody read the old value before the barrier above
        if (!SN_DBG(s, 4)) __syncthreads();
        SN_TACC(7, t_sw);
        tile = next;
    }
    SN_T(5);
}

size_t lds_bytes(const Shape& s, int ypb, bool stage) {
    const size_t KT = s.KS + kTablePad;
    const size_t rows = (size_t)(s.TZ + s.kz - 1) * (s.TX + s.kx - 1 + s.XPAD);
    const size_t halo = 4 * (size_t)s.CB + (stage ? rows * ypb + 16 : 0);
    const size_t staged_bank = ((size_t)s.G * s.kz * s.kx * s.ky + 1) * sizeof(float);  // aliases the halo area (+ 1 pad)
    return KT * 3 * 64 * 16 + KT * 4 * 16 + 16 * 4 + 16 * 8 + 16 * 8 + 16 * 4 + (halo > staged_bank ? halo : staged_bank);
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

}  // namespace

namespace sn {

int conv_bank_group(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X, int Y,
                    int G, int Gtot, int g0, int head, int kz, int kx, int ky, void* act, void* out, int out_dtype,
                    sn_stream_t stream);   // conv.hip

// returns SN_OK, an error, or 1 when this shape is not served by the int8 kernel (caller falls back to fp32)
int conv_occ_i8(const uint8_t* x, const float* bank, const float* lambdas, int B, int Z, int X, int Y, int G, int Gtot,
                int g0, int head, int kz, int kx, int ky, void* act, void* out, int out_dtype, hipStream_t stream) {
    if (Y % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 3) != 0 || G > 16) return 1;
    Shape s;
    s.B = B; s.Z = Z; s.X = X; s.Y = Y; s.G = G; s.kz = kz; s.kx = kx; s.ky = ky;
    s.Gtot = Gtot; s.g0 = g0; s.head = head;
    s.gate = sn::current_gate();
    s.C = (ky + 3) / 4;
    if (s.C > kMaxC) return 1;
    s.R = kz * kx;
    s.KS = (((s.R * s.C + 15) / 16) + 1) & ~1;
    const int py = (ky - 1) / 2;
    s.PYA = (py + 3) & ~3;
    s.delta = s.PYA - py;
    s.nyt = (Y + TY - 1) / TY;
    s.dbg = sn::debug_env_int("SN_CONV_I8_DBG");   // (0 in the product: common.h)
    s.skip_empty = sn::option_conv_skip_empty_tiles();
    const bool nostage = sn::option_extra(sn::kOptConvI8NoStage) != 0;
    const int cus = num_cus();
    const int need = s.delta + 15 + 48 + 4 * s.C + 3;  // bytes of a halo row the reads can touch
    // variants in order of preference: (row stride, LDS-DMA staging)
    const int variants[][2] = {{80, 1}, {96, 0}};
    static const int cand[][2] = {{8, 8}, {4, 8}, {4, 4}, {2, 4}, {1, 4}, {1, 2}};
    for (const auto& var : variants) {
        const int ypb = var[0];
        const bool stage = var[1] != 0;
        if (need > ypb) continue;
        if (stage && nostage) continue;
        bool found = false;
        for (const auto& c : cand) {
            s.TZ = c[0]; s.TX = c[1];
            s.nzt = (Z + s.TZ - 1) / s.TZ; s.nxt = (X + s.TX - 1) / s.TX;
            s.ntiles = B * s.nzt * s.nxt * s.nyt;
            // The two lane groups one LDS cycle serves read kernel rows 4 list entries apart: 4 halo rows = 16 banks
            // apart inside a z plane, but XP - kx + 4 rows apart when the pair straddles a plane.  Pad the plane
            // with unused rows until that distance is = 16 banks (mod 32) too; at kx = 9, 80-byte rows: 16 -> 17
            // rows ([measured] SQ_LDS_BANK_CONFLICT was 23 % of the LDS-active cycles without it).
            s.XPAD = 0;
            for (int pad = 0; pad < 8; ++pad) {
                const int xp = s.TX + kx - 1 + pad;
                const int apart = (ypb == 96) ? 2 : 4;   // slot_index: list distance of the paired lane groups
                if (((xp - kx + apart) * (ypb / 4)) % 32 != 16) continue;
                s.XPAD = pad;
                s.CB = (s.TZ + kz - 1) * xp * ypb + kCopyPad;
                if (lds_bytes(s, ypb, stage) <= (size_t)kMaxLds) break;
                s.XPAD = 0;
            }
            s.CB = (s.TZ + kz - 1) * (s.TX + kx - 1 + s.XPAD) * ypb + kCopyPad;
            if (lds_bytes(s, ypb, stage) > (size_t)kMaxLds) continue;
            found = true;
            if (s.ntiles >= 4 * cus) break;
        }
        // staging only pays with the big tile (several rounds per wave between barriers); otherwise try the next
        if (!found || (stage && (s.TZ != 8 || s.TX != 8))) continue;
        const int grid = cus < s.ntiles ? cus : s.ntiles;
        const size_t lds = lds_bytes(s, ypb, stage);
        s.perm = 1;
        for (int cand_p : {97, 101, 103, 107, 109, 113, 127, 131})
            if (s.ntiles % cand_p != 0) { s.perm = cand_p; break; }  // primes: coprime unless they divide ntiles
        // dynamic tile scheduling when tile costs are data dependent: a ticket counter, zeroed on the launch stream.
        // Counters come from a per-device ring allocated once (the only state this opt-in mode keeps): 1024 launches
        // may be in flight before a slot is reused.
        int* ticket = nullptr;
        if (s.skip_empty && s.ntiles > grid) {
            ticket = sn::device_flag_slot(stream);
            if (ticket && hipMemsetAsync(ticket, 0, sizeof(int), stream) != hipSuccess) {
                (void)hipGetLastError();
                ticket = nullptr;  // static order still gives the right answer
            }
        }
        s.tol = sn::option_conv_i8_tolerance();
        s.route = nullptr;
        if (s.tol > 0.0f) {
            s.route = sn::device_flag_slot(stream);
            if (!s.route) s.tol = 0.0f;   // no flag memory: run unguarded rather than fail
        }
#define SN_LAUNCH_I8(OT, YPBV, STG)                                                                              \
    do {                                                                                                         \
        auto kern = conv_occ_i8_kernel<OT, YPBV, STG>;                                                           \
        if (sn::ensure_dynamic_lds((const void*)kern, kMaxLds) != hipSuccess)                                    \
            return check_launch("sn_conv_bank(i8: hipFuncSetAttribute)");                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, stream, x, bank, lambdas, s, ticket,           \
                           (OT*)act, (OT*)out);                                                                  \
    } while (0)
        if (out_dtype == SN_F32) {
            if (stage) SN_LAUNCH_I8(float, 80, true); else SN_LAUNCH_I8(float, 96, false);
        } else {
            if (stage) SN_LAUNCH_I8(double, 80, true); else SN_LAUNCH_I8(double, 96, false);
        }
#undef SN_LAUNCH_I8
        if (int rc = check_launch("sn_conv_bank(i8)")) return rc;
        if (s.route) {   // the same launch on the fp32 matrix pipe, enqueued behind: runs only if the guard sent it there
            sn::GateScope guard(s.route, 1);
            return sn::conv_bank_group(x, SN_U8, bank, lambdas, B, Z, X, Y, G, Gtot, g0, head, kz, kx, ky, act, out,
                                       out_dtype, reinterpret_cast<sn_stream_t>(stream));
        }
        return SN_OK;
    }
    return 1;
}

}  // namespace sn

#ifdef SN_CONV_TIMING
extern "C" void sn_debug_conv_times(unsigned long long* host) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_t), sizeof(unsigned long long) * 1024 * 16);
    (void)hipMemcpyFromSymbol(host + 1024 * 16, HIP_SYMBOL(g_conv_w), sizeof(unsigned long long) * 1024 * 8);
    (void)hipMemset((void*)nullptr, 0, 0);
}
#endif
