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
#include <hip/hip_ext.h>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace {

#include "conv_prep.h"
#include "conv_fp32.inc"   // fp32k::conv_bank_body: the fp32 form, for the combined fallback launch (conv_i8_fallback_kernel)

#include "conv_i8_common.inc"   // geometry, Shape, halo ring, quantisation, finish_round: shared by the three generations
#include "conv_i8s_kernel.inc"  // stride-4 kernel (any 9 x 9-row bank, ky = 9)
#include "conv_i8f.inc"         // folded tile kernel (x/y-symmetric 9^3 banks, no blob)

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
    s.sticky = sn::sticky_device_ptr(stream);
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
    z.sticky = sn::sticky_device_ptr(stream);
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
    hipEvent_t t_start = nullptr, t_stop = nullptr;
    sn::take_timing_events(t_start, t_stop);   // (one-shot: set by sn_launch_timing_events for the next walk launch of this thread)
#define SN_LAUNCH_I8Z(OT, KH, KR, KW)                                                                            \
    do {                                                                                                         \
        auto kern = act ? conv_occ_i8z_kernel<OT, KH, KR, KW, true> : conv_occ_i8z_kernel<OT, KH, KR, KW, false>; \
        if (sn::ensure_dynamic_lds((const void*)kern, kMaxLds) != hipSuccess)                                    \
            return check_launch("sn_conv_bank_prepared(i8z: hipFuncSetAttribute)");                              \
        if (t_start || t_stop)   /* sn_launch_timing_events: this kernel's own start / stop timestamps */            \
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * KW), (uint32_t)lds, stream, t_start, t_stop, 0u, x, lambdas,  \
                                  (const uint8_t*)prep, z, (OT*)act, (OT*)out);                                   \
        else                                                                                                     \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * KW), lds, stream, x, lambdas, (const uint8_t*)prep, z,   \
                               (OT*)act, (OT*)out);                                                              \
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
