// Shared by conv_i8s.hip (the int8 kernels) and bank.hip (the GENEO bank builder): the folded operand plan of a 9 x 9 x 9
// bank that is symmetric in x and y, and the per-bank preparation of the int8 contraction -- symmetry verdict, 24-bit
// fixed-point weights, worst-case quantisation error, digit table of the z-walk plan -- written into the caller-owned
// blob of sn_conv_bank_prep / sn_geneo_bank_prep (include/scenenet_hip.h).  Included INSIDE each translation unit's
// anonymous namespace (device code only; nothing here has linkage).
//
// Reference: SceneNet.forward rebuilds its kernels from the parameters at every call (core/models/SCENE_Net.py:322-327);
// everything below is a function of those weights alone.
#pragma once

constexpr double kQMax = 8355711.0;   // 127 * 65793: the largest magnitude three balanced base-256 digits hold

// The folded kernels' operand plan: kFoldSteps MFMA steps of kFoldRows folded kernel rows (dz, dx' <= 4) per lane group;
// slot (st, j) is SINGLE (dx' = 4: one halo row) for j = 2 of steps 0..2 and DOUBLE (halo rows dx' and 8 - dx' summed)
// elsewhere.  Regular slots: lane group q carries planes dz = 2 q + a.
constexpr int kFoldSteps = 4, kFoldRows = 3, kFoldSlots = kFoldSteps * kFoldRows;
constexpr bool fold_slot_single(int st, int j) { return j == 2 && st < 3; }
constexpr int fold_slot_a(int slot) { return slot >= 5 && slot != 8 && slot != 11 ? 1 : 0; }   // slots 0..4: a = 0; 5, 6, 7, 9, 10: a = 1
constexpr int fold_slot_dx(int slot) {
    return slot == 2 || slot == 5 ? 4 : slot == 0 || slot == 6 ? 0 : slot == 1 || slot == 7 ? 1 : slot == 3 || slot == 9 ? 2 : 3;
}
constexpr bool fold_slot_irregular(int slot) { return slot == 8 || slot == 11; }

// the preparation blob (caller-owned device memory, SN_CONV_PREP_BYTES per group of 16 kernels)
constexpr int kPrepWd = 0;                     // uint4 [4][3][64]   folded digit table of the z-walk plan
constexpr int kPrepScale = 12288;              // float [16]         max|W_g| / 8355711
constexpr int kPrepBnd = 12352;                // double [16]        worst-case quantisation error per kernel
constexpr int kPrepSym = 12480;                // int [16]           1: kernel g is bit-for-bit symmetric in x and y
constexpr int kPrepFit = 12544;                // int [16]           1: every partial digit recombination fits int32
constexpr int kPrepMagic = 12608;              // int                0x5a57414c once written
constexpr int kPrepZero = 12800;               // 16 zero bytes (the z-walk's LDS-DMA source for pieces outside the grid)
constexpr int kPrepRoute = 12672;              // int                the caller-owned route flag of the launches using this blob
                                               //                    (-1 fresh from the preparation; 0 / 1 / 2 a walk's verdict)
static_assert(kPrepRoute + 4 <= kPrepZero && kPrepZero + 16 <= SN_CONV_PREP_BYTES && kPrepZero % 16 == 0, "blob layout");

// slot -> folded kernel row of lane group qq (the z-walk plan; regular slots as FoldPlan: planes dz = 2 qq + a)
__host__ __device__ constexpr int zplan_krow(int qq, int slot) {
    if (slot == 8) return qq == 0 ? 8 * 9 + 4 : -1;       // the single of plane 8; pads elsewhere
    if (slot == 11) return 8 * 9 + qq;                    // plane 8's doubles: dx' = qq (rows one apart for the lane
                                                          // groups an LDS cycle serves: 16 banks in Yc)
    return (2 * qq + fold_slot_a(slot)) * 9 + fold_slot_dx(slot);
}

// one kernel's part of quantise_kernels_folded on the 32 lanes of a half wave (l32): `w` = the kernel's 729 fp32 weights in
// LDS (overwritten by Q at the unique taps); shared with the stand-alone preparation kernel (conv_i8z.inc), so that both
// produce the same bits
__device__ __forceinline__ void quantise_folded_half(float* w, bool valid, int l32, float& scale_out, double& bnd_out,
                                                     double& pos_out, double& neg_out) {
    constexpr int nuniq = 9 * 5 * 5;
    auto tap_of = [](int u, int& mult) -> int {   // u = (dz * 5 + dx) * 5 + dy, dx, dy <= 4
        const int dy = u % 5, r = u / 5, dx = r % 5, dz = r / 5;
        mult = (dx < 4 ? 2 : 1) * (dy < 4 ? 2 : 1);
        return (dz * 9 + dx) * 9 + dy;
    };
    float m = 0.0f;
    if (valid)
        for (int u = l32; u < nuniq; u += 32) {
            int mult;
            const float a = fabsf(w[tap_of(u, mult)]);
            m = (a <= 3.0e38f) ? fmaxf(m, a) : __int_as_float(0x7fc00000);
        }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        const float u = __shfl_xor(m, o, 64);
        m = (m != m || u != u) ? __int_as_float(0x7fc00000) : fmaxf(m, u);
    }
    const double S = (m > 0.0f) ? kQMax / (double)m : 0.0;
    const double invS = (double)m / kQMax;
    double ep = 0.0, en = 0.0, qp = 0.0, qn = 0.0;
    if (valid)
        for (int u = l32; u < nuniq; u += 32) {
            int mult;
            const int t = tap_of(u, mult);
            if (m > 0.0f) {
                const double wv = (double)w[t];
                const int Q = __double2int_rn(wv * S);
                const double e = ((double)Q * invS - wv) * (double)mult;
                ep += e > 0.0 ? e : 0.0;
                en += e < 0.0 ? -e : 0.0;
                qp += Q > 0 ? (double)Q * (double)mult : 0.0;   // (exact: |Q| < 2^23, 729 taps)
                qn += Q < 0 ? -(double)Q * (double)mult : 0.0;
                w[t] = __int_as_float(Q);
            } else {
                w[t] = 0.0f;   // all-zero or poisoned kernel: Q = 0 (scale carries a NaN)
            }
        }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        ep += __shfl_xor(ep, o, 64);
        en += __shfl_xor(en, o, 64);
        qp += __shfl_xor(qp, o, 64);
        qn += __shfl_xor(qn, o, 64);
    }
    scale_out = (m != m) ? m : (float)((double)m / kQMax);
    bnd_out = ep > en ? ep : en;
    pos_out = qp;
    neg_out = qn;
}


// The preparation of ONE kernel g by a workgroup of >= 64 threads (all of them call; contains barriers): w = the kernel's
// 729 fp32 weights in LDS (zeros when !valid; overwritten), asym_s = one LDS int.  Symmetry verdict bitwise on the fp32
// weights, quantisation and error bound by quantise_folded_half (the folded kernel's own arithmetic), digit entries of
// the z-walk plan for lane (qq, g) of every step.
__device__ __forceinline__ void prep_one_kernel(float* w, int* asym_s, bool valid, int g, uint8_t* __restrict__ prep,
                                                int tid) {
    if (tid == 0) *asym_s = 0;
    __syncthreads();
    {
        bool asym = false;
        const uint32_t* wb = reinterpret_cast<const uint32_t*>(w);
        if (tid < 45) {
            const int dx = tid % 5, dz = tid / 5;
            const uint32_t* ra = wb + (dz * 9 + dx) * 9;
            const uint32_t* rb = wb + (dz * 9 + 8 - dx) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) asym |= ra[k] != rb[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) asym |= (ra[k] != ra[8 - k]) | (rb[k] != rb[8 - k]);
        }
        if (asym) *asym_s = 1;
    }
    __syncthreads();
    float sc = 0.0f;
    double bd = 0.0, qp = 0.0, qn = 0.0;
    if (tid < 64) quantise_folded_half(w, valid && tid < 32, tid & 31, sc, bd, qp, qn);   // lanes 0..31 carry the kernel
    __syncthreads();
    if (tid < 16) {
        const int st = tid >> 2, qq = tid & 3;
        uint32_t w0[4] = {0u, 0u, 0u, 0u}, w1[4] = {0u, 0u, 0u, 0u}, w2[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (j == 3 && b == 3) continue;
                const int slot = st * kFoldRows + (j < 3 ? j : b);
                const int krow = zplan_krow(qq, slot);
                const int dy = j < 3 ? b : 4;
                int Q = __float_as_int(w[(krow < 0 ? 0 : krow) * 9 + dy]);
                Q = (krow < 0 || !valid) ? 0 : Q;
                const int d0 = ((Q + 128) & 255) - 128;
                Q = (Q - d0) >> 8;
                const int d1 = ((Q + 128) & 255) - 128;
                const int d2 = (Q - d1) >> 8;
                w0[j] |= (uint32_t)(d0 & 255) << (8 * b);
                w1[j] |= (uint32_t)(d1 & 255) << (8 * b);
                w2[j] |= (uint32_t)(d2 & 255) << (8 * b);
            }
        uint4* Wd = reinterpret_cast<uint4*>(prep + kPrepWd);
        const int l = qq * 16 + g;
        Wd[(st * 3 + 0) * 64 + l] = make_uint4(w0[0], w0[1], w0[2], w0[3]);
        Wd[(st * 3 + 1) * 64 + l] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
        Wd[(st * 3 + 2) * 64 + l] = make_uint4(w2[0], w2[1], w2[2], w2[3]);
    }
    if (tid == 0) {
        reinterpret_cast<float*>(prep + kPrepScale)[g] = sc;
        reinterpret_cast<double*>(prep + kPrepBnd)[g] = bd;
        reinterpret_cast<int*>(prep + kPrepSym)[g] = *asym_s ? 0 : 1;
        // |sum of any subset of the 729 signed weights x {0,1}| stays below 2^31: the digit sums may be recombined in int32
        reinterpret_cast<int*>(prep + kPrepFit)[g] = (qp < 2147483000.0 && qn < 2147483000.0) ? 1 : 0;
        if (g == 0) {
            *reinterpret_cast<int*>(prep + kPrepMagic) = 0x5a57414c;
            *reinterpret_cast<uint4*>(prep + kPrepZero) = make_uint4(0u, 0u, 0u, 0u);
            // the verdict word starts every preparation as "no walk has decided on THIS bank yet" (-1): a caller that learns
            // verdicts by reading the word back (PreparedVerdict) can then never take the previous bank's 0 -- or the zero a
            // fresh buffer holds -- for this bank's, e.g. after a call whose shape the walk does not serve
            *reinterpret_cast<int*>(prep + kPrepRoute) = -1;
        }
    }
}
