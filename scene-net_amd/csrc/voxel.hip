// K1 -- point cloud -> voxel grid: fp64 bounding box, numpy.linspace-exact edge tables, per-point
// scatter into the occupancy grid, column min-max normalisation.
//
// Follows pyntcloud 0.1.6 VoxelGrid.compute as called from utils/pcd_processing.py:341-372,
// utils/voxelization.py:164-204 (hist_on_voxel), :244-300 (reg_on_voxel),
// utils/pcd_processing.py:305-321 (normalize_xyz), core/datasets/torch_transforms.py:33-34.
//
// All box / edge arithmetic is fp64 with explicit round-to-nearest mul/add (no FMA contraction) so
// that every edge equals numpy.linspace's bit for bit; the UTM-scale coordinates (|y| ~ 4.6e6) are
// never narrowed.
//
// Two scatter forms:
//   * counting  (sn_voxel_scatter): one global int32 atomicAdd per point.  Needed for the density /
//     ratio outputs.  Scattered device atomics execute at the memory side (~20 G atomics/s chip-wide
//     on MI355X, measured), so this form is atomic-rate bound, not HBM bound.
//   * occupancy (sn_voxel_occupancy): the network only consumes ToFullDense(density), i.e. one BIT per
//     voxel.  Each workgroup privatises the whole tile's bitmap in LDS (64^3 bits = 32 KiB), sets
//     bits with LDS atomics while streaming its share of the points with 16-byte loads, and writes
//     its partial bitmap with coalesced stores; a second kernel ORs the partials and expands to the
//     u8 / f32 grid.  No global atomics.  ToFullDense(density) is (count > column minimum), which
//     differs from (count > 0) only if some y column is occupied in EVERY (z, x) row; the finalize
//     kernel proves per tile that this cannot be the case (an empty row exists) or raises the tile's
//     flag, and flagged tiles are redone by the (gated) counting kernels.
#include "common.h"

#include <atomic>
#include <type_traits>
#include <cfloat>
#include <climits>

namespace {

// K2's device code rides in K1's first launch (bbox_partial_bank_kernel).  This file is compiled with -ffp-contract=off
// (numpy's rounding sequence, Makefile); the bank builder and the preparation are compiled with hipcc's default
// contraction in bank.hip, and must give the same bits here: the pragma restores that default for their code only.
#pragma clang fp contract(fast)
#include "conv_prep.h"
#include "bank_body.inc"
#pragma clang fp contract(off)

constexpr int kThreads = 256;
static_assert(kThreads == kBankThreads, "the rider workgroups have the bank builder's thread count");
constexpr int kMaxKeep = 16;
#ifndef SN_OCC_THREADS
#define SN_OCC_THREADS 512
#endif
constexpr int kOccThreads = SN_OCC_THREADS;
// partial bitmaps per tile: SN_OCC_PARTS sizes the workspace; fewer, larger workgroups write (and the finalize kernel
// ORs) fewer of them
constexpr int kOccParts = SN_OCC_PARTS * 512 / SN_OCC_THREADS;
constexpr int kMaxOccWords = 16 * 1024;  // 64 KiB of LDS bitmap per workgroup (occ + tower share it)

struct KeepLabels {
    double v[kMaxKeep];
    int n;
};

__device__ __forceinline__ bool is_kept(double label, const KeepLabels& keep) {
    bool k = false;
    for (int i = 0; i < keep.n; ++i) k |= (label == keep.v[i]);
    return k;
}

// ---------------------------------------------------------------- streaming a tile's points
// Tile b owns points [p0, p1).  A thread-iteration takes TWO points = 48 contiguous bytes as three
// 16-byte loads; an odd first point is peeled so the pairs start 16-byte aligned.  f(x, y, z, i).
//
// A callback that also takes the point's label -- f(x, y, z, i, label) -- gets it loaded next to the coordinates
// (lab may be null: label 0), instead of fetching labels[i] behind its own branches, one exposed latency per point.
template <typename F>
__device__ __forceinline__ void call_point(F&& f, double x, double y, double z, long i, double l) {
    if constexpr (std::is_invocable_v<F, double, double, double, long, double>) f(x, y, z, i, l);
    else f(x, y, z, i);
}
template <bool kAligned, typename F>
__device__ __forceinline__ void for_each_point(const double* __restrict__ pts, long p0, long p1, long gtid,
                                               long gstride, F&& f, const double* __restrict__ lab = nullptr) {
    constexpr bool kLab = std::is_invocable_v<F, double, double, double, long, double>;
#ifndef SN_VOX_DEPTH
#define SN_VOX_DEPTH 2
#endif
    constexpr int kDepth = SN_VOX_DEPTH;  // pairs in flight per iteration
    auto label = [&](long i) { return (kLab && lab) ? lab[i] : 0.0; };
    if (kAligned) {
        long q0 = p0 + (p0 & 1);  // first even point index >= p0
        if (q0 > p1) q0 = p1;
        if ((p0 & 1) && gtid == 0 && p0 < p1)
            call_point(f, pts[3 * p0], pts[3 * p0 + 1], pts[3 * p0 + 2], p0, label(p0));
        const long npair = (p1 - q0) >> 1;
        const double2* src = reinterpret_cast<const double2*>(pts + 3 * q0);
        long i = gtid;
        for (; i + (kDepth - 1) * gstride < npair; i += kDepth * gstride) {
            double2 a[kDepth][3];
            double l[kDepth][2];
#pragma unroll
            for (int k = 0; k < kDepth; ++k) {
                const long j = i + k * gstride;
                a[k][0] = src[3 * j]; a[k][1] = src[3 * j + 1]; a[k][2] = src[3 * j + 2];
                l[k][0] = label(q0 + 2 * j); l[k][1] = label(q0 + 2 * j + 1);
            }
#pragma unroll
            for (int k = 0; k < kDepth; ++k) {
                const long j = i + k * gstride;
                call_point(f, a[k][0].x, a[k][0].y, a[k][1].x, q0 + 2 * j, l[k][0]);
                call_point(f, a[k][1].y, a[k][2].x, a[k][2].y, q0 + 2 * j + 1, l[k][1]);
            }
        }
        for (; i < npair; i += gstride) {
            double2 a0 = src[3 * i], a1 = src[3 * i + 1], a2 = src[3 * i + 2];
            const double l0 = label(q0 + 2 * i), l1 = label(q0 + 2 * i + 1);
            call_point(f, a0.x, a0.y, a1.x, q0 + 2 * i, l0);
            call_point(f, a1.y, a2.x, a2.y, q0 + 2 * i + 1, l1);
        }
        if (((p1 - q0) & 1) && gtid == 0) {
            const long k = p1 - 1;
            call_point(f, pts[3 * k], pts[3 * k + 1], pts[3 * k + 2], k, label(k));
        }
    } else {
        for (long i = p0 + gtid; i < p1; i += gstride)
            call_point(f, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], i, label(i));
    }
}

// ---------------------------------------------------------------- order-preserving double <-> u64
__device__ __forceinline__ unsigned long long enc_f64(double d) {
    unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_f64(unsigned long long u) {
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    return __longlong_as_double((long long)u);
}

__global__ void bbox_init_kernel(unsigned long long* enc, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * 6) enc[i] = ((i % 6) < 3) ? ~0ull : 0ull;
}

template <bool kAligned>
__global__ __launch_bounds__(kThreads) void bbox_reduce_kernel(const double* __restrict__ pts,
                                                               const int64_t* __restrict__ offsets,
                                                               unsigned long long* __restrict__ enc) {
    __shared__ double red[kThreads / 64][6];
    const int b = blockIdx.y;
    const long p0 = offsets[b], p1 = offsets[b + 1];
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for_each_point<kAligned>(pts, p0, p1, (long)blockIdx.x * kThreads + threadIdx.x, (long)gridDim.x * kThreads,
                             [&](double x, double y, double z, long) {
                                 mn[0] = fmin(mn[0], x); mx[0] = fmax(mx[0], x);
                                 mn[1] = fmin(mn[1], y); mx[1] = fmax(mx[1], y);
                                 mn[2] = fmin(mn[2], z); mx[2] = fmax(mx[2], z);
                             });
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fmin(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmax(mx[c], __shfl_xor(mx[c], o, 64));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            red[wave][c] = mn[c];
            red[wave][3 + c] = mx[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6 && p1 > p0) {  // one atomic per workgroup and slot
        const int c = threadIdx.x;
        double v = red[0][c];
        for (int w = 1; w < kThreads / 64; ++w) v = (c < 3) ? fmin(v, red[w][c]) : fmax(v, red[w][c]);
        if (c < 3) {
            if (v != DBL_MAX) atomicMin(&enc[b * 6 + c], enc_f64(v));
        } else {
            if (v != -DBL_MAX) atomicMax(&enc[b * 6 + c], enc_f64(v));
        }
    }
}

__global__ void bbox_decode_kernel(unsigned long long* enc, const int64_t* __restrict__ offsets, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 6) return;
    int b = i / 6;
    double v = (offsets[b + 1] > offsets[b]) ? dec_f64(enc[i]) : 0.0;
    reinterpret_cast<double*>(enc)[i] = v;
}

// Fused form for the batch pipeline: per-workgroup partial boxes (no atomics, nothing to initialise) ...
template <bool kAligned>
__device__ __forceinline__ void bbox_partial_body(const double* __restrict__ pts, const int64_t* __restrict__ offsets,
                                                  double* __restrict__ partial, const int b) {
    __shared__ double red[kThreads / 64][6];
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for_each_point<kAligned>(pts, offsets[b], offsets[b + 1], (long)blockIdx.x * kThreads + threadIdx.x,
                             (long)gridDim.x * kThreads, [&](double x, double y, double z, long) {
                                 mn[0] = fmin(mn[0], x); mx[0] = fmax(mx[0], x);
                                 mn[1] = fmin(mn[1], y); mx[1] = fmax(mx[1], y);
                                 mn[2] = fmin(mn[2], z); mx[2] = fmax(mx[2], z);
                             });
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fmin(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmax(mx[c], __shfl_xor(mx[c], o, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            red[threadIdx.x >> 6][c] = mn[c];
            red[threadIdx.x >> 6][3 + c] = mx[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        double v = red[0][c];
        for (int w = 1; w < kThreads / 64; ++w) v = (c < 3) ? fmin(v, red[w][c]) : fmax(v, red[w][c]);
        partial[((size_t)b * gridDim.x + blockIdx.x) * 6 + c] = v;
    }
}

template <bool kAligned>
__global__ __launch_bounds__(kThreads) void bbox_partial_kernel(const double* __restrict__ pts,
                                                                const int64_t* __restrict__ offsets,
                                                                double* __restrict__ partial) {
    bbox_partial_body<kAligned>(pts, offsets, partial, blockIdx.y);
}

// The same launch with K2 as RIDERS (sn_voxel_occupancy_fused_bank): the FIRST grid rows (blockIdx.y < rider_rows: dispatched
// first -- a rider runs ~6 us, and started last it would be the launch's tail) are not tiles -- workgroup y gridDim.x + x
// builds GENEO kernel g of the bank and its share of the int8 contraction's preparation blob
// (geneo_bank_body<true>, exactly what sn_geneo_bank_prep launches), beside the HBM-bound box pass.  [measured, round 3,
// tools/debug/fork_cost.py] K2 on a side stream (one event record + one wait on the main stream) left 10.8 of its 12.4
// serial microseconds on the step's critical path: 152.2 us forked, 153.8 serial, 141.4 with K2 left out.
struct BankRider {
    const float* params;
    const int32_t* kinds;
    float* bank;
    int32_t* status;
    uint8_t* prep;
    float* lambdas;          // nullable: the effective coefficients ride too (one more workgroup; sn_geneo_bank_prep's)
    const int32_t* order;
    float* lambdas_out;
    int last;
    int G, nblocks;          // nblocks = 16 ceil(G / 16) (+ 1 with lambdas)
};

template <bool kAligned>
__global__ __launch_bounds__(kThreads) void bbox_partial_bank_kernel(const double* __restrict__ pts,
                                                                     const int64_t* __restrict__ offsets,
                                                                     double* __restrict__ partial, int rider_rows,
                                                                     BankRider r) {
    if ((int)blockIdx.y < rider_rows) {
        __shared__ float bank_lds[729 + 9 + 1 + 8 + 1];
        const int g = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
        if (g < r.nblocks)
            geneo_bank_body<true>(bank_lds, g, threadIdx.x, r.params, r.kinds, 9, 9, 9, r.bank, r.status, r.G, r.lambdas,
                                  r.order, r.last, r.lambdas_out, r.prep);
        return;
    }
    bbox_partial_body<kAligned>(pts, offsets, partial, (int)blockIdx.y - rider_rows);
}

// ---------------------------------------------------------------- grid descriptor
// desc[b] = lo[3], hi[3], edges_x[nx+1], edges_y[ny+1], edges_z[nz+1]
// numpy.linspace(lo, hi, n+1): step = (hi-lo)/n ; e_k = k*step + lo ; e_n = hi.
// box: [B,6] (nparts == 0) or the per-workgroup partial boxes [B, nparts, 6] of bbox_partial_kernel.
__global__ void desc_kernel(const double* __restrict__ box, int nparts, int nx, int ny, int nz, int regular,
                            int from_bounds, double* __restrict__ bbox_out, double* __restrict__ desc) {
    const int b = blockIdx.x;
    const int len = SN_DESC_LEN(nx, ny, nz);
    double* d = desc + (size_t)b * len;
    double lo[3], hi[3];
    if (nparts == 0) {
        const double* bb = box + (size_t)b * 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = bb[c];
            hi[c] = bb[3 + c];
        }
    } else {  // ... reduced here: one load per thread (all in flight at once), then a 64-lane shuffle tree
        __shared__ double red[6];
        const double* bb = box + (size_t)b * nparts * 6;
        if (threadIdx.x < 64) {  // lane l: slots l, l+64, ... of the [nparts][6] block; slot % 6 = component
            for (int c = 0; c < 6; ++c) {
                double v = (c < 3) ? DBL_MAX : -DBL_MAX;
                for (int pp = threadIdx.x; pp < nparts; pp += 64) {
                    const double u = bb[pp * 6 + c];
                    v = (c < 3) ? fmin(v, u) : fmax(v, u);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double u = __shfl_xor(v, o, 64);
                    v = (c < 3) ? fmin(v, u) : fmax(v, u);
                }
                if (threadIdx.x == 0) red[c] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = red[c];
            hi[c] = red[3 + c];
        }
        if (bbox_out && threadIdx.x < 3) {
            bbox_out[b * 6 + threadIdx.x] = lo[threadIdx.x];
            bbox_out[b * 6 + 3 + threadIdx.x] = hi[threadIdx.x];
        }
    }
    if (!from_bounds && regular) {
        // pyntcloud regular_bounding_box: margin = max(range) - range; min -= margin/2; max += margin/2
        double r[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = __dsub_rn(hi[c], lo[c]);
        double rmax = fmax(r[0], fmax(r[1], r[2]));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double half = __ddiv_rn(__dsub_rn(rmax, r[c]), 2.0);
            lo[c] = __dsub_rn(lo[c], half);
            hi[c] = __dadd_rn(hi[c], half);
        }
    }
    if (threadIdx.x < 3) {
        d[threadIdx.x] = lo[threadIdx.x];
        d[3 + threadIdx.x] = hi[threadIdx.x];
    }
    const int n[3] = {nx, ny, nz};
    int base = 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double step = __ddiv_rn(__dsub_rn(hi[c], lo[c]), (double)n[c]);
        for (int k = threadIdx.x; k <= n[c]; k += blockDim.x)
            d[base + k] = (k == n[c]) ? hi[c] : __dadd_rn(__dmul_rn((double)k, step), lo[c]);
        base += n[c] + 1;
    }
}

// ---------------------------------------------------------------- grid descriptor, size_x / size_y / size_z mode
// pyntcloud VoxelGrid.compute with sizes (utils/pcd_processing.py:365-367, core/datasets/semKITTI.py:453-455): the box
// is cubed, every axis is then extended by m = ((range // size) + 1) * size - range (range = the ORIGINAL extent of
// that axis) and gets n = int((max - min) / size) voxels -- n is data dependent, so a batch is voxelised into grids of
// a caller-given maximum (nx, ny, nz) and the descriptor carries each tile's own table: edges 0..n_a are
// numpy.linspace(lo, hi, n_a + 1) bit for bit, edges beyond are +inf (no point is ever above them, so the binning,
// scatter and gather kernels run unchanged on the padded table; a row / column of the grid is REAL iff its upper edge
// is finite).  dims (nullable) [B,3] i32 receives (n_x, n_y, n_z); status (nullable) [B] i32 is 1 where a tile needs
// more voxels than the maximum (its points beyond the table are dropped and counted in `dropped`).
struct Vec3s { double v[3]; };
// numpy's floor_divide for positive doubles (npy_divmod): fmod is exact, the quotient of (a - mod) by b is rounded to
// the nearest integer
__device__ __forceinline__ double np_floor_divide(double a, double b) {
    const double mod = fmod(a, b);
    const double div = __ddiv_rn(__dsub_rn(a, mod), b);
    if (div == 0.0) return 0.0;
    double fl = floor(div);
    if (__dsub_rn(div, fl) > 0.5) fl = __dadd_rn(fl, 1.0);
    return fl;
}
// box: [B,6] (nparts == 0), or the per-workgroup partial boxes [B, nparts, 6] of bbox_partial_kernel (min / max are exact:
// any order of reduction gives the same box); bbox_out (nullable): the reduced raw box.
__global__ void desc_sized_kernel(const double* __restrict__ box, int nparts, Vec3s size, int nx, int ny, int nz,
                                  double* __restrict__ desc, int32_t* __restrict__ dims,
                                  int32_t* __restrict__ status, double* __restrict__ bbox_out) {
    const int b = blockIdx.x;
    double* d = desc + (size_t)b * SN_DESC_LEN(nx, ny, nz);
    __shared__ double red6[6];
    if (nparts > 0) {
        const double* pb = box + (size_t)b * nparts * 6;
        if (threadIdx.x < 6) {
            double v = (threadIdx.x < 3) ? DBL_MAX : -DBL_MAX;
            for (int pp = 0; pp < nparts; ++pp) {
                const double u = pb[pp * 6 + threadIdx.x];
                v = (threadIdx.x < 3) ? fmin(v, u) : fmax(v, u);
            }
            red6[threadIdx.x] = v;
        }
    } else if (threadIdx.x < 6) {
        red6[threadIdx.x] = box[(size_t)b * 6 + threadIdx.x];
    }
    __syncthreads();
    if (bbox_out && threadIdx.x < 6) bbox_out[b * 6 + threadIdx.x] = red6[threadIdx.x];
    double lo[3], hi[3], r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lo[c] = red6[c];
        hi[c] = red6[3 + c];
        r[c] = __dsub_rn(hi[c], lo[c]);
    }
    const double rmax = fmax(r[0], fmax(r[1], r[2]));
    const int nmax[3] = {nx, ny, nz};
    int n[3];
    bool over = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double half = __ddiv_rn(__dsub_rn(rmax, r[c]), 2.0);   // regular_bounding_box: the cube
        lo[c] = __dsub_rn(lo[c], half);
        hi[c] = __dadd_rn(hi[c], half);
        const double m = __dsub_rn(__dmul_rn(__dadd_rn(np_floor_divide(r[c], size.v[c]), 1.0), size.v[c]), r[c]);
        const double mh = __ddiv_rn(m, 2.0);
        lo[c] = __dsub_rn(lo[c], mh);
        hi[c] = __dadd_rn(hi[c], mh);
        n[c] = (int)__ddiv_rn(__dsub_rn(hi[c], lo[c]), size.v[c]);   // int(): truncation
        over |= n[c] > nmax[c];
    }
    if (threadIdx.x < 3) {
        d[threadIdx.x] = lo[threadIdx.x];
        d[3 + threadIdx.x] = hi[threadIdx.x];
        if (dims) dims[b * 3 + threadIdx.x] = n[threadIdx.x];
    }
    if (status && threadIdx.x == 0) status[b] = over ? 1 : 0;
    int base = 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double step = __ddiv_rn(__dsub_rn(hi[c], lo[c]), (double)n[c]);
        for (int k = threadIdx.x; k <= nmax[c]; k += blockDim.x)
            d[base + k] = (k > n[c]) ? __longlong_as_double(0x7ff0000000000000ll)
                                     : (k == n[c]) ? hi[c] : __dadd_rn(__dmul_rn((double)k, step), lo[c]);
        base += nmax[c] + 1;
    }
}

// The descriptor of tile b from its partial boxes, written by the calling workgroup (nthreads threads, >= 64) into
// `lohi` [6] and `edges` [nx+ny+nz+3] (LDS or global) -- the same instruction sequence as desc_kernel, so every
// workgroup that derives it gets the same bits.  Ends with a __syncthreads().
__device__ void derive_desc(const double* __restrict__ box, int nparts, int b, int nx, int ny, int nz, int regular,
                            double* lohi, double* edges, int nthreads) {
    __shared__ double red6[6];
    const double* bb = box + (size_t)b * nparts * 6;
    if (threadIdx.x < 64) {
        double v[6] = {DBL_MAX, DBL_MAX, DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
        for (int pp = threadIdx.x; pp < nparts; pp += 64) {   // a part's six numbers in one go (independent loads)
            double u[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) u[c] = bb[pp * 6 + c];
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = (c < 3) ? fmin(v[c], u[c]) : fmax(v[c], u[c]);
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double u = __shfl_xor(v[c], o, 64);
                v[c] = (c < 3) ? fmin(v[c], u) : fmax(v[c], u);
            }
            if (threadIdx.x == 0) red6[c] = v[c];
        }
    }
    __syncthreads();
    double lo[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lo[c] = red6[c];
        hi[c] = red6[3 + c];
    }
    if (regular) {
        double r[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = __dsub_rn(hi[c], lo[c]);
        double rmax = fmax(r[0], fmax(r[1], r[2]));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double half = __ddiv_rn(__dsub_rn(rmax, r[c]), 2.0);
            lo[c] = __dsub_rn(lo[c], half);
            hi[c] = __dadd_rn(hi[c], half);
        }
    }
    if (threadIdx.x < 3) {
        lohi[threadIdx.x] = lo[threadIdx.x];
        lohi[3 + threadIdx.x] = hi[threadIdx.x];
    }
    const int n[3] = {nx, ny, nz};
    int base = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double step = __ddiv_rn(__dsub_rn(hi[c], lo[c]), (double)n[c]);
        for (int k = threadIdx.x; k <= n[c]; k += nthreads)
            edges[base + k] = (k == n[c]) ? hi[c] : __dadd_rn(__dmul_rn((double)k, step), lo[c]);
        base += n[c] + 1;
    }
    __syncthreads();
}

// ---------------------------------------------------------------- binning
// largest j with e[j] < p  (== numpy.searchsorted(e, p, side='left') - 1), in [-1, n].  bin_guess is the arithmetic
// estimate (right except for points sitting on an edge and last-bit rounding), bin_search walks from any start to the
// exact answer with dependent LDS reads; Binner::flat confirms the three guesses with one round of independent reads
// and only walks when a guess fails.
// The guess is clamped to [0, n-1] so that e[k] and e[k+1] both exist: the confirmation then needs no special cases.
__device__ __forceinline__ int bin_guess(double p, int n, double lo, double inv_step) {
    const int g = (int)((p - lo) * inv_step);   // v_cvt_i32_f64 saturates; NaN -> 0
    return max(0, min(g, n - 1));
}
__device__ __forceinline__ int bin_search(double p, const double* e, int n, int k) {
    while (k < n && e[k + 1] < p) ++k;
    while (k >= 0 && !(e[k] < p)) --k;
    return k;
}

struct Binner {
    const double *ex, *ey, *ez;
    double lo[3], inv[3];
    int nx, ny, nz;

    // edges must already sit in LDS at `edges`; d = this tile's descriptor
    __device__ __forceinline__ void init(const double* edges, const double* d, int nx_, int ny_, int nz_) {
        ex = edges; ey = edges + nx_ + 1; ez = edges + nx_ + ny_ + 2;
        nx = nx_; ny = ny_; nz = nz_;
        // the guess only needs the bin width: the first step of each edge table (== (hi - lo) / n up to rounding, and
        // right for the padded per-tile tables of the size mode, whose real n is not the table's)
        const double* e[3] = {ex, ey, ez};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = d[a];
            const double step = e[a][1] - e[a][0];
            inv[a] = (step > 0.0 && step < DBL_MAX) ? 1.0 / step : 0.0;
        }
    }
    // flat [z][x][y] index, or -1 when the point is NaN / outside the edge table (np.clip to n).
    // Fast path, branch-free: three clamped guesses, six independent LDS reads, six compares -- a guess k stands iff
    // e[k] < p and not e[k+1] < p, exactly where bin_search(k) would stop.  Everything else (a point sitting on an
    // edge, the tile's minimum, NaN, a point beyond the table) takes the exact walk; a wave skips that branch as a
    // whole when none of its points needs it ([measured] the first form, with its per-axis special cases, spent
    // ~100 VALU instructions per point).
    __device__ __forceinline__ int flat(double x, double y, double z) const {
        int ix = bin_guess(x, nx, lo[0], inv[0]);
        int iy = bin_guess(y, ny, lo[1], inv[1]);
        int iz = bin_guess(z, nz, lo[2], inv[2]);
        const double x0 = ex[ix], x1 = ex[ix + 1], y0 = ey[iy], y1 = ey[iy + 1], z0 = ez[iz], z1 = ez[iz + 1];
        const bool ok = (x0 < x) & !(x1 < x) & (y0 < y) & !(y1 < y) & (z0 < z) & !(z1 < z);
        if (!ok) {
            if (x != x || y != y || z != z) return -1;
            ix = max(bin_search(x, ex, nx, ix), 0);
            iy = max(bin_search(y, ey, ny, iy), 0);
            iz = max(bin_search(z, ez, nz, iz), 0);
            if (ix >= nx || iy >= ny || iz >= nz) return -1;
        }
        return (iz * nx + ix) * ny + iy;
    }
};

__device__ __forceinline__ void load_edges(double* edges, const double* d, int ne, int nthreads) {
    for (int i = threadIdx.x; i < ne; i += nthreads) edges[i] = d[6 + i];
    __syncthreads();
}

// ---------------------------------------------------------------- counting scatter (global atomics)
__global__ void gated_zero_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b2, size_t V,
                                  const int32_t* __restrict__ gate) {
    const int b = blockIdx.y;
    if (gate && !gate[b]) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (size_t)gridDim.x * blockDim.x) {
        a[(size_t)b * V + i] = 0;
        if (b2) b2[(size_t)b * V + i] = 0;
    }
}

template <bool kAligned>
__global__ __launch_bounds__(kThreads) void scatter_kernel(const double* __restrict__ pts,
                                                           const double* __restrict__ labels,
                                                           const int64_t* __restrict__ offsets,
                                                           const double* __restrict__ desc, int nx, int ny, int nz,
                                                           int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ towers, KeepLabels keep,
                                                           int32_t* __restrict__ dropped_out,
                                                           const int32_t* __restrict__ gate) {
    extern __shared__ double edges[];  // [nx+1 + ny+1 + nz+1]
    const int b = blockIdx.y;
    if (gate && !gate[b]) return;
    const double* d = desc + (size_t)b * SN_DESC_LEN(nx, ny, nz);
    load_edges(edges, d, nx + ny + nz + 3, kThreads);
    Binner bin;
    bin.init(edges, d, nx, ny, nz);
    const size_t V = (size_t)nx * ny * nz;
    int32_t* c = counts + (size_t)b * V;
    int32_t* t = towers ? towers + (size_t)b * V : nullptr;
    const bool want_tower = (towers != nullptr) && (labels != nullptr);
    int dropped = 0;
    for_each_point<kAligned>(pts, offsets[b], offsets[b + 1], (long)blockIdx.x * kThreads + threadIdx.x,
                             (long)gridDim.x * kThreads, [&](double x, double y, double z, long, double label) {
                                 const int f = bin.flat(x, y, z);
                                 if (f < 0) { ++dropped; return; }
                                 atomicAdd(&c[f], 1);
                                 if (want_tower && is_kept(label, keep)) atomicAdd(&t[f], 1);
                             }, want_tower ? labels : nullptr);
    if (dropped_out && dropped) atomicAdd(&dropped_out[b], dropped);
}

// ---------------------------------------------------------------- occupancy bitmap (LDS atomics)
// grid (parts * slabs, B).  Workgroup (part, slab) streams part's share of the tile's points and owns the bits of
// z-slab `slab` (bitmaps larger than the LDS budget -- 128^3 -- are split into slabs; every slab's workgroup then
// re-reads the points, which come from L2 / Infinity Cache).  LDS: edge table, then `swords` words of occupancy
// bits, then (optionally) `swords` of tower bits.  Part p writes bits_ws[(b*parts + p) * planes * words ...].
template <bool kAligned>
__global__ __launch_bounds__(kOccThreads) void occ_partial_kernel(const double* __restrict__ pts,
                                                                  const double* __restrict__ labels,
                                                                  const int64_t* __restrict__ offsets,
                                                                  const double* __restrict__ desc, int nx, int ny,
                                                                  int nz, int words, int planes, int parts,
                                                                  int slabs, KeepLabels keep,
                                                                  uint32_t* __restrict__ bits_ws,
                                                                  int32_t* __restrict__ dropped_parts,
                                                                  int32_t* __restrict__ flags,
                                                                  const double* __restrict__ box_parts, int nbparts,
                                                                  int regular, double* __restrict__ desc_out,
                                                                  double* __restrict__ bbox_out) {
    __shared__ int dropped_blk;
    __shared__ double lohi[6];
    if (threadIdx.x == 0) dropped_blk = 0;
    extern __shared__ double smem[];
    const int ne = nx + ny + nz + 3;
    double* edges = smem;
    uint32_t* bits = reinterpret_cast<uint32_t*>(smem + ((ne + 1) & ~1));  // 16-byte aligned
    const int b = blockIdx.y;
    const int part = blockIdx.x / slabs, slab = blockIdx.x - part * slabs;
    const int swords = words / slabs;
    const int f_lo = slab * swords * 32, f_hi = f_lo + swords * 32;  // flat voxel range of this slab
    for (int i = threadIdx.x; i < swords * planes; i += kOccThreads) bits[i] = 0u;
    const double* d;
    if (box_parts) {
        // fused prepare: every workgroup derives the tile's descriptor from the partial boxes (3 (n+1) edges: cheaper
        // than one more dependent launch); workgroup (part 0, slab 0) publishes it for the later kernels
        derive_desc(box_parts, nbparts, b, nx, ny, nz, regular, lohi, edges, kOccThreads);
        d = lohi;
        if (blockIdx.x == 0) {
            double* dd = desc_out + (size_t)b * SN_DESC_LEN(nx, ny, nz);
            for (int i = threadIdx.x; i < 6 + ne; i += kOccThreads) dd[i] = (i < 6) ? lohi[i] : edges[i - 6];
            if (bbox_out) {   // the raw box (before the cube regularisation), as sn_voxel_prepare reports it
                const double* bb = box_parts + (size_t)b * nbparts * 6;
                if (threadIdx.x < 6) {
                    double v = (threadIdx.x < 3) ? DBL_MAX : -DBL_MAX;
                    for (int pp = 0; pp < nbparts; ++pp) {
                        const double u = bb[pp * 6 + threadIdx.x];
                        v = (threadIdx.x < 3) ? fmin(v, u) : fmax(v, u);
                    }
                    bbox_out[b * 6 + threadIdx.x] = v;
                }
            }
        }
    } else {
        d = desc + (size_t)b * SN_DESC_LEN(nx, ny, nz);
        load_edges(edges, d, ne, kOccThreads);  // ends with __syncthreads()
    }
    Binner bin;
    bin.init(edges, d, nx, ny, nz);
    const bool want_tower = (planes == 2);
    int dropped = 0;
    for_each_point<kAligned>(pts, offsets[b], offsets[b + 1], (long)part * kOccThreads + threadIdx.x,
                             (long)parts * kOccThreads, [&](double x, double y, double z, long, double label) {
                                 const int f = bin.flat(x, y, z);
                                 if (f < 0) { dropped += (slab == 0); return; }
                                 if (f < f_lo || f >= f_hi) return;
                                 const int g = f - f_lo;
                                 atomicOr(&bits[g >> 5], 1u << (g & 31));
                                 if (want_tower && is_kept(label, keep))
                                     atomicOr(&bits[swords + (g >> 5)], 1u << (g & 31));
                             }, want_tower ? labels : nullptr);
    __syncthreads();
    uint32_t* out = bits_ws + ((size_t)b * parts + part) * (size_t)planes * words + (size_t)slab * swords;
    for (int i = threadIdx.x; i < swords; i += kOccThreads) {
        out[i] = bits[i];
        if (want_tower) out[words + i] = bits[swords + i];
    }
    if (dropped) atomicAdd(&dropped_blk, dropped);  // LDS
    __syncthreads();
    if (threadIdx.x == 0 && slab == 0) {
        dropped_parts[b * parts + part] = dropped_blk;
        if (flags && part == 0) flags[b] = 1;  // cleared by occ_finalize_kernel
    }
}

// ---------------------------------------------------------------- one pass over the points (round 4)
// bbox_partial_kernel + occ_partial_kernel read every point twice -- the box pass from HBM, the binning pass from the
// Infinity Cache -- and the binning pass is latency bound (six dependent trips of two points per thread).  Here a thread
// keeps its share of the tile in REGISTERS: kOnePairs pairs of points (48 B each, three 16-byte loads, all requested before
// the first one is used), takes their min / max, the tile's 16 workgroups exchange partial boxes through memory, every
// workgroup derives the descriptor (the same code, the same bits, as before) and bins its points out of the registers into
// its LDS bitmap.  One launch instead of two, 77 MB of point traffic instead of 154.
//
// The exchange: workgroup `part` of tile b publishes {box[6], tag} in partial_ws[b][part][0..7] -- the seven words with
// agent-scope atomic stores (the 16 workgroups of a tile sit on all eight XCDs, each behind its own L2), the tag last,
// behind a release -- and waits until all kOccParts tags of its tile carry this launch's tag, then reads the boxes with
// agent-scope loads.  The tag is (magic | epoch), the epoch a process-wide launch counter: whatever the scratch block held
// before cannot pass for it; occ_finalize_kernel, which runs behind this launch, zeroes the tags, so a captured graph (same
// epoch at every replay) starts clean each time.
// Forward progress: a workgroup publishes before it waits, and waits only for workgroups of ITS tile.  Each XCD is handed
// its share of the grid in order (tile-major: blockIdx.y = tile), so the lowest unfinished tile always has every workgroup
// dispatched or about to be; at C2 (8 x 32 workgroups of 1024 threads, one per CU) the whole grid is resident at once.
// And the wait is bounded: after `spin_max` polls a workgroup computes the tile's box from the points alone (same bits) --
// so the launch cannot hang or go wrong even where that argument does not hold (grids of two processes interleaved).
// Tiles with more than kOnePairs x 2 x 512 x 16 = 114 688 points: the surplus pairs are streamed from memory in both
// phases, as the two-kernel form does.
constexpr int kOnePairs = 7;
#ifndef SN_ONE_THREADS
#define SN_ONE_THREADS 1024
#endif
constexpr int kOneThreads = SN_ONE_THREADS;   // [measured, C2] 1024 threads x 8 parts per tile: stage 34.1 us; 512 x 16: 35.7-36.5 (half the partial bitmaps to write and to OR)
constexpr int kOneParts = kOccParts * kOccThreads / kOneThreads;
#ifdef SN_CONV_TIMING   // make -B EXTRA=-DSN_CONV_TIMING; read by tools/vox_timing.py
__device__ unsigned long long g_vox_t[1024 * 8];   // per workgroup: 0 start, 1 points in + min/max, 2 published, 3 all tags seen, 4 descriptor, 5 binned, 6 end
#define SN_VT(k) do { if (threadIdx.x == 0) g_vox_t[(blockIdx.y * gridDim.x + blockIdx.x) % 1024 * 8 + (k)] = wall_clock64(); } while (0)
#else
#define SN_VT(k) do {} while (0)
#endif
constexpr unsigned long long kOneMagic = 0x5ce7e000ull << 32;

template <bool kAligned>
__global__ __launch_bounds__(kOneThreads) void occ_onepass_kernel(const double* __restrict__ pts,
                                                                  const double* __restrict__ labels,
                                                                  const int64_t* __restrict__ offsets, int nx, int ny,
                                                                  int nz, int words, int planes, KeepLabels keep,
                                                                  uint32_t* __restrict__ bits_ws,
                                                                  int32_t* __restrict__ dropped_parts,
                                                                  int32_t* __restrict__ flags,
                                                                  double* __restrict__ box_parts, unsigned epoch,
                                                                  int regular, double* __restrict__ desc_out,
                                                                  double* __restrict__ bbox_out, int spin_max,
                                                                  int rider_rows, BankRider r) {
    if ((int)blockIdx.y < rider_rows) {
        // K2 as riders (the first grid rows: dispatched first), 256 of the 512 threads build kernel g + its preparation
        __shared__ float bank_lds[729 + 9 + 1 + 8 + 1];
        const int g = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
        if (g < r.nblocks && threadIdx.x < kBankThreads)
            geneo_bank_body<true>(bank_lds, g, threadIdx.x, r.params, r.kinds, 9, 9, 9, r.bank, r.status, r.G, r.lambdas,
                                  r.order, r.last, r.lambdas_out, r.prep);
        return;
    }
    __shared__ int dropped_blk;
    __shared__ double lohi[6];
    __shared__ double red[kOneThreads / 64][6];
    __shared__ double box16[kOneParts][6];
    extern __shared__ double smem[];
    const int ne = nx + ny + nz + 3;
    double* edges = smem;
    uint32_t* bits = reinterpret_cast<uint32_t*>(smem + ((ne + 1) & ~1));
    const int b = (int)blockIdx.y - rider_rows, part = blockIdx.x, tid = threadIdx.x;
    const bool want_tower = (planes == 2);
    if (tid == 0) dropped_blk = 0;
    SN_VT(0);

    // ---- phase 1: this thread's points into registers, their min / max
    const long p0 = offsets[b], p1 = offsets[b + 1];
    long q0 = kAligned ? p0 + (p0 & 1) : p0;
    if (q0 > p1) q0 = p1;
    const long npair = (p1 - q0) >> 1;
    const long gtid = (long)part * kOneThreads + tid, gstride = (long)kOneParts * kOneThreads;
    const double2* src = reinterpret_cast<const double2*>(pts + 3 * q0);
    double2 a[kOnePairs][3];
    unsigned kept = 0u;      // bit 2k / 2k + 1: the pair's first / second point carries a kept label (GT plane)
#pragma unroll
    for (int k = 0; k < kOnePairs; ++k) {
        const long j = gtid + k * gstride;
        if (j < npair) {
            if constexpr (kAligned) {
                a[k][0] = src[3 * j]; a[k][1] = src[3 * j + 1]; a[k][2] = src[3 * j + 2];
            } else {
                const double* q = pts + 3 * (q0 + 2 * j);
                a[k][0] = make_double2(q[0], q[1]); a[k][1] = make_double2(q[2], q[3]); a[k][2] = make_double2(q[4], q[5]);
            }
        }
    }
    if (want_tower) {
#pragma unroll
        for (int k = 0; k < kOnePairs; ++k) {
            const long j = gtid + k * gstride;
            if (j < npair) {
                const double l0 = labels[q0 + 2 * j], l1 = labels[q0 + 2 * j + 1];
                kept |= (is_kept(l0, keep) ? 1u : 0u) << (2 * k) | (is_kept(l1, keep) ? 2u : 0u) << (2 * k);
            }
        }
    }
    // (the bitmap is cleared while the loads are in flight)
    for (int i = tid; i < words * planes; i += kOneThreads) bits[i] = 0u;
    // the points that are nobody's pair: an odd first one (aligned form), an odd last one -- thread 0 of part 0
    const bool lone_first = kAligned && (p0 & 1) && p0 < p1, lone_last = ((p1 - q0) & 1) != 0;
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    auto take = [&](double x, double y, double z) {
        mn[0] = fmin(mn[0], x); mx[0] = fmax(mx[0], x);
        mn[1] = fmin(mn[1], y); mx[1] = fmax(mx[1], y);
        mn[2] = fmin(mn[2], z); mx[2] = fmax(mx[2], z);
    };
#pragma unroll
    for (int k = 0; k < kOnePairs; ++k) {
        if (gtid + k * gstride < npair) {
            take(a[k][0].x, a[k][0].y, a[k][1].x);
            take(a[k][1].y, a[k][2].x, a[k][2].y);
        }
    }
    for (long j = gtid + (long)kOnePairs * gstride; j < npair; j += gstride) {   // (tiles beyond the register budget)
        const double* q = pts + 3 * (q0 + 2 * j);
        take(q[0], q[1], q[2]);
        take(q[3], q[4], q[5]);
    }
    if (gtid == 0) {
        if (lone_first) take(pts[3 * p0], pts[3 * p0 + 1], pts[3 * p0 + 2]);
        if (lone_last) take(pts[3 * (p1 - 1)], pts[3 * (p1 - 1) + 1], pts[3 * (p1 - 1) + 2]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fmin(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmax(mx[c], __shfl_xor(mx[c], o, 64));
        }
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            red[tid >> 6][c] = mn[c];
            red[tid >> 6][3 + c] = mx[c];
        }
    }
    __syncthreads();
    SN_VT(1);
    // ---- the exchange
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(box_parts) + (size_t)b * kOneParts * 8;
    const unsigned long long tag = kOneMagic | epoch;
    if (tid < 6) {
        double v = red[0][tid];
        for (int w = 1; w < kOneThreads / 64; ++w) v = (tid < 3) ? fmin(v, red[w][tid]) : fmax(v, red[w][tid]);
        __hip_atomic_store(&slots[part * 8 + tid], (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    // The tag goes out behind the six values WITHOUT a release fence: all seven words are agent-scope atomic stores (write
    // through: nothing else this workgroup has written needs to be visible to anybody), the six values leave in ONE wave
    // instruction of wave 0, and s_waitcnt vmcnt(0) in that wave waits for their acknowledgement before lane 0 issues the tag.
    // [measured] with `release` on the tag and an agent-scope `acquire` fence behind the wait -- an L2 write-back and an L2
    // invalidate per workgroup, 512 of each -- the kernel took 58.9 us against 36 us for the two kernels it replaces.
    if (tid < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) __hip_atomic_store(&slots[part * 8 + 6], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    SN_VT(2);
    // The wait is BOUNDED and nobody depends on its outcome: a workgroup whose siblings have not all published after
    // `spin_max` polls (~1 us each) stops waiting and takes the tile's box from the points itself (one streaming pass over
    // the whole tile: min / max are exact and order-free, so it gets the very bits the exchange would have delivered).
    // In-order dispatch makes that a path for pathological cases only -- two processes' grids interleaved on one GPU can
    // leave each other's workgroups without the slots their siblings need -- but it means the launch can neither hang nor
    // compute a wrong box, whatever shares the device (spin_max 0: never wait; tested).
    bool gave_up = false;
    if (tid < kOneParts) {
        int spins = 0;
        while (__hip_atomic_load(&slots[tid * 8 + 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
            if (++spins > spin_max) { gave_up = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    const bool alone = __syncthreads_or(gave_up ? 1 : 0) != 0;
    SN_VT(3);
    if (!alone) {
        // (the boxes are read with agent-scope atomic loads, issued behind the tags' loads: no stale line of an earlier
        // launch's exchange can be taken for them, and no cache needs invalidating)
        if (tid < kOneParts * 6)
            box16[tid / 6][tid % 6] = __longlong_as_double((long long)__hip_atomic_load(
                &slots[(tid / 6) * 8 + tid % 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    } else {
        double m2[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, x2[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
        for (long i = p0 + tid; i < p1; i += kOneThreads) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double v = pts[3 * i + c];
                m2[c] = fmin(m2[c], v);
                x2[c] = fmax(x2[c], v);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                m2[c] = fmin(m2[c], __shfl_xor(m2[c], o, 64));
                x2[c] = fmax(x2[c], __shfl_xor(x2[c], o, 64));
            }
        }
        __syncthreads();   // (red[] was read by the publishing threads above)
        if ((tid & 63) == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                red[tid >> 6][c] = m2[c];
                red[tid >> 6][3 + c] = x2[c];
            }
        }
        __syncthreads();
        if (tid < kOneParts * 6) {
            const int c = tid % 6;
            double v = (c < 3) ? DBL_MAX : -DBL_MAX;          // parts 1.. are neutral; part 0 carries the tile's box
            if (tid < 6)
                for (int w = 0; w < kOneThreads / 64; ++w) v = (c < 3) ? fmin(v, red[w][c]) : fmax(v, red[w][c]);
            box16[tid / 6][c] = v;
        }
    }
    __syncthreads();
    // the descriptor from the 16 partial boxes: derive_desc's own code on the LDS copy (same bits in every workgroup)
    derive_desc(&box16[0][0], kOneParts, 0, nx, ny, nz, regular, lohi, edges, kOneThreads);
    if (part == 0) {
        double* dd = desc_out + (size_t)b * SN_DESC_LEN(nx, ny, nz);
        for (int i = tid; i < 6 + ne; i += kOneThreads) dd[i] = (i < 6) ? lohi[i] : edges[i - 6];
        if (bbox_out && tid < 6) {
            double v = (tid < 3) ? DBL_MAX : -DBL_MAX;
            for (int pp = 0; pp < kOneParts; ++pp) v = (tid < 3) ? fmin(v, box16[pp][tid]) : fmax(v, box16[pp][tid]);
            bbox_out[b * 6 + tid] = v;
        }
    }
    SN_VT(4);
    // ---- phase 2: bin out of the registers
    Binner bin;
    bin.init(edges, lohi, nx, ny, nz);
    int dropped = 0;
    auto put = [&](double x, double y, double z, bool tower) {
        const int f = bin.flat(x, y, z);
        if (f < 0) { ++dropped; return; }
        atomicOr(&bits[f >> 5], 1u << (f & 31));
        if (tower) atomicOr(&bits[words + (f >> 5)], 1u << (f & 31));
    };
#pragma unroll
    for (int k = 0; k < kOnePairs; ++k) {
        if (gtid + k * gstride < npair) {
            put(a[k][0].x, a[k][0].y, a[k][1].x, want_tower && ((kept >> (2 * k)) & 1u));
            put(a[k][1].y, a[k][2].x, a[k][2].y, want_tower && ((kept >> (2 * k + 1)) & 1u));
        }
    }
    for (long j = gtid + (long)kOnePairs * gstride; j < npair; j += gstride) {
        const double* q = pts + 3 * (q0 + 2 * j);
        put(q[0], q[1], q[2], want_tower && is_kept(labels[q0 + 2 * j], keep));
        put(q[3], q[4], q[5], want_tower && is_kept(labels[q0 + 2 * j + 1], keep));
    }
    if (gtid == 0) {
        if (lone_first) put(pts[3 * p0], pts[3 * p0 + 1], pts[3 * p0 + 2], want_tower && is_kept(labels[p0], keep));
        if (lone_last) {
            const long k = p1 - 1;
            put(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2], want_tower && is_kept(labels[k], keep));
        }
    }
    __syncthreads();
    SN_VT(5);
    uint32_t* out = bits_ws + ((size_t)b * kOneParts + part) * (size_t)planes * words;
    for (int i = tid; i < words; i += kOneThreads) {
        out[i] = bits[i];
        if (want_tower) out[words + i] = bits[words + i];
    }
    if (dropped) atomicAdd(&dropped_blk, dropped);  // LDS
    __syncthreads();
    if (tid == 0) {
        dropped_parts[b * kOneParts + part] = dropped_blk;
        if (flags && part == 0) flags[b] = 1;  // cleared by occ_finalize_kernel
    }
    SN_VT(6);
}

// OR of the `parts` partial bitmaps of one tile (plane 0 = occupancy, 1 = towers), word w
__device__ __forceinline__ uint32_t merged_word(const uint32_t* __restrict__ src, int parts, int planes, int words,
                                                int plane, long w) {
    uint32_t m = 0u;
    for (int p = 0; p < parts; ++p) m |= src[((size_t)p * planes + plane) * words + w];
    return m;
}

template <typename OT>
__device__ __forceinline__ void expand_word(uint32_t m, OT* __restrict__ dst);

// 32 bits -> 32 bytes {0,1}: (nibble * 0x00204081) & 0x01010101 spreads 4 bits over 4 bytes
template <>
__device__ __forceinline__ void expand_word<uint8_t>(uint32_t m, uint8_t* __restrict__ dst) {
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (((m >> (4 * k)) & 0xFu) * 0x00204081u) & 0x01010101u;
    uint4* d = reinterpret_cast<uint4*>(dst);
    d[0] = make_uint4(o[0], o[1], o[2], o[3]);
    d[1] = make_uint4(o[4], o[5], o[6], o[7]);
}
template <>
__device__ __forceinline__ void expand_word<float>(uint32_t m, float* __restrict__ dst) {
    float4* d = reinterpret_cast<float4*>(dst);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        d[k] = make_float4((float)((m >> (4 * k)) & 1u), (float)((m >> (4 * k + 1)) & 1u),
                           (float)((m >> (4 * k + 2)) & 1u), (float)((m >> (4 * k + 3)) & 1u));
}

// grid (C, B): OR the partial bitmaps, expand bits to the u8 / f32 grids (one 32-voxel word per thread
// iteration, 16-byte stores), and prove "no y column is full" (an empty (z,x) row exists) -- the tile's
// flag was raised by occ_partial_kernel and is cleared here by whoever finds an empty row.
template <typename OT>
__global__ __launch_bounds__(kThreads) void occ_finalize_kernel(const uint32_t* __restrict__ bits_ws, int words,
                                                                int planes, int parts, int rows, int ny, size_t V,
                                                                OT* __restrict__ occ, OT* __restrict__ gt_occ,
                                                                int32_t* __restrict__ flags,
                                                                const int32_t* __restrict__ dropped_parts,
                                                                int32_t* __restrict__ dropped,
                                                                const int32_t* __restrict__ dims, int nx,
                                                                unsigned long long* __restrict__ exchange_slots) {
    const int b = blockIdx.y;
    // (the one-pass kernel's exchange tags of this tile: zeroed here, behind that launch, so that a replayed graph --
    // whose launches carry the same tag every time -- never finds the previous replay's)
    if (exchange_slots && blockIdx.x == 0 && threadIdx.x < parts)
        exchange_slots[((size_t)b * parts + threadIdx.x) * 8 + 6] = 0ull;
    if (dropped && blockIdx.x == 0 && threadIdx.x == 0) {
        int t = 0;
        for (int p = 0; p < parts; ++p) t += dropped_parts[b * parts + p];
        dropped[b] = t;
    }
    const uint32_t* src = bits_ws + (size_t)b * parts * planes * words;
    const int gtid = blockIdx.x * kThreads + threadIdx.x, gstride = gridDim.x * kThreads;
    // When a (z,x) row is a power-of-two number of whole words that sit in consecutive lanes (ny = 32, 64, ..., and
    // words % 64 == 0), the emptiness proof rides on the words this loop has merged anyway; otherwise a second pass
    // re-merges the words of each row.
    const int wpr = ny >> 5;   // words per row
    // (voxel-size mode, dims != null: the grid is padded to the maximum and only the tile's own n_z x n_x rows, n_y bits
    // each, are the reference's grid -- the padding must not count as an empty row: second pass)
    const bool fused_proof = flags && !dims && (ny & 31) == 0 && wpr >= 1 && wpr <= 64 && (wpr & (wpr - 1)) == 0 &&
                             (words & 63) == 0;
    const int own_nx = dims ? dims[b * 3 + 0] : nx, own_ny = dims ? dims[b * 3 + 1] : ny,
              own_nz = dims ? dims[b * 3 + 2] : rows / (nx > 0 ? nx : 1);
    bool empty = false;
    // Four words (128 voxels) per thread when the layout allows: 16-byte loads of the partial bitmaps, eight 16-byte stores
    // per plane -- a quarter of the memory instructions of the word-per-thread loop below ([measured] round 4: 7.0 -> see
    // DESIGN K1).  The emptiness proof of a row rides on the merged words: inside the thread for rows of <= 4 words, by
    // shuffles over the row's lanes for longer rows (then every lane of a row group runs the same trips: words % 256 == 0).
    const bool vec4 = (words & 3) == 0 && ((uintptr_t)bits_ws & 15) == 0 && (!fused_proof || wpr <= 4 || (words & 255) == 0);
    if (vec4) {
        const int words4 = words >> 2;
        for (int q = gtid; q < words4; q += gstride) {
            uint4 m = make_uint4(0u, 0u, 0u, 0u), g4 = make_uint4(0u, 0u, 0u, 0u);
            for (int p = 0; p < parts; ++p) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + ((size_t)p * planes + 0) * words + 4 * (size_t)q);
                m.x |= v.x; m.y |= v.y; m.z |= v.z; m.w |= v.w;
            }
            OT* o = occ + (size_t)b * V + (size_t)q * 128;
            expand_word<OT>(m.x, o); expand_word<OT>(m.y, o + 32); expand_word<OT>(m.z, o + 64); expand_word<OT>(m.w, o + 96);
            if (gt_occ) {
                for (int p = 0; p < parts; ++p) {
                    const uint4 v = *reinterpret_cast<const uint4*>(src + ((size_t)p * planes + 1) * words + 4 * (size_t)q);
                    g4.x |= v.x; g4.y |= v.y; g4.z |= v.z; g4.w |= v.w;
                }
                OT* g = gt_occ + (size_t)b * V + (size_t)q * 128;
                expand_word<OT>(g4.x, g); expand_word<OT>(g4.y, g + 32); expand_word<OT>(g4.z, g + 64); expand_word<OT>(g4.w, g + 96);
            }
            if (fused_proof) {
                if (wpr == 1) empty |= (m.x == 0u) | (m.y == 0u) | (m.z == 0u) | (m.w == 0u);
                else if (wpr == 2) empty |= ((m.x | m.y) == 0u) | ((m.z | m.w) == 0u);
                else {
                    uint32_t any = m.x | m.y | m.z | m.w;
                    for (int o2 = wpr >> 3; o2 > 0; o2 >>= 1) any |= __shfl_xor(any, o2, 64);
                    empty |= (any == 0u);
                }
            }
        }
    } else
    for (int w = gtid; w < words; w += gstride) {
        const uint32_t m0 = merged_word(src, parts, planes, words, 0, w);
        expand_word<OT>(m0, occ + (size_t)b * V + (size_t)w * 32);
        if (gt_occ) expand_word<OT>(merged_word(src, parts, planes, words, 1, w), gt_occ + (size_t)b * V + (size_t)w * 32);
        if (fused_proof) {
            uint32_t any = m0;
            for (int o = wpr >> 1; o > 0; o >>= 1) any |= __shfl_xor(any, o, 64);
            empty |= (any == 0u);
        }
    }
    if (!flags) return;
    if (!fused_proof) {
        // row r owns bits [r*ny, (r+1)*ny)
        for (int r = gtid; r < rows && !empty; r += gstride) {
            if (dims && (r / nx >= own_nz || r % nx >= own_nx)) continue;   // a padding row
            const long lo = (long)r * ny, hi = lo + own_ny;
            uint32_t any = 0u;
            for (long w = lo >> 5; w <= (hi - 1) >> 5; ++w) {
                uint32_t m = merged_word(src, parts, planes, words, 0, w);
                const long wlo = w << 5;
                if (lo > wlo) m &= ~0u << (lo - wlo);
                if (hi < wlo + 32) m &= ~0u >> (wlo + 32 - hi);
                any |= m;
            }
            empty = (any == 0u);
        }
    }
    if (empty) flags[b] = 0;  // benign race: every writer stores 0
}

// Rare path of sn_voxel_occupancy: a flagged tile (no empty (z,x) row, so a y column may be full) is redone
// exactly -- counts by global atomics, column minima, ToFullDense -- by ONE workgroup, in one launch that
// exits at once for every other tile (256 threads: the launch is unconditional, so what counts is how fast it is
// dispatched and retired when no flag is set -- [measured] 4.5 us with 1024 threads).  Counts are re-read with
// agent-scope atomic loads (the atomics execute beyond this CU's L1, which may still hold the zeroed lines).
template <typename OT, bool kAligned>
__global__ __launch_bounds__(256) void occ_fallback_kernel(const double* __restrict__ pts,
                                                            const double* __restrict__ labels,
                                                            const int64_t* __restrict__ offsets,
                                                            const double* __restrict__ desc, int nx, int ny, int nz,
                                                            KeepLabels keep, const int32_t* __restrict__ flags,
                                                            int32_t* __restrict__ counts_ws,
                                                            int32_t* __restrict__ towers_ws, OT* __restrict__ occ,
                                                            OT* __restrict__ gt_occ, const int32_t* __restrict__ dims) {
    const int b = blockIdx.x;
    if (!flags[b]) return;
    const int own_nx = dims ? dims[b * 3 + 0] : nx, own_ny = dims ? dims[b * 3 + 1] : ny, own_nz = dims ? dims[b * 3 + 2] : nz;
    extern __shared__ double smem[];
    const int ne = nx + ny + nz + 3;
    double* edges = smem;
    int* cmin = reinterpret_cast<int*>(smem + ne);  // [ny]
    const size_t V = (size_t)nx * ny * nz;
    int32_t* c = counts_ws + (size_t)b * V;
    int32_t* t = (gt_occ && towers_ws) ? towers_ws + (size_t)b * V : nullptr;
    for (size_t i = threadIdx.x; i < V; i += blockDim.x) {
        c[i] = 0;
        if (t) t[i] = 0;
    }
    for (int y = threadIdx.x; y < ny; y += blockDim.x) cmin[y] = INT_MAX;
    const double* d = desc + (size_t)b * SN_DESC_LEN(nx, ny, nz);
    for (int i = threadIdx.x; i < ne; i += blockDim.x) edges[i] = d[6 + i];
    __threadfence();
    __syncthreads();
    Binner bin;
    bin.init(edges, d, nx, ny, nz);
    for_each_point<kAligned>(pts, offsets[b], offsets[b + 1], (long)threadIdx.x, (long)blockDim.x,
                             [&](double x, double y, double z, long, double label) {
                                 const int f = bin.flat(x, y, z);
                                 if (f < 0) return;
                                 atomicAdd(&c[f], 1);
                                 if (t && is_kept(label, keep)) atomicAdd(&t[f], 1);
                             }, t ? labels : nullptr);
    __threadfence();
    __syncthreads();
    for (size_t i = threadIdx.x; i < V; i += blockDim.x) {
        // (voxel-size mode: the column statistics are over the tile's own n_z x n_x x n_y part of the padded grid)
        const int yy = (int)(i % ny), xx = (int)((i / ny) % nx), zz = (int)(i / ((size_t)ny * nx));
        if (yy < own_ny && xx < own_nx && zz < own_nz)
            atomicMin(&cmin[yy], __hip_atomic_load(&c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();
    for (size_t i = threadIdx.x; i < V; i += blockDim.x) {
        const int cnt = __hip_atomic_load(&c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        occ[(size_t)b * V + i] = (cnt > cmin[i % ny]) ? (OT)1 : (OT)0;  // == ToFullDense(normalize_xyz(counts))
        if (t)
            gt_occ[(size_t)b * V + i] =
                (__hip_atomic_load(&t[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) ? (OT)1 : (OT)0;
    }
}

// ---------------------------------------------------------------- finalize (counting path)
__global__ void colstats_init_kernel(int32_t* cs, int B, int ny) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * 2 * ny) cs[i] = (((i / ny) & 1) == 0) ? INT_MAX : 0;
}

// grid (S, B): block s takes rows r = s, s+S, ... of the [nz*nx, ny] count matrix of tile b;
// threads run along y (coalesced), kThreads/ny' row lanes deep.
// desc (nullable): size-mode descriptor -- rows (z, x) beyond the tile's own dims are not part of its grid and stay out
// of the column statistics (a row is real iff the upper edges of its z and x bins are finite).
__device__ __forceinline__ bool real_row(const double* __restrict__ d, int nx, int ny, int r) {
    const int z = r / nx, x = r - z * nx;
    return d[6 + x + 1] < DBL_MAX && d[6 + nx + 1 + ny + 1 + z + 1] < DBL_MAX;
}
__global__ __launch_bounds__(kThreads) void colstats_kernel(const int32_t* __restrict__ counts, int rows, int ny,
                                                            int32_t* __restrict__ cs,
                                                            const int32_t* __restrict__ gate,
                                                            const double* __restrict__ desc, int nx, int desc_len) {
    const int b = blockIdx.y;
    if (gate && !gate[b]) return;
    const double* d = desc ? desc + (size_t)b * desc_len : nullptr;
    const int32_t* c = counts + (size_t)b * rows * ny;
    for (int y0 = 0; y0 < ny; y0 += kThreads) {
        const int width = min(ny - y0, kThreads);
        const int depth = kThreads / width;  // rows processed in parallel by this block
        const int ty = threadIdx.x % width, tr = threadIdx.x / width;
        if (tr >= depth) continue;
        int mn = INT_MAX, mx = 0;
        for (int r = blockIdx.x * depth + tr; r < rows; r += gridDim.x * depth) {
            if (d && !real_row(d, nx, ny, r)) continue;
            int v = c[(size_t)r * ny + y0 + ty];
            mn = min(mn, v);
            mx = max(mx, v);
        }
        if (mn != INT_MAX) {
            atomicMin(&cs[(b * 2 + 0) * ny + y0 + ty], mn);
            atomicMax(&cs[(b * 2 + 1) * ny + y0 + ty], mx);
        }
    }
}

// grid (blocks, B)
template <typename OT>
__global__ __launch_bounds__(kThreads) void finalize_kernel(const int32_t* __restrict__ counts,
                                                            const int32_t* __restrict__ towers,
                                                            const int32_t* __restrict__ cs, size_t V, int ny,
                                                            double* __restrict__ density, double* __restrict__ gt,
                                                            OT* __restrict__ occ, OT* __restrict__ gt_occ,
                                                            const int32_t* __restrict__ gate,
                                                            const double* __restrict__ desc, int nx, int desc_len) {
    const int b = blockIdx.y;
    if (gate && !gate[b]) return;
    const double* d = desc ? desc + (size_t)b * desc_len : nullptr;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t v = (size_t)blockIdx.x * kThreads + threadIdx.x; v < V; v += stride) {
        const size_t i = (size_t)b * V + v;
        const int y = (int)(v % ny);
        if (d && !(real_row(d, nx, ny, (int)(v / ny)) && d[6 + nx + 1 + y + 1] < DBL_MAX)) {
            // a voxel beyond this tile's own dims (size mode: grids are allocated at the batch maximum): not part of the grid
            if (density) density[i] = 0.0;
            if (occ) occ[i] = (OT)0;
            if (gt) gt[i] = 0.0;
            if (gt_occ) gt_occ[i] = (OT)0;
            continue;
        }
        const int c = counts[i];
        if (density || occ) {
            // sklearn MinMaxScaler: scale = 1/range (range < 10 eps -> 1); X*scale + (0 - min*scale)
            const int mn = cs[(b * 2 + 0) * ny + y], mx = cs[(b * 2 + 1) * ny + y];
            double rng = (double)(mx - mn);
            if (rng < 10.0 * DBL_EPSILON) rng = 1.0;
            const double scale = __ddiv_rn(1.0, rng);
            const double min_ = __dsub_rn(0.0, __dmul_rn((double)mn, scale));
            const double val = __dadd_rn(__dmul_rn((double)c, scale), min_);
            if (density) density[i] = val;
            if (occ) occ[i] = (val > 0.0) ? (OT)1 : (OT)0;
        }
        if (gt || gt_occ) {
            const int t = towers[i];
            const double r = (c > 0) ? __ddiv_rn((double)t, (double)c) : 0.0;
            if (gt) gt[i] = r;
            if (gt_occ) gt_occ[i] = (r > 0.0) ? (OT)1 : (OT)0;
        }
    }
}

// ---------------------------------------------------------------- grid -> points (per-point gather)
// out[i] = grid[b, vz(i), vx(i), vy(i)] with the SAME binning as the scatter (so a point reads the voxel it fell
// into); points outside the edge table get `fill`.
// vxg_to_xyz: one thread per cell, one 32-byte row (two 16-byte stores; a wave writes 2 KB contiguous)
struct Vec3d { double v[3]; };
template <typename T>
__global__ __launch_bounds__(kThreads) void grid_to_points_kernel(const T* __restrict__ grid, int n1, int n2, long V,
                                                                  Vec3d origin, Vec3d size, double* __restrict__ out) {
    for (long n = (long)blockIdx.x * kThreads + threadIdx.x; n < V; n += (long)gridDim.x * kThreads) {
        const long i01 = n / n2;
        const int i2 = (int)(n - i01 * n2), i1 = (int)(i01 % n1), i0 = (int)(i01 / n1);
        double2* o = reinterpret_cast<double2*>(out + 4 * n);
        o[0] = make_double2(__dadd_rn(origin.v[0], __dmul_rn((double)i0, size.v[0])),
                            __dadd_rn(origin.v[1], __dmul_rn((double)i1, size.v[1])));
        o[1] = make_double2(__dadd_rn(origin.v[2], __dmul_rn((double)i2, size.v[2])), (double)grid[n]);
    }
}

template <typename T, bool kAligned>
__global__ __launch_bounds__(kThreads) void gather_points_kernel(const T* __restrict__ grid, int channels,
                                                                 const double* __restrict__ pts,
                                                                 const int64_t* __restrict__ offsets,
                                                                 const double* __restrict__ desc, int nx, int ny,
                                                                 int nz, T fill, T* __restrict__ out) {
    extern __shared__ double edges[];
    const int b = blockIdx.y;
    const double* d = desc + (size_t)b * SN_DESC_LEN(nx, ny, nz);
    load_edges(edges, d, nx + ny + nz + 3, kThreads);
    Binner bin;
    bin.init(edges, d, nx, ny, nz);
    const size_t V = (size_t)nx * ny * nz;
    const T* g = grid + (size_t)b * channels * V;
    const long total = offsets[gridDim.y];
    for_each_point<kAligned>(pts, offsets[b], offsets[b + 1], (long)blockIdx.x * kThreads + threadIdx.x,
                             (long)gridDim.x * kThreads, [&](double x, double y, double z, long i) {
                                 const int f = bin.flat(x, y, z);
                                 for (int c = 0; c < channels; ++c)
                                     out[(size_t)c * total + i] = (f < 0) ? fill : g[(size_t)c * V + f];
                             });
}

inline int blocks_per_tile(int B, int target) {
    int per = (target + B - 1) / B;
    return per < 1 ? 1 : (per > 256 ? 256 : per);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int fill_keep(KeepLabels& keep, const double* keep_labels_host, int n_keep) {
    if (n_keep < 0 || n_keep > kMaxKeep) return -1;
    keep.n = n_keep;
    for (int i = 0; i < kMaxKeep; ++i) keep.v[i] = (i < n_keep) ? keep_labels_host[i] : 0.0;
    return 0;
}

// counting scatter + colstats + finalize, optionally gated per tile; shared by sn_voxel_scatter /
// sn_voxel_finalize / the fallback of sn_voxel_occupancy
void launch_scatter(const double* pts, const double* labels, const int64_t* offsets, int B, const double* desc,
                    int nx, int ny, int nz, int32_t* counts, int32_t* towers, const KeepLabels& keep,
                    int32_t* dropped, const int32_t* gate, hipStream_t s) {
    const size_t V = (size_t)nx * ny * nz;
    hipLaunchKernelGGL(gated_zero_kernel, dim3(64, B), dim3(256), 0, s, counts, towers, V, gate);
    const size_t lds = (size_t)(nx + ny + nz + 3) * sizeof(double);
    dim3 grid(blocks_per_tile(B, 1024), B);
    const bool al = aligned16(pts);
    if (al)
        hipLaunchKernelGGL(scatter_kernel<true>, grid, dim3(kThreads), lds, s, pts, labels, offsets, desc, nx, ny, nz,
                           counts, towers, keep, dropped, gate);
    else
        hipLaunchKernelGGL(scatter_kernel<false>, grid, dim3(kThreads), lds, s, pts, labels, offsets, desc, nx, ny,
                           nz, counts, towers, keep, dropped, gate);
}

template <typename OT>
void launch_finalize(const int32_t* counts, const int32_t* towers, int B, int nx, int ny, int nz, int32_t* colstats,
                     double* density, double* gt, OT* occ, OT* gt_occ, const int32_t* gate, hipStream_t s,
                     const double* desc = nullptr) {
    const size_t V = (size_t)nx * ny * nz;
    const int dlen = SN_DESC_LEN(nx, ny, nz);
    if (density || occ) {
        hipLaunchKernelGGL(colstats_init_kernel, dim3((B * 2 * ny + 255) / 256), dim3(256), 0, s, colstats, B, ny);
        const int rows = nz * nx;
        int S = (rows + 63) / 64;
        if (S > 64) S = 64;
        hipLaunchKernelGGL(colstats_kernel, dim3(S, B), dim3(kThreads), 0, s, counts, rows, ny, colstats, gate, desc, nx,
                           dlen);
    }
    size_t blocks = (V + kThreads - 1) / kThreads;
    const size_t cap = (size_t)blocks_per_tile(B, 4096);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(finalize_kernel<OT>, dim3((unsigned)blocks, B), dim3(kThreads), 0, s, counts, towers, colstats,
                       V, ny, density, gt, occ, gt_occ, gate, desc, nx, dlen);
}

}  // namespace

extern "C" int sn_voxel_bbox(const double* pts, const int64_t* offsets, int B, double* bbox, sn_stream_t stream) {
    if (!pts || !offsets || !bbox) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_bbox: null pointer");
    if (B <= 0) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_bbox: B=%d", B);
    hipStream_t s = sn::as_stream(stream);
    auto* enc = reinterpret_cast<unsigned long long*>(bbox);
    hipLaunchKernelGGL(bbox_init_kernel, dim3((B * 6 + 255) / 256), dim3(256), 0, s, enc, B);
    dim3 grid(blocks_per_tile(B, 1024), B);
    if (aligned16(pts))
        hipLaunchKernelGGL(bbox_reduce_kernel<true>, grid, dim3(kThreads), 0, s, pts, offsets, enc);
    else
        hipLaunchKernelGGL(bbox_reduce_kernel<false>, grid, dim3(kThreads), 0, s, pts, offsets, enc);
    hipLaunchKernelGGL(bbox_decode_kernel, dim3((B * 6 + 255) / 256), dim3(256), 0, s, enc, offsets, B);
    return sn::check_launch("sn_voxel_bbox");
}

static int desc_common(const double* box, int B, int nx, int ny, int nz, int regular, int from_bounds, double* desc,
                       sn_stream_t stream, const char* who) {
    if (!box || !desc) return sn::fail(SN_ERR_INVALID_ARG, "%s: null pointer", who);
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "%s: non-positive extent (B=%d n=%d,%d,%d)", who, B, nx, ny, nz);
    hipLaunchKernelGGL(desc_kernel, dim3(B), dim3(128), 0, sn::as_stream(stream), box, 0, nx, ny, nz, regular,
                       from_bounds, (double*)nullptr, desc);
    return sn::check_launch(who);
}

extern "C" int sn_voxel_desc(const double* bbox, int B, int nx, int ny, int nz, int regular, double* desc,
                             sn_stream_t stream) {
    return desc_common(bbox, B, nx, ny, nz, regular, 0, desc, stream, "sn_voxel_desc");
}

extern "C" int sn_voxel_desc_from_bounds(const double* bounds, int B, int nx, int ny, int nz, double* desc,
                                         sn_stream_t stream) {
    return desc_common(bounds, B, nx, ny, nz, 0, 1, desc, stream, "sn_voxel_desc_from_bounds");
}

extern "C" int sn_voxel_desc_sized(const double* bbox, int B, const double* size_xyz_host, int nx, int ny, int nz,
                                   double* desc, int32_t* dims, int32_t* status, sn_stream_t stream) {
    if (!bbox || !desc || !size_xyz_host) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_desc_sized: null pointer");
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_desc_sized: non-positive extent (B=%d n=%d,%d,%d)", B, nx, ny, nz);
    Vec3s sz;
    for (int c = 0; c < 3; ++c) {
        sz.v[c] = size_xyz_host[c];
        if (!(sz.v[c] > 0.0)) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_desc_sized: voxel size must be > 0");
    }
    hipLaunchKernelGGL(desc_sized_kernel, dim3(B), dim3(128), 0, sn::as_stream(stream), bbox, 0, sz, nx, ny, nz, desc, dims,
                       status, (double*)nullptr);
    return sn::check_launch("sn_voxel_desc_sized");
}

extern "C" int sn_voxel_finalize_sized(const int32_t* counts, const int32_t* tower_counts, int B, int nx, int ny, int nz,
                                       const double* desc, int32_t* colstats, double* density, double* gt, float* occ,
                                       float* gt_occ, sn_stream_t stream) {
    if (!counts || !desc) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_finalize_sized: null counts or desc");
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_finalize_sized: non-positive extent (B=%d n=%d,%d,%d)", B, nx, ny,
                        nz);
    if ((gt || gt_occ) && !tower_counts)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_finalize_sized: gt outputs need tower_counts");
    if ((density || occ) && !colstats)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_finalize_sized: density/occ need the colstats workspace");
    launch_finalize<float>(counts, tower_counts, B, nx, ny, nz, colstats, density, gt, occ, gt_occ, nullptr,
                           sn::as_stream(stream), desc);
    return sn::check_launch("sn_voxel_finalize_sized");
}

extern "C" int sn_voxel_prepare(const double* pts, const int64_t* offsets, int B, int nx, int ny, int nz,
                                int regular, double* partial_ws, double* bbox, double* desc, sn_stream_t stream) {
    if (!pts || !offsets || !partial_ws || !desc) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_prepare: null pointer");
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_prepare: non-positive extent (B=%d n=%d,%d,%d)", B, nx, ny, nz);
    hipStream_t s = sn::as_stream(stream);
    const int nparts = SN_BBOX_PARTS;
    dim3 grid(nparts, B);
    if (aligned16(pts))
        hipLaunchKernelGGL(bbox_partial_kernel<true>, grid, dim3(kThreads), 0, s, pts, offsets, partial_ws);
    else
        hipLaunchKernelGGL(bbox_partial_kernel<false>, grid, dim3(kThreads), 0, s, pts, offsets, partial_ws);
    hipLaunchKernelGGL(desc_kernel, dim3(B), dim3(128), 0, s, (const double*)partial_ws, nparts, nx, ny, nz, regular,
                       0, bbox, desc);
    return sn::check_launch("sn_voxel_prepare");
}

extern "C" int sn_voxel_scatter(const double* pts, const double* labels, const int64_t* offsets, int B,
                                const double* desc, int nx, int ny, int nz, int32_t* counts, int32_t* tower_counts,
                                const double* keep_labels_host, int n_keep, int32_t* dropped, sn_stream_t stream) {
    if (!pts || !offsets || !desc || !counts) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_scatter: null pointer");
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_scatter: non-positive extent (B=%d n=%d,%d,%d)", B, nx, ny, nz);
    KeepLabels keep;
    if (fill_keep(keep, keep_labels_host, (n_keep > 0 && !keep_labels_host) ? -1 : n_keep))
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_voxel_scatter: n_keep=%d outside [0,%d] or null list", n_keep,
                        kMaxKeep);
    if (tower_counts && !labels)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_scatter: tower_counts needs labels and keep_labels_host");
    if ((size_t)(nx + ny + nz + 3) * sizeof(double) > 64 * 1024)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_voxel_scatter: edge table > 64 KiB");
    hipStream_t s = sn::as_stream(stream);
    if (dropped && hipMemsetAsync(dropped, 0, (size_t)B * sizeof(int32_t), s) != hipSuccess)
        return sn::check_launch("sn_voxel_scatter(memset)");
    launch_scatter(pts, labels, offsets, B, desc, nx, ny, nz, counts, tower_counts, keep, dropped, nullptr, s);
    return sn::check_launch("sn_voxel_scatter");
}

extern "C" int sn_voxel_finalize(const int32_t* counts, const int32_t* tower_counts, int B, int nx, int ny, int nz,
                                 int32_t* colstats, double* density, double* gt, float* occ, float* gt_occ,
                                 sn_stream_t stream) {
    if (!counts) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_finalize: null counts");
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_finalize: non-positive extent (B=%d n=%d,%d,%d)", B, nx, ny, nz);
    if ((gt || gt_occ) && !tower_counts)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_finalize: gt outputs need tower_counts");
    if ((density || occ) && !colstats)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_finalize: density/occ need the colstats workspace");
    launch_finalize<float>(counts, tower_counts, B, nx, ny, nz, colstats, density, gt, occ, gt_occ, nullptr,
                           sn::as_stream(stream));
    return sn::check_launch("sn_voxel_finalize");
}

// the one-pass kernel serves grids whose bitmap(s) fit one workgroup's LDS (no z-slabs): 64^3 with or without the GT plane
static bool onepass_eligible(int nx, int ny, int nz, int planes) {
    const size_t V = (size_t)nx * ny * nz;
    return sn::option_voxel_onepass() && V % 32 == 0 && (V / 32) * planes <= (size_t)kMaxOccWords;
}

// sn_voxel_occupancy (descriptor given) and sn_voxel_occupancy_fused (box_parts given: the binning kernel derives the
// descriptor itself and writes it to `desc`)
static int occupancy_impl(const double* pts, const double* labels, const int64_t* offsets, int B, double* desc, int nx,
                          int ny, int nz, const double* keep_labels_host, int n_keep, uint32_t* bits_ws, void* occ,
                          void* gt_occ, int out_dtype, int32_t* flags, int32_t* dropped, int32_t* counts_ws,
                          int32_t* towers_ws, const double* box_parts, int nbparts, int regular, double* bbox_out,
                          sn_stream_t stream, const int32_t* dims = nullptr, const BankRider* onepass_rider = nullptr,
                          bool onepass = false) {
    if (!pts || !offsets || !desc || !bits_ws || !occ)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_occupancy: null pointer");
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_occupancy: non-positive extent (B=%d n=%d,%d,%d)", B, nx, ny,
                        nz);
    if (out_dtype != SN_U8 && out_dtype != SN_F32)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_occupancy: out_dtype %d (SN_U8 | SN_F32)", out_dtype);
    const int planes = gt_occ ? 2 : 1;
    if (gt_occ && !labels) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_occupancy: gt_occ needs labels");
    KeepLabels keep;
    if (fill_keep(keep, keep_labels_host, (n_keep > 0 && !keep_labels_host) ? -1 : n_keep))
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_voxel_occupancy: n_keep=%d outside [0,%d] or null list", n_keep,
                        kMaxKeep);
    const size_t V = (size_t)nx * ny * nz;
    // z-slabs: the bitmap(s) of one slab must fit the 64 KiB LDS budget and start on a word boundary
    int slabs = 0;
    if (V % 32 == 0)
        for (int sl = 1; sl <= kOccParts; sl *= 2)
            if (nz % sl == 0 && (V / 32) % sl == 0 && ((size_t)nz / sl) * nx * ny % 32 == 0 &&
                (V / 32 / sl) * planes <= (size_t)kMaxOccWords) {
                slabs = sl;
                break;
            }
    if (!slabs)
        return sn::fail(SN_ERR_UNSUPPORTED,
                        "sn_voxel_occupancy: %zu voxels x %d planes do not fit the LDS bitmap in <= %d z-slabs "
                        "(use sn_voxel_scatter + sn_voxel_finalize)", V, planes, kOccParts);
    const int parts = onepass ? kOneParts : kOccParts / slabs;   // (eligibility of the one-pass form implies slabs == 1)
    if (gt_occ && flags && counts_ws && !towers_ws)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_occupancy: the counting fallback needs towers_ws for gt_occ");
    const int words = (int)(V / 32);
    const int ne = nx + ny + nz + 3;
    hipStream_t s = sn::as_stream(stream);
    // per-part dropped counts live behind the partial bitmaps in bits_ws (SN_OCC_WS_WORDS accounts for them)
    int32_t* dropped_parts = reinterpret_cast<int32_t*>(bits_ws + (size_t)B * kOccParts * planes * words);
    const size_t lds1 =
        (size_t)((ne + 1) & ~1) * sizeof(double) + (size_t)(words / slabs) * planes * sizeof(uint32_t);
    const bool al = aligned16(pts) && (!labels || aligned16(labels));
    unsigned long long* exchange = nullptr;
    if (onepass) {
        // ONE pass over the points: the box, the descriptor and the bitmap in one launch (occ_onepass_kernel); `box_parts` is
        // the exchange area.  Riders (K2) in the first grid rows when a bank was handed in.
        static std::atomic<unsigned> epoch_ctr{0};
        const unsigned epoch = epoch_ctr.fetch_add(1, std::memory_order_relaxed) + 1;
        BankRider none{};
        const BankRider& r = onepass_rider ? *onepass_rider : none;
        const int rider_rows = onepass_rider ? (r.nblocks + kOneParts - 1) / kOneParts : 0;
        exchange = reinterpret_cast<unsigned long long*>(const_cast<double*>(box_parts));
        const bool al_p = aligned16(pts);
        auto kern = al_p ? occ_onepass_kernel<true> : occ_onepass_kernel<false>;
        if (sn::ensure_dynamic_lds((const void*)kern, 96 * 1024) != hipSuccess)
            return sn::check_launch("sn_voxel_occupancy_fused(hipFuncSetAttribute)");
        hipLaunchKernelGGL(kern, dim3(kOneParts, B + rider_rows), dim3(kOneThreads), lds1, s, pts, labels, offsets, nx, ny, nz,
                           words, planes, keep, bits_ws, dropped_parts, flags, const_cast<double*>(box_parts), epoch, regular,
                           desc, bbox_out, sn::option_voxel_onepass_spin(), rider_rows, r);
    } else {
        auto kern = al ? occ_partial_kernel<true> : occ_partial_kernel<false>;
        if (sn::ensure_dynamic_lds((const void*)kern, 96 * 1024) != hipSuccess)
            return sn::check_launch("sn_voxel_occupancy(hipFuncSetAttribute)");
        hipLaunchKernelGGL(kern, dim3(parts * slabs, B), dim3(kOccThreads), lds1, s, pts, labels, offsets, desc, nx,
                           ny, nz, words, planes, parts, slabs, keep, bits_ws, dropped_parts, flags, box_parts, nbparts,
                           regular, box_parts ? desc : nullptr, bbox_out);
    }
    const int rows = nz * nx;
    // (four words per thread where the finalize kernel can: see its vec4 path; any grid is correct, the loops stride)
    const bool fin4 = (words & 3) == 0 && ((uintptr_t)bits_ws & 15) == 0;
    int C = ((fin4 ? words / 4 : words) + kThreads - 1) / kThreads;
    if (C > blocks_per_tile(B, 2048)) C = blocks_per_tile(B, 2048);
    if (out_dtype == SN_U8)
        hipLaunchKernelGGL(occ_finalize_kernel<uint8_t>, dim3(C, B), dim3(kThreads), 0, s, bits_ws, words, planes,
                           parts, rows, ny, V, (uint8_t*)occ, (uint8_t*)gt_occ, flags, dropped_parts, dropped, dims, nx,
                           exchange);
    else
        hipLaunchKernelGGL(occ_finalize_kernel<float>, dim3(C, B), dim3(kThreads), 0, s, bits_ws, words, planes, parts,
                           rows, ny, V, (float*)occ, (float*)gt_occ, flags, dropped_parts, dropped, dims, nx, exchange);
    // flagged tiles (a y column might be full): redone exactly by one gated launch
    if (flags && counts_ws) {
        const size_t lds3 = (size_t)ne * sizeof(double) + (size_t)ny * sizeof(int);
        if (lds3 > 64 * 1024) return sn::fail(SN_ERR_UNSUPPORTED, "sn_voxel_occupancy: edge table too large");
#define SN_FALLBACK(OT, AL)                                                                                        \
    hipLaunchKernelGGL((occ_fallback_kernel<OT, AL>), dim3(B), dim3(256), lds3, s, pts, gt_occ ? labels : nullptr, \
                       offsets, desc, nx, ny, nz, keep, flags, counts_ws, towers_ws, (OT*)occ, (OT*)gt_occ, dims)
        if (out_dtype == SN_U8) { if (al) SN_FALLBACK(uint8_t, true); else SN_FALLBACK(uint8_t, false); }
        else { if (al) SN_FALLBACK(float, true); else SN_FALLBACK(float, false); }
#undef SN_FALLBACK
    }
    return sn::check_launch("sn_voxel_occupancy");
}

extern "C" int sn_voxel_occupancy(const double* pts, const double* labels, const int64_t* offsets, int B,
                                  const double* desc, int nx, int ny, int nz, const double* keep_labels_host,
                                  int n_keep, uint32_t* bits_ws, void* occ, void* gt_occ, int out_dtype,
                                  int32_t* flags, int32_t* dropped, int32_t* counts_ws, int32_t* towers_ws,
                                  sn_stream_t stream) {
    return occupancy_impl(pts, labels, offsets, B, const_cast<double*>(desc), nx, ny, nz, keep_labels_host, n_keep,
                          bits_ws, occ, gt_occ, out_dtype, flags, dropped, counts_ws, towers_ws, nullptr, 0, 0, nullptr,
                          stream);
}

extern "C" int sn_voxel_occupancy_fused(const double* pts, const double* labels, const int64_t* offsets, int B, int nx,
                                        int ny, int nz, int regular, const double* keep_labels_host, int n_keep,
                                        double* partial_ws, double* bbox, double* desc, uint32_t* bits_ws, void* occ,
                                        void* gt_occ, int out_dtype, int32_t* flags, int32_t* dropped,
                                        int32_t* counts_ws, int32_t* towers_ws, sn_stream_t stream) {
    if (!pts || !offsets || !partial_ws || !desc)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_occupancy_fused: null pointer");
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_occupancy_fused: non-positive extent (B=%d n=%d,%d,%d)", B, nx,
                        ny, nz);
    hipStream_t s = sn::as_stream(stream);
    const bool one = onepass_eligible(nx, ny, nz, gt_occ ? 2 : 1);
    if (!one) {
        dim3 grid(SN_BBOX_PARTS, B);
        if (aligned16(pts))
            hipLaunchKernelGGL(bbox_partial_kernel<true>, grid, dim3(kThreads), 0, s, pts, offsets, partial_ws);
        else
            hipLaunchKernelGGL(bbox_partial_kernel<false>, grid, dim3(kThreads), 0, s, pts, offsets, partial_ws);
        if (int rc = sn::check_launch("sn_voxel_occupancy_fused(bbox)")) return rc;
    }
    return occupancy_impl(pts, labels, offsets, B, desc, nx, ny, nz, keep_labels_host, n_keep, bits_ws, occ, gt_occ,
                          out_dtype, flags, dropped, counts_ws, towers_ws, partial_ws, SN_BBOX_PARTS, regular ? 1 : 0,
                          bbox, stream, nullptr, nullptr, one);
}

// the bounding-box launch of the fused entry points, with K2's workgroups as riders when a bank is handed in
static int check_rider(const char* who, const float* params, const int32_t* kinds, int G, int kz, int kx, int ky, float* bank,
                       const float* lambdas, const int32_t* order, int last, const float* lambdas_out, void* prep) {
    if (!params || !kinds || !bank || !prep) return sn::fail(SN_ERR_INVALID_ARG, "%s: null bank argument", who);
    if (lambdas && (!order || !lambdas_out)) return sn::fail(SN_ERR_INVALID_ARG, "%s: lambdas without order / out", who);
    if (G <= 0 || (lambdas && (last < 0 || last >= G))) return sn::fail(SN_ERR_INVALID_ARG, "%s: bad G / last", who);
    if (kz != 9 || kx != 9 || ky != 9)
        return sn::fail(SN_ERR_UNSUPPORTED, "%s: the prepared contraction serves 9 x 9 x 9 kernels (got %d,%d,%d)", who, kz, kx,
                        ky);
    if (reinterpret_cast<uintptr_t>(prep) & 15) return sn::fail(SN_ERR_INVALID_ARG, "%s: prep must be 16-byte aligned", who);
    return SN_OK;
}

static void launch_bbox_with_riders(const double* pts, const int64_t* offsets, double* partial_ws, int B, const float* params,
                                    const int32_t* kinds, int G, float* bank, int32_t* status, float* lambdas,
                                    const int32_t* order, int last, float* lambdas_out, void* prep, hipStream_t s) {
    BankRider r{params, kinds, bank, status, static_cast<uint8_t*>(prep), lambdas, order, lambdas_out, last, G,
                16 * ((G + 15) / 16) + (lambdas ? 1 : 0)};
    const int extra_rows = (r.nblocks + SN_BBOX_PARTS - 1) / SN_BBOX_PARTS;
    dim3 grid(SN_BBOX_PARTS, B + extra_rows);
    if (aligned16(pts))
        hipLaunchKernelGGL(bbox_partial_bank_kernel<true>, grid, dim3(kThreads), 0, s, pts, offsets, partial_ws, extra_rows, r);
    else
        hipLaunchKernelGGL(bbox_partial_bank_kernel<false>, grid, dim3(kThreads), 0, s, pts, offsets, partial_ws, extra_rows,
                           r);
}

extern "C" int sn_voxel_occupancy_fused_bank(const double* pts, const double* labels, const int64_t* offsets, int B, int nx,
                                             int ny, int nz, int regular, const double* keep_labels_host, int n_keep,
                                             double* partial_ws, double* bbox, double* desc, uint32_t* bits_ws, void* occ,
                                             void* gt_occ, int out_dtype, int32_t* flags, int32_t* dropped,
                                             int32_t* counts_ws, int32_t* towers_ws, const float* params,
                                             const int32_t* kinds, int G, int kz, int kx, int ky, float* bank,
                                             int32_t* status, float* lambdas, const int32_t* order, int last,
                                             float* lambdas_out, void* prep, sn_stream_t stream) {
    if (!pts || !offsets || !partial_ws || !desc)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_occupancy_fused_bank: null pointer");
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_occupancy_fused_bank: non-positive extent (B=%d n=%d,%d,%d)", B, nx,
                        ny, nz);
    if (int rc = check_rider("sn_voxel_occupancy_fused_bank", params, kinds, G, kz, kx, ky, bank, lambdas, order, last,
                             lambdas_out, prep))
        return rc;
    if (onepass_eligible(nx, ny, nz, gt_occ ? 2 : 1)) {   // K2 rides in the one-pass kernel's first grid rows
        const BankRider r{params, kinds, bank, status, static_cast<uint8_t*>(prep), lambdas, order, lambdas_out, last, G,
                          16 * ((G + 15) / 16) + (lambdas ? 1 : 0)};
        return occupancy_impl(pts, labels, offsets, B, desc, nx, ny, nz, keep_labels_host, n_keep, bits_ws, occ, gt_occ,
                              out_dtype, flags, dropped, counts_ws, towers_ws, partial_ws, SN_BBOX_PARTS, regular ? 1 : 0,
                              bbox, stream, nullptr, &r, true);
    }
    launch_bbox_with_riders(pts, offsets, partial_ws, B, params, kinds, G, bank, status, lambdas, order, last, lambdas_out,
                            prep, sn::as_stream(stream));
    if (int rc = sn::check_launch("sn_voxel_occupancy_fused_bank(bbox + bank)")) return rc;
    return occupancy_impl(pts, labels, offsets, B, desc, nx, ny, nz, keep_labels_host, n_keep, bits_ws, occ, gt_occ,
                          out_dtype, flags, dropped, counts_ws, towers_ws, partial_ws, SN_BBOX_PARTS, regular ? 1 : 0,
                          bbox, stream);
}

static int occupancy_sized_impl(const char* who, const double* pts, const double* labels, const int64_t* offsets, int B,
                                const double* size_xyz_host, int nx, int ny, int nz, const double* keep_labels_host,
                                int n_keep, double* partial_ws, double* bbox, double* desc, int32_t* dims, int32_t* status,
                                uint32_t* bits_ws, void* occ, void* gt_occ, int out_dtype, int32_t* flags, int32_t* dropped,
                                int32_t* counts_ws, int32_t* towers_ws, bool rider, const float* params,
                                const int32_t* kinds, int G, float* bank, int32_t* bank_status, float* lambdas,
                                const int32_t* order, int last, float* lambdas_out, void* prep, sn_stream_t stream) {
    if (!pts || !offsets || !partial_ws || !desc || !dims || !size_xyz_host)
        return sn::fail(SN_ERR_INVALID_ARG, "%s: null pointer", who);
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "%s: non-positive extent (B=%d n=%d,%d,%d)", who, B, nx, ny, nz);
    Vec3s sz;
    for (int c = 0; c < 3; ++c) {
        sz.v[c] = size_xyz_host[c];
        if (!(sz.v[c] > 0.0)) return sn::fail(SN_ERR_INVALID_ARG, "%s: voxel size must be > 0", who);
    }
    hipStream_t s = sn::as_stream(stream);
    if (rider) {
        launch_bbox_with_riders(pts, offsets, partial_ws, B, params, kinds, G, bank, bank_status, lambdas, order, last,
                                lambdas_out, prep, s);
    } else {
        dim3 grid(SN_BBOX_PARTS, B);
        if (aligned16(pts))
            hipLaunchKernelGGL(bbox_partial_kernel<true>, grid, dim3(kThreads), 0, s, pts, offsets, partial_ws);
        else
            hipLaunchKernelGGL(bbox_partial_kernel<false>, grid, dim3(kThreads), 0, s, pts, offsets, partial_ws);
    }
    // per-tile grid extents and padded edge tables from the partial boxes, on the device (no host round trip)
    hipLaunchKernelGGL(desc_sized_kernel, dim3(B), dim3(128), 0, s, partial_ws, SN_BBOX_PARTS, sz, nx, ny, nz, desc, dims,
                       status, bbox);
    if (int rc = sn::check_launch("sn_voxel_occupancy_sized(bbox, descriptor)")) return rc;
    // the LDS-bitmap kernels run unchanged on the padded tables (a point never bins beyond its tile's own dims: the
    // edges there are +inf); the column rule of ToFullDense(normalize_xyz(.)) looks at each tile's own part only
    return occupancy_impl(pts, labels, offsets, B, desc, nx, ny, nz, keep_labels_host, n_keep, bits_ws, occ, gt_occ,
                          out_dtype, flags, dropped, counts_ws, towers_ws, nullptr, 0, 0, nullptr, stream, dims);
}

extern "C" int sn_voxel_occupancy_sized(const double* pts, const double* labels, const int64_t* offsets, int B,
                                        const double* size_xyz_host, int nx, int ny, int nz,
                                        const double* keep_labels_host, int n_keep, double* partial_ws, double* bbox,
                                        double* desc, int32_t* dims, int32_t* status, uint32_t* bits_ws, void* occ,
                                        void* gt_occ, int out_dtype, int32_t* flags, int32_t* dropped,
                                        int32_t* counts_ws, int32_t* towers_ws, sn_stream_t stream) {
    return occupancy_sized_impl("sn_voxel_occupancy_sized", pts, labels, offsets, B, size_xyz_host, nx, ny, nz,
                                keep_labels_host, n_keep, partial_ws, bbox, desc, dims, status, bits_ws, occ, gt_occ, out_dtype,
                                flags, dropped, counts_ws, towers_ws, false, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                                nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int sn_voxel_occupancy_sized_bank(const double* pts, const double* labels, const int64_t* offsets, int B,
                                             const double* size_xyz_host, int nx, int ny, int nz,
                                             const double* keep_labels_host, int n_keep, double* partial_ws, double* bbox,
                                             double* desc, int32_t* dims, int32_t* status, uint32_t* bits_ws, void* occ,
                                             void* gt_occ, int out_dtype, int32_t* flags, int32_t* dropped,
                                             int32_t* counts_ws, int32_t* towers_ws, const float* params,
                                             const int32_t* kinds, int G, int kz, int kx, int ky, float* bank,
                                             int32_t* bank_status, float* lambdas, const int32_t* order, int last,
                                             float* lambdas_out, void* prep, sn_stream_t stream) {
    if (int rc = check_rider("sn_voxel_occupancy_sized_bank", params, kinds, G, kz, kx, ky, bank, lambdas, order, last,
                             lambdas_out, prep))
        return rc;
    return occupancy_sized_impl("sn_voxel_occupancy_sized_bank", pts, labels, offsets, B, size_xyz_host, nx, ny, nz,
                                keep_labels_host, n_keep, partial_ws, bbox, desc, dims, status, bits_ws, occ, gt_occ, out_dtype,
                                flags, dropped, counts_ws, towers_ws, true, params, kinds, G, bank, bank_status, lambdas, order,
                                last, lambdas_out, prep, stream);
}

extern "C" int sn_gather_points(const void* grid, int dtype, int channels, const double* pts, const int64_t* offsets,
                                int B, const double* desc, int nx, int ny, int nz, double fill, void* out,
                                sn_stream_t stream) {
    if (!grid || !pts || !offsets || !desc || !out) return sn::fail(SN_ERR_INVALID_ARG, "sn_gather_points: null pointer");
    if (B <= 0 || channels <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_gather_points: non-positive extent");
    if (dtype != SN_F32 && dtype != SN_F64)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_gather_points: dtype %d (SN_F32 | SN_F64)", dtype);
    const size_t lds = (size_t)(nx + ny + nz + 3) * sizeof(double);
    if (lds > 64 * 1024) return sn::fail(SN_ERR_UNSUPPORTED, "sn_gather_points: edge table > 64 KiB");
    hipStream_t s = sn::as_stream(stream);
    dim3 grd(blocks_per_tile(B, 1024), B);
    const bool al = aligned16(pts);
#define SN_GATHER(T, AL)                                                                                              \
    hipLaunchKernelGGL((gather_points_kernel<T, AL>), grd, dim3(kThreads), lds, s, (const T*)grid, channels, pts,    \
                       offsets, desc, nx, ny, nz, (T)fill, (T*)out)
    if (dtype == SN_F32) { if (al) SN_GATHER(float, true); else SN_GATHER(float, false); }
    else { if (al) SN_GATHER(double, true); else SN_GATHER(double, false); }
#undef SN_GATHER
    return sn::check_launch("sn_gather_points");
}

extern "C" int sn_grid_to_points(const void* grid, int dtype, int n0, int n1, int n2, const double* origin_host,
                                 const double* voxel_size_host, double* out, sn_stream_t stream) {
    if (!grid || !out) return sn::fail(SN_ERR_INVALID_ARG, "sn_grid_to_points: null pointer");
    if (n0 <= 0 || n1 <= 0 || n2 <= 0) return sn::fail(SN_ERR_INVALID_ARG, "sn_grid_to_points: empty grid");
    if (reinterpret_cast<uintptr_t>(out) & 15) return sn::fail(SN_ERR_INVALID_ARG, "sn_grid_to_points: out not 16-byte aligned");
    Vec3d o{{0.0, 0.0, 0.0}}, sz{{1.0, 1.0, 1.0}};
    for (int c = 0; c < 3; ++c) {
        if (origin_host) o.v[c] = origin_host[c];
        if (voxel_size_host) sz.v[c] = voxel_size_host[c];
    }
    const long V = (long)n0 * n1 * n2;
    hipStream_t s = sn::as_stream(stream);
    const int blocks = (int)std::min<long>((V + kThreads - 1) / kThreads, 256 * 16);
#define SN_G2P(T) hipLaunchKernelGGL((grid_to_points_kernel<T>), dim3(blocks), dim3(kThreads), 0, s, (const T*)grid, n1, n2, V, o, sz, out)
    switch (dtype) {
        case SN_F32: SN_G2P(float); break;
        case SN_F64: SN_G2P(double); break;
        case SN_U8:
        case SN_OCC8: SN_G2P(uint8_t); break;
        default: return sn::fail(SN_ERR_INVALID_ARG, "sn_grid_to_points: dtype");
    }
#undef SN_G2P
    return sn::check_launch("sn_grid_to_points");
}

#ifdef SN_CONV_TIMING
extern "C" void sn_debug_vox_times(unsigned long long* host) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_vox_t), sizeof(unsigned long long) * 1024 * 8);
}
#endif
