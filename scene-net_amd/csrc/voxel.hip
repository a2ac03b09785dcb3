// K1 -- point cloud -> voxel grid: fp64 bounding box, numpy.linspace-exact edge tables,
// per-point atomic scatter into the int32 occupancy grid, column min-max normalisation.
//
// Follows pyntcloud 0.1.6 VoxelGrid.compute as called from utils/pcd_processing.py:341-372,
// utils/voxelization.py:164-204 (hist_on_voxel), :244-300 (reg_on_voxel),
// utils/pcd_processing.py:305-321 (normalize_xyz), core/datasets/torch_transforms.py:33-34.
//
// All box / edge arithmetic is fp64 with explicit round-to-nearest mul/add (no FMA
// contraction) so that every edge equals numpy.linspace's bit for bit; the UTM-scale
// coordinates (|y| ~ 4.6e6) are never narrowed.
#include "common.h"
#include <cfloat>
#include <climits>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxKeep = 16;

struct KeepLabels {
    double v[kMaxKeep];
    int n;
};

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

struct MinMax3 {
    double mn[3], mx[3];
};

__device__ __forceinline__ void mm_init(MinMax3& m) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        m.mn[c] = DBL_MAX;
        m.mx[c] = -DBL_MAX;
    }
}
__device__ __forceinline__ void mm_point(MinMax3& m, double x, double y, double z) {
    m.mn[0] = fmin(m.mn[0], x); m.mx[0] = fmax(m.mx[0], x);
    m.mn[1] = fmin(m.mn[1], y); m.mx[1] = fmax(m.mx[1], y);
    m.mn[2] = fmin(m.mn[2], z); m.mx[2] = fmax(m.mx[2], z);
}

// Tile b owns points [p0, p1).  A thread-iteration takes TWO points = 48 contiguous bytes as
// three 16-byte loads; an odd first point is peeled so the pairs start 16-byte aligned.
template <bool kAligned>
__global__ __launch_bounds__(kThreads) void bbox_reduce_kernel(const double* __restrict__ pts,
                                                               const int64_t* __restrict__ offsets,
                                                               unsigned long long* __restrict__ enc) {
    const int b = blockIdx.y;
    const long p0 = offsets[b], p1 = offsets[b + 1];
    MinMax3 m;
    mm_init(m);
    const long gtid = (long)blockIdx.x * kThreads + threadIdx.x;
    const long gstride = (long)gridDim.x * kThreads;
    if (kAligned) {
        long q0 = p0 + (p0 & 1);  // first even point index >= p0
        if (q0 > p1) q0 = p1;
        if ((p0 & 1) && gtid == 0 && p0 < p1) mm_point(m, pts[3 * p0], pts[3 * p0 + 1], pts[3 * p0 + 2]);
        const long npair = (p1 - q0) >> 1;
        const double2* src = reinterpret_cast<const double2*>(pts + 3 * q0);
        for (long i = gtid; i < npair; i += gstride) {
            double2 a = src[3 * i], c = src[3 * i + 1], d = src[3 * i + 2];
            mm_point(m, a.x, a.y, c.x);
            mm_point(m, c.y, d.x, d.y);
        }
        if (((p1 - q0) & 1) && gtid == 0) mm_point(m, pts[3 * (p1 - 1)], pts[3 * (p1 - 1) + 1], pts[3 * (p1 - 1) + 2]);
    } else {
        for (long i = p0 + gtid; i < p1; i += gstride) mm_point(m, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    }
    // wave reduce, then one atomic per wave and slot
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m.mn[c] = fmin(m.mn[c], __shfl_xor(m.mn[c], o, 64));
            m.mx[c] = fmax(m.mx[c], __shfl_xor(m.mx[c], o, 64));
        }
    }
    if ((threadIdx.x & 63) == 0 && p1 > p0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (m.mn[c] != DBL_MAX) atomicMin(&enc[b * 6 + c], enc_f64(m.mn[c]));
            if (m.mx[c] != -DBL_MAX) atomicMax(&enc[b * 6 + 3 + c], enc_f64(m.mx[c]));
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

// ---------------------------------------------------------------- grid descriptor
// desc[b] = lo[3], hi[3], edges_x[nx+1], edges_y[ny+1], edges_z[nz+1]
// numpy.linspace(lo, hi, n+1): step = (hi-lo)/n ; e_k = k*step + lo ; e_n = hi.
__global__ void desc_kernel(const double* __restrict__ box, int nx, int ny, int nz, int regular, int from_bounds,
                            double* __restrict__ desc) {
    const int b = blockIdx.x;
    const int len = SN_DESC_LEN(nx, ny, nz);
    double* d = desc + (size_t)b * len;
    const double* bb = box + (size_t)b * 6;
    double lo[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lo[c] = bb[c];
        hi[c] = bb[3 + c];
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

// ---------------------------------------------------------------- scatter
// largest j with e[j] < p  (== numpy.searchsorted(e, p, side='left') - 1), in [-1, n]
__device__ __forceinline__ int bin_axis(double p, const double* e, int n, double lo, double inv_step) {
    double f = (p - lo) * inv_step;
    int k = (f > 0.0) ? (int)fmin(f, (double)n) : 0;
    while (k < n && e[k + 1] < p) ++k;
    while (k >= 0 && !(e[k] < p)) --k;
    return k;
}

struct ScatterCtx {
    const double *ex, *ey, *ez;
    double lo[3], inv[3];
    int nx, ny, nz;
    int32_t* counts;
    int32_t* towers;
};

__device__ __forceinline__ void scatter_point(const ScatterCtx& c, double x, double y, double z, bool tower,
                                              int& dropped) {
    if (x != x || y != y || z != z) { ++dropped; return; }
    int ix = bin_axis(x, c.ex, c.nx, c.lo[0], c.inv[0]);
    int iy = bin_axis(y, c.ey, c.ny, c.lo[1], c.inv[1]);
    int iz = bin_axis(z, c.ez, c.nz, c.lo[2], c.inv[2]);
    ix = max(ix, 0); iy = max(iy, 0); iz = max(iz, 0);  // np.clip(., 0, n)
    if (ix >= c.nx || iy >= c.ny || iz >= c.nz) { ++dropped; return; }  // index n: outside the table
    int flat = (iz * c.nx + ix) * c.ny + iy;
    atomicAdd(&c.counts[flat], 1);
    if (tower) atomicAdd(&c.towers[flat], 1);
}

__device__ __forceinline__ bool is_kept(double label, const KeepLabels& keep) {
    bool k = false;
    for (int i = 0; i < keep.n; ++i) k |= (label == keep.v[i]);
    return k;
}

template <bool kAligned>
__global__ __launch_bounds__(kThreads) void scatter_kernel(const double* __restrict__ pts,
                                                           const double* __restrict__ labels,
                                                           const int64_t* __restrict__ offsets,
                                                           const double* __restrict__ desc, int nx, int ny, int nz,
                                                           int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ towers, KeepLabels keep,
                                                           int32_t* __restrict__ dropped_out) {
    extern __shared__ double edges[];  // [nx+1 + ny+1 + nz+1]
    const int b = blockIdx.y;
    const int len = SN_DESC_LEN(nx, ny, nz);
    const double* d = desc + (size_t)b * len;
    const int ne = nx + ny + nz + 3;
    for (int i = threadIdx.x; i < ne; i += kThreads) edges[i] = d[6 + i];
    __syncthreads();

    ScatterCtx c;
    c.ex = edges; c.ey = edges + nx + 1; c.ez = edges + nx + ny + 2;
    c.nx = nx; c.ny = ny; c.nz = nz;
    const int n[3] = {nx, ny, nz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c.lo[a] = d[a];
        double step = (d[3 + a] - d[a]) / (double)n[a];
        c.inv[a] = (step > 0.0) ? 1.0 / step : 0.0;
    }
    const size_t V = (size_t)nx * ny * nz;
    c.counts = counts + (size_t)b * V;
    c.towers = towers ? towers + (size_t)b * V : nullptr;
    const bool want_tower = (towers != nullptr) && (labels != nullptr);

    const long p0 = offsets[b], p1 = offsets[b + 1];
    const long gtid = (long)blockIdx.x * kThreads + threadIdx.x;
    const long gstride = (long)gridDim.x * kThreads;
    int dropped = 0;
    if (kAligned) {
        long q0 = p0 + (p0 & 1);
        if (q0 > p1) q0 = p1;
        if ((p0 & 1) && gtid == 0 && p0 < p1)
            scatter_point(c, pts[3 * p0], pts[3 * p0 + 1], pts[3 * p0 + 2], want_tower && is_kept(labels[p0], keep),
                          dropped);
        const long npair = (p1 - q0) >> 1;
        const double2* src = reinterpret_cast<const double2*>(pts + 3 * q0);
        const double2* lsrc = want_tower ? reinterpret_cast<const double2*>(labels + q0) : nullptr;
        for (long i = gtid; i < npair; i += gstride) {
            double2 u = src[3 * i], v = src[3 * i + 1], w = src[3 * i + 2];
            bool t0 = false, t1 = false;
            if (want_tower) {
                double2 l = lsrc[i];
                t0 = is_kept(l.x, keep);
                t1 = is_kept(l.y, keep);
            }
            scatter_point(c, u.x, u.y, v.x, t0, dropped);
            scatter_point(c, v.y, w.x, w.y, t1, dropped);
        }
        if (((p1 - q0) & 1) && gtid == 0) {
            long i = p1 - 1;
            scatter_point(c, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], want_tower && is_kept(labels[i], keep),
                          dropped);
        }
    } else {
        for (long i = p0 + gtid; i < p1; i += gstride)
            scatter_point(c, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], want_tower && is_kept(labels[i], keep),
                          dropped);
    }
    if (dropped_out && dropped) atomicAdd(&dropped_out[b], dropped);
}

// ---------------------------------------------------------------- finalize
__global__ void colstats_init_kernel(int32_t* cs, int B, int ny) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * 2 * ny) cs[i] = (((i / ny) & 1) == 0) ? INT_MAX : 0;
}

// grid (S, B): block s takes rows r = s, s+S, ... of the [nz*nx, ny] count matrix of tile b;
// threads run along y (coalesced), kThreads/ny' row lanes deep.
__global__ __launch_bounds__(kThreads) void colstats_kernel(const int32_t* __restrict__ counts, int rows, int ny,
                                                            int32_t* __restrict__ cs) {
    const int b = blockIdx.y;
    const int32_t* c = counts + (size_t)b * rows * ny;
    for (int y0 = 0; y0 < ny; y0 += kThreads) {
        const int width = min(ny - y0, kThreads);
        const int depth = kThreads / width;  // rows processed in parallel by this block
        const int ty = threadIdx.x % width, tr = threadIdx.x / width;
        if (tr >= depth) continue;
        int mn = INT_MAX, mx = 0;
        for (int r = blockIdx.x * depth + tr; r < rows; r += gridDim.x * depth) {
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

__global__ __launch_bounds__(kThreads) void finalize_kernel(const int32_t* __restrict__ counts,
                                                            const int32_t* __restrict__ towers,
                                                            const int32_t* __restrict__ cs, size_t total, size_t V,
                                                            int ny, double* __restrict__ density,
                                                            double* __restrict__ gt, float* __restrict__ occ,
                                                            float* __restrict__ gt_occ) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
        const int b = (int)(i / V);
        const int y = (int)(i % ny);
        const int c = counts[i];
        if (density || occ) {
            // sklearn MinMaxScaler: scale = 1/range (range < 10 eps -> 1); X*scale + (0 - min*scale)
            const int mn = cs[(b * 2 + 0) * ny + y], mx = cs[(b * 2 + 1) * ny + y];
            double rng = (double)(mx - mn);
            if (rng < 10.0 * DBL_EPSILON) rng = 1.0;
            const double scale = __ddiv_rn(1.0, rng);
            const double min_ = __dsub_rn(0.0, __dmul_rn((double)mn, scale));
            const double v = __dadd_rn(__dmul_rn((double)c, scale), min_);
            if (density) density[i] = v;
            if (occ) occ[i] = (v > 0.0) ? 1.0f : 0.0f;
        }
        if (gt || gt_occ) {
            const int t = towers[i];
            const double r = (c > 0) ? __ddiv_rn((double)t, (double)c) : 0.0;
            if (gt) gt[i] = r;
            if (gt_occ) gt_occ[i] = (r > 0.0) ? 1.0f : 0.0f;
        }
    }
}

inline int blocks_per_tile(int B) {
    // enough workgroups to fill 256 CUs a few times over, whatever the batch
    int per = (2048 + B - 1) / B;
    return per < 1 ? 1 : (per > 256 ? 256 : per);
}

}  // namespace

extern "C" int sn_voxel_bbox(const double* pts, const int64_t* offsets, int B, double* bbox, sn_stream_t stream) {
    if (!pts || !offsets || !bbox) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_bbox: null pointer");
    if (B <= 0) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_bbox: B=%d", B);
    hipStream_t s = sn::as_stream(stream);
    auto* enc = reinterpret_cast<unsigned long long*>(bbox);
    hipLaunchKernelGGL(bbox_init_kernel, dim3((B * 6 + 255) / 256), dim3(256), 0, s, enc, B);
    dim3 grid(blocks_per_tile(B), B);
    if ((reinterpret_cast<uintptr_t>(pts) & 15) == 0)
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
    hipLaunchKernelGGL(desc_kernel, dim3(B), dim3(128), 0, sn::as_stream(stream), box, nx, ny, nz, regular,
                       from_bounds, desc);
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

extern "C" int sn_voxel_scatter(const double* pts, const double* labels, const int64_t* offsets, int B,
                                const double* desc, int nx, int ny, int nz, int32_t* counts, int32_t* tower_counts,
                                const double* keep_labels_host, int n_keep, int32_t* dropped, sn_stream_t stream) {
    if (!pts || !offsets || !desc || !counts) return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_scatter: null pointer");
    if (B <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_scatter: non-positive extent (B=%d n=%d,%d,%d)", B, nx, ny, nz);
    if (n_keep < 0 || n_keep > kMaxKeep)
        return sn::fail(SN_ERR_UNSUPPORTED, "sn_voxel_scatter: n_keep=%d outside [0,%d]", n_keep, kMaxKeep);
    if (tower_counts && (!labels || (n_keep > 0 && !keep_labels_host)))
        return sn::fail(SN_ERR_INVALID_ARG, "sn_voxel_scatter: tower_counts needs labels and keep_labels_host");
    const size_t lds = (size_t)(nx + ny + nz + 3) * sizeof(double);
    if (lds > 64 * 1024) return sn::fail(SN_ERR_UNSUPPORTED, "sn_voxel_scatter: edge table %zu B > 64 KiB", lds);
    hipStream_t s = sn::as_stream(stream);
    const size_t bytes = (size_t)B * nx * ny * nz * sizeof(int32_t);
    if (hipMemsetAsync(counts, 0, bytes, s) != hipSuccess) return sn::check_launch("sn_voxel_scatter(memset)");
    if (tower_counts && hipMemsetAsync(tower_counts, 0, bytes, s) != hipSuccess)
        return sn::check_launch("sn_voxel_scatter(memset)");
    if (dropped && hipMemsetAsync(dropped, 0, (size_t)B * sizeof(int32_t), s) != hipSuccess)
        return sn::check_launch("sn_voxel_scatter(memset)");
    KeepLabels keep;
    keep.n = n_keep;
    for (int i = 0; i < kMaxKeep; ++i) keep.v[i] = (i < n_keep) ? keep_labels_host[i] : 0.0;
    dim3 grid(blocks_per_tile(B), B);
    const bool aligned = ((reinterpret_cast<uintptr_t>(pts) & 15) == 0) &&
                         (!labels || (reinterpret_cast<uintptr_t>(labels) & 15) == 0);
    if (aligned)
        hipLaunchKernelGGL(scatter_kernel<true>, grid, dim3(kThreads), lds, s, pts, labels, offsets, desc, nx, ny, nz,
                           counts, tower_counts, keep, dropped);
    else
        hipLaunchKernelGGL(scatter_kernel<false>, grid, dim3(kThreads), lds, s, pts, labels, offsets, desc, nx, ny,
                           nz, counts, tower_counts, keep, dropped);
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
    hipStream_t s = sn::as_stream(stream);
    const size_t V = (size_t)nx * ny * nz;
    if (density || occ) {
        hipLaunchKernelGGL(colstats_init_kernel, dim3((B * 2 * ny + 255) / 256), dim3(256), 0, s, colstats, B, ny);
        const int rows = nz * nx;
        int S = (rows + 63) / 64;
        if (S > 64) S = 64;
        hipLaunchKernelGGL(colstats_kernel, dim3(S, B), dim3(kThreads), 0, s, counts, rows, ny, colstats);
    }
    const size_t total = (size_t)B * V;
    size_t blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, s, counts, tower_counts, colstats,
                       total, V, ny, density, gt, occ, gt_occ);
    return sn::check_launch("sn_voxel_finalize");
}
