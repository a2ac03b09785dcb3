/*
 * scenenet_hip.h -- C ABI of the MI355X-native SCENE-Net GENEO hot path (forward, and the training rows of SURVEY 8f).
 *
 * The reference (dlavado/scene-net) is pure Python and has no FFI layer; its
 * "operator API" for this path is a set of Python classes/functions.  Each
 * entry point below names the reference interface it replaces (file:line,
 * relative to the reference tree).  INTEGRATION.md shows the ctypes stub a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - plain `extern "C"`, no torch / C++ types in any signature;
 *   - every pointer is a caller-allocated DEVICE pointer (e.g. tensor.data_ptr())
 *     unless the parameter name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every call only enqueues work on it (no allocation, no sync, graph-capturable; the only process state is a
 *     cache of per-kernel LDS attributes and, with the opt-in "conv_skip_empty_tiles", a device ring of ticket
 *     counters allocated once);
 *   - return value: 0 = SN_OK, negative = sn_status; sn_last_error() gives the
 *     message of the calling thread's last failure;
 *   - grids are [B, C, Z, X, Y] row-major, y fastest (utils/voxelization.py:193);
 *     GENEO kernels are [G, kz, kx, ky] row-major.
 */
#ifndef SCENENET_HIP_H
#define SCENENET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sn_stream_t;

typedef enum {
    SN_OK = 0,
    SN_ERR_INVALID_ARG = -1,   /* null pointer, non-positive extent, bad enum        */
    SN_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels are built for       */
    SN_ERR_LAUNCH = -3,        /* hipGetLastError() after a launch                   */
    SN_ERR_NO_DEVICE = -4,     /* no gfx950 device / HIP runtime unusable            */
    SN_ERR_DEVICE_STATUS = -5  /* a kernel latched the sticky status: sn_device_status */
} sn_status;

/* SN_BF16: bfloat16 STORAGE of activations / their gradients in the training path (reference: `precision: 16`,
 * experiments/scenenet_ts40k/defaults_config.yml:83-84): accepted as out_dtype of sn_conv_fused, as pred / grad
 * dtype of sn_loss_forward / sn_loss_backward and as the gradient dtype of sn_conv_corr_t; every sum stays fp32 / fp64.
 * SN_OCC8: uint8 grid whose values are all 0 or 1 (binary occupancy, torch.bool) -- lets sn_conv_bank use the
 * int8 matrix cores.  SN_U8 is a general 0..255 byte grid. */
typedef enum { SN_F32 = 0, SN_F64 = 1, SN_U8 = 2, SN_OCC8 = 3, SN_BF16 = 4 } sn_dtype;

/* GENEO kinds: 0-2 = cylinderv2 / arrow / negSpherev2 of SceneNet (core/models/SCENE_Net.py:259-272);
 * 3-5 = cylinder_kernel / cone_kernel / neg_sphere_kernel of the v1 module SCENE_Net (SCENE_Net.py:158-170). */
typedef enum {
    SN_GENEO_CY = 0, SN_GENEO_CONE = 1, SN_GENEO_NEG = 2,
    SN_GENEO_CY_V1 = 3, SN_GENEO_CONE_V1 = 4, SN_GENEO_NEG_V1 = 5
} sn_geneo_kind;

/* Parameter slots of one GENEO: params[g * SN_NPARAM + slot] (fp32). */
enum {
    SN_P_RADIUS = 0,      /* cy, cone, neg : cylinder.py:55, arrow.py:78, neg_sphere.py:62 */
    SN_P_SIGMA = 1,       /* cy, cone, neg : default 1                                      */
    SN_P_APEX = 2,        /* cone          : truncated to int, arrow.py:235                 */
    SN_P_CONE_RADIUS = 3, /* cone          : arrow.py:84-87                                 */
    SN_P_CONE_INC = 4,    /* cone          : clamped to [0, 0.499], arrow.py:244            */
    SN_P_NEG_FACTOR = 5,  /* neg           : neg_sphere.py:63                               */
    SN_NPARAM = 8
};

int sn_version(void);
/* Allocates the current device's pool of flag / ticket words (20 KB) now -- the only allocation the library ever makes.
 * Optional: the first int8 launch on a device does it too, but that launch must then not be inside a stream capture
 * (hipMalloc / hipMemset are illegal there).  Launches made while their stream is capturing draw words that are never
 * handed out again, so a captured graph owns its flags (see csrc/cabi.hip). */
int sn_prepare_device(void);
/* The sticky device status (round 4).  A kernel that cannot keep its contract does not return plausible numbers: it
 * overwrites what it owns with NaN and latches a status in host-pinned words of its device --
 *   1  a dependency spin of the z-walk contraction gave up (sn_conv_bank_prepared[_served]);
 *   2  sn_conv_bank_prepared_served was called for a bank whose verdict is NOT "served" (a stale cached verdict);
 *   3  a hand-over spin of an int8 tile kernel gave up (sn_conv_bank, x_dtype SN_OCC8 / SN_U8)
 * -- and from then on EVERY sn_* entry that launches work fails with SN_ERR_DEVICE_STATUS (sn_last_error() says which code,
 * which workgroup) until sn_device_status_clear().  Looking at the words is a host memory read: no synchronisation, nothing
 * on the launch path.  sn_device_status: *code = 0 when healthy; detail2 (nullable) = {workgroup, ticket | verdict}.
 * sn_device_status_clear synchronises the device first.  The reference has no counterpart (its conv3d cannot fail this
 * way); this is the error behaviour of the hand-written path. */
int sn_device_status(int* code, int* detail2);
int sn_device_status_clear(void);
/* Diagnostics: how many sn_conv_bank launches (this device, this process) the folded int8 kernel served [0], declined
 * because the bank was not symmetric in x and y [1], and handed to the fp32 kernel because the quantisation bound
 * exceeded the tolerance [2].  Synchronises the device.  (No reference counterpart.) */
int sn_conv_i8_path_counts(unsigned long long* counts3);
const char* sn_last_error(void);

/* Number of gfx950 devices visible (0 when none); never throws. */
int sn_device_count(void);

/* Process-wide options.
 *   "conv_skip_empty_tiles" (default 0): sn_conv_bank on SN_OCC8 input skips the MFMA steps of workgroup tiles whose
 *       halo holds no set voxel (their response is exactly 0 for every kernel).  Output unchanged; run time becomes
 *       data dependent, so benchmarks quote it separately from the dense figure.
 *   "conv_i8_tolerance_ppb" (default 90000 = 9e-5): the int8 kernels (sn_conv_bank / sn_conv_fused / sn_forward_auto on
 *       binary occupancy) quantise the weights to 24-bit fixed point and compute, on the device, the exact worst case
 *       of the resulting activation error over all binary inputs.  A bank whose bound exceeds value * 1e-9 is computed
 *       by the fp32 kernel instead (decided on the device, no host synchronisation).  0 switches the guard off.
 *   "conv_i8_fold" (default 1): 9 x 9 x 9 banks that are bit-for-bit symmetric in x and y (every GENEO bank) are
 *       contracted over 9 x 5 x 5 folded taps (exactly the same integer sums, a third of the MFMAs); the symmetry is
 *       checked on the device at every call, other banks take the unfolded kernel; 0 = never try
 *   "conv_i8_legacy" (default 0): 1 = sn_conv_bank uses the four-copy int8 kernel (conv_i8.hip) for every shape
 *       instead of the stride-4 kernel (conv_i8s.hip) it prefers for ky = 9 (A/B timing, parity tests of both).
 *   "conv_i8z_variant" (default 2): the shape of the z-walk's tickets -- 0: rounds of two x-rows on 8 waves, 1: one round of
 *       one x-row per ticket on 12 waves, 2: two such rounds per ticket.  Same results bit for bit (tested on all three).
 *   "voxel_onepass" (default 1): sn_voxel_occupancy_fused[_bank] on grids whose bitmap(s) fit one workgroup's LDS (64^3)
 *       read the points ONCE -- bounding box, descriptor and binning in one launch whose workgroups exchange partial boxes;
 *       0 = the two-kernel form (box pass, then binning pass).  Same results bit for bit.
 *   "voxel_onepass_spin" (default 64): polls (~1 us each) a workgroup of that launch waits for its tile's other workgroups
 *       before it computes the tile's box from the points alone (same bits; 0 = never wait).
 *   "conv_i8z_inject_fault" (default 0): TEST HOOK.  1 = the next z-walk launches never report plane 0's first raw rows
 *       as landed, so a dependency spin gives up (~0.5 s per launch): the way to see the loud failure path -- NaN outputs
 *       and the sticky device status -- on the product build (tests/test_gpu_conv_zwalk.py). */
int sn_set_option(const char* name, int value);
int sn_get_option(const char* name);

/* ------------------------------------------------------------------------- *
 * K2  GENEO bank builder
 * replaces: GENEO_Layer.compute_kernel (core/models/SCENE_Net.py:103-106) over
 *           cylinderv2.compute_kernel (core/models/geneos/cylinder.py:162-176),
 *           arrow.compute_kernel      (core/models/geneos/arrow.py:228-252),
 *           negSpherev2.compute_kernel(core/models/geneos/neg_sphere.py:185-199),
 *           the v1 generators cylinder.py:84-103, arrow.py:173-205, neg_sphere.py:133-158,
 *           and the torch.stack at SCENE_Net.py:324 / :211.
 * params [G, SN_NPARAM] f32, kinds [G] i32 -> bank [G, kz, kx, ky] f32.
 * status (nullable) [G] i32: 0 ok, 1 = int(apex) outside [0, kz] (the reference
 * raises from torch.stack there; the kernel clamps and flags).
 * ------------------------------------------------------------------------- */
int sn_geneo_bank(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky,
                  float* bank, int32_t* status, sn_stream_t stream);

/* The effective convex coefficients of SceneNet.forward (core/models/SCENE_Net.py:329-335):
 * out[g] = lambdas[g] except out[last] = (1 - sum_i lambdas[order[i]]) + lambdas[last], the sum taken sequentially in
 * fp32 in the order given (nn.ParameterDict order: names sorted) -- bit for bit what the reference's python `sum`
 * yields.  lambdas [G] f32 is in/out: lambdas[last] is overwritten with out[last], the side effect of
 * SCENE_Net.py:333 (the frozen parameter is refreshed by every forward).  order [G] i32, out [G] f32. */
int sn_effective_lambdas(float* lambdas, const int32_t* order, int G, int last, float* out, sn_stream_t stream);
/* sn_geneo_bank and sn_effective_lambdas in one launch (the two openers of a training forward: both read the packed
 * parameter vector, neither depends on the other). */
int sn_geneo_bank_lambdas(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky, float* bank,
                          int32_t* status, float* lambdas, const int32_t* order, int last, float* lambdas_out,
                          sn_stream_t stream);
/* sn_geneo_bank_lambdas (lambdas nullable: then sn_geneo_bank) and sn_conv_bank_prep in ONE launch, for 9 x 9 x 9 kernels:
 * the workgroup that has just built kernel g prepares it for the int8 contraction from LDS (symmetry verdict, fixed-point
 * weights, error bound, digit-table entries; see sn_conv_bank_prep below).  prep: SN_CONV_PREP_BYTES x ceil(G / 16),
 * 16-byte aligned.  Neither the bank nor the blob depends on the voxel grid: the launch can run on a side stream next to
 * the voxelisation and be joined in front of sn_conv_bank_prepared. */
int sn_geneo_bank_prep(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky, float* bank,
                       int32_t* status, float* lambdas, const int32_t* order, int last, float* lambdas_out,
                       void* prep, sn_stream_t stream);

/* ------------------------------------------------------------------------- *
 * K3  GENEO bank convolution + convex-combination head
 * replaces: SceneNet.forward (core/models/SCENE_Net.py:322-339):
 *           F.conv3d(x, kernels, padding='same') (cross-correlation, zero pad
 *           left (k-1)/2, right k/2), sum_i lambda_i * conv_i, relu(tanh(.)).
 * x [B,1,Z,X,Y] of x_dtype; bank [G,kz,kx,ky] f32; lambdas [G] f32 = the
 * EFFECTIVE coefficients (last one already 1 - sum(others)), nullable iff out
 * is null.  act (nullable) [B,G,Z,X,Y] and out (nullable) [B,1,Z,X,Y] are of
 * out_dtype (SN_F32 or SN_F64).  x_dtype SN_F32 / SN_F64 / SN_U8: fp32 MFMA accumulation
 * (v_mfma_f32_16x16x4_f32).  x_dtype SN_OCC8 (values in {0,1}): weights as 24-bit fixed point in three
 * int8 digits, exact int32 accumulation on v_mfma_i32_16x16x64_i8, recombined in fp32.
 * Any G: one MFMA row block holds 16 kernels, larger banks run as ceil(G/16) launches with the raw partial sum
 * carried in `out` (the last launch applies relu(tanh)).
 * ------------------------------------------------------------------------- */
int sn_conv_bank(const void* x, int x_dtype, const float* bank, const float* lambdas,
                 int B, int Z, int X, int Y, int G, int kz, int kx, int ky,
                 void* act, void* out, int out_dtype, sn_stream_t stream);

/* The same contraction for a bank whose per-bank work was done once, ahead of the launch (round 3).
 * SceneNet.forward rebuilds its kernels from the parameters on every call (SCENE_Net.py:322-327); everything the int8
 * contraction derives from the bank alone -- the x/y symmetry verdict, the 24-bit fixed-point weights, the worst-case
 * quantisation error per kernel, the digit table of the folded operand plan -- is a function of those weights, so
 * sn_conv_bank_prep computes it in one small launch (16 workgroups; it can run on a side stream next to the
 * voxelisation) into `prep`: caller-owned device memory, 16-byte aligned, SN_CONV_PREP_BYTES per group of 16 kernels
 * (ceil(G / 16) groups).  sn_conv_bank_prepared is sn_conv_bank with that blob: SN_OCC8 input and a 9 x 9 x 9 bank run
 * the z-walk kernel (csrc/conv_i8z.inc: every input plane fetched and y-folded once per 8 x 64 column, no per-workgroup
 * prologue); the blob also holds the launch's route flag (bank not symmetric: the unfolded body runs in the same
 * launch; quantisation bound over the tolerance: the gated fp32 launch behind it takes over), so a captured graph owns
 * its flag.  One blob serves the launches of ONE stream at a time.  Results are bit-identical to sn_conv_bank's.
 * Every other dtype / shape: forwarded to sn_conv_bank (prep may be null). */
#define SN_CONV_PREP_BYTES 16384
int sn_conv_bank_prep(const float* bank, int G, int kz, int kx, int ky, void* prep, sn_stream_t stream);
int sn_conv_bank_prepared(const void* x, int x_dtype, const float* bank, const float* lambdas, void* prep,
                          int B, int Z, int X, int Y, int G, int kz, int kx, int ky,
                          void* act, void* out, int out_dtype, sn_stream_t stream);
/* sn_conv_bank_prepared without the fallback launch behind the walk.  The walk's verdict -- 0 served, 1 quantisation bound
 * over the tolerance (fp32 form), 2 bank not symmetric (unfolded body) -- is an int32 at byte offset
 * sn_conv_prep_verdict_offset() of each group's blob, written by every walk launch; it depends on the weights, the
 * coefficients, the tolerance and on which outputs are asked for, on nothing else.  A caller that has READ it as 0 for
 * exactly these (e.g. a serving loop whose parameters do not change: read it back once, asynchronously) may use this entry:
 * the ~3 us empty dispatch of the fallback launch disappears.  If the verdict is NOT 0 -- the caller's knowledge was stale --
 * the launch fills `out` / `act` with NaN and latches the sticky device status (code 2, sn_device_status): the call itself
 * has returned SN_OK by then (it is asynchronous), the NEXT sn_* call on the device fails with SN_ERR_DEVICE_STATUS.  It
 * never leaves the outputs unwritten. */
int sn_conv_bank_prepared_served(const void* x, int x_dtype, const float* bank, const float* lambdas, void* prep,
                                 int B, int Z, int X, int Y, int G, int kz, int kx, int ky,
                                 void* act, void* out, int out_dtype, sn_stream_t stream);
int sn_conv_prep_verdict_offset(void);
/* Diagnostics: how many waves of the int8 kernels ever gave up a bounded LDS hand-over spin (must stay 0).  A give-up is
 * not silent: it latches the sticky device status (code 1 / 3, above), and the z-walk overwrites its workgroup's outputs
 * with NaN.  Synchronises the device. */
int sn_conv_i8_spin_timeouts(unsigned long long* count);

/* Measurement hook: the NEXT z-walk launch of the calling thread (sn_conv_bank_prepared[_served] on a 9^3 bank) is made with
 * hipExtLaunchKernel(..., start_event, stop_event): the two hipEvent_t (created by the caller; either may be null) receive
 * the kernel's own start and stop timestamps -- what a profiler's kernel trace reports -- instead of the interval between two
 * hipEventRecord around the call, which includes the records' own ~2.5 us on the stream and the launch's dispatch.  One-shot:
 * consumed by that launch (or by a later one if this one takes another kernel).  bench.py times the dominant kernel with it.
 * The reference has no counterpart (its profiler is torch.profiler around core/models/SCENE_Net.py:322-339). */
int sn_launch_timing_events(void* start_event, void* stop_event);

/* The same forward output through linearity, without materialising the bank activations:
 *   relu(tanh(sum_g lambda_g conv3d(x, K_g))) == relu(tanh(conv3d(x, sum_g lambda_g K_g)))
 * (equal to 5e-16 in the reference's fp64; core/models/SCENE_Net.py:322-339).  One combined kernel
 * K* = sum_g lambda_g K_g (any G), 24-bit fixed point, Toeplitz-along-y implicit GEMM on v_mfma_i32_16x16x64_i8:
 * 0.375 MFMA per voxel at 9^3 (kernel rows packed at 24 K-bytes for ky <= 9; 0.48 otherwise) instead of 3.  x: SN_OCC8 only ([B,1,Z,X,Y], values in {0,1}, Y % 4 == 0, ky <= 17);
 * out [B,1,Z,X,Y] SN_F32 | SN_F64 | SN_BF16.  Returns SN_ERR_UNSUPPORTED for other inputs / shapes: call sn_conv_bank then.
 * Error vs the 16-kernel contraction: the fixed-point step of K* (<= 2^-23 max|K*| per tap) and fp32 rounding of
 * the combination, ~1e-6 on the output. */
int sn_conv_fused(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X, int Y,
                  int G, int kz, int kx, int ky, void* out, int out_dtype, sn_stream_t stream);
/* sn_conv_fused with the guard's verdict in a caller-owned device word (0: served by the combined int8 kernel, 1: its
 * quantisation bound exceeded the tolerance and the gated fp32 launches behind took over) -- the verdict depends on the
 * weights, the coefficients and the tolerance only, so a caller that has READ 0 for these very parameters may pass
 * assume_served = 1 and the (then empty) gated launches are left out, as sn_conv_bank_prepared_served does for the
 * contraction.  bf16 output runs unguarded (the verdict is then not written). */
int sn_conv_fused_v(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X, int Y, int G,
                    int kz, int kx, int ky, void* out, int out_dtype, int32_t* verdict, int assume_served,
                    sn_stream_t stream);
/* The per-(bank, coefficients) work of sn_conv_fused -- K* = sum_g lambda_g K_g, its 24-bit fixed point, the worst-case
 * error bound and the Toeplitz digit tables: ~8 us that every workgroup of the kernel otherwise spends for itself -- once,
 * into a caller-owned blob of sn_conv_fused_prep_bytes(kz, kx, ky) bytes (16-byte aligned; 0: kernel extent not served),
 * and the forward on that blob: same results bit for bit.  The blob carries the guard's verdict at the tolerance in force
 * when it was written (sn_conv_fused_v's word: byte offset bytes - 12); assume_served as there. */
size_t sn_conv_fused_prep_bytes(int kz, int kx, int ky);
int sn_conv_fused_prep(const float* bank, const float* lambdas, int G, int kz, int kx, int ky, void* blob,
                       sn_stream_t stream);
int sn_conv_fused_prepared(const void* x, int x_dtype, const float* bank, const float* lambdas, const void* blob, int B,
                           int Z, int X, int Y, int G, int kz, int kx, int ky, void* out, int out_dtype, int assume_served,
                           sn_stream_t stream);


/* 1 when sn_conv_fused serves a [B,1,Z,X,Y] SN_OCC8 grid with a [kz,kx,ky] kernel (Y % 4, the ky window, the tables and
 * the halo within 160 KiB of LDS), else 0: the caller's dispatch predicate, from the same plan the launch uses. */
int sn_conv_fused_supported(int B, int Z, int X, int Y, int kz, int kx, int ky);

/* SceneNet.forward for FLOAT grids that are usually binary occupancy -- what the reference itself feeds: f64 {0., 1.}
 * out of ToFullDense (core/datasets/torch_transforms.py:33-34).  One pass writes (x != 0) into occ_ws [B*Z*X*Y] bytes
 * and raises not_binary[0] when an element is neither 0 nor 1; then the int8 form (sn_conv_fused on the bytes, or
 * sn_conv_bank where that does not serve the shape) and the fp32 form (sn_conv_bank on x) of the same forward are
 * BOTH enqueued, each gated on that device flag: one runs, the other exits in its first instruction.  No host
 * synchronisation, same output contract as sn_conv_bank(out only).  x: SN_F32 | SN_F64, 16-byte aligned. */
int sn_forward_auto(const void* x, int x_dtype, const float* bank, const float* lambdas, int B, int Z, int X, int Y,
                    int G, int kz, int kx, int ky, uint8_t* occ_ws, int32_t* not_binary, void* out, int out_dtype,
                    sn_stream_t stream);

/* ------------------------------------------------------------------------- *
 * K1  point cloud -> voxel grid
 * replaces: pyntcloud VoxelGrid.compute as called by eda.voxelize_ply
 *           (utils/pcd_processing.py:341-372), hist_on_voxel
 *           (utils/voxelization.py:164-204), reg_on_voxel (:244-300),
 *           normalize_xyz (utils/pcd_processing.py:305-321) and
 *           ToFullDense.densify (core/datasets/torch_transforms.py:33-34).
 *
 * A batch is ragged: pts [total,3] f64 (x,y,z), labels [total] f64 (nullable),
 * offsets [B+1] i64 (CSR: tile b owns points offsets[b] .. offsets[b+1]-1).
 * ------------------------------------------------------------------------- */

/* doubles per tile in a grid descriptor: lo[3], hi[3], then the linspace edge
 * tables of x (nx+1), y (ny+1), z (nz+1). */
#define SN_DESC_LEN(nx, ny, nz) (6 + (nx) + (ny) + (nz) + 3)

/* Per-tile bounding box: bbox [B,6] f64 = (min x,y,z, max x,y,z). */
int sn_voxel_bbox(const double* pts, const int64_t* offsets, int B, double* bbox, sn_stream_t stream);

/* sn_voxel_bbox + sn_voxel_desc in two launches (no atomics, nothing to initialise) for the batch pipeline:
 * partial_ws is scratch [B, SN_BBOX_PARTS, 6] f64; bbox (nullable) receives the raw boxes. */
#define SN_BBOX_PARTS 32
int sn_voxel_prepare(const double* pts, const int64_t* offsets, int B, int nx, int ny, int nz, int regular,
                     double* partial_ws, double* bbox, double* desc, sn_stream_t stream);

/* Grid descriptor from a bounding box, n_x/n_y/n_z mode
 * (add_structure("voxelgrid", n_x, n_y, n_z), pcd_processing.py:362-363):
 * regular != 0 pads the box to a cube (pyntcloud regular_bounding_box=True);
 * edges are numpy.linspace(lo, hi, n+1) bit for bit.  desc [B, SN_DESC_LEN]. */
int sn_voxel_desc(const double* bbox, int B, int nx, int ny, int nz, int regular,
                  double* desc, sn_stream_t stream);

/* Same, from explicit per-tile bounds [B,6] (lo xyz, hi xyz) the caller has
 * already extended -- used for the size_x/size_y/size_z mode
 * (pcd_processing.py:365-367), whose grid extents are data dependent. */
int sn_voxel_desc_from_bounds(const double* bounds, int B, int nx, int ny, int nz,
                              double* desc, sn_stream_t stream);

/* Grid descriptor in the size_x/size_y/size_z mode for a whole batch, computed on the device
 * (replaces: pyntcloud VoxelGrid.compute with sizes, utils/pcd_processing.py:365-367, as used by
 * core/datasets/semKITTI.py:453-455): cube the box, extend every axis by ((range // size) + 1) * size - range, n =
 * int((max - min) / size) -- numpy's fp64 arithmetic bit for bit.  n is data dependent, so grids are allocated at a
 * caller-given maximum (nx, ny, nz) and desc [B, SN_DESC_LEN(nx,ny,nz)] carries each tile's own edge tables: edges
 * 0..n_a as numpy.linspace, +inf beyond.  sn_voxel_scatter / sn_gather_points take this descriptor unchanged;
 * sn_voxel_finalize_sized keeps the voxels beyond a tile's own dims out of the column statistics and zero.
 *   size_xyz_host  3 doubles on the host (voxel size per axis, > 0)
 *   dims           (nullable) [B,3] i32 out: n_x, n_y, n_z of every tile
 *   status         (nullable) [B] i32 out: 1 = the tile needs more voxels than the maximum (points beyond the table
 *                  are dropped and counted by the scatter) */
int sn_voxel_desc_sized(const double* bbox, int B, const double* size_xyz_host, int nx, int ny, int nz,
                        double* desc, int32_t* dims, int32_t* status, sn_stream_t stream);

/* sn_voxel_finalize for grids voxelised with a size-mode descriptor (normalize_xyz / ToFullDense over each tile's
 * own [n_z, n_x, n_y] part of the padded grid; the rest is 0). */
int sn_voxel_finalize_sized(const int32_t* counts, const int32_t* tower_counts, int B, int nx, int ny, int nz,
                            const double* desc, int32_t* colstats, double* density, double* gt, float* occ,
                            float* gt_occ, sn_stream_t stream);

/* Atomic scatter: counts[b,z,x,y] += 1 per point; tower_counts (nullable) += 1
 * per point whose label equals one of keep_labels_host[0..n_keep) (<= 16).
 * Both grids [B,nz,nx,ny] i32 are zeroed by the call.
 * dropped (nullable) [B] i32: points that fall outside the edge table. */
int sn_voxel_scatter(const double* pts, const double* labels, const int64_t* offsets, int B,
                     const double* desc, int nx, int ny, int nz,
                     int32_t* counts, int32_t* tower_counts,
                     const double* keep_labels_host, int n_keep,
                     int32_t* dropped, sn_stream_t stream);

/* counts -> outputs (each nullable):
 *   density [B,1,nz,nx,ny] f64 : hist_on_voxel's per-y-column min-max normalised counts
 *   gt      [B,1,nz,nx,ny] f64 : reg_on_voxel's tower/total ratio (needs tower_counts)
 *   occ     [B,1,nz,nx,ny] f32 : ToFullDense(density)  == (density > 0)
 *   gt_occ  [B,1,nz,nx,ny] f32 : ToFullDense(gt)       == (tower > 0)
 * colstats: workspace [B, 2, ny] i32. */
int sn_voxel_finalize(const int32_t* counts, const int32_t* tower_counts, int B, int nx, int ny, int nz,
                      int32_t* colstats, double* density, double* gt, float* occ, float* gt_occ,
                      sn_stream_t stream);

/* Occupancy-only form of the same step, for the fused pipeline: what SceneNet consumes is
 * ToFullDense(Voxelization(points)) (scripts/main.py:138-140), one bit per voxel.  Workgroups build the tile's
 * bitmap in LDS (no global atomics) and the result is expanded to occ / gt_occ [B,1,nz,nx,ny] of out_dtype
 * (SN_U8 or SN_F32); gt_occ nullable.  Needs nx*ny*nz % 32 == 0 and the bitmap(s) of one of <= SN_OCC_PARTS
 * z-slabs to fit 64 KiB of LDS (64^3: one slab; 128^3: 4 slabs, 8 with gt_occ); otherwise SN_ERR_UNSUPPORTED ->
 * use sn_voxel_scatter + sn_voxel_finalize.
 *   bits_ws     scratch, SN_OCC_WS_WORDS(B, nx*ny*nz, planes) uint32 (planes = 2 with gt_occ, else 1)
 *   flags       (nullable) [B] i32 out: 1 = the tile could not be proven free of a fully occupied y column
 *               (where ToFullDense(density) != (count > 0)); such tiles are recomputed exactly (counts by
 *               global atomics, column minima) by one gated launch when counts_ws (and towers_ws with
 *               gt_occ) is given: counts_ws, towers_ws [B,nz,nx,ny] i32 scratch (nullable).
 *   dropped     (nullable) [B] i32: points outside the edge table. */
#define SN_OCC_PARTS 16
#define SN_OCC_WS_WORDS(B, V, planes) ((size_t)(B) * SN_OCC_PARTS * ((planes) * ((V) / 32) + 1))
int sn_voxel_occupancy(const double* pts, const double* labels, const int64_t* offsets, int B,
                       const double* desc, int nx, int ny, int nz,
                       const double* keep_labels_host, int n_keep,
                       uint32_t* bits_ws, void* occ, void* gt_occ, int out_dtype,
                       int32_t* flags, int32_t* dropped,
                       int32_t* counts_ws, int32_t* towers_ws, sn_stream_t stream);

/* sn_voxel_prepare + sn_voxel_occupancy in four launches instead of five: the binning kernel derives the tile's
 * descriptor (cube regularisation, numpy.linspace edges: the same instruction sequence, the same bits) from the bbox
 * partials itself and publishes it in `desc` [B, SN_DESC_LEN] (and the raw box in `bbox` [B,6], nullable) for the later
 * kernels and callers.  partial_ws: scratch [B, SN_BBOX_PARTS, 6] f64.  Everything else as sn_voxel_occupancy. */
int sn_voxel_occupancy_fused(const double* pts, const double* labels, const int64_t* offsets, int B, int nx, int ny,
                             int nz, int regular, const double* keep_labels_host, int n_keep, double* partial_ws,
                             double* bbox, double* desc, uint32_t* bits_ws, void* occ, void* gt_occ, int out_dtype,
                             int32_t* flags, int32_t* dropped, int32_t* counts_ws, int32_t* towers_ws,
                             sn_stream_t stream);

/* sn_voxel_occupancy_fused with K2 riding in its first launch: the workgroups that build the 9 x 9 x 9 GENEO bank and the
 * int8 contraction's preparation blob (exactly sn_geneo_bank_prep's: the same arguments from `params` to `prep`, the same
 * bits; with `lambdas` the effective coefficients ride too, as in a training forward's opener sn_geneo_bank_lambdas) are extra grid rows of the bounding-box kernel, so the inference step has no launch, no stream fork and no
 * event for K2 at all ([measured] the forked form left 10.8 of K2's 12.4 serial microseconds on the critical path).
 * Replaces, for the hot path, GENEO_Layer.compute_kernel x G (SCENE_Net.py:103-106, 322-324) next to
 * Voxelization.__call__ (torch_transforms.py:74-81). */
int sn_voxel_occupancy_fused_bank(const double* pts, const double* labels, const int64_t* offsets, int B, int nx, int ny,
                                  int nz, int regular, const double* keep_labels_host, int n_keep, double* partial_ws,
                                  double* bbox, double* desc, uint32_t* bits_ws, void* occ, void* gt_occ, int out_dtype,
                                  int32_t* flags, int32_t* dropped, int32_t* counts_ws, int32_t* towers_ws,
                                  const float* params, const int32_t* kinds, int G, int kz, int kx, int ky, float* bank,
                                  int32_t* status, float* lambdas, const int32_t* order, int last, float* lambdas_out,
                                  void* prep, sn_stream_t stream);

/* sn_voxel_occupancy_fused in VOXEL-SIZE mode (voxelize_ply with size_x / size_y / size_z, utils/pcd_processing.py:365-367;
 * the mode SemanticKITTI uses, core/datasets/semKITTI.py:453-455): every tile's grid extents follow from its own bounding
 * box -- computed on the device as in sn_voxel_desc_sized, no host round trip -- and the binary grids are written padded
 * to the maximum (nx, ny, nz): dims [B,3] i32 = each tile's own (n_x, n_y, n_z), status [B] i32 (nullable) = 1 where a
 * tile needs more than the maximum (its points beyond are dropped), voxels beyond a tile's own dims are 0.  Same
 * LDS-bitmap kernels (no global atomic per point), same workspaces and outputs as sn_voxel_occupancy_fused; the
 * "count > column minimum" rule of ToFullDense(normalize_xyz(.)) is evaluated over each tile's OWN part of the grid. */
int sn_voxel_occupancy_sized(const double* pts, const double* labels, const int64_t* offsets, int B,
                             const double* size_xyz_host, int nx, int ny, int nz,
                             const double* keep_labels_host, int n_keep, double* partial_ws, double* bbox,
                             double* desc, int32_t* dims, int32_t* status, uint32_t* bits_ws, void* occ,
                             void* gt_occ, int out_dtype, int32_t* flags, int32_t* dropped,
                             int32_t* counts_ws, int32_t* towers_ws, sn_stream_t stream);
/* sn_voxel_occupancy_sized with the same riders as sn_voxel_occupancy_fused_bank (C4's chain: voxel-size mode -> conv). */
int sn_voxel_occupancy_sized_bank(const double* pts, const double* labels, const int64_t* offsets, int B,
                                  const double* size_xyz_host, int nx, int ny, int nz, const double* keep_labels_host,
                                  int n_keep, double* partial_ws, double* bbox, double* desc, int32_t* dims, int32_t* status,
                                  uint32_t* bits_ws, void* occ, void* gt_occ, int out_dtype, int32_t* flags,
                                  int32_t* dropped, int32_t* counts_ws, int32_t* towers_ws, const float* params,
                                  const int32_t* kinds, int G, int kz, int kx, int ky, float* bank, int32_t* bank_status,
                                  float* lambdas, const int32_t* order, int last, float* lambdas_out, void* prep,
                                  sn_stream_t stream);


/* Grid -> points: out[c, i] = grid[b(i), c, vz, vx, vy] for every point i of the batch, binned exactly as the
 * scatter binned it (same desc); points outside the edge table get `fill`.  grid [B,channels,nz,nx,ny] and
 * out [channels, total] are of `dtype` (SN_F32 | SN_F64).  The reference only has the voxel-list direction
 * (vxg_to_xyz, utils/voxelization.py:328-360) and prob_to_label (:304-323); per-point read-back of the
 * prediction (BASELINE config 4) is defined here. */
int sn_gather_points(const void* grid, int dtype, int channels, const double* pts, const int64_t* offsets, int B,
                     const double* desc, int nx, int ny, int nz, double fill, void* out, sn_stream_t stream);

/* Grid -> voxel list (vxg_to_xyz, utils/voxelization.py:328-360): EVERY cell of the [n0,n1,n2] grid, in C order
 * of its indices (np.indices(shape).reshape(3,-1).T), as a row  out[n] = (origin + index * voxel_size, grid[index]),
 * fp64 like the reference's np.concatenate result.  origin / voxel_size: 3 host doubles each, null = (0,0,0) /
 * (1,1,1) (the reference defaults).  grid of `dtype` (SN_F32 | SN_F64 | SN_U8 | SN_OCC8); out [n0*n1*n2, 4] f64,
 * 16-byte aligned.  The product is formed and then added (two roundings, as numpy does). */
int sn_grid_to_points(const void* grid, int dtype, int n0, int n1, int n2, const double* origin_host,
                      const double* voxel_size_host, double* out, sn_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Backward (training config; reference: autograd through SceneNet.forward, SCENE_Net.py:322-339, and the
 * generator graphs cylinder.py / arrow.py / neg_sphere.py).  By linearity the whole backward needs ONE
 * correlation C[t] = sum_{b,v} delta[b,v] x[b, v + t - p]:  dL/dK_g = lambda_g C,  dL/dlambda_g = <K_g, C>.
 * ------------------------------------------------------------------------- */

/* C [kz,kx,ky] f32.  gout = dL/dout [B,1,Z,X,Y] f32; when `out` (the forward output, f32) is given,
 * delta = gout * (out > 0) * (1 - out^2) (the relu(tanh) derivative) else delta = gout.  x as in sn_conv_bank.
 * partial_ws: scratch [sn_conv_corr_blocks(B,Z,X,Y), kz*kx*ky] f32; the block partials are summed in a fixed
 * order (bit-reproducible). */
int sn_conv_corr(const void* x, int x_dtype, const float* gout, const float* out, int B, int Z, int X, int Y,
                 int kz, int kx, int ky, float* partial_ws, float* C, sn_stream_t stream);
int sn_conv_corr_blocks(int B, int Z, int X, int Y);
/* The same with gout / out of g_dtype SN_F32 or SN_BF16 (bf16 activations: half the bytes of the two grids this pass
 * reads; the products are formed and summed in fp32 either way). */
int sn_conv_corr_t(const void* x, int x_dtype, const void* gout, const void* out, int g_dtype, int B, int Z, int X, int Y,
                   int kz, int kx, int ky, float* partial_ws, float* C, sn_stream_t stream);

/* The same behind ONE caller-owned workspace of sn_conv_corr_ws_bytes(...) bytes (256-byte aligned), and the entry point
 * that serves binary occupancy (x_dtype SN_OCC8) as a GATHER over the set voxels (K4s, csrc/corr.hip: a list kernel +
 * a gather kernel; no matrix core; nnz x kz kx ky terms instead of V x kz x 256 products): the workspace then also holds
 * the voxel lists.  Any other input, option "corr_dense" = 1, or a workspace with room for the partial rows only
 * (>= sn_conv_corr_blocks x kz kx ky floats) takes sn_conv_corr_t's kernels.  Both forms sum in a fixed order
 * (bit-reproducible per form; the two forms differ by fp32 summation order).  Option "corr_sparse_tile_bytes"
 * (0 = 2048): input bytes per gather job. */
size_t sn_conv_corr_ws_bytes(int x_dtype, int B, int Z, int X, int Y, int kz, int kx, int ky);
int sn_conv_corr_ws(const void* x, int x_dtype, const void* gout, const void* out, int g_dtype, int B, int Z, int X,
                    int Y, int kz, int kx, int ky, void* ws, size_t ws_bytes, float* C, sn_stream_t stream);

/* Generator Jacobians: dparams [G, SN_NPARAM] f32 = d<dW, bank(params)>/dparams for dW [G,kz,kx,ky] f32
 * (apex has no gradient: it is truncated to an index, arrow.py:235, and non-trainable, arrow.py:134). */
int sn_geneo_bank_bwd(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky,
                      const float* dW, float* dparams, sn_stream_t stream);

/* The parameter side of SceneNet's backward in one launch, from the correlation C of sn_conv_corr (by linearity
 * dL/dK_g = lambda_g C): dparams [G, SN_NPARAM] as sn_geneo_bank_bwd would give for dW_g = lambda_g C, and
 * dlambdas [G] = <K_g, C> - <K_last, C> -- the gradient of the trainable convex coefficients when coefficient `last`
 * is 1 - sum(others) (SCENE_Net.py:331; entry `last` comes out 0).  bank [G,kz,kx,ky] and lambdas [G] (effective
 * coefficients) are the forward's. */
int sn_geneo_backward(const float* params, const int32_t* kinds, int G, int kz, int kx, int ky, const float* bank,
                      const float* lambdas, const float* corr, int last, float* dparams, float* dlambdas,
                      sn_stream_t stream);

/* ------------------------------------------------------------------------- *
 * K5  training criterion on the prediction grid
 * replaces: WeightedMSE.forward + get_weight_target/get_dens_target (core/criterions/w_mse.py:114-151),
 *           FocalTverskyLoss.forward / TverskyLoss.forward (core/criterions/tversky_loss.py:81-95, :35-51),
 *           BinaryDiceLoss.forward, p = 2, reduction 'mean' (core/criterions/dice_loss.py:33-51),
 *           summed as in GENEO_Loss / GENEO_Dice_Loss / GENEO_Tversky_Loss.forward (geneo_loss.py:72-81, :131-161).
 * The scalar penalties over the ~50 parameters (cvx_loss, positive_regularizer, geneo_loss.py:36-70) stay on the
 * host side.
 *
 * Everything the criterion needs from the B*n_per elements is a handful of sums, taken in ONE pass over
 * (pred, gt): per weight bin k (bin = argmin_k |gt - ranges[k]|, first minimum, w_mse.py:122) the element count and
 * sum (gt - pred)^2; per sample sum p*t, sum p, sum t, sum p^2, sum t^2.  fp64 accumulation, fixed summation order
 * (bit-reproducible).  The weights' mean (w_mse.py:144) is sum_k cnt_k w_k / n.
 * ------------------------------------------------------------------------- */
#define SN_LOSS_MAX_BINS 16
#define SN_LOSS_NSTAT(H) (3 * (H) + 5)   /* cnt[H], sq_err[H], sum_pt, sum_p, sum_t, sum_pp, sum_tt, bce[H] */
#define SN_LOSS_PARTS(n_per) ((n_per) <= 16384 ? 1 : ((n_per) >= 16384 * 256 ? 256 : (int)(((n_per) + 16383) / 16384)))
#define SN_LOSS_WS_DOUBLES(B, n_per, H) ((int64_t)(B) * SN_LOSS_PARTS(n_per) * SN_LOSS_NSTAT(H))
#define SN_LOSS_NCOEF(B) (2 * SN_LOSS_MAX_BINS + 3 * (B))
typedef enum { SN_LOSS_WMSE = 1, SN_LOSS_FOCAL_TVERSKY = 2, SN_LOSS_DICE = 4, SN_LOSS_WBCE = 8 } sn_loss_term;

/* Forward.  pred [B, n_per] (SN_F32 | SN_F64), gt [B, n_per] (SN_F32 | SN_F64 | SN_U8 | SN_OCC8),
 * ranges [H] f32 (left bin edges, w_mse.py:100), bin_w [H] f32 = max(1 - alpha*dens_k, eps) per bin BEFORE the
 * division by the mean (w_mse.py:128-142; ~10 numbers the host derives from the frequency table once).
 * terms = OR of sn_loss_term (SN_LOSS_WBCE = mean(weights * BCELoss(pred, gt)) of BinaryDiceLoss_BCE,
 * core/criterions/dice_loss.py:71-80, with torch's clamps: log >= -100, gradient denominator >= 1e-12).
 * Outputs: stats [B, SN_LOSS_NSTAT(H)] f64; loss [5] f64 = {sum of the requested terms, weighted MSE, focal Tversky,
 * dice, weighted BCE}; coef [SN_LOSS_NCOEF(B)] f64 for sn_loss_backward.
 * parts_ws: scratch [SN_LOSS_WS_DOUBLES(B, n_per, H)] f64. */
int sn_loss_forward(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                    const float* ranges, const float* bin_w, int H, int terms, double mse_weight,
                    double tversky_alpha, double tversky_beta, double focal_gamma, double tversky_smooth,
                    double dice_smooth, double* parts_ws, double* stats, double* loss, double* coef,
                    sn_stream_t stream);

/* The penalties of GENEO_Loss over the model's scalars (core/criterions/geneo_loss.py:36-70) in one launch:
 * value[0] = weight * ( sum_{mask[i] >= 1} relu(-P[i]) + [with_sum] relu(-(1 - sum_{mask[i] == 2} P[i])) ),
 * grad[i] = d value / d P[i].  P [N] f32 (the packed parameter vector), mask [N] i8: 0 = not a parameter,
 * 1 = GENEO parameter (positive_regularizer), 2 = trainable convex coefficient (cvx_loss: its own relu(-phi) and the
 * relu(-(1 - sum)) of the frozen last one). */
int sn_param_penalty(const float* P, const int8_t* mask, int N, float weight, int with_sum, float* value, float* grad,
                     sn_stream_t stream);

/* Backward: grad_pred[b,i] = up * (c[bin(gt)] (p - t) + e[bin(gt)] (p - t) / max((1 - p) p, 1e-12) + A_b t + B_b + C_b p),
 * in pred's dtype (c, e, A, B, C from coef).
 * upstream: device scalar f64 (dL/dloss), NULL = 1. */
int sn_loss_backward(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                     const float* ranges, int H, const double* coef, const double* upstream, void* grad_pred,
                     sn_stream_t stream);
/* The same two with what a float32 criterion needs to stay free of cast launches (a training step replayed from a hipGraph
 * pays ~4 us of GPU time for every one-element kernel): sn_loss_forward_m also writes the five losses rounded to float32
 * (loss_f32 [5], nullable); sn_loss_backward_u takes the upstream scalar as SN_F64 or SN_F32. */
int sn_loss_forward_m(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                      const float* ranges, const float* bin_w, int H, int terms, double mse_weight, double tversky_alpha,
                      double tversky_beta, double focal_gamma, double tversky_smooth, double dice_smooth, double* parts_ws,
                      double* stats, double* loss, float* loss_f32, double* coef, sn_stream_t stream);
int sn_loss_backward_u(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                       const float* ranges, int H, const double* coef, const void* upstream, int up_dtype, void* grad_pred,
                       sn_stream_t stream);
/* The criterion of a GENEO_Loss family member (dense terms + cvx_loss + positive_regularizer, geneo_loss.py:36-91, 145-161)
 * in the launches of sn_loss_forward / sn_loss_backward alone: sn_criterion_forward = sn_loss_forward_m with
 * sn_param_penalty's work opening the combine launch, and total_f32 [1] = (float)loss[0] + pen_value -- the scalar the
 * criterion returns; sn_criterion_backward = sn_loss_backward_u with pen_out [N] = pen_grad [N] x upstream riding in the
 * gradient launch.  Same numbers, bit for bit, as the three forward launches + the torch add / multiply they replace. */
int sn_criterion_forward(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                         const float* ranges, const float* bin_w, int H, int terms, double mse_weight, double tversky_alpha,
                         double tversky_beta, double focal_gamma, double tversky_smooth, double dice_smooth,
                         double* parts_ws, double* stats, double* loss, float* loss_f32, double* coef, const float* P,
                         const int8_t* mask, int N, float weight, int with_sum, float* pen_value, float* pen_grad,
                         float* total_f32, sn_stream_t stream);
int sn_criterion_backward(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int B, int64_t n_per,
                          const float* ranges, int H, const double* coef, const void* upstream, int up_dtype,
                          void* grad_pred, const float* pen_grad, int N, float* pen_out, sn_stream_t stream);



#ifdef __cplusplus
}
#endif
#endif /* SCENENET_HIP_H */
