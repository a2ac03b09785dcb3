#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline: voxel-tiles/s of the SCENE-Net GENEO forward hot path on MI355X.

One step = one pass of the hot path over one HBM-resident batch of synthetic tiles:
    bbox -> (edge tables, LDS-bitmap scatter) -> finalize (occupancy) [K1, csrc/voxel.hip]
    -> GENEO bank build                                                [K2, csrc/bank.hip]
    -> bank conv on MFMA + fused convex head                           [K3, csrc/conv.hip]
Workload = BASELINE configs[1] ("C2"): 32 tiles x 100k points, 64^3 grid, 16 GENEO kernels of 9^3, per GPU.
Multi-GPU: one process per GPU (torchrun), tiles sharded per rank, NO data-path collective; the barrier and the
max-over-ranks of the timed region are the only collectives (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (K3) with its duration measured live with
HIP events on the launch stream; `cpu_baseline` is the oracle (CPU restatement of the reference path, the only
place this file touches oracle/) timed on the host cores on a bounded sample.
"""
import argparse
import gc
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _load_launcher():
    """scene-net_amd/launch.py by path: stdlib only, so the launching parent never imports torch or the HIP library
    (children must be started by a process that has not initialised the device)."""
    spec = importlib.util.spec_from_file_location("_sn_launch", os.path.join(ROOT, "scene-net_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_I8_MFMA_TOPS = 5000.0     # MI355X_MICROARCH.md: i8 MFMA = 2x the bf16 rate per clock, bf16 ~2.5 PF dense
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
MEASURED_I8_MFMA_TOPS = 4050.0  # tools/micro/mfma_i8_peak.hip: a loop of nothing but independent i8 MFMAs on RANDOM operands
                                # sustains 4.0-4.06 POP/s once the clocks have settled (3.5-3.7 started cold; 4.7-4.8 on
                                # constant operands: the chip clocks down with operand toggling)
GENEO_NUM = {"cy": 6, "cone": 5, "neg": 5}
KERNEL_SIZE = (9, 9, 9)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU per step")
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--spinup-ms", type=float, default=200.0,
                    help="untimed spin-up of the same step before the warm-up steps (device clocks settle)")
    ap.add_argument("--event-every", type=int, default=0,
                    help="record the HIP events that time the stages on every N-th timed step; 0 (default) = "
                         "max(5, steps / 8) (1 below 8 steps): at least four timed launches at the default 20 steps, "
                         "eight at 64 and more.  [measured] a record costs ~2.5 us of GPU time plus host work next to "
                         "a queue that is barely ahead: 187.3 / 159.1 us per step with the records on every 4th / "
                         "every 50th of 200 steps")
    ap.add_argument("--sustain-ms", type=float, default=6000.0,
                    help="also time the same eager step back to back over at least this much wall time "
                         "(`value_sustained`; 0 = skip).  6 s by default: an external sampler with a ~5 s period (the "
                         "driver's gpu_busy probe, rocm-smi) then has at least one sample inside the loop; the loop also "
                         "samples the device's own gpu_busy_percent (sysfs) every 100 ms and reports what it saw")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 code path on a one-GPU box: every rank uses cuda:0, process group over gloo (not a "
                         "measurement)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the graph-replay and skip-empty extras (profiling runs: keeps per-kernel averages clean)")
    return ap.parse_args()


def launch_if_needed(args):
    """`python bench.py --gpus N` with N > 1 and no launcher above it (torchrun sets RANK/WORLD_SIZE): start the N
    ranks ourselves -- fresh children, one per GPU, before this process has made any GPU call -- and exit with their
    verdict.  Under torchrun (the driver's N > 1 form) this is a no-op."""
    launch = _load_launcher()
    if args.gpus <= 1 or launch.under_launcher():
        return
    rc = launch.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    sys.exit(rc)


def host_cores():
    """Threads the CPU baseline may use: the process's affinity / cgroup quota, capped at the GPU box's per-GPU
    CPU share (16) -- oversubscribing a shared host would only slow the baseline down."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get("BENCH_CPU_THREADS", min(n, 16)))


def cpu_baseline(args, specs, names, lambdas, last, budget_s=20.0):
    """The reference path restated on the CPU (oracle/), timed on this host's cores on a bounded sample of the
    same workload.  fp64 conv3d is what the reference runs (SCENE_Net.py:105,325); the fp32 variant is reported
    beside it because ATen's fp64 conv3d is far off the host's own roofline."""
    from oracle import geneo_oracle as go  # cpu_baseline leg only
    from oracle import voxel_oracle as vo
    cores = host_cores()
    torch.set_num_threads(cores)
    dims = (args.grid,) * 3
    done, t_vox, t_conv64, t_conv32 = 0, 0.0, 0.0, 0.0
    t_start = time.perf_counter()
    for t in range(64):
        xyz, labels = synthetic_tile(10_000 + t, args.points)
        t0 = time.perf_counter()
        counts, _, _ = vo.voxel_counts(xyz, dims)
        occ = vo.to_full_dense(vo.normalize_xyz(counts.astype(np.float64)))
        t1 = time.perf_counter()
        x = torch.from_numpy(occ)[None, None]
        with torch.no_grad():
            go.scenenet_forward(x, specs, KERNEL_SIZE, lambdas, last, names=names)
            t2 = time.perf_counter()
            go.scenenet_forward(x.float(), specs, KERNEL_SIZE, lambdas, last, names=names)
            t3 = time.perf_counter()
        if t == 0:
            continue  # warm-up tile
        done += 1
        t_vox += t1 - t0
        t_conv64 += t2 - t1
        t_conv32 += t3 - t2
        if time.perf_counter() - t_start > budget_s:
            break
    return {
        "value": done / (t_vox + t_conv64), "unit": "tiles/s", "cores": cores, "kind": "port",
        "sample": f"{done} tiles of the same workload (after 1 warm-up): numpy voxelisation + torch fp64 conv3d + head, "
                  f"{cores} threads",
        "voxel_points_per_s": done * args.points / t_vox,
        "fp32_conv_value": done / (t_vox + t_conv32),
    }


def _late_imports():
    """torch / numpy / the package, imported only in a process that is a rank (never in the launching parent)."""
    global np, torch, sna, job_sum, job_time_max, apply_bank_spec, synthetic_bank_spec, synthetic_tile
    import numpy as np
    import torch
    import scene_net_amd as sna
    from scene_net_amd.pipeline import job_sum, job_time_max
    from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile


class BusySampler:
    """gpu_busy_percent of this rank's device (amdgpu sysfs), sampled every 100 ms on a thread while the sustained loop runs:
    corroboration from OUTSIDE the HIP event / host clock pair every other number here rests on."""

    def __init__(self, index):
        import glob
        import threading
        self.paths = sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))
        self.index, self.samples, self._stop = index, [], threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.wait(0.1):
            vals = []
            for p in self.paths:
                try:
                    vals.append(int(open(p).read().strip()))
                except (OSError, ValueError):
                    pass
            if vals:
                self.samples.append(max(vals))   # (one GPU per box here; the busiest card otherwise)

    def start(self):
        if self.paths:
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self.paths and self._thread.is_alive():
            self._thread.join(timeout=1.0)
        if not self.samples:
            return {"samples": 0, "note": "no readable gpu_busy_percent in sysfs"}
        s_ = self.samples
        return {"samples": len(s_), "mean": sum(s_) / len(s_), "min": min(s_), "max": max(s_),
                "frac_samples_busy_ge_90": sum(1 for v in s_ if v >= 90) / len(s_)}


REFERENCE_CHECKPOINT = {   # experiments/scenenet_ts40k/.../checkpoints/FBetaScore.ckpt, the 13 trained scalars
    "cy_0": dict(radius=0.998896, sigma=1.199054),
    "cone_0": dict(apex=0.0, cone_inc=0.565547, cone_radius=4.000988, radius=1.5, sigma=0.955910),
    "neg_0": dict(neg_factor=0.127053, radius=3.000918, sigma=0.605097),
    "lambdas": {"cy_0": 0.024178, "cone_0": 0.608911, "neg_0": 0.366911},
}


def reference_defaults_extra(sna, dev, ev, spin, args, B=64):
    """ms per call and roofline fractions of the module as the reference configures it: SceneNet({'cy':1,'cone':1,'neg':1},
    (9,5,5)) with the checkpoint's scalars on 64 resident tiles of 64^3 -- `model(x)` (the head: what LitSceneNet.forward
    returns) and `model(x, return_bank_activations=True)`; plus the whole step from points."""
    import torch
    geneo_num, ks = {"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)
    model = sna.SceneNet(geneo_num, ks)
    with torch.no_grad():
        for name in ("cy_0", "cone_0", "neg_0"):
            for k, v in REFERENCE_CHECKPOINT[name].items():
                model.geneos[name].geneo_params[k].fill_(v)
            model.lambdas_dict[f"lambda_{name}"].fill_(REFERENCE_CHECKPOINT["lambdas"][name])
    model.last_lambda = "lambda_cy_0"
    model = model.to(dev)
    tiles, labels = zip(*[synthetic_tile(50_000 + i, args.points) for i in range(B)])
    batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
    pipe = sna.ScenePipeline(model, (64,) * 3)
    del tiles, labels
    V, G, ntaps = 64 ** 3, 3, 9 * 5 * 5
    flops = 2.0 * V * ntaps * G * B
    out = {"config": f"{B} tiles x {args.points} points, 64^3 grid, kernel (9,5,5), 3 GENEOs (cy 1, cone 1, neg 1), the "
                     "checkpoint's 13 scalars; defaults_config.yml:16-19,33-40", "flops_per_call_dense": flops}
    with torch.no_grad():
        occ = pipe.voxelize(batch).occ
        calls = {"head_only": lambda: model(occ), "with_activations": lambda: model(occ, return_bank_activations=True),
                 "whole_step_from_points": lambda: pipe(batch)}
        for name, fn in calls.items():
            for _ in range(3):
                fn()
            spin(fn, args.spinup_ms / 4)
            n = 30
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            # compulsory bytes: occupancy in (1 B / voxel), head out (4 B), activations out (4 B x G)
            bytes_ = B * V * (1 + 4 + (4 * G if name == "with_activations" else 0))
            if name == "whole_step_from_points":
                bytes_ += B * args.points * 24
            out[name] = {"ms_per_call": ms, "tiles_per_s": B / (ms * 1e-3),
                         "algorithmic_tflops": flops / (ms * 1e-3) / 1e12,
                         "frac_of_int8_peak_dense": flops / (ms * 1e-3) / 1e12 / PEAK_I8_MFMA_TOPS,
                         "compulsory_GBps": bytes_ / (ms * 1e-3) / 1e9,
                         "frac_of_hbm_peak": bytes_ / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
        fused = bool(model.fused_forward and sna._hip.conv_fused_supported(occ, ks))
        out["head_only"]["kernel"] = ("conv_lin_i8_kernel (K3L: one combined kernel sum_i lambda_i K_i, int8 Toeplitz GEMM)"
                                      if fused else "sn_conv_bank")
        out["with_activations"]["kernel"] = ("conv_occ_i8_kernel (four-copy int8 kernel: the only int8 form for ky != 9; "
                                             "3 of its 16 kernel rows carry a kernel)")
        out["bound"] = ("hbm: 270 flop per compulsory byte (head only) is under the int8 matrix pipe's balance of 625 -- "
                        "the dense-convention fraction of the int8 peak cannot pass ~0.43 on this configuration")
    return out


def main():
    args = parse()
    launch_if_needed(args)
    _late_imports()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
              f"(or plain `python bench.py --gpus {args.gpus}`, which starts the ranks itself)", file=sys.stderr)
        sys.exit(2)
    need = 1 if args.rehearse_on_one_gpu else local_rank + 1
    if torch.cuda.device_count() < need:
        print(f"bench.py: rank {rank} needs HIP device {need - 1}, {torch.cuda.device_count()} visible",
              file=sys.stderr)
        sys.exit(3)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU path)"
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL; used only for barrier + max/sum of scalars
        rccl_ranks = dist.get_world_size()
    n_gpus = world

    # ---- model with explicit parameters and this rank's resident batch
    specs, names, lambdas, last = synthetic_bank_spec(GENEO_NUM)
    torch.manual_seed(0)
    model = sna.SceneNet(GENEO_NUM, KERNEL_SIZE)
    apply_bank_spec(model, specs, names, lambdas, last)
    model = model.to(dev)
    dims = (args.grid,) * 3
    B = args.batch
    tiles, labels = zip(*[synthetic_tile(rank * B + i, args.points) for i in range(B)])
    batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)  # inputs resident in HBM before timing
    pipe = sna.ScenePipeline(model, dims)
    del tiles, labels

    # HIP events that time the stages live, inside the timed loop, on the launch stream.  They are created BEFORE the loop
    # (building one costs the host ~5-10 us, and the eager step is within 10 % of being host bound) and recorded on every
    # --event-every'th step only: [measured] round 3, 200 steps: 187.3 us/step with the records on every 4th step, 159.1 on
    # every 50th -- a record is ~2.5 us of GPU time and the host work around it starves the queue.
    _pool = [torch.cuda.Event(enable_timing=True) for _ in range(6 * (args.steps + 8))]
    ev = lambda: _pool.pop() if _pool else torch.cuda.Event(enable_timing=True)  # noqa: E731
    conv_ev, vox_ev, conv_kev = [], [], []
    for e_ in _pool:   # (recorded once: torch creates the hipEvent_t at the first record -- sn_launch_timing_events needs the handle)
        e_.record()
    # the z-walk launch can carry its own start / stop events (sn_launch_timing_events: the kernel's timestamps, what rocprofv3's
    # kernel trace reports); every other kernel of the path is timed between two records on the stream
    walk_events = tuple(KERNEL_SIZE) == (9, 9, 9) and args.grid % 16 == 0

    def step(timed):
        # K2 (GENEO bank + the int8 contraction's per-bank preparation) RIDES in K1's first launch: 16 extra workgroups of the
        # bounding-box kernel (sn_voxel_occupancy_fused_bank; scene-net_amd/pipeline.py does the same) -- it reads only the
        # model's scalars.  [measured] forked onto a side stream it left 10.8 of its 12.4 serial us on the critical path.
        _, _, bank, prep = rider = model.bank_rider(dev)
        lam = model.effective_lambdas(dev)
        if timed:
            a, b, c = ev(), ev(), ev()
            a.record()
        grids = pipe.voxelize(batch, bank_rider=rider)
        if timed:
            b.record()   # (the voxel stage's end IS the contraction's start: one record between them, not two -- a record is
            # ~2.5 us of GPU time of its own, and the second one used to sit inside the contraction's interval)
        # (the walk's verdict for these parameters is learnt asynchronously during the warm-up; from then on the empty
        # fallback launch behind the walk is left out: scene_net.py, contract_prepared)
        with_kev = timed and walk_events and len(vox_ev) % 2 == 0   # (every other timed step: the launch with events costs ~9 us of stream time)
        if with_kev:
            ks, ke = ev(), ev()
            sna._hip.launch_timing_events(ks, ke)
            conv_kev.append((ks, ke))
        _, out = model.contract_prepared(grids.occ, bank, lam, prep)
        if timed:
            c.record()
            vox_ev.append((a, b))
            if not with_kev:   # (an interval that holds a launch with events is not an ordinary one)
                conv_ev.append((b, c))
        return out

    def step_fp32():
        """the same step with the contraction on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32): the occupancy bytes
        are handed over as u8, which sn_conv_bank takes as a general (non-binary) input"""
        grids = pipe.voxelize(batch)
        bank = model.compute_bank(dev)
        lam = model.effective_lambdas(dev)
        return sna._hip.conv_bank(grids.occ.view(torch.uint8), bank, lam, want_act=False, want_out=True)[1]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed_job(fn, n):
        """EXACTLY n steps of fn between two (barrier + synchronize) fences; returns (max-over-ranks seconds, this
        rank's seconds)."""
        fence()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        fence()
        local = time.perf_counter() - t_
        return job_time_max(local, dev), local

    def spin(fn, ms):
        """untimed: run fn under sustained load for `ms` so the device clocks are where a long job would have them"""
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < ms:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()

    # host hygiene: a full python GC pass over torch's ~10^6 objects costs tens of ms and would land at a random
    # point of the timed loop; park everything allocated so far in the permanent generation
    gc.collect()
    gc.freeze()
    # ---- cold figure first: W warm-up steps (lazy initialisation, allocator growth -- ~2 ms of GPU work after the
    # seconds of idling the set-up above took), then K steps timed with the clocks wherever that leaves them
    for _ in range(args.warmup):
        step(False)
    dt_cold, _ = timed_job(lambda: step(False), args.steps)
    # ---- headline: the chip's clocks take ~100 ms of sustained load to settle after idling ([measured] the same
    # kernels: 0.317 ms/launch in a 20-step run started cold, 0.296 in the second half of a 100-step run); the
    # untimed spin-up runs the same step until that much GPU time has passed, then the W warm-up steps follow
    if args.spinup_ms > 0:
        spin(lambda: step(False), args.spinup_ms)
    for _ in range(args.warmup):
        step(False)
    served_before = sna._hip.conv_i8_path_counts()[0]   # launches the folded int8 kernel has served so far
    event_every = args.event_every if args.event_every > 0 else (max(5, args.steps // 8) if args.steps >= 8 else 1)
    fence()
    t0 = time.perf_counter()
    for i_ in range(args.steps):
        out = step(i_ % event_every == 0)   # HIP events around the stages: on every --event-every'th timed step
    fence()
    dt_local = time.perf_counter() - t0
    dt = job_time_max(dt_local, dev)
    tiles_done = job_sum(float(B * args.steps), dev)
    if world > 1:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, B * args.steps / dt_local)
    else:
        per_rank = [B * args.steps / dt_local]
    # ---- the same job on the fp32 matrix pipe, end to end (K1 -> K2 -> fp32 conv + head), same loop, same fences
    n32 = max(3, min(10, args.steps))
    for _ in range(2):
        step_fp32()
    if args.spinup_ms > 0:
        spin(step_fp32, args.spinup_ms / 4)
    dt_fp32, _ = timed_job(step_fp32, n32)

    conv_stream_ms = float(np.mean([a.elapsed_time(b) for a, b in conv_ev])) if conv_ev else float("nan")   # end of the voxel stage -> end of the contraction
    conv_ms, conv_clock, conv_n = conv_stream_ms, "interval between two event records on the stream", len(conv_ev)
    if conv_kev:
        kms = [a.elapsed_time(b) for a, b in conv_kev]
        if all(0.0 < k <= 1.05 * conv_stream_ms for k in kms) or not conv_ev:   # (a launch that took another kernel leaves its pair unwritten)
            conv_ms, conv_n = float(np.mean(kms)), len(kms)
            conv_clock = "the kernel's own start/stop timestamps (hipExtLaunchKernel events; sn_launch_timing_events)"
    vox_ms = float(np.mean([a.elapsed_time(b) for a, b in vox_ev]))
    V = args.grid ** 3
    ntaps = int(np.prod(KERNEL_SIZE))
    G = sum(GENEO_NUM.values())
    conv_flops = 2.0 * V * ntaps * G * B                  # 6.115 GFLOP/tile @ 64^3
    vox_bytes = (24.0 * args.points + 4.0 * V) * B        # 3.449 MB/tile @ (100k, 64^3)
    conv_tflops = conv_flops / (conv_ms * 1e-3) / 1e12
    vox_gbs = vox_bytes / (vox_ms * 1e-3) / 1e9
    # what the int8 kernel really issues per output: 64-slot MFMA steps x 3 digit planes instead of ntaps.  Stride-4
    # kernel (conv_i8s.hip, ky = 9, 9 x 9 kernel rows): a row costs 2 1/4 dwords of K -> 183 dwords -> 12 steps;
    # four-copy kernel (conv_i8.hip, everything else): ky padded to a multiple of 4 -> 243 dwords -> 16 steps
    stride4 = (KERNEL_SIZE[2] == 9 and KERNEL_SIZE[0] * KERNEL_SIZE[1] == 81 and args.grid % 16 == 0
               and not sna._hip.get_option("conv_i8_legacy"))
    # folded kernel (conv_i8s.hip, conv_occ_i8f_kernel): a 9^3 bank that is symmetric in x and y -- every GENEO bank;
    # the device checks it per call and counts what it served -- is contracted over 9 x 5 x 5 folded taps: 45 folded
    # rows x 1 1/4 dwords -> 4 steps.  `achieved` keeps SURVEY 8d's dense convention (2 * V * 729 * 16 flops per tile).
    served_now = sna._hip.conv_i8_path_counts()[0]
    folded = stride4 and tuple(KERNEL_SIZE) == (9, 9, 9) and served_now - served_before >= args.steps
    zwalk = folded   # the step calls sn_conv_bank_prepared: the folded contraction runs as the z-walk kernel (conv_i8z.inc)
    if folded:
        i8_steps = 4
    elif stride4:
        rq = (KERNEL_SIZE[0] * KERNEL_SIZE[1] + 3) // 4
        i8_steps = rq // 2 + ((rq + 3) // 4 + 2 * (rq & 1) + 3) // 4
    else:
        chunks = KERNEL_SIZE[0] * KERNEL_SIZE[1] * ((KERNEL_SIZE[2] + 3) // 4)
        i8_steps = (chunks + 15) // 16
    executed_ops = 2.0 * V * (i8_steps * 64 * 3) * 16 * B
    executed_tops = executed_ops / (conv_ms * 1e-3) / 1e12

    # the same batch through the fp32-MFMA kernel (general-input path), timed the same way
    occ_u8 = pipe.voxelize(batch).occ.view(torch.uint8)
    bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)
    for _ in range(2):
        sna._hip.conv_bank(occ_u8, bank, lam, want_act=False, want_out=True)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(n32):
        sna._hip.conv_bank(occ_u8, bank, lam, want_act=False, want_out=True)
    e1.record()
    torch.cuda.synchronize()
    conv32_ms = e0.elapsed_time(e1) / n32
    conv32_tflops = conv_flops / (conv32_ms * 1e-3) / 1e12

    skip_info = None
    if not args.no_extras:
        # opt-in data-dependent mode (sn_set_option): tiles whose halo is empty skip their MFMA loop, tiles are handed
        # out by ticket.  Same results bit for bit; quoted beside the dense headline, never as `value`.
        sna._hip.set_option("conv_skip_empty_tiles", 1)
        try:
            out_skip = None
            for _ in range(3):   # also grows torch's allocator to the two extra output blocks the loop below ping-pongs
                out_skip = step(False)
            spin(lambda: step(False), args.spinup_ms / 4)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(n32):
                out_skip = step(False)
            torch.cuda.synchronize()
            skip_ms = (time.perf_counter() - ts) / n32 * 1e3
            skip_same = bool(torch.equal(out_skip, out))
        finally:
            sna._hip.set_option("conv_skip_empty_tiles", 0)
        skip_info = {"ms_per_step": skip_ms, "tiles_per_s_per_gpu": B / (skip_ms * 1e-3), "identical_output": skip_same,
                     "note": "opt-in conv_skip_empty_tiles=1 on this rank's synthetic LiDAR-shaped batch; data dependent, "
                             "not the headline"}

    # the forward THROUGH LINEARITY (sn_conv_fused: conv(x, sum_i lambda_i K_i), what SceneNet.forward uses when the
    # bank activations are not asked for).  Not the 16-kernel contraction BASELINE's roofline is quoted
    # on, so: an extra, never `value`.
    fused_info = None
    if not args.no_extras and sna._hip.conv_fused_supported(pipe.voxelize(batch).occ, KERNEL_SIZE):
        def fused_step():   # (what ScenePipeline runs for the module's default forward: K2 riding in K1's first launch, the
            # guard's gated launches left out once its verdict has been read for these parameters)
            _, _, bank_f, _ = rider_f = model.bank_rider(dev)
            grids = pipe.voxelize(batch, bank_rider=rider_f)
            return model.fused_served(grids.occ, bank_f, model.effective_lambdas(dev), torch.float32)
        nf = max(30, args.steps)   # a 0.12 ms step: ten of them are too few to time against the host clock
        out_fused = None
        for _ in range(3):
            out_fused = fused_step()
        spin(fused_step, args.spinup_ms / 4)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(nf):
            out_fused = fused_step()
        torch.cuda.synchronize()
        f_ms = (time.perf_counter() - ts) / nf * 1e3
        e0, e1, e2 = ev(), ev(), ev()
        g_ = pipe.voxelize(batch)
        b_, l_ = model.compute_bank(dev), model.effective_lambdas(dev)
        for _ in range(4):   # (the verdict is learnt asynchronously: after a few calls the served form runs)
            model.fused_served(g_.occ, b_, l_, torch.float32)
            torch.cuda.synchronize()
        e0.record()
        for _ in range(nf):   # K3L as the step runs it: prepared tables, the guard's gated launches left out
            model.fused_served(g_.occ, b_, l_, torch.float32)
        e1.record()
        for _ in range(nf):   # the self-contained entry: every workgroup builds the tables, gated fp32 launches behind
            sna._hip.conv_fused(g_.occ, b_, l_)
        e2.record()
        torch.cuda.synchronize()
        fused_info = {"ms_per_step": f_ms, "tiles_per_s_per_gpu": B / (f_ms * 1e-3),
                      "conv_launch_ms": e0.elapsed_time(e1) / nf,
                      "conv_launch_unprepared_ms": e1.elapsed_time(e2) / nf,
                      "max_abs_diff_vs_headline_output": float((out_fused - out).abs().max()),
                      "note": "forward through linearity: one combined 24-bit kernel, Toeplitz implicit GEMM on int8 "
                              "MFMA, kernel rows packed at 24 K-bytes (0.375 MFMA/voxel instead of 3); both outputs are within 1e-4 of the fp64 reference"}

    # the same step captured once into a hipGraph and replayed (nothing on the path synchronises or allocates outside
    # torch's allocator, every launch goes to the current stream): removes the ~20 us of dispatch gaps per step.
    # Reported beside the eager headline, never as `value`.
    graph_info = None
    try:
        if args.no_extras:
            raise RuntimeError("skipped (--no-extras)")
        # under a live process group the RCCL watchdog thread may call into HIP while this thread captures: relaxed
        # (thread-local) capture mode keeps its calls out of the capture's legality checks
        cap_mode = "thread_local" if world > 1 else "global"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(False)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode=cap_mode):
            out_graph = step(False)
        for _ in range(3):
            graph.replay()
        spin(graph.replay, args.spinup_ms / 4)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        torch.cuda.synchronize()
        g_ms = (time.perf_counter() - ts) / args.steps * 1e3
        graph_info = {"ms_per_step": g_ms, "tiles_per_s_per_gpu": B / (g_ms * 1e-3),
                      "identical_output": bool(torch.equal(out_graph, out)),
                      "note": "whole step (3 voxel launches -- one pass over the points with the bank + preparation riding in it, finalize, gated fallback --; conv) replayed "
                              "from one hipGraph"}
        del graph
        if fused_info is not None:   # the 0.12 ms fused step is at the edge of being host bound when launched eagerly
            with torch.cuda.stream(side):
                fused_step()
            torch.cuda.current_stream().wait_stream(side)
            fgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(fgraph, capture_error_mode=cap_mode):
                out_fg = fused_step()
            for _ in range(3):
                fgraph.replay()
            spin(fgraph.replay, args.spinup_ms / 4)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(nf):
                fgraph.replay()
            torch.cuda.synchronize()
            fg_ms = (time.perf_counter() - ts) / nf * 1e3
            fused_info["graph_replay_ms_per_step"] = fg_ms
            fused_info["graph_replay_tiles_per_s_per_gpu"] = B / (fg_ms * 1e-3)
            fused_info["graph_replay_identical_output"] = bool(torch.equal(out_fg, out_fused))
            del fgraph
    except Exception as exc:  # noqa: BLE001 -- an extra, never fatal to the headline
        msg = f"{type(exc).__name__}: {exc}"[:200]
        graph_info = {"skipped": msg} if "skipped" in msg else {"error": msg}

    # ---- what the REFERENCE'S OWN DEFAULTS run (experiments/scenenet_ts40k/defaults_config.yml:16-19,33-40 and the committed
    # checkpoint): kernel (9, 5, 5), one GENEO per family, the checkpoint's 13 trained scalars, 64 tiles of 64^3.
    # A user who drops this package into core/lit_modules unchanged gets THIS, not C2's 16 x 9^3 bank.  2 V 225 x 3 = 354
    # MFLOP per tile against 0.26 MB in + 1.05 MB out: 270 flop/B, under the int8 pipe's balance (625) -- the bound is HBM.
    ref_defaults = None
    if not args.no_extras and args.grid == 64:
        ref_defaults = reference_defaults_extra(sna, dev, ev, spin, args)

    # which BASELINE config this run is (derived from the arguments, never assumed)
    if (B, args.points, args.grid) == (32, 100_000, 64):
        wl_tag = "C2"
    elif args.grid == 128 and B == 32:
        wl_tag = "C3 per-GPU share (256 tiles / 8 GPUs)"
    else:
        wl_tag = "custom"

    # ---- sustained figure: the same eager step, back to back, over >= --sustain-ms of wall time (no events inside):
    # long enough for an external sampler (rocm-smi, the driver's gpu_busy probe) to see the GPU busy, and a check on the
    # headline's clock state -- `value` is 20 steps = 3-4 ms
    sustained = None
    if args.sustain_ms > 0:
        n_s = max(args.steps, int(args.sustain_ms / max(dt / args.steps * 1e3, 1e-3)) + 1)
        busy = BusySampler(local_rank)
        busy.start()
        dt_s, _ = timed_job(lambda: step(False), n_s)
        busy_info = busy.stop()
        sustained = {"value": B * n_s * n_gpus / dt_s, "ms_per_step": dt_s / n_s * 1e3, "steps": n_s,
                     "wall_s": dt_s, "gpu_busy_percent": busy_info}

    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes/launch from separate rocprofv3 --pmc passes
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f)

    res = {
        "metric": f"voxel-tiles/sec (point cloud -> {args.grid}^3 occupancy -> {G}-GENEO bank conv -> head)",
        "value": tiles_done / dt, "unit": "tiles/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i8", "data": "synthetic",
        # the same K steps timed BEFORE the spin-up (clocks as the set-up left them), and the same job with the
        # contraction on the fp32 matrix pipe (K1 -> K2 -> conv_bank_kernel, same fences, same batch)
        "value_cold": tiles_done / dt_cold, "ms_per_step_cold": dt_cold / args.steps * 1e3,
        "value_fp32": B * n32 * n_gpus / dt_fp32, "ms_per_step_fp32": dt_fp32 / n32 * 1e3, "steps_fp32": n32,
        "value_sustained": sustained["value"] if sustained else None,
        "sustained": sustained,
        "clock_state": f"value: after {args.spinup_ms:.0f} ms untimed spin-up of the same step; value_cold: none",
        "rccl_ranks": rccl_ranks, "per_rank_tiles_per_s": per_rank,
        "config": {"workload": f"{wl_tag}: {B} tiles/GPU x {args.points} points (fp64 xyz, UTM-scale), {args.grid}^3 voxel "
                               f"grid, {G} GENEO kernels {KERNEL_SIZE[0]}^3 (cy 6, cone 5, neg 5); conv on the int8 "
                               f"matrix cores: binary occupancy x 24-bit fixed-point weights (3 int8 digits), exact "
                               f"int32 accumulation, fp32 head",
                   "tiles_per_gpu": B, "points_per_tile": args.points, "grid": list(dims), "geneo_kernels": G,
                   "kernel_size": list(KERNEL_SIZE), "parallelism": f"tile-sharded x{n_gpus}, no collectives"},
        "points_per_s": B * args.points * n_gpus / (vox_ms * 1e-3),
        # dominant kernel.  `achieved` = ALGORITHMIC flops (2*V*729*16 per tile) / launch time; `peak` = dense int8
        # MFMA peak.  `executed` counts what the kernel really issues (3 digit planes x 64-slot steps): that is the
        # matrix-pipe utilisation figure.
        "roofline": {"kernel": ("conv_occ_i8z_kernel" if zwalk else "conv_occ_i8s_kernel" if stride4 else "conv_occ_i8_kernel")
                               + " (K3', v_mfma_i32_16x16x64_i8)",
                     "algorithm": ("bank symmetric in x and y (verdict of the device-side preparation, per call): 9x5x5 folded "
                                   "taps, the same integer sums as the 729-tap contraction; z-walk over planes y-folded once "
                                   "per column" if folded else "729 taps per kernel"),
                     "bound": "mfma", "mfma_steps_per_16x16_outputs": i8_steps,
                     "achieved": conv_tflops, "peak": PEAK_I8_MFMA_TOPS, "unit": "TFLOP/s",
                     "frac": conv_tflops / PEAK_I8_MFMA_TOPS,
                     "traffic": traffic.get("conv_occ_i8z_kernel" if zwalk and "conv_occ_i8z_kernel" in traffic
                                            else "conv_occ_i8f_kernel" if folded and "conv_occ_i8f_kernel" in traffic
                                            else "conv_occ_i8s_kernel" if stride4 else "conv_occ_i8_kernel"),
                     "launch_ms": conv_ms, "launch_clock": conv_clock, "launch_ms_between_stream_events": conv_stream_ms,
                     "launches_timed": conv_n, "flops_per_launch": conv_flops, "executed": executed_tops,
                     "executed_frac": executed_tops / PEAK_I8_MFMA_TOPS,
                     "executed_frac_of_measured_ceiling": executed_tops / MEASURED_I8_MFMA_TOPS},
        "roofline_fp32": {"kernel": "conv_bank_kernel (K3, v_mfma_f32_16x16x4_f32; same batch, general-input path)",
                          "bound": "mfma", "achieved": conv32_tflops, "peak": PEAK_F32_MFMA_TFLOPS,
                          "unit": "TFLOP/s", "frac": conv32_tflops / PEAK_F32_MFMA_TFLOPS,
                          "traffic": traffic.get("conv_bank_kernel"), "launch_ms": conv32_ms,
                          "flops_per_launch": conv_flops, "tiles_per_s_conv_only": B / (conv32_ms * 1e-3)},
        "roofline_voxel": {"kernel": "K1: one pass over the points (box, in-launch exchange, descriptor, LDS-bitmap occupancy; K2 riding) + "
                                     "finalize + gated fallback (3 launches at 64^3; larger grids: box pass + z-slab binning, 4)", "bound": "hbm", "achieved": vox_gbs, "peak": PEAK_HBM_GBS,
                           "unit": "GB/s", "frac": vox_gbs / PEAK_HBM_GBS, "traffic": traffic.get("voxel_stage"),
                           "stage_ms": vox_ms, "stages_timed": len(vox_ev), "bytes_per_stage": vox_bytes},
        "fused_linear": fused_info,
        "reference_defaults": ref_defaults,
        "graph_replay": graph_info,
        "skip_empty_tiles": skip_info,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, specs, names, lambdas, last)
            res["speedup_vs_cpu_fp64"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
