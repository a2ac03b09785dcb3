#!/usr/bin/env python3
"""Training-step rates at BASELINE C2 (not the headline): voxelise(+GT) -> SceneNet forward -> GENEO_Tversky_Loss ->
backward, with HIP-event timings per stage.  python tools/train_step_bench.py [--batch 32] [--iters 10]"""
import argparse
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd import _hip  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile  # noqa: E402
from scene_net_amd.training import backward_seeded, voxelize_and_forward  # noqa: E402


def timed(fn, iters, warm=2, spin_ms=150.0):
    gc.collect()
    gc.freeze()   # keep Python's full GC passes out of the timed loop
    t_spin = time.perf_counter()   # the chip's clocks settle after ~100 ms of sustained load (see bench.py)
    while (time.perf_counter() - t_spin) * 1e3 < spin_ms:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--graph", action="store_true", help="also capture the whole step in a hipGraph and replay it")
    ap.add_argument("--bf16", action="store_true", help="bf16 activation storage (SceneNet.activation_dtype; BASELINE C5)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    geneo_num = {"cy": 6, "cone": 5, "neg": 5}
    specs, names, lambdas, last = synthetic_bank_spec(geneo_num)
    torch.manual_seed(0)
    model = sna.SceneNet(geneo_num, (9, 9, 9))
    apply_bank_spec(model, specs, names, lambdas, last)
    model = model.to(dev)
    if args.bf16:
        model.activation_dtype = torch.bfloat16
    tiles, labels = zip(*[synthetic_tile(i, args.points) for i in range(args.batch)])
    batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
    pipe = sna.ScenePipeline(model, (args.grid,) * 3, keep_labels=[15.0])
    grids = pipe.voxelize(batch, want_gt=True)
    x, gt = grids.occ, grids.gt_occ
    crit = sna.GENEO_Tversky_Loss(targets=gt.float().cpu(), weighting_scheme_path=None, save_weighting_scheme=False)
    V = args.grid ** 3
    n = args.batch * V
    pred = model(x).detach()
    ranges, bin_w = crit._device_tables(dev)
    terms = _hip.SN_LOSS_WMSE | _hip.SN_LOSS_FOCAL_TVERSKY

    t_fwd = timed(lambda: _hip.loss_forward(pred, gt, ranges, bin_w, terms), args.iters)
    _, _, coef, _ = _hip.loss_forward(pred, gt, ranges, bin_w, terms)
    t_bwd = timed(lambda: _hip.loss_backward(pred, gt, ranges, coef), args.iters)
    fwd_bytes = n * (pred.element_size() + gt.element_size())
    bwd_bytes = fwd_bytes + n * pred.element_size()
    print(f"loss forward  {t_fwd * 1e3:8.1f} us  {fwd_bytes / t_fwd / 1e6:8.1f} GB/s (algorithmic {fwd_bytes / 1e6:.1f} MB)")
    print(f"loss backward {t_bwd * 1e3:8.1f} us  {bwd_bytes / t_bwd / 1e6:8.1f} GB/s (algorithmic {bwd_bytes / 1e6:.1f} MB)")

    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        g, out = voxelize_and_forward(pipe, batch)   # (the forward's opener rides in the voxelisation's first launch)
        loss = crit(out, g.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
        backward_seeded(loss)   # (loss.backward() with a cached seed: no ones_like launch)
        opt.step()
        return loss

    def fwd_only():
        with torch.no_grad():
            g = pipe.voxelize(batch, want_gt=True)
            return model(g.occ)

    t_step = timed(step, args.iters)
    t_inf = timed(fwd_only, args.iters)
    print(f"activations   {str(pred.dtype)}")
    print(f"training step {t_step:8.3f} ms  -> {args.batch / t_step * 1e3:9.0f} tiles/s   "
          f"(inference through the module: {t_inf:.3f} ms)")

    if args.graph:
        # whole-step capture (voxelise -> forward -> criterion -> backward -> SGD): every launch of the C ABI goes
        # to torch's current stream and nothing on the path synchronises, so the step is graph-capturable as is
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = step()
        torch.cuda.synchronize()
        before = [p.detach().clone() for p in model.parameters()]
        t_graph = timed(graph.replay, args.iters)
        moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(model.parameters(), before))
        print(f"graph replay  {t_graph:8.3f} ms  -> {args.batch / t_graph * 1e3:9.0f} tiles/s   "
              f"(loss {static_loss.item():.6f}, parameters moved by {moved:.3e} over the replays)")


if __name__ == "__main__":
    main()
