#!/usr/bin/env python3
"""A/B of the three forms of K3' (folded kernel vs stride-4 kernel vs four-copy kernel) and of the guard's gated fp32 launch, interleaved
rounds in ONE process on the bench's C2 batch (cdna_hip_programming.md rule 24).
    python tools/conv_ab.py [--rounds 5] [--iters 40] [--batch 32] [--grid 64]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd import _hip  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--random-occ", type=float, default=0.0, help="random occupancy of this density instead of tiles")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    specs, names, lambdas, last = synthetic_bank_spec()
    torch.manual_seed(0)
    model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
    apply_bank_spec(model, specs, names, lambdas, last)
    model = model.to(dev)
    bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)
    if args.random_occ > 0:
        occ = torch.rand((args.batch, 1) + (args.grid,) * 3, device=dev) < args.random_occ
    else:
        tiles = [synthetic_tile(i, args.points)[0] for i in range(args.batch)]
        occ = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles, device=dev), (args.grid,) * 3,
                                 occ_dtype=torch.bool).occ

    prep = _hip.conv_bank_prep(bank)
    use_prep = [False]

    def run(n):
        for _ in range(n):
            _hip.conv_bank(occ, bank, lam, want_act=False, want_out=True, prep=prep if use_prep[0] else None)

    variants = {
        "zwalk 2x1x8+guard": dict(legacy=0, tol=90000, fold=1, prep=True, zv=0),
        "zwalk 1x1x12+guard": dict(legacy=0, tol=90000, fold=1, prep=True, zv=1),
        "zwalk 1x2x12+guard": dict(legacy=0, tol=90000, fold=1, prep=True, zv=2),
        "zwalk 1x2x12 noguard": dict(legacy=0, tol=0, fold=1, prep=True, zv=2),
        "folded+guard": dict(legacy=0, tol=90000, fold=1),
        "folded noguard": dict(legacy=0, tol=0, fold=1),
        "stride4+guard": dict(legacy=0, tol=90000, fold=0),
        "stride4 noguard": dict(legacy=0, tol=0, fold=0),
        "legacy+guard": dict(legacy=1, tol=90000, fold=0),
        "legacy noguard": dict(legacy=1, tol=0, fold=0),
    }
    for k, v in os.environ.items():
        if k.startswith("SN_CONV"):
            print("env", k, v)
    res = {k: [] for k in variants}
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        run(10)
        torch.cuda.synchronize()
    for r in range(args.rounds):
        for name, v in variants.items():
            _hip.set_option("conv_i8_legacy", v["legacy"])
            _hip.set_option("conv_i8_tolerance_ppb", v["tol"])
            _hip.set_option("conv_i8_fold", v["fold"])
            use_prep[0] = bool(v.get("prep"))
            if "zv" in v:
                _hip.set_option("conv_i8z_variant", v["zv"])
            run(5)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(args.iters)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / args.iters)
    _hip.set_option("conv_i8_legacy", 0)
    _hip.set_option("conv_i8_tolerance_ppb", 90000)
    _hip.set_option("conv_i8_fold", 1)
    flops = 2.0 * args.grid ** 3 * 729 * 16 * args.batch
    for name, ts in res.items():
        med, mn = float(np.median(ts)), float(np.min(ts))
        print(f"{name:18s} median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us  {flops / (med * 1e-3) / 1e12:7.1f} TFLOP/s "
              f"algorithmic  rounds {[round(t * 1e3, 1) for t in ts]}")


if __name__ == "__main__":
    main()
