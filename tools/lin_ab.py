#!/usr/bin/env python3
"""K3L (sn_conv_fused_prepared, the module's default forward) timed alone between events, on the prepared tables and with the
guard's verdict learnt -- what SceneNet.forward runs: BASELINE C2 (32 x 64^3, 16 kernels 9^3), the reference's defaults
(64 x 64^3, kernel (9, 5, 5)) and C3's per-GPU share (32 x 128^3).  python tools/lin_ab.py [--iters 50]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd import _hip  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()   # (the chip's clocks settle after ~100 ms of sustained load)
    while time.perf_counter() - t < 0.2:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


def model_for(ks, geneo_num):
    specs, names, lambdas, last = synthetic_bank_spec(geneo_num)
    m = sna.SceneNet(geneo_num, ks)
    apply_bank_spec(m, specs, names, lambdas, last)
    return m.to(dev)


for label, ks, gn, B, n in (("C2 32 x 64^3, 9^3", (9, 9, 9), {"cy": 6, "cone": 5, "neg": 5}, 32, 64),
                            ("defaults 64 x 64^3, (9,5,5)", (9, 5, 5), {"cy": 1, "cone": 1, "neg": 1}, 64, 64),
                            ("C3 share 32 x 128^3, 9^3", (9, 9, 9), {"cy": 6, "cone": 5, "neg": 5}, 32, 128)):
    model = model_for(ks, gn)
    bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)
    x = torch.rand((B, 1, n, n, n), device=dev) < 0.035
    with torch.no_grad():
        for _ in range(4):   # the verdict is learnt asynchronously: a few calls, then the served form
            out = model.fused_served(x, bank, lam, torch.float32)
            torch.cuda.synchronize()
        t = timed(lambda: model.fused_served(x, bank, lam, torch.float32), args.iters)
        ref = _hip.conv_fused(x, bank, lam)   # tables built by the kernel itself, gated launches behind it
        same = torch.equal(out, ref)
    print(f"{label:30s} {t:8.1f} us   (== the unprepared entry bit for bit: {same}; spin give-ups {_hip.conv_i8_spin_timeouts()}, "
          f"status {_hip.device_status()[0]})", flush=True)
