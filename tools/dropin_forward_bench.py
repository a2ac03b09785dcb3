#!/usr/bin/env python3
"""The literal drop-in case: SceneNet.forward on the reference's own input format, f64 {0., 1.} grids [B,1,64,64,64]
already on the device (lit_model_wrappers.py:56-57).  python tools/dropin_forward_bench.py"""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec  # noqa: E402

dev = torch.device("cuda:0")
specs, names, lambdas, last = synthetic_bank_spec({"cy": 6, "cone": 5, "neg": 5})
model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
apply_bank_spec(model, specs, names, lambdas, last)
model = model.to(dev)
x = (torch.rand((32, 1, 64, 64, 64), device=dev) < 0.035).double()


def timed(fn, iters=20):
    gc.collect()
    gc.freeze()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    import time as _time
    _t = _time.perf_counter()   # the chip's clocks settle after ~100 ms of sustained load (see bench.py)
    while _time.perf_counter() - _t < 0.15:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


with torch.no_grad():
    t_auto = timed(lambda: model(x))
    model.fused_forward = False
    t_fp32 = timed(lambda: model(x))
    model.fused_forward = True
    t_bool = timed(lambda: model(x.bool()))
print(f"f64 {{0,1}} input, B=32 64^3: device-side check + int8 kernels {t_auto:.3f} ms; fp32 contraction {t_fp32:.3f} ms "
      f"({t_fp32 / t_auto:.1f}x); bool input incl. the cast {t_bool:.3f} ms")
