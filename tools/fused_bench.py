#!/usr/bin/env python3
"""sn_conv_fused (forward through linearity) vs sn_conv_bank (16-kernel contraction) at BASELINE C2 / C3 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd import _hip  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec  # noqa: E402

dev = torch.device("cuda:0")
specs, names, lambdas, last = synthetic_bank_spec({"cy": 6, "cone": 5, "neg": 5})
model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
apply_bank_spec(model, specs, names, lambdas, last)
model = model.to(dev)
bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)


def timed(fn, iters=20):
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
    return a.elapsed_time(b) / iters * 1e3


for n, B in ((64, 32), (128, 8)):
    x = torch.rand((B, 1, n, n, n), device=dev) < 0.035
    t_d = timed(lambda: _hip.conv_bank(x, bank, lam, want_act=False, want_out=True))
    t_f = timed(lambda: _hip.conv_fused(x, bank, lam))
    d = (_hip.conv_fused(x, bank, lam) - _hip.conv_bank(x, bank, lam, want_act=False, want_out=True)[1]).abs().max().item()
    print(f"{n}^3 B={B}: 16-kernel contraction {t_d:8.1f} us   fused-linear {t_f:8.1f} us   ({t_d / t_f:.1f}x)   "
          f"max |diff| {d:.2e}")
