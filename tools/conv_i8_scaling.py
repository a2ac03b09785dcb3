#!/usr/bin/env python3
"""int8 conv launch time vs batch size at 64^3 / 16 kernels 9^3: slope = per-tile cost, intercept = per-launch cost
(prologue + launch).  python tools/conv_i8_scaling.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd import _hip  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec  # noqa: E402

dev = torch.device("cuda:0")
geneo_num = {"cy": 6, "cone": 5, "neg": 5}
specs, names, lambdas, last = synthetic_bank_spec(geneo_num)
model = sna.SceneNet(geneo_num, (9, 9, 9))
apply_bank_spec(model, specs, names, lambdas, last)
model = model.to(dev)
bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)
res = []
for B in (8, 16, 32, 64, 96, 128):
    x = torch.rand((B, 1, 64, 64, 64), device=dev) < 0.3
    for _ in range(3):
        _hip.conv_bank(x, bank, lam, want_act=False, want_out=True)
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _hip.conv_bank(x, bank, lam, want_act=False, want_out=True)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    res.append((B, float(np.median(ts))))
    print(f"B={B:4d}  {res[-1][1]:8.1f} us   {res[-1][1] / B:6.2f} us/tile")
Bs, t = np.array([r[0] for r in res], float), np.array([r[1] for r in res])
k, c = np.polyfit(Bs[2:], t[2:], 1)
print(f"fit over B>=32: {k:.2f} us/tile (= {k * 4:.2f} us per round of 256 workgroup tiles) + {c:.1f} us per launch")
