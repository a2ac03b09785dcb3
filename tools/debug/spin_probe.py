#!/usr/bin/env python3
"""Which int8 kernel gives up a bounded spin on which shape (sn_conv_i8_spin_timeouts)?  python tools/debug/spin_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scene_net_amd import _hip
dev = torch.device("cuda:0")
def bank(G, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.rand((G, 9, 9, 9), generator=g) - 0.5
    w = w + w.flip(2); w = w + w.flip(3)
    return w.float().contiguous().to(dev)
for shape, G in [((2, 1, 16, 16, 64), 16), ((1, 1, 20, 18, 64), 5), ((1, 1, 12, 10, 64), 16), ((1, 1, 9, 24, 128), 16),
                 ((3, 1, 7, 5, 16), 16), ((1, 1, 1, 1, 16), 3), ((2, 1, 40, 9, 48), 16), ((1, 1, 128, 16, 32), 16)]:
    x = (torch.rand(shape) < 0.3).to(dev)
    b = bank(G, 1); lam = ((torch.rand(G) - 0.3) / G).to(dev)
    prep = _hip.conv_bank_prep(b)
    for name, kw in [("legacy-api", {}), ("zwalk v0", dict(prep=prep, zv=0)), ("zwalk v1", dict(prep=prep, zv=1))]:
        if "zv" in kw:
            _hip.set_option("conv_i8z_variant", kw.pop("zv"))
        c0 = _hip.conv_i8_spin_timeouts()
        t0 = time.perf_counter()
        _hip.conv_bank(x, b, lam, want_act=True, want_out=True, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(shape, G, name, "timeouts", _hip.conv_i8_spin_timeouts() - c0, f"{dt * 1e3:.1f} ms", flush=True)
