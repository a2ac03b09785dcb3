#!/usr/bin/env python3
"""Where does a z-walk variant differ from sn_conv_bank?  python tools/debug/zwalk_diff.py <variant>"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scene_net_amd import _hip
dev = torch.device("cuda:0")
v = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator().manual_seed(1)
w = torch.rand((16, 9, 9, 9), generator=g) - 0.5
w = w + w.flip(2); w = (w + w.flip(3)).float().contiguous().to(dev)
lam = ((torch.rand(16) - 0.3) / 16).to(dev)
for shape in [(2, 1, 16, 16, 64), (1, 1, 64, 64, 64)]:
    x = (torch.rand(shape) < 0.3).to(dev)
    prep = _hip.conv_bank_prep(w)
    _hip.set_option("conv_i8z_variant", v)
    for rep in range(3):
        a_z, o_z = _hip.conv_bank(x, w, lam, want_act=True, want_out=True, prep=prep)
        a_r, o_r = _hip.conv_bank(x, w, lam, want_act=True, want_out=True)
        bad = (a_z != a_r).nonzero()
        print(shape, "rep", rep, "mismatching act elements:", bad.shape[0], "timeouts", _hip.conv_i8_spin_timeouts())
        if bad.shape[0]:
            print("  b", bad[:, 0].unique().tolist()[:8], "g", bad[:, 1].unique().tolist()[:8], "z", bad[:, 2].unique().tolist()[:16],
                  "x", bad[:, 3].unique().tolist()[:16], "y", bad[:, 4].unique().tolist()[:8], "...")
