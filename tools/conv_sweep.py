#!/usr/bin/env python3
"""Times sn_conv_bank alone over batch sizes (per-CU tile counts) -- prologue vs per-tile cost."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna
from scene_net_amd import _hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
bank = ((torch.rand(16, 9, 9, 9) - 0.5)).to(dev).contiguous()
lam = (torch.rand(16) / 16).to(dev)
for dt in (torch.bool, torch.uint8):
    for B in (4, 8, 16, 32, 64):
        x = (torch.rand(B, 1, 64, 64, 64, device=dev) < 0.035).to(dt)
        for _ in range(3):
            _hip.conv_bank(x, bank, lam)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            _hip.conv_bank(x, bank, lam)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{str(dt):12s} B={B:3d} tiles/CU={B*64/256:5.1f}  {ms*1000:8.1f} us   {2*64**3*729*16*B/ms/1e9:7.1f} TFLOP/s", flush=True)
