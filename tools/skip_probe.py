#!/usr/bin/env python3
"""conv_skip_empty_tiles on/off on all-zero, half-empty and LiDAR-shaped occupancy at 64^3 and 128^3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd import _hip  # noqa: E402
from scene_net_amd.synthetic import synthetic_tile  # noqa: E402

dev = torch.device("cuda:0")
bank = (torch.rand(16, 9, 9, 9) - 0.5).to(dev).contiguous()
lam = (torch.rand(16) / 16).to(dev)
grids = [int(g) for g in (sys.argv[1:] or ["64", "128"])]
for n in grids:
    B = int(os.environ.get("PROBE_B", 32 if n == 64 else 8))
    tiles = [synthetic_tile(t, 100_000)[0] for t in range(B)]
    lidar = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles, device=dev), (n, n, n), occ_dtype=torch.bool).occ
    half = torch.cat([torch.zeros(B, 1, n, n // 2, n, dtype=torch.bool, device=dev),
                      torch.rand(B, 1, n, n // 2, n, device=dev) < 0.03], 3).contiguous()
    for name, x in (("zeros", torch.zeros(B, 1, n, n, n, dtype=torch.bool, device=dev)), ("half", half),
                    ("lidar", lidar)):
        for opt in (0, 1):
            _hip.set_option("conv_skip_empty_tiles", opt)
            for _ in range(5):
                _hip.conv_bank(x, bank, lam)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _hip.conv_bank(x, bank, lam)
            e1.record()
            torch.cuda.synchronize()
            print(f"{n}^3 B={B} {name:6s} occupancy {x.float().mean().item():.4f} skip {opt}: "
                  f"{e0.elapsed_time(e1) / 20 * 1000:8.1f} us", flush=True)
_hip.set_option("conv_skip_empty_tiles", 0)
