#!/usr/bin/env python3
"""A/B timing of sn_conv_bank (C2 batch, occupancy input) under values of one env knob: interleaved rounds in
one process, min and median per value (cdna guide rule 24)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene_net_amd import _hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
bank = ((torch.rand(16, 9, 9, 9) - 0.5)).to(dev).contiguous()
lam = (torch.rand(16) / 16).to(dev)
dt = torch.uint8 if os.environ.get("KNOB_DTYPE") == "u8" else torch.bool
if os.environ.get("KNOB_DATA") == "lidar":  # real-shaped occupancy: synthetic LiDAR tiles through the voxeliser
    import scene_net_amd as sna
    from scene_net_amd.synthetic import synthetic_tile
    x = sna.voxelize_batch(sna.PointBatch.from_tiles([synthetic_tile(t, 100_000)[0] for t in range(32)], device=dev),
                           (64, 64, 64), occ_dtype=torch.bool).occ.to(dt)
else:
    x = (torch.rand(32, 1, 64, 64, 64, device=dev) < 0.035).to(dt)
knob, vals = sys.argv[1], sys.argv[2:]
for _ in range(60):
    _hip.conv_bank(x, bank, lam)
torch.cuda.synchronize()
res = {v: [] for v in vals}
for rnd in range(6):
    for v in vals:
        os.environ[knob] = v
        if knob == "OPT_SKIP_EMPTY":
            _hip.set_option("conv_skip_empty_tiles", int(v))
        _hip.conv_bank(x, bank, lam)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(15):
            _hip.conv_bank(x, bank, lam)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 15 * 1000)
for v in vals:
    print(f"{knob}={v}: min {min(res[v]):.1f} us  median {statistics.median(res[v]):.1f} us", flush=True)
