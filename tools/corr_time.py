#!/usr/bin/env python3
"""Backward correlation at the training shape (C2: 32 tiles of 64^3, LiDAR-shaped occupancy from the synthetic tiles, 9^3):
the gather over the set voxels (K4s) at several tile sizes against the GEMM form (K4).  HIP-event times per call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd import _hip  # noqa: E402
from scene_net_amd.synthetic import synthetic_tile  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch = sna.PointBatch.from_tiles([synthetic_tile(t)[0] for t in range(B)], device=dev)
x = sna.voxelize_batch(batch, (64, 64, 64), occ_dtype=torch.bool).occ.reshape(B, 1, 64, 64, 64)
print("set voxels per tile:", x.sum().item() / B)
g = torch.randn(B, 1, 64, 64, 64, device=dev)
o = torch.tanh(torch.randn(B, 1, 64, 64, 64, device=dev)).clamp_min(0)

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for dt in (torch.float32, torch.bfloat16):
    gg, oo = g.to(dt), o.to(dt)
    _hip.set_option("corr_dense", 1)
    ref = _hip.conv_corr(x, gg, oo, (9, 9, 9))
    print(f"{dt}: GEMM form (K4)            {timed(lambda: _hip.conv_corr(x, gg, oo, (9, 9, 9))):8.1f} us")
    _hip.set_option("corr_dense", 0)
    for tb in (0, 512, 1024, 2048):
        _hip.set_option("corr_sparse_tile_bytes", tb)
        c = _hip.conv_corr(x, gg, oo, (9, 9, 9))
        err = (c - ref).abs().max().item() / ref.abs().max().item()
        print(f"{dt}: gather (K4s) tile {tb:5d} B   {timed(lambda: _hip.conv_corr(x, gg, oo, (9, 9, 9))):8.1f} us   rel diff {err:.2e}")
    _hip.set_option("corr_sparse_tile_bytes", 0)
