#!/usr/bin/env python3
"""PCIe-inclusive rates of the reference-signature entry points (numpy in / numpy out), for DESIGN.md.  Not `value`."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna
from scene_net_amd.synthetic import synthetic_tile
xyz, lab = synthetic_tile(0, 100_000)
tf = sna.Voxelization([15], vxg_size=(64, 64, 64))
for _ in range(3):
    tf((xyz, lab))
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    vox, gt = tf((xyz, lab))
dt = (time.perf_counter() - t0) / n
print(f"Voxelization.__call__ (100k pts -> density+gt f64 [1,64,64,64], H2D + kernels + D2H): {dt*1e3:.2f} ms/tile = {1/dt:.0f} tiles/s, {1e5/dt/1e6:.1f} Mpoints/s")
torch.manual_seed(0)
model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9)).cuda()
x = torch.from_numpy((vox > 0).astype(np.float64))[None]  # [1,1,64,64,64] f64 host, like the reference's batch
with torch.no_grad():
    for _ in range(3):
        model(x.cuda()).cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = model(x.cuda()).cpu()
    dt = (time.perf_counter() - t0) / n
print(f"SceneNet.forward, f64 host tensor in/out (H2D + fp32-MFMA conv + D2H), B=1: {dt*1e3:.2f} ms = {1/dt:.0f} tiles/s")
