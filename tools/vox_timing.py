#!/usr/bin/env python3
"""Debug only: per-workgroup phase times of occ_onepass_kernel (K1) at C2, from the wall_clock64 stamps a
`make -B EXTRA=-DSN_CONV_TIMING` build records (100 MHz clock)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna
from scene_net_amd import _hip
from scene_net_amd.synthetic import synthetic_tile
dev = torch.device("cuda:0")
tiles = [synthetic_tile(i, 100_000)[0] for i in range(32)]
batch = sna.PointBatch.from_tiles(tiles, device=dev)
for _ in range(50):
    g = sna.voxelize_batch(batch, (64, 64, 64), occ_dtype=torch.bool)
torch.cuda.synchronize()
buf = np.zeros(1024 * 8, dtype=np.uint64)
_hip.load().sn_debug_vox_times(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(1024, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = lambda a: a / 100.0
def show(name, a):
    print(f"{name:34s} min {a.min():8.2f} med {np.median(a):8.2f} max {a.max():8.2f}")
print("workgroups", len(t))
show("start (after the first)", us(t[:, 0] - t0))
show("points in registers + min/max", us(t[:, 1] - t[:, 0]))
show("publish", us(t[:, 2] - t[:, 1]))
show("wait for the tile's 16 tags", us(t[:, 3] - t[:, 2]))
show("boxes + descriptor", us(t[:, 4] - t[:, 3]))
show("binning out of registers", us(t[:, 5] - t[:, 4]))
show("bitmap write-out", us(t[:, 6] - t[:, 5]))
show("whole workgroup", us(t[:, 6] - t[:, 0]))
show("end (after the first start)", us(t[:, 6] - t0))
