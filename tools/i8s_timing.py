#!/usr/bin/env python3
"""Debug only: per-workgroup / per-wave phase times of conv_occ_i8s_kernel at C2 from the wall_clock64 stamps of a
`make -B EXTRA=-DSN_CONV_TIMING OUT=build/timing OBJDIR=build/obj_timing` build:
    SN_HIP_LIB=build/timing/libscenenet_hip.so python tools/i8s_timing.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna
from scene_net_amd import _hip
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile
dev = torch.device("cuda:0")
specs, names, lambdas, last = synthetic_bank_spec()
model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
apply_bank_spec(model, specs, names, lambdas, last)
model = model.to(dev)
bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)
tiles = [synthetic_tile(i, 100_000)[0] for i in range(32)]
occ = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles, device=dev), (64,) * 3, occ_dtype=torch.bool).occ
for _ in range(200):
    _hip.conv_bank(occ, bank, lam, want_act=False, want_out=True)
torch.cuda.synchronize()
lib = _hip.load()
buf = np.zeros(1024 * 16 + 1024 * 64, dtype=np.uint64)
lib.sn_debug_i8s_times(buf.ctypes.data_as(ctypes.c_void_p))
t = buf[:1024 * 16].reshape(1024, 16).astype(np.int64)
w = buf[1024 * 16:].reshape(1024, 8, 8).astype(np.int64)
n = int((t[:, 0] > 0).sum())
t, w = t[:n], w[:n]
t0 = t[:, 0].min()
us = lambda a: a / 100.0
def show(name, a):
    print(f"{name:34s} min {a.min():8.2f} med {np.median(a):8.2f} max {a.max():8.2f}")
print("workgroups", n)
show("start", us(t[:, 0] - t0))
show("bank staged", us(t[:, 1] - t[:, 0]))
if (t[:, 6] > 0).all():   # the folded kernel stamps the end of its symmetry check
    show("symmetry check (folded kernel)", us(t[:, 6] - t[:, 1]))
show("scales + bounds", us(t[:, 2] - t[:, 1]))
show("tables", us(t[:, 3] - t[:, 2]))
show("first halos in + barrier", us(t[:, 4] - t[:, 3]))
show("prologue total", us(t[:, 4] - t[:, 0]))
show("end", us(t[:, 5] - t0))
names = ["wait landed", "pair steps", "tail steps", "epilogue", "dma issue (+done wait)", "tile end (waits, signals)", "tile loop total"]
for half, sel in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    print(half)
    for k, nm in enumerate(names):
        show("  " + nm, us(w[:, sel, k]).reshape(-1))
