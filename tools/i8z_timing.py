#!/usr/bin/env python3
"""Debug only: per-wave phase cycles of the z-walk kernel (conv_occ_i8z_kernel) at C2 from the s_memtime stamps of a
timing build:
    make -B EXTRA=-DSN_CONV_TIMING OUT=build/timing OBJDIR=build/obj_timing
    SN_HIP_LIB=build/timing/libscenenet_hip.so python tools/i8z_timing.py"""
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
prep = _hip.conv_bank_prep(bank)
for _ in range(200):
    _hip.conv_bank(occ, bank, lam, want_act=False, want_out=True, prep=prep)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    _hip.conv_bank(occ, bank, lam, want_act=False, want_out=True, prep=prep)
e1.record()
torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (with the stamps in)")
lib = _hip.load()
buf = np.zeros(1024 * 16 + 1024 * 64, dtype=np.uint64)
lib.sn_debug_i8s_times(buf.ctypes.data_as(ctypes.c_void_p))
t = buf[:1024 * 16].reshape(1024, 16).astype(np.int64)
w = buf[1024 * 16:].reshape(1024, 8, 8).astype(np.int64)
n = int((t[:, 0] > 0).sum())
t, w = t[:n], w[:n]
t0 = t[:, 0].min()
def show(name, a, unit=""):
    print(f"{name:40s} min {a.min():9.1f} med {np.median(a):9.1f} max {a.max():9.1f} {unit}")
print("workgroups", n)
show("start (wall, us)", (t[:, 0] - t0) / 100.0)
show("prologue (wall, us)", (t[:, 4] - t[:, 0]) / 100.0)
show("end (wall, us)", (t[:, 5] - t0) / 100.0)
names = ["claim + pending signal", "dependency check", "dma issue + fold pass", "round setup + step 0 operands",
         "MFMA steps 0-3", "epilogue", "round tickets total"]
for half, sel in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    print(half, "(cycles of s_memtime summed over the wave's tickets)")
    for k, nm in enumerate(names):
        show("  " + nm, w[:, sel, k].reshape(-1).astype(np.float64))
