#!/usr/bin/env python3
"""Debug only: per-workgroup / per-wave phase times of conv_occ_i8_kernel at C2, from the wall_clock64 stamps a
`make -B EXTRA=-DSN_CONV_TIMING` build records (rebuild without EXTRA afterwards: the stamps cost ~1 %)."""
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
model.fused_forward = False
tiles, labels = zip(*[synthetic_tile(i, 100_000) for i in range(32)])
batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
pipe = sna.ScenePipeline(model, (64,) * 3, keep_labels=[15.0])
with torch.no_grad():
    for _ in range(5):
        out = pipe(batch)
torch.cuda.synchronize()
lib = _hip.load()
buf = np.zeros(1024 * 24, dtype=np.uint64)
lib.sn_debug_conv_times(buf.ctypes.data_as(ctypes.c_void_p))
t = buf[:1024 * 16].reshape(1024, 16).astype(np.int64)
w = buf[1024 * 16:].reshape(1024, 8).astype(np.int64)
n = int((t[:, 0] > 0).sum())
t = t[:n]
t0 = t[:, 0].min()
us = lambda a: a / 100.0
print("workgroups", n, "tiles per wg min/max", t[:, 8].min(), t[:, 8].max())
def show(name, a):
    print(f"{name:28s} min {a.min():8.2f} med {np.median(a):8.2f} max {a.max():8.2f}")
show("start", us(t[:, 0] - t0))
show("bank staged", us(t[:, 1] - t[:, 0]))
show("scales", us(t[:, 2] - t[:, 1]))
show("digit table", us(t[:, 3] - t[:, 2]))
show("first halo", us(t[:, 4] - t[:, 3]))
show("prologue total", us(t[:, 4] - t[:, 0]))
show("tiles: rounds (sum)", us(t[:, 6]))
show("  wave0 own rounds (sum)", us(t[:, 10]))
show("tiles: switch (sum)", us(t[:, 7]))
show("end", us(t[:, 5] - t0))
show("per tile rounds", us(t[:, 6]) / t[:, 8])
show("per tile switch", us(t[:, 7]) / t[:, 8])
w = w[:n]
# g_conv_w accumulates over all 5 launches (never reset): divide
for k in range(8):
    show(f"wave {k} rounds (sum/launch)", us(w[:, k]) / 5.0)
