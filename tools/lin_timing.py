#!/usr/bin/env python3
"""Debug only: per-workgroup / per-wave phase times of conv_lin_i8_kernel (K3L) at C2, from the wall_clock64 stamps a
`make -B EXTRA=-DSN_CONV_TIMING` build records."""
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
tiles, labels = zip(*[synthetic_tile(i, 100_000) for i in range(32)])
batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
pipe = sna.ScenePipeline(model, (64,) * 3, keep_labels=[15.0])
with torch.no_grad():
    if "--unprepared" in sys.argv:   # the self-contained entry: every workgroup builds the tables (phases 4..6 are stamped there)
        occ = pipe.voxelize(batch).occ
        bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)
        for _ in range(5):
            out = _hip.conv_fused(occ, bank, lam)
    else:
        for _ in range(5):
            out = pipe(batch)
torch.cuda.synchronize()
buf = np.zeros(1024 * 16, dtype=np.uint64)
_hip.load().sn_debug_lin_times(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(1024, 16).astype(np.int64)
n = int((t[:, 0] > 0).sum())
t = t[:n]
t0 = t[:, 0].min()
us = lambda a: a / 100.0
def show(name, a):
    print(f"{name:28s} min {a.min():8.2f} med {np.median(a):8.2f} max {a.max():8.2f}")
print("workgroups", n, "tiles per wg min/max", t[:, 3].min(), t[:, 3].max())
show("start", us(t[:, 0] - t0))
show("prologue (tables)", us(t[:, 1] - t[:, 0]))
show("  first halo issued", us(t[:, 4] - t[:, 0]))
show("  K* + max", us(t[:, 5] - t[:, 4]))
show("  scale, digits rows", us(t[:, 6] - t[:, 5]))
show("  Toeplitz table", us(t[:, 1] - t[:, 6]))
show("tile loop + last epilogue", us(t[:, 2] - t[:, 1]))
show("  halo commit + barrier (sum)", us(t[:, 7]))
show("end", us(t[:, 2] - t0))
for k in range(8):
    hw = t[:, 8 + k] >> 40          # HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8]
    simd = (hw >> 4) & 3
    show(f"wave {k} busy (sum), SIMD {np.bincount(simd, minlength=4).tolist()}", us(t[:, 8 + k] & ((1 << 40) - 1)))
# the two waves of a SIMD: which one is the slow one?
for sd in range(4):
    pairs = []
    for wg in range(n):
        ws = [k for k in range(8) if ((t[wg, 8 + k] >> 44) & 3) == sd]
        if len(ws) == 2:
            a, b = (t[wg, 8 + ws[0]] & ((1 << 40) - 1)) / 100.0, (t[wg, 8 + ws[1]] & ((1 << 40) - 1)) / 100.0
            pairs.append((ws[0], ws[1], a, b))
    if pairs:
        import collections
        c = collections.Counter((p[0], p[1]) for p in pairs)
        print(f"SIMD {sd}: wave pairs {dict(c)}; busy of the lower / higher wave id: "
              f"{np.median([p[2] for p in pairs]):.2f} / {np.median([p[3] for p in pairs]):.2f} us")
