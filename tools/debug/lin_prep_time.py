#!/usr/bin/env python3
"""sn_conv_fused against sn_conv_fused_prepared at C2 (events around 50 calls each, no gated launch: assume_served)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_net_amd as sna
from scene_net_amd import _hip
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile
dev = torch.device("cuda:0")
GENEO = {"cy": 6, "cone": 5, "neg": 5}
specs, names, lambdas, last = synthetic_bank_spec(GENEO)
model = sna.SceneNet(GENEO, (9, 9, 9)); apply_bank_spec(model, specs, names, lambdas, last); model = model.to(dev)
batch = sna.PointBatch.from_tiles([synthetic_tile(t)[0] for t in range(32)], device=dev)
x = sna.voxelize_batch(batch, (64, 64, 64), occ_dtype=torch.bool).occ
bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)
blob = _hip.conv_fused_prep(bank, lam)
word = torch.zeros(1, dtype=torch.int32, device=dev)
def timed(fn, n=50):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rep in range(3):
    print(f"tables built by every workgroup: {timed(lambda: _hip.conv_fused(x, bank, lam, verdict=word, assume_served=True)):6.1f} us   "
          f"from the blob: {timed(lambda: _hip.conv_fused(x, bank, lam, prep=blob, assume_served=True)):6.1f} us   "
          f"(the preparation alone: {timed(lambda: _hip.conv_fused_prep(bank, lam, blob)):5.1f} us)")
