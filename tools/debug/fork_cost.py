#!/usr/bin/env python3
"""What the fork / join around K2 costs the main stream (C2, eager, sustained): the step with K2 (a) forked beside K1 as the
pipeline does, (b) serial on the main stream, (c) left out (bank / blob of the first step reused: a probe, not a mode)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_net_amd as sna
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile

dev = torch.device("cuda:0"); B = 32
batch = sna.PointBatch.from_tiles([synthetic_tile(t)[0] for t in range(B)], device=dev)
specs, names, lambdas, last = synthetic_bank_spec()
model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9)).to(dev).eval()
apply_bank_spec(model, specs, names, lambdas, last)
pipe = sna.ScenePipeline(model, (64, 64, 64))

def forked():
    bank, lam, prep, join = pipe.bank_beside(dev)
    g = pipe.voxelize(batch); join()
    return model.contract_prepared(g.occ, bank, lam, prep)[1]
def serial():
    bank, prep = model.compute_bank_prepared(dev); lam = model.effective_lambdas(dev)
    g = pipe.voxelize(batch)
    return model.contract_prepared(g.occ, bank, lam, prep)[1]
with torch.no_grad():
    bank0, prep0 = model.compute_bank_prepared(dev); lam0 = model.effective_lambdas(dev)
    def reused():
        g = pipe.voxelize(batch)
        return model.contract_prepared(g.occ, bank0, lam0, prep0)[1]
    for name, fn in (("forked", forked), ("serial", serial), ("K2 left out", reused), ("forked", forked)):
        for _ in range(50): fn()
        torch.cuda.synchronize(); t = time.perf_counter(); n = 0
        while time.perf_counter() - t < 0.4:
            for _ in range(50): fn()
            n += 50
        torch.cuda.synchronize()
        print(f"{name:12s} {(time.perf_counter() - t) / n * 1e6:8.1f} us/step")
