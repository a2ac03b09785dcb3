#!/usr/bin/env python3
"""Is the eager forward step host bound?  Enqueue time vs GPU time of bench.py's step at C2, and a cProfile of the enqueue.
    python tools/step_host_profile.py [--profile]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile
dev = torch.device("cuda:0")
specs, names, lambdas, last = synthetic_bank_spec({"cy": 6, "cone": 5, "neg": 5})
model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
apply_bank_spec(model, specs, names, lambdas, last)
model = model.to(dev)
tiles, labels = zip(*[synthetic_tile(i, 100_000) for i in range(32)])
batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
pipe = sna.ScenePipeline(model, (64, 64, 64))
def step():   # (bench.py's step: K2 rides in K1's first launch)
    _, _, bank, prep = rider = model.bank_rider(dev)
    lam = model.effective_lambdas(dev)
    grids = pipe.voxelize(batch, bank_rider=rider)
    return model.contract_prepared(grids.occ, bank, lam, prep)[1]
import gc
gc.collect(); gc.freeze()
with torch.no_grad():
    for _ in range(50): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    while time.perf_counter() - t < 0.3:
        for _ in range(20): step()
        torch.cuda.synchronize()
    for n in (50, 300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{n} steps: enqueue {(t1 - t0) / n * 1e6:.1f} us/step, until the GPU is done {(t2 - t0) / n * 1e6:.1f} us/step")
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(300): step()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
