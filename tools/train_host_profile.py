#!/usr/bin/env python3
"""Where the host time of an EAGER training step goes (cProfile over 200 steps at C2; the GPU work is ~0.3 ms/step, the
eager step ~1 ms: host bound).  python tools/train_host_profile.py"""
import cProfile
import gc
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile  # noqa: E402

dev = torch.device("cuda:0")
geneo_num = {"cy": 6, "cone": 5, "neg": 5}
specs, names, lambdas, last = synthetic_bank_spec(geneo_num)
torch.manual_seed(0)
model = sna.SceneNet(geneo_num, (9, 9, 9))
apply_bank_spec(model, specs, names, lambdas, last)
model = model.to(dev)
tiles, labels = zip(*[synthetic_tile(i, 100_000) for i in range(32)])
batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
pipe = sna.ScenePipeline(model, (64,) * 3, keep_labels=[15.0])
g0 = pipe.voxelize(batch, want_gt=True)
crit = sna.GENEO_Tversky_Loss(targets=g0.gt_occ.float().cpu(), weighting_scheme_path=None, save_weighting_scheme=False)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4)


def step():
    opt.zero_grad(set_to_none=True)
    g = pipe.voxelize(batch, want_gt=True)
    out = model(g.occ)
    loss = crit(out, g.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
    loss.backward()
    opt.step()
    return loss


for _ in range(10):
    step()
torch.cuda.synchronize()
gc.collect()
gc.freeze()
import time
t = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print(f"eager step {(time.perf_counter() - t) / 200 * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
