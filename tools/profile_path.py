#!/usr/bin/env python3
"""Runs the hot path a few times at BASELINE C2 size; meant to sit behind `rocprofv3 ... -- python tools/profile_path.py`
(kernel trace or one --pmc pass at a time).  No timing here: bench.py measures, this only provides launches."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--points", type=int, default=100_000)
ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--train", action="store_true", help="also run training steps (correlation, criterion kernels)")
args = ap.parse_args()

dev = torch.device("cuda:0")
specs, names, lambdas, last = synthetic_bank_spec()
torch.manual_seed(0)
model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
apply_bank_spec(model, specs, names, lambdas, last)
model = model.to(dev)
tiles, labels = zip(*[synthetic_tile(i, args.points) for i in range(args.batch)])
batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
pipe = sna.ScenePipeline(model, (args.grid,) * 3, keep_labels=[15.0])
with torch.no_grad():
    for _ in range(args.iters):
        out = pipe(batch)                     # forward through linearity (K3L)
    model.fused_forward = False
    for _ in range(args.iters):
        out = pipe(batch)                     # the 16-kernel contraction (K3')
    model.fused_forward = True
if args.train:
    grids = pipe.voxelize(batch, want_gt=True)
    crit = sna.GENEO_Tversky_Loss(targets=grids.gt_occ.float().cpu(), weighting_scheme_path=None,
                                  save_weighting_scheme=False)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    for _ in range(args.iters):
        opt.zero_grad(set_to_none=True)
        g = pipe.voxelize(batch, want_gt=True)
        loss = crit(model(g.occ), g.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
        loss.backward()
        opt.step()
torch.cuda.synchronize()
print("done", float(out.sum()))
