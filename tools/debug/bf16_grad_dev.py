#!/usr/bin/env python3
"""How far are the bf16-activation step's gradients from the fp32 step's?  (sizing the bar of tests/test_gpu_bf16.py)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_net_amd as sna
from scene_net_amd.synthetic import synthetic_tile
dev = torch.device("cuda:0")
for seed in (5, 6, 7):
    tiles, labels = zip(*[synthetic_tile(40 + i + 10 * seed, 20_000) for i in range(4)])
    batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
    grads = {}
    for dt in (None, torch.bfloat16):
        torch.manual_seed(seed)
        model = sna.SceneNet({"cy": 2, "cone": 2, "neg": 1}, (9, 9, 9)).to(dev)
        model.activation_dtype = dt
        pipe = sna.ScenePipeline(model, (32, 32, 64), keep_labels=[15.0])
        grids = pipe.voxelize(batch, want_gt=True)
        crit = sna.GENEO_Tversky_Loss(targets=torch.tensor([0.0, 1.0]), weighting_scheme_path=None, save_weighting_scheme=False)
        loss = crit(model(grids.occ), grids.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
        loss.backward()
        grads[dt] = {n: p.grad.item() for n, p in model.named_parameters() if p.grad is not None}
    g32, g16 = grads[None], grads[torch.bfloat16]
    rel = {n: abs(g16[n] - g32[n]) / (abs(g32[n]) + 1e-12) for n in g32}
    print("seed", seed, "max rel", max(rel.values()), sorted(rel.items(), key=lambda kv: -kv[1])[:3], "min |g32|", min(abs(v) for v in g32.values()))
