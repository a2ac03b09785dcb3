#!/usr/bin/env python3
"""Sizing the per-parameter bar of tests/test_gpu_bf16.py::test_training_step_with_bf16_activations.

A scalar's gradient is  g_p = sum_g sum_tap lambda_g dK_g/dp[tap] * C[tap],  C[tap] = sum_v delta_v x[v + tap]  (binary x),
delta_v = dL/dpred_v * (1 - out_v^2) [out_v > 0].  bf16 storage rounds out_v and dL/dpred_v once each (2^-9 relative), so
|dC[tap]| <= eps * C_abs[tap] with C_abs = the same correlation over |delta|, and
    |g16_p - g32_p| <= eps * B_p,   B_p = sum_g sum_tap |lambda_g dK_g/dp[tap]| C_abs[tap]      (no cancellation left).
This script prints |g16 - g32| / B_p per parameter in units of 2^-9 -- the measured eps -- for three seeds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_net_amd as sna
from scene_net_amd import _hip
from scene_net_amd.synthetic import synthetic_tile
dev = torch.device("cuda:0")


def abs_sum_bounds(model, x, gout, out, ks=(9, 9, 9)):
    """B_p per trainable scalar (name -> float), see above; Jacobians of the bank by central differences"""
    C_abs = _hip.conv_corr(x, gout.float().abs().contiguous(), out.float().contiguous(), ks)        # [kz,kx,ky] >= 0
    lam = model.effective_lambdas(dev).clone()
    names = list(model.geneos)
    last = names.index(model.last_lambda.replace("lambda_", "", 1))
    bank0 = model.compute_bank(dev).clone()
    B = {}
    with torch.no_grad():
        for g, gname in enumerate(names):
            for pname, p in model.geneos[gname].geneo_params.items():
                if not p.requires_grad:
                    continue
                v0 = float(p)
                h = 1e-3 * max(1.0, abs(v0))
                p.fill_(v0 + h); kp = model.compute_bank(dev)[g].clone()
                p.fill_(v0 - h); km = model.compute_bank(dev)[g].clone()
                p.fill_(v0)
                J = (kp - km) / (2 * h)
                B[f"geneos.{gname}.geneo_params.{pname}"] = float((lam[g].abs() * J.abs() * C_abs).sum())
        for g, gname in enumerate(names):   # dL/dlambda_g = <K_g, C> - <K_last, C> (the frozen coefficient is 1 - sum of the others)
            if g == last:
                continue
            B[f"lambdas_dict.lambda_{gname}"] = float(((bank0[g].abs() + bank0[last].abs()) * C_abs).sum())
    return B


if __name__ == "__main__":
    for seed in (5, 6, 7):
        tiles, labels = zip(*[synthetic_tile(40 + i + 10 * seed, 20_000) for i in range(4)])
        batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
        grads, keep = {}, {}
        for dt in (None, torch.bfloat16):
            torch.manual_seed(seed)
            model = sna.SceneNet({"cy": 2, "cone": 2, "neg": 1}, (9, 9, 9)).to(dev)
            model.activation_dtype = dt
            pipe = sna.ScenePipeline(model, (32, 32, 64), keep_labels=[15.0])
            grids = pipe.voxelize(batch, want_gt=True)
            crit = sna.GENEO_Tversky_Loss(targets=torch.tensor([0.0, 1.0]), weighting_scheme_path=None, save_weighting_scheme=False)
            pred = model(grids.occ)
            pred.retain_grad()
            loss = crit(pred, grids.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
            loss.backward()
            grads[dt] = {n: p.grad.item() for n, p in model.named_parameters() if p.grad is not None}
            if dt is None:
                keep = dict(model=model, x=grids.occ, gout=pred.grad.detach(), out=pred.detach())
        B = abs_sum_bounds(keep["model"], keep["x"], keep["gout"], keep["out"])
        g32, g16 = grads[None], grads[torch.bfloat16]
        rows = []
        for n in g32:
            if n in B and B[n] > 0:
                rows.append((abs(g16[n] - g32[n]) / B[n] * 512, abs(g16[n] - g32[n]) / (abs(g32[n]) + 1e-30), abs(g32[n]) / B[n], n))
        rows.sort(reverse=True)
        print(f"seed {seed}: |g16-g32| / B_p in units of 2^-9 (then: relative to |g32|, |g32| / B_p)")
        for r in rows:
            print(f"   {r[0]:8.3f}  {r[1]:9.5f}  {r[2]:8.4f}  {r[3]}")
