#!/usr/bin/env python3
"""Host profile of the eager training step WITH the rider path (training.voxelize_and_forward), cProfile by own time."""
import cProfile, gc, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_net_amd as sna
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile
from scene_net_amd.training import backward_seeded, voxelize_and_forward
dev = torch.device("cuda:0")
geneo_num = {"cy": 6, "cone": 5, "neg": 5}
specs, names, lambdas, last = synthetic_bank_spec(geneo_num)
torch.manual_seed(0)
model = sna.SceneNet(geneo_num, (9, 9, 9)); apply_bank_spec(model, specs, names, lambdas, last); model = model.to(dev)
tiles, labels = zip(*[synthetic_tile(i, 100_000) for i in range(32)])
batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
pipe = sna.ScenePipeline(model, (64,) * 3, keep_labels=[15.0])
g0 = pipe.voxelize(batch, want_gt=True)
crit = sna.GENEO_Tversky_Loss(targets=g0.gt_occ.float().cpu(), weighting_scheme_path=None, save_weighting_scheme=False)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    g, out = voxelize_and_forward(pipe, batch)
    loss = crit(out, g.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
    backward_seeded(loss)
    opt.step()
    return loss
for _ in range(20): step()
torch.cuda.synchronize(); gc.collect(); gc.freeze()
t = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
print(f"eager step {(time.perf_counter() - t) / 300 * 1e3:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
