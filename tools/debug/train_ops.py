#!/usr/bin/env python3
"""Which torch operators (hence which small launches) one eager training step issues besides the C ABI's kernels."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_net_amd as sna
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile
from scene_net_amd.training import voxelize_and_forward
dev = torch.device("cuda:0")
GENEO = {"cy": 6, "cone": 5, "neg": 5}
specs, names, lambdas, last = synthetic_bank_spec(GENEO)
model = sna.SceneNet(GENEO, (9, 9, 9)); apply_bank_spec(model, specs, names, lambdas, last); model = model.to(dev)
tiles, labels = zip(*[synthetic_tile(i, 100_000) for i in range(8)])
batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
pipe = sna.ScenePipeline(model, (64, 64, 64), keep_labels=[15.0])
gt = pipe.voxelize(batch, want_gt=True).gt_occ
crit = sna.GENEO_Tversky_Loss(targets=gt.float(), weighting_scheme_path=None, save_weighting_scheme=False)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    g, out = voxelize_and_forward(pipe, batch)
    loss = crit(out, g.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print(f"{e.count:4d}  {e.key}")

for ev in prof.events():
    if ev.name in ("aten::zeros", "aten::fill_", "aten::_to_copy", "aten::add", "aten::mul", "aten::add_", "aten::clone"):
        print(ev.name, [f.split("/")[-1] for f in (ev.stack or [])[:6]])
