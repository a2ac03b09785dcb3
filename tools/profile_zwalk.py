#!/usr/bin/env python3
"""A few launches of the prepared int8 contraction (z-walk kernel) at C2 size behind `rocprofv3 --pmc ... -- python3
tools/profile_zwalk.py [--variant V]` (one counter group per pass)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd import _hip  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=-1)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
specs, names, lambdas, last = synthetic_bank_spec()
model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
apply_bank_spec(model, specs, names, lambdas, last)
model = model.to(dev)
bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)
tiles = [synthetic_tile(i, 100_000)[0] for i in range(32)]
occ = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles, device=dev), (64,) * 3, occ_dtype=torch.bool).occ
prep = _hip.conv_bank_prep(bank)
if args.variant >= 0:
    _hip.set_option("conv_i8z_variant", args.variant)
_hip.set_option("conv_i8_tolerance_ppb", 0)
for _ in range(args.iters):
    _hip.conv_bank(occ, bank, lam, want_act=False, want_out=True, prep=prep)
torch.cuda.synchronize()
print("done")
