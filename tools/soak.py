#!/usr/bin/env python3
"""Soak: the hot path (contraction, linear, skip-empty) and a captured training step run for N iterations each; outputs
must stay bit-identical to the first iteration's (deterministic kernels), no launch may fail.
python tools/soak.py [--iters 3000]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scene_net_amd as sna  # noqa: E402
from scene_net_amd import _hip  # noqa: E402
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000)
args = ap.parse_args()
dev = torch.device("cuda:0")
specs, names, lambdas, last = synthetic_bank_spec()
torch.manual_seed(0)
model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
apply_bank_spec(model, specs, names, lambdas, last)
model = model.to(dev)
tiles, labels = zip(*[synthetic_tile(i, 100_000) for i in range(32)])
batch = sna.PointBatch.from_tiles(tiles, labels, device=dev)
pipe = sna.ScenePipeline(model, (64,) * 3, keep_labels=[15.0])


def contraction():
    g = pipe.voxelize(batch)
    return _hip.conv_bank(g.occ, model.compute_bank(dev), model.effective_lambdas(dev), want_act=False, want_out=True)[1]


def linear():
    with torch.no_grad():
        return pipe(batch)


def headline():   # bench.py's step: K2 riding in K1's first launch, the z-walk on the prepared blob
    with torch.no_grad():
        _, _, bank, prep = rider = model.bank_rider(dev)
        g = pipe.voxelize(batch, bank_rider=rider)
        return model.contract_prepared(g.occ, bank, model.effective_lambdas(dev), prep)[1]


def run(name, fn):
    ref = fn().clone()
    bad = 0
    for i in range(args.iters):
        out = fn()
        if i % 25 == 24:
            bad += int(not torch.equal(out, ref))
        if i % 500 == 499:
            print(f"{name}: {i + 1} iterations, mismatches so far {bad}", flush=True)
    torch.cuda.synchronize()
    assert bad == 0, name


run("headline step (K1 + riders, K3'z)", headline)
assert _hip.conv_i8_spin_timeouts() == 0 and _hip.device_status()[0] == 0
run("contraction (K3')", contraction)
run("linear (K3L)", linear)
_hip.set_option("conv_skip_empty_tiles", 1)
try:
    run("contraction, skip-empty", contraction)
finally:
    _hip.set_option("conv_skip_empty_tiles", 0)
print("soak ok")
