"""K2 parity on the MI355X: the HIP bank builder (through the C ABI) against every golden kernel dumped from
the reference and against the oracle.  Tolerance: 2e-6 absolute (fp32 closed form vs the reference's
sqrt->square / pairwise-sum op order; kernels are O(1))."""
import json
import os

import numpy as np
import pytest
import torch

import scene_net_amd as sna
from scene_net_amd import _hip
from scene_net_amd.geneos import KIND_OF_CLASS, pack_params
from oracle import geneo_oracle as go

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _build(kind, ks, params, dev):
    p = pack_params(KIND_OF_CLASS[kind], params, dev).unsqueeze(0).contiguous()
    kinds = torch.tensor([KIND_OF_CLASS[kind]], dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    bank = _hip.geneo_bank(p, kinds, ks, status)
    return bank[0].cpu().numpy(), int(status.item())


def test_every_golden_kernel(hip_device, golden_dir):
    K = np.load(os.path.join(golden_dir, "geneo_kernels.npz"))
    with open(os.path.join(golden_dir, "geneo_kernels_meta.json")) as f:
        cases = json.load(f)
    worst = 0.0
    for m in cases:
        got, status = _build(m["kind"], m["kernel_size"], m["params"], hip_device)
        assert status == 0
        ref = K[m["key"]]
        assert got.shape == ref.shape
        err = np.abs(got - ref).max()
        worst = max(worst, err)
        assert err < TOL, (m["key"], err)
    print("worst kernel error", worst)


def test_whole_bank_in_one_launch(hip_device):
    rng = np.random.default_rng(7)
    specs = []
    for kind, n in (("cy", 6), ("cone", 5), ("neg", 5)):
        for _ in range(n):
            p = dict(radius=float(rng.uniform(0.5, 4)), sigma=float(rng.uniform(0.5, 2)))
            if kind == "cone":
                p.update(apex=float(rng.integers(4, 8)), cone_radius=float(rng.uniform(0.5, 4)),
                         cone_inc=float(rng.uniform(0.05, 0.45)))
            if kind == "neg":
                p.update(neg_factor=float(rng.uniform(0.1, 0.9)))
            specs.append((kind, p))
    for ks in [(9, 9, 9), (9, 5, 5), (6, 5, 6), (3, 7, 4)]:
        sp = [(k, dict(p, apex=min(p["apex"], float(ks[0]))) if k == "cone" else p) for k, p in specs]
        params = torch.stack([pack_params(KIND_OF_CLASS[k], p, hip_device) for k, p in sp]).contiguous()
        kinds = torch.tensor([KIND_OF_CLASS[k] for k, _ in sp], dtype=torch.int32, device=hip_device)
        bank = _hip.geneo_bank(params, kinds, ks).cpu()
        ref = go.geneo_bank(sp, ks)[:, 0].float()
        assert (bank - ref).abs().max() < TOL, ks


def test_apex_out_of_range_is_flagged_and_raises(hip_device):
    p = dict(radius=1.0, sigma=1.0, apex=12.0, cone_radius=2.0, cone_inc=0.2)
    _, status = _build("cone", (9, 9, 9), p, hip_device)
    assert status == 1
    with pytest.raises(RuntimeError):
        sna.arrow("cone", (9, 9, 9), **{k: torch.tensor(v) for k, v in p.items()})
    # apex == kz is legal: pure cylinder (hc = kz, no cone slices)
    got, status = _build("cone", (9, 9, 9), dict(p, apex=9.0), hip_device)
    assert status == 0
    ref = go.arrow_kernel((9, 9, 9), 1.0, 1.0, 9.0, 2.0, 0.2).numpy()
    assert np.abs(got - ref).max() < TOL


def test_geneo_classes_build_on_device(hip_device):
    cy = sna.cylinderv2("cy", (6, 7, 7), radius=torch.tensor(2.5), sigma=torch.tensor(5.0))  # cylinder.py:212
    assert cy.kernel.is_cuda and cy.kernel.shape == (6, 7, 7) and cy.kernel.dtype == torch.float32
    ref = go.cylinderv2_kernel((6, 7, 7), 2.5, 5.0)
    assert (cy.kernel.cpu() - ref).abs().max() < 1e-5  # sigma = 5 scales the absolute error
    layer = sna.GENEO_Layer(sna.negSpherev2, kernel_size=(9, 5, 5))
    k = layer.compute_kernel()
    assert k.shape == (1, 9, 5, 5) and k.dtype == torch.float64
    p = {n: float(v) for n, v in layer.geneo_params.items()}
    assert (k[0].cpu() - go.negspherev2_kernel((9, 5, 5), p["radius"], p["sigma"], p["neg_factor"]).double()).abs().max() < TOL
