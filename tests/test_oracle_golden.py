"""Pins the oracle: every golden vector dumped from the reference (tests/golden/make_golden.py) must be
reproduced by oracle/ -- kernels bit for bit, SceneNet.forward bit for bit (fp64), normalize_xyz / ToFullDense
bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import geneo_oracle as go
from oracle import voxel_oracle as vo


def _kernel_cases(golden_dir):
    with open(os.path.join(golden_dir, "geneo_kernels_meta.json")) as f:
        return json.load(f)


def test_every_reference_kernel_bit_exact(golden_dir):
    K = np.load(os.path.join(golden_dir, "geneo_kernels.npz"))
    cases = _kernel_cases(golden_dir)
    assert len(cases) >= 90
    for m in cases:
        mine = go.geneo_kernel(m["kind"], m["kernel_size"], m["params"]).numpy()
        ref = K[m["key"]]
        assert mine.shape == ref.shape == tuple(m["kernel_size"]), m["key"]
        assert mine.dtype == np.float32
        assert np.array_equal(mine, ref), m["key"]


def test_kernel_invariants(golden_dir):
    # the reference's commented-out asserts (cylinder.py:98-101) and neg-sphere's sum (neg_sphere.py:181-182)
    for m in _kernel_cases(golden_dir):
        k = go.geneo_kernel(m["kind"], m["kernel_size"], m["params"]).double()
        if m["kind"] in ("cy", "cone", "cy_v1", "cone_v1"):
            assert k.sum(dim=(1, 2)).abs().max() < 1e-4, m["key"]  # every z-slice sums to 0 (fp32 rounding)
        elif m["kind"] == "neg":
            assert abs(k.sum().item() + m["params"]["neg_factor"]) < 1e-4, m["key"]
        else:  # neg_v1: sum_zero(.) - neg_factor on every element
            assert abs(k.sum().item() + m["params"]["neg_factor"] * k.numel()) < 1e-3, m["key"]
        if m["kind"] in ("cy", "cy_v1"):
            assert torch.equal(k[0], k[-1])


@pytest.mark.parametrize("tag", ["ckpt955", "c1_999", "g16_999", "even_656"])
def test_scenenet_forward_bit_exact(golden_dir, tag):
    F = np.load(os.path.join(golden_dir, "geneo_forward.npz"))
    with open(os.path.join(golden_dir, "geneo_forward_meta.json")) as f:
        meta = json.load(f)[tag]
    names = [str(n) for n in F[f"{tag}/names"]]
    specs = [(n.split("_")[0], meta["geneo_params"][n]) for n in names]
    ks = tuple(int(k) for k in F[f"{tag}/kernel_size"])
    x = torch.from_numpy(F[f"{tag}/x"].astype(np.float64))
    bank = go.geneo_bank(specs, ks)
    assert np.array_equal(bank.numpy(), F[f"{tag}/bank"])
    out, conv = go.scenenet_forward(x, specs, ks, F[f"{tag}/lambdas"], int(F[f"{tag}/last"]), return_bank=True,
                                    names=names)
    assert np.array_equal(conv.numpy(), F[f"{tag}/conv"])
    assert np.array_equal(out.numpy(), F[f"{tag}/out"])
    lam = go.effective_lambdas(F[f"{tag}/lambdas"], int(F[f"{tag}/last"]), names)
    assert lam[int(F[f"{tag}/last"])].item() == pytest.approx(float(F[f"{tag}/lambda_last_after"]), abs=0)
    assert out.min() >= 0 and out.max() < 1  # relu(tanh(.))


def test_v1_module_forward_bit_exact(golden_dir):
    """SCENE_Net (v1 module) forward, SCENE_Net.py:209-226, with the v1 generators."""
    F = np.load(os.path.join(golden_dir, "geneo_forward_v1.npz"))
    with open(os.path.join(golden_dir, "geneo_forward_v1_meta.json")) as f:
        meta = json.load(f)
    names = [str(n) for n in F["names"]]
    specs = [(n.split("_")[0] + "_v1", meta["geneo_params"][n]) for n in names]
    ks = tuple(int(k) for k in F["kernel_size"])
    x = torch.from_numpy(F["x"].astype(np.float64))
    out, conv = go.scenenet_forward(x, specs, ks, F["lambdas"], int(F["last"]), return_bank=True, names=names)
    assert np.array_equal(go.geneo_bank(specs, ks).numpy(), F["bank"])
    assert np.array_equal(conv.numpy(), F["conv"]) and np.array_equal(out.numpy(), F["out"])


def test_linearity_fast_path_matches(golden_dir):
    # out == relu(tanh(conv3d(x, sum_i lambda_i K_i)))
    F = np.load(os.path.join(golden_dir, "geneo_forward.npz"))
    tag = "g16_999"
    bank = torch.from_numpy(F[f"{tag}/bank"])
    names = [str(n) for n in F[f"{tag}/names"]]
    lam = go.effective_lambdas(F[f"{tag}/lambdas"], int(F[f"{tag}/last"]), names).double()
    mixed = (lam.view(-1, 1, 1, 1, 1) * bank).sum(0, keepdim=True)
    x = torch.from_numpy(F[f"{tag}/x"].astype(np.float64))
    out = torch.relu(torch.tanh(go.conv_bank(x, mixed)))
    assert (out - torch.from_numpy(F[f"{tag}/out"])).abs().max() < 1e-12


@pytest.mark.parametrize("case", ["sparse", "dense", "fullcol", "constcol", "empty"])
def test_normalize_and_fulldense_bit_exact(golden_dir, case):
    N = np.load(os.path.join(golden_dir, "voxel_normalize.npz"))
    counts = N[f"{case}/counts"]
    norm = vo.normalize_xyz(counts)
    assert np.array_equal(norm, N[f"{case}/norm"])
    assert np.array_equal(vo.to_full_dense(norm), N[f"{case}/dense"])
    # occupancy == (count > column minimum): the rule the HIP finalize kernel implements
    colmin = counts.reshape(-1, counts.shape[-1]).min(0)
    assert np.array_equal(N[f"{case}/dense"], (counts > colmin).astype(np.float64))
