"""The folded int8 kernel (conv_occ_i8f_kernel, conv_i8s.hip): banks that are bit-for-bit symmetric in x and y -- every
GENEO bank: the generators are radial in (x, y), cylinder.py:152-176, arrow.py:214-252, neg_sphere.py:166-199 -- are
contracted over 9 x 5 x 5 folded taps.  The integer sums are the SAME integers, so the result must equal the unfolded
stride-4 kernel's bit for bit; the device decides per call (symmetry check on the fp32 weights), and a bank that is
off by one ulp anywhere must take the unfolded kernel."""
import numpy as np
import pytest
import torch

import scene_net_amd as sna
from scene_net_amd import _hip
from oracle import geneo_oracle as go

from conftest import act_err_ok

pytestmark = pytest.mark.gpu
TOL = 1e-4


class fold:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        _hip.set_option("conv_i8_fold", 1 if self.on else 0)

    def __exit__(self, *a):
        _hip.set_option("conv_i8_fold", 1)


def _symmetric_bank(G, seed, scale=None):
    g = torch.Generator().manual_seed(seed)
    w = torch.rand((G, 9, 9, 9), generator=g) - 0.5
    w = w + w.flip(2)           # a + b == b + a: symmetric in x bit for bit
    w = w + w.flip(3)           # and in y, keeping the x symmetry (both mirrors add the same two numbers)
    if scale is None:
        scale = torch.logspace(-2, 0.3, G)
    return (w * scale.view(G, 1, 1, 1)).float().contiguous()


def _run(x, bank, lam, want_act=True, dt=torch.float32):
    return _hip.conv_bank(x, bank, lam, want_act=want_act, want_out=True, out_dtype=dt)


def _delta(before, after):
    return tuple(a - b for a, b in zip(after, before))


def test_bank_helper_is_symmetric():
    b = _symmetric_bank(16, 0)
    assert torch.equal(b, b.flip(2)) and torch.equal(b, b.flip(3))


@pytest.mark.parametrize("shape,G", [((2, 1, 16, 16, 64), 16), ((1, 1, 20, 18, 64), 5), ((4, 1, 32, 32, 64), 16),
                                     ((1, 1, 64, 64, 64), 16), ((1, 1, 12, 10, 64), 33), ((1, 1, 9, 24, 128), 16)])
def test_folded_equals_unfolded_and_oracle(hip_device, shape, G):
    torch.manual_seed(hash((shape, G)) % 2**31)
    occ = torch.rand(shape) < 0.3
    bank = _symmetric_bank(G, G + shape[2])
    lam = (torch.rand(G) - 0.3) / G
    ref_act = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
    ref_out = torch.relu(torch.tanh((lam.double().view(1, G, 1, 1, 1) * ref_act).sum(1, keepdim=True)))
    x, b, l = occ.to(hip_device), bank.to(hip_device), lam.to(hip_device)
    c0 = _hip.conv_i8_path_counts()
    with fold(True):
        act_f, out_f = _run(x, b, l)
        _, only_f = _run(x, b, l, want_act=False)
        act_d, out_d = _run(x, b, l, dt=torch.float64)
    served, declined, routed = _delta(c0, _hip.conv_i8_path_counts())
    groups = (G + 15) // 16
    assert served == 3 * groups and declined == 0 and routed == 0, (served, declined, routed)
    with fold(False):
        act_u, out_u = _run(x, b, l)
        _, only_u = _run(x, b, l, want_act=False)
    assert _delta(c0, _hip.conv_i8_path_counts())[0] == 3 * groups   # the folded kernel was not even tried
    assert torch.equal(act_f, act_u) and torch.equal(out_f, out_u) and torch.equal(only_f, only_u)
    assert torch.equal(only_f, out_f)
    assert act_err_ok(act_f, ref_act, TOL)
    assert (out_f.cpu().double() - ref_out).abs().max().item() < TOL
    assert act_err_ok(act_d, ref_act, TOL)
    assert (out_d.cpu() - ref_out).abs().max().item() < TOL


def test_one_ulp_off_symmetry_takes_the_unfolded_kernel(hip_device):
    torch.manual_seed(5)
    occ = torch.rand((1, 1, 16, 16, 64)) < 0.4
    bank = _symmetric_bank(16, 77)
    lam = (torch.rand(16) - 0.3) / 16
    x, l = occ.to(hip_device), lam.to(hip_device)
    for where in [(0, 0, 0, 0), (7, 4, 8, 3), (15, 8, 2, 8), (3, 5, 4, 0)]:
        b = bank.clone()
        v = b[where]
        b[where] = torch.nextafter(v, v + 1)   # one ulp: the mirrored tap no longer matches (or it is its own mirror)
        own_mirror = where[2] == 4 and where[3] == 4
        c0 = _hip.conv_i8_path_counts()
        with fold(True):
            act, out = _run(x, b.to(hip_device), l)
        served, declined, routed = _delta(c0, _hip.conv_i8_path_counts())
        assert (served, declined) == ((1, 0) if own_mirror else (0, 1)), (where, served, declined)
        with fold(False):
            act_u, out_u = _run(x, b.to(hip_device), l)
        assert torch.equal(act, act_u) and torch.equal(out, out_u)
        ref = go.conv_bank(occ.double(), b.double().unsqueeze(1))
        assert act_err_ok(act, ref, TOL)


def test_geneo_banks_are_served_folded(hip_device):
    """The bank the module builds on the device (sn_geneo_bank) is symmetric bit for bit, every kernel family."""
    from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile
    specs, names, lambdas, last = synthetic_bank_spec()
    model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
    apply_bank_spec(model, specs, names, lambdas, last)
    model = model.to(hip_device)
    bank, lam = model.compute_bank(hip_device), model.effective_lambdas(hip_device)
    assert torch.equal(bank, bank.flip(2)) and torch.equal(bank, bank.flip(3))
    tiles = [synthetic_tile(i, 40_000)[0] for i in range(2)]
    occ = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles, device=hip_device), (64,) * 3, occ_dtype=torch.bool).occ
    c0 = _hip.conv_i8_path_counts()
    act, out = _run(occ, bank, lam)
    assert _delta(c0, _hip.conv_i8_path_counts()) == (1, 0, 0)
    with fold(False):
        act_u, out_u = _run(occ, bank, lam)
    assert torch.equal(act, act_u) and torch.equal(out, out_u)
    ref = go.conv_bank(occ.cpu().double(), bank.cpu().double().unsqueeze(1))
    assert act_err_ok(act, ref, TOL)


def test_guard_routes_a_symmetric_bank_it_cannot_serve(hip_device):
    """A symmetric bank whose quantisation bound exceeds the tolerance: the folded kernel hands it to the fp32 kernel."""
    occ = torch.rand((1, 1, 16, 16, 64)) < 0.5
    bank = _symmetric_bank(16, 3, scale=torch.full((16,), 40.0))
    lam = torch.full((16,), 1.0 / 16)
    x, b, l = occ.to(hip_device), bank.to(hip_device), lam.to(hip_device)
    c0 = _hip.conv_i8_path_counts()
    act, out = _run(x, b, l)
    served, declined, routed = _delta(c0, _hip.conv_i8_path_counts())
    assert (served, declined, routed) == (0, 0, 1)
    ref = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
    # (routed to the fp32 matrix pipe: its accumulation error scales with the sum of |terms|, ~300 here: tensor-scale bar)
    assert (act.cpu().double() - ref).abs().max().item() < TOL * max(1.0, ref.abs().max().item())
    fp32 = _hip.conv_bank(x.view(torch.uint8), b, l, want_act=True, want_out=True)
    assert torch.equal(act, fp32[0]) and torch.equal(out, fp32[1])


def test_full_c2_batch_and_128_cubed_folded_equals_unfolded(hip_device):
    """BASELINE C2's batch (32 x 64^3: eight tiles per workgroup, the halo ring wraps twice) and two 128^3 tiles (C3's tile
    size): the folded kernel serves them and accumulates the same integers as the stride-4 kernel."""
    from scene_net_amd.synthetic import synthetic_bank_spec
    specs, names, lambdas, last = synthetic_bank_spec()
    bank = go.geneo_bank(specs, (9, 9, 9))[:, 0].float().contiguous()
    lam = go.effective_lambdas(lambdas, last, names).float()
    assert torch.equal(bank, bank.flip(2)) and torch.equal(bank, bank.flip(3))   # the oracle's bank is symmetric too
    b, l = bank.to(hip_device), lam.to(hip_device)
    torch.manual_seed(11)
    for shape in [(32, 1, 64, 64, 64), (2, 1, 128, 128, 128)]:
        x = torch.rand(shape, device=hip_device) < 0.04
        x[1] = True      # one completely full tile
        x[0, :, : shape[2] // 2] = False
        c0 = _hip.conv_i8_path_counts()
        outs = [_hip.conv_bank(x, b, l, want_act=False, want_out=True)[1] for _ in range(3)]
        assert _delta(c0, _hip.conv_i8_path_counts()) == (3, 0, 0)
        with fold(False):
            ref = _hip.conv_bank(x, b, l, want_act=False, want_out=True)[1]
        for o in outs:
            assert torch.equal(o, ref)


def test_random_shapes_folded_equals_unfolded(hip_device):
    """Seeded sweep over grid extents (ragged in z and x, y any multiple of 16 -- partial y tiles, tiles smaller than a
    workgroup's), batch sizes and kernel counts: folded == unfolded bit for bit, and both within tolerance of the oracle."""
    rng = np.random.default_rng(20260104)
    for case in range(14):
        B = int(rng.integers(1, 4))
        Z, X = int(rng.integers(1, 41)), int(rng.integers(1, 41))
        Y = 16 * int(rng.integers(1, 8))
        G = int(rng.choice([1, 3, 7, 16, 17]))
        occ = torch.from_numpy(rng.random((B, 1, Z, X, Y)) < rng.choice([0.02, 0.3, 0.9]))
        bank = _symmetric_bank(G, 1000 + case)
        lam = torch.from_numpy(((rng.random(G) - 0.3) / G).astype(np.float32))
        x, b, l = occ.to(hip_device), bank.to(hip_device), lam.to(hip_device)
        c0 = _hip.conv_i8_path_counts()
        act_f, out_f = _run(x, b, l)
        served, declined, routed = _delta(c0, _hip.conv_i8_path_counts())
        assert (served, declined, routed) == ((G + 15) // 16, 0, 0), (case, (B, Z, X, Y, G), served, declined, routed)
        with fold(False):
            act_u, out_u = _run(x, b, l)
        assert torch.equal(act_f, act_u) and torch.equal(out_f, out_u), (case, (B, Z, X, Y, G))
        ref_act = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
        ref_out = torch.relu(torch.tanh((lam.double().view(1, G, 1, 1, 1) * ref_act).sum(1, keepdim=True)))
        assert act_err_ok(act_f, ref_act, TOL), case
        assert (out_f.cpu().double() - ref_out).abs().max().item() < TOL, case
