"""K3 parity on the MI355X: the MFMA bank convolution + fused head (through the C ABI) against the reference's
golden forward vectors, against the oracle on seeded inputs, and through size-independent properties at
BASELINE's full size.  Tolerance (north_star): activations within 1e-4 of the fp64 reference."""
import json
import os

import numpy as np
import pytest
import torch

import scene_net_amd as sna
from scene_net_amd import _hip
from oracle import geneo_oracle as go

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rand_bank(G, ks, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand((G,) + tuple(ks), generator=g) - 0.5).to(dev)


@pytest.mark.parametrize("tag", ["ckpt955", "c1_999", "g16_999", "even_656"])
def test_golden_forward(hip_device, golden_dir, tag):
    F = np.load(os.path.join(golden_dir, "geneo_forward.npz"))
    names = [str(n) for n in F[f"{tag}/names"]]
    bank = torch.from_numpy(F[f"{tag}/bank"][:, 0]).float().to(hip_device).contiguous()
    lam = go.effective_lambdas(F[f"{tag}/lambdas"], int(F[f"{tag}/last"]), names).to(hip_device)
    x = torch.from_numpy(F[f"{tag}/x"].astype(np.float64)).to(hip_device)
    act, out = _hip.conv_bank(x, bank, lam, want_act=True, want_out=True)
    assert out.dtype == torch.float64 and out.shape == x.shape
    e_act = (act.cpu() - torch.from_numpy(F[f"{tag}/conv"])).abs().max().item()
    e_out = (out.cpu() - torch.from_numpy(F[f"{tag}/out"])).abs().max().item()
    print(tag, "act err", e_act, "out err", e_out)
    assert e_act < TOL and e_out < TOL


@pytest.mark.parametrize("shape,ks,G", [
    ((2, 1, 16, 16, 16), (9, 9, 9), 16),
    ((1, 1, 13, 9, 70), (9, 9, 9), 16),      # ragged: not multiples of the tile, y > 64
    ((3, 1, 8, 10, 33), (9, 5, 5), 3),       # reference default bank
    ((1, 1, 20, 6, 130), (6, 5, 6), 5),      # even dims: pad left (k-1)//2, right k//2
    ((2, 1, 5, 4, 3), (3, 3, 3), 1),         # grid smaller than the kernel halo
    ((1, 1, 9, 9, 9), (1, 1, 1), 2),         # pointwise
    ((1, 1, 12, 12, 12), (5, 7, 3), 7),
    ((1, 1, 7, 7, 40), (3, 3, 25), 4),       # widest supported ky
])
def test_against_oracle_random_fp32(hip_device, shape, ks, G):
    torch.manual_seed(hash((shape, ks, G)) % 2**31)
    x = torch.randn(shape)  # arbitrary (non-binary) input
    bank = _rand_bank(G, ks, 5, "cpu")
    lam = (torch.rand(G) - 0.3) / G
    ref_act = go.conv_bank(x.double(), bank.double().unsqueeze(1))
    ref_out = torch.relu(torch.tanh((lam.double().view(1, G, 1, 1, 1) * ref_act).sum(1, keepdim=True)))
    act, out = _hip.conv_bank(x.to(hip_device), bank.to(hip_device).contiguous(), lam.to(hip_device),
                              want_act=True, want_out=True)
    scale = max(1.0, ref_act.abs().max().item())
    assert (act.cpu().double() - ref_act).abs().max().item() < TOL * scale
    assert (out.cpu().double() - ref_out).abs().max().item() < TOL


def test_input_and_output_dtypes(hip_device):
    torch.manual_seed(1)
    occ = (torch.rand(2, 1, 16, 12, 20) < 0.1)
    bank = _rand_bank(16, (9, 9, 9), 2, hip_device).contiguous()
    lam = torch.full((16,), 1 / 16, device=hip_device)
    outs = {}
    for name, x in (("u8", occ.to(torch.uint8)), ("f32", occ.float()), ("f64", occ.double())):
        act, out = _hip.conv_bank(x.to(hip_device), bank, lam, want_act=True, want_out=True)
        outs[name] = (act.double().cpu(), out.double().cpu(), out.dtype)
    assert outs["u8"][2] == torch.float32 and outs["f32"][2] == torch.float32 and outs["f64"][2] == torch.float64
    for name in ("f32", "f64"):
        assert torch.equal(outs[name][0].float(), outs["u8"][0].float())  # same fp32 arithmetic inside
        assert torch.equal(outs[name][1].float(), outs["u8"][1].float())
    _, o32 = _hip.conv_bank(occ.double().to(hip_device), bank, lam, out_dtype=torch.float32)
    assert o32.dtype == torch.float32


def test_only_act_or_only_out(hip_device):
    x = (torch.rand(1, 1, 10, 10, 10) < 0.2).float().to(hip_device)
    bank = _rand_bank(4, (3, 3, 3), 3, hip_device).contiguous()
    lam = torch.tensor([0.1, 0.2, 0.3, 0.4], device=hip_device)
    act, none = _hip.conv_bank(x, bank, None, want_act=True, want_out=False)
    assert none is None and act.shape == (1, 4, 10, 10, 10)
    none, out = _hip.conv_bank(x, bank, lam, want_act=False, want_out=True)
    assert none is None
    ref = torch.relu(torch.tanh((lam.view(1, 4, 1, 1, 1) * act).sum(1, keepdim=True)))
    assert (out - ref).abs().max().item() < 1e-6
    with pytest.raises(sna.HipLibraryError):
        _hip.conv_bank(x, bank, None, want_act=False, want_out=True)  # out needs lambdas


def test_full_size_c2_tile_against_oracle(hip_device):
    """One tile at BASELINE C2 size (64^3, 16 kernels of 9^3) against the fp64 oracle."""
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
    G = len(specs)
    lam = rng.uniform(-2 / G, 1 / G, G).astype(np.float32)
    names = [f"{k}_{i}" for i, (k, _) in enumerate(specs)]
    x = torch.from_numpy((rng.random((1, 1, 64, 64, 64)) < 0.035).astype(np.float64))
    ref_out, ref_act = go.scenenet_forward(x, specs, (9, 9, 9), lam, 3, return_bank=True, names=names)
    bank = go.geneo_bank(specs, (9, 9, 9))[:, 0].float().to(hip_device).contiguous()
    lam_eff = go.effective_lambdas(lam, 3, names).to(hip_device)
    act, out = _hip.conv_bank(x.to(hip_device), bank, lam_eff, want_act=True, want_out=True)
    e_act = (act.cpu() - ref_act).abs().max().item()
    e_out = (out.cpu() - ref_out).abs().max().item()
    print("C2 tile: act err", e_act, "out err", e_out, "act max", ref_act.abs().max().item())
    assert e_act < TOL and e_out < TOL


def test_full_size_properties_batch32(hip_device):
    """BASELINE C2 batch (32 x 64^3, G = 16, 9^3): linearity over disjoint occupancy, batch independence,
    shift equivariance, output range -- no oracle needed at this size."""
    torch.manual_seed(3)
    B = 32
    bank = _rand_bank(16, (9, 9, 9), 11, hip_device).contiguous() * 0.05
    lam = ((torch.rand(16) - 0.6) / 4).to(hip_device)
    occ = torch.rand(B, 1, 64, 64, 64, device=hip_device) < 0.035
    half = torch.rand(B, 1, 64, 64, 64, device=hip_device) < 0.5
    x, x1, x2 = occ.to(torch.uint8), (occ & half).to(torch.uint8), (occ & ~half).to(torch.uint8)
    act, out = _hip.conv_bank(x, bank, lam, want_act=True, want_out=True)
    a1, _ = _hip.conv_bank(x1, bank, lam, want_act=True, want_out=False)
    a2, _ = _hip.conv_bank(x2, bank, lam, want_act=True, want_out=False)
    assert (act - (a1 + a2)).abs().max().item() < 1e-4  # linearity
    del a1, a2
    # the head is exactly relu(tanh(sum lambda_i act_i))
    ref = torch.relu(torch.tanh((lam.view(1, 16, 1, 1, 1) * act).sum(1, keepdim=True)))
    assert (out - ref).abs().max().item() < 2e-6
    assert out.min().item() >= 0 and out.max().item() < 1
    # batch independence: tile 5 alone == tile 5 in the batch, bit for bit
    a5, o5 = _hip.conv_bank(x[5:6].contiguous(), bank, lam, want_act=True, want_out=True)
    assert torch.equal(a5, act[5:6]) and torch.equal(o5, out[5:6])
    # shift equivariance away from the borders: moving the occupancy by (3,-2,5) moves the response
    xs = torch.zeros_like(x[:1])
    xs[:, :, 11:51, 8:48, 13:53] = x[:1, :, 8:48, 10:50, 8:48]
    a_s, _ = _hip.conv_bank(xs, bank, lam, want_act=True, want_out=False)
    x0 = torch.zeros_like(x[:1])
    x0[:, :, 8:48, 10:50, 8:48] = x[:1, :, 8:48, 10:50, 8:48]
    a_0, _ = _hip.conv_bank(x0, bank, lam, want_act=True, want_out=False)
    assert (a_s[:, :, 7:55, 4:52, 9:57] - a_0[:, :, 4:52, 6:54, 4:52]).abs().max().item() < 1e-5
    # determinism
    act2, out2 = _hip.conv_bank(x, bank, lam, want_act=True, want_out=True)
    assert torch.equal(act, act2) and torch.equal(out, out2)


def test_128_cubed_tile_properties(hip_device):
    """BASELINE C3 grid (128^3): single impulse reproduces the flipped bank (cross-correlation, no flip)."""
    bank = _rand_bank(16, (9, 9, 9), 13, hip_device).contiguous()
    x = torch.zeros(1, 1, 128, 128, 128, dtype=torch.uint8, device=hip_device)
    x[0, 0, 64, 70, 100] = 1
    x[0, 0, 0, 0, 0] = 1
    x[0, 0, 127, 127, 127] = 1
    act, _ = _hip.conv_bank(x, bank, None, want_act=True, want_out=False)
    # out[v] = sum_t W[t] x[v + t - p]  ->  impulse at c puts W[t] at v = c - t + p  (p = 4)
    patch = act[0, :, 60:69, 66:75, 96:105]
    assert torch.equal(patch, bank.flip(1, 2, 3))
    assert torch.equal(act[0, :, 0:5, 0:5, 0:5], bank.flip(1, 2, 3)[:, 4:, 4:, 4:])  # corner: zero padding
    assert torch.equal(act[0, :, 123:, 123:, 123:], bank.flip(1, 2, 3)[:, :5, :5, :5])
    assert act.abs().sum().item() == pytest.approx(
        (bank.abs().sum() + bank[:, :5, :5, :5].abs().sum() + bank[:, 4:, 4:, 4:].abs().sum()).item(), rel=1e-5)


# ------------------------------------------------------------------ binary occupancy on the int8 matrix cores
@pytest.mark.parametrize("tag", ["ckpt955", "c1_999", "g16_999", "even_656"])
def test_golden_forward_occupancy_i8(hip_device, golden_dir, tag):
    """The golden inputs are binary: as torch.bool they take the int8 (24-bit fixed-point weight) kernel."""
    F = np.load(os.path.join(golden_dir, "geneo_forward.npz"))
    names = [str(n) for n in F[f"{tag}/names"]]
    bank = torch.from_numpy(F[f"{tag}/bank"][:, 0]).float().to(hip_device).contiguous()
    lam = go.effective_lambdas(F[f"{tag}/lambdas"], int(F[f"{tag}/last"]), names).to(hip_device)
    x = torch.from_numpy(F[f"{tag}/x"]).to(torch.bool).to(hip_device)
    act, out = _hip.conv_bank(x, bank, lam, want_act=True, want_out=True, out_dtype=torch.float64)
    e_act = (act.cpu() - torch.from_numpy(F[f"{tag}/conv"])).abs().max().item()
    e_out = (out.cpu() - torch.from_numpy(F[f"{tag}/out"])).abs().max().item()
    print(tag, "i8 act err", e_act, "out err", e_out)
    assert e_act < TOL and e_out < TOL


@pytest.mark.parametrize("shape,ks,G", [
    ((2, 1, 16, 16, 16), (9, 9, 9), 16),
    ((1, 1, 13, 9, 72), (9, 9, 9), 16),      # ragged z/x, y > 64
    ((3, 1, 8, 10, 36), (9, 5, 5), 3),
    ((1, 1, 20, 6, 132), (6, 5, 6), 5),      # even dims
    ((2, 1, 5, 4, 4), (3, 3, 3), 1),
    ((1, 1, 9, 9, 12), (1, 1, 1), 2),
    ((1, 1, 12, 12, 12), (5, 7, 3), 7),
    ((1, 1, 7, 7, 40), (3, 3, 24), 4),       # widest ky the int8 kernel takes
    ((1, 1, 7, 7, 40), (3, 3, 25), 4),       # ky = 25: served by the fp32 kernel (same answer)
    ((1, 1, 6, 6, 30), (3, 3, 3), 4),        # Y % 4 != 0: served by the fp32 kernel
])
def test_occupancy_i8_against_oracle(hip_device, shape, ks, G):
    torch.manual_seed(hash((shape, ks, G)) % 2**31)
    occ = torch.rand(shape) < 0.3
    bank = _rand_bank(G, ks, 5, "cpu") * torch.logspace(-3, 1, G).view(G, 1, 1, 1)  # kernels of very different scale
    lam = (torch.rand(G) - 0.3) / G
    ref_act = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
    ref_out = torch.relu(torch.tanh((lam.double().view(1, G, 1, 1, 1) * ref_act).sum(1, keepdim=True)))
    act, out = _hip.conv_bank(occ.to(hip_device), bank.to(hip_device).contiguous(), lam.to(hip_device),
                              want_act=True, want_out=True)
    assert act.dtype == torch.float32
    # per kernel: the fixed point is per kernel, so the error scales with that kernel's own largest weight; the parity
    # bar itself is relative to max(1, |act|) of that kernel
    for g in range(G):
        scale = max(1.0, ref_act[:, g].abs().max().item())
        wmax = bank[g].abs().max().item()
        err = (act[:, g].cpu().double() - ref_act[:, g]).abs().max().item()
        assert err < TOL * max(wmax, 1e-30) * 10 and err < TOL * scale, (g, err, wmax)
    assert (out.cpu().double() - ref_out).abs().max().item() < TOL
    # and the fp32 kernel on the same bytes agrees
    act32, out32 = _hip.conv_bank(occ.to(torch.uint8).to(hip_device), bank.to(hip_device).contiguous(),
                                  lam.to(hip_device), want_act=True, want_out=True)
    assert (act32 - act).abs().max().item() < TOL and (out32 - out).abs().max().item() < TOL


def test_occupancy_i8_integer_accumulation_is_exact(hip_device):
    """Integer accumulation: a bank whose weights are integer multiples of max|W| / 8355711 is represented exactly by
    the fixed point (the scale uses the whole range of three balanced digits), so the only error left is the fp32
    recombination of the sum -- and the result does not depend on the order of accumulation."""
    torch.manual_seed(9)
    occ = torch.rand(4, 1, 32, 32, 64, device=hip_device) < 0.2
    q = torch.randint(-4194304, 4194305, (16, 9, 9, 9))   # |q| <= 2^22: the fp32 rounding of W moves W * S by < 0.5
    q[:, 0, 0, 0] = 8355711                           # max|Q| = 8355711 -> S = 8355711 / max|W| reproduces q
    bank64 = q.double() * (2.0 / 8355711.0)            # max|W| = 2
    bank = bank64.float().to(hip_device).contiguous()   # fp32 rounding of the weights: rint(W * S) is still q
    act, _ = _hip.conv_bank(occ, bank, None, want_act=True, want_out=False)
    ref = go.conv_bank(occ.cpu().double(), bank64.unsqueeze(1))
    rel = ((act.cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
    assert rel < 1e-6, rel                            # a few fp32 roundings of an exactly accumulated integer (unit 2.4e-7)
    act2, _ = _hip.conv_bank(occ, bank, None, want_act=True, want_out=False)
    assert torch.equal(act, act2)


def test_full_size_c2_tile_occupancy_i8(hip_device):
    from scene_net_amd.synthetic import synthetic_bank_spec
    specs, names, lambdas, last = synthetic_bank_spec()
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.random((2, 1, 64, 64, 64)) < 0.035)
    ref_out, ref_act = go.scenenet_forward(x.double(), specs, (9, 9, 9), lambdas, last, return_bank=True, names=names)
    bank = go.geneo_bank(specs, (9, 9, 9))[:, 0].float().to(hip_device).contiguous()
    lam_eff = go.effective_lambdas(lambdas, last, names).to(hip_device)
    act, out = _hip.conv_bank(x.to(hip_device), bank, lam_eff, want_act=True, want_out=True)
    e_act = (act.cpu().double() - ref_act).abs().max().item()
    e_out = (out.cpu().double() - ref_out).abs().max().item()
    print("C2 tiles (i8): act err", e_act, "out err", e_out)
    assert e_act < TOL and e_out < TOL


@pytest.mark.parametrize("G", [17, 20, 33])
def test_more_than_16_kernels(hip_device, G):
    """Banks larger than one MFMA row block: grouped launches, partial sums carried in `out`."""
    torch.manual_seed(G)
    occ = torch.rand(2, 1, 12, 10, 24) < 0.25
    bank = _rand_bank(G, (5, 5, 5), 7, "cpu")
    lam = (torch.rand(G) - 0.4) / G
    ref_act = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
    ref_out = torch.relu(torch.tanh((lam.double().view(1, G, 1, 1, 1) * ref_act).sum(1, keepdim=True)))
    for x in (occ, occ.float(), occ.double()):  # int8 kernel, fp32 kernel (f32 and f64 I/O)
        act, out = _hip.conv_bank(x.to(hip_device), bank.to(hip_device).contiguous(), lam.to(hip_device),
                                  want_act=True, want_out=True)
        assert act.shape == (2, G, 12, 10, 24)
        assert (act.cpu().double() - ref_act).abs().max().item() < TOL
        assert (out.cpu().double() - ref_out).abs().max().item() < TOL
    torch.manual_seed(0)
    model = sna.SceneNet({"cy": 8, "cone": 6, "neg": 6}, (9, 5, 5)).to(hip_device)
    y = model(occ.to(hip_device))
    assert y.shape == occ.shape and y.min().item() >= 0 and y.max().item() <= 1  # fp32 tanh saturates to 1.0


def test_non_finite_weights_propagate_like_conv3d(hip_device):
    """A NaN weight makes that kernel's whole response NaN (0 * NaN = NaN at every voxel) and, through the head,
    the output; torch.relu keeps NaN.  Holds for the fp32 and the int8 kernel."""
    torch.manual_seed(4)
    occ = torch.rand(1, 1, 10, 10, 16) < 0.3
    bank = _rand_bank(4, (3, 3, 3), 1, "cpu")
    bank[2, 1, 1, 1] = float("nan")
    lam = torch.tensor([0.2, 0.3, 0.25, 0.25])
    ref_act = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
    assert torch.isnan(ref_act[:, 2]).all() and not torch.isnan(ref_act[:, [0, 1, 3]]).any()
    for x in (occ, occ.float()):
        act, out = _hip.conv_bank(x.to(hip_device), bank.to(hip_device).contiguous(), lam.to(hip_device),
                                  want_act=True, want_out=True)
        assert torch.isnan(act[:, 2]).all() and not torch.isnan(act[:, [0, 1, 3]]).any()
        assert torch.isnan(out).all()


def test_skip_empty_tiles_option_changes_nothing_but_time(hip_device):
    """conv_skip_empty_tiles: bit-identical output on a LiDAR-like batch with large empty regions."""
    from scene_net_amd.synthetic import synthetic_bank_spec, synthetic_tile
    specs, names, lambdas, last = synthetic_bank_spec()
    bank = go.geneo_bank(specs, (9, 9, 9))[:, 0].float().to(hip_device).contiguous()
    lam = go.effective_lambdas(lambdas, last, names).to(hip_device)
    tiles = [synthetic_tile(t, 50_000)[0] for t in range(4)]
    occ = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles, device=hip_device), (64, 64, 64),
                             occ_dtype=torch.bool).occ
    assert _hip.get_option("conv_skip_empty_tiles") == 0
    act0, out0 = _hip.conv_bank(occ, bank, lam, want_act=True, want_out=True)
    _hip.set_option("conv_skip_empty_tiles", 1)
    try:
        act1, out1 = _hip.conv_bank(occ, bank, lam, want_act=True, want_out=True)
        empty = torch.zeros_like(occ)
        a2, o2 = _hip.conv_bank(empty, bank, lam, want_act=True, want_out=True)
    finally:
        _hip.set_option("conv_skip_empty_tiles", 0)
    assert torch.equal(act0, act1) and torch.equal(out0, out1)
    assert a2.abs().max().item() == 0 and o2.abs().max().item() == 0
    with pytest.raises(sna.HipLibraryError):
        _hip.set_option("no_such_option", 1)


# ------------------------------------------------------------------ forward through linearity (sn_conv_fused)
@pytest.mark.parametrize("tag", ["ckpt955", "c1_999", "g16_999", "even_656"])
def test_golden_forward_fused_linear(hip_device, golden_dir, tag):
    """relu(tanh(conv3d(x, sum_g lambda_g K_g))) on the reference's golden forward cases: same output within 1e-4."""
    F = np.load(os.path.join(golden_dir, "geneo_forward.npz"))
    names = [str(n) for n in F[f"{tag}/names"]]
    bank = torch.from_numpy(F[f"{tag}/bank"][:, 0]).float().to(hip_device).contiguous()
    lam = go.effective_lambdas(F[f"{tag}/lambdas"], int(F[f"{tag}/last"]), names).to(hip_device)
    x = torch.from_numpy(F[f"{tag}/x"]).to(torch.bool).to(hip_device)
    if x.shape[-1] % 4:   # the kernel reads aligned dwords along y; such grids go through sn_conv_bank
        assert not _hip.conv_fused_supported(x, bank.shape[1:])
        x = torch.nn.functional.pad(x, (0, 4 - x.shape[-1] % 4))
        ref = go.conv_bank(x.double().cpu(), bank.double().cpu().unsqueeze(1))
        ref = torch.relu(torch.tanh((lam.double().cpu().view(1, -1, 1, 1, 1) * ref).sum(1, keepdim=True))).numpy()
    else:
        ref = F[f"{tag}/out"]
    assert _hip.conv_fused_supported(x, bank.shape[1:])
    for dt in (torch.float32, torch.float64):
        out = _hip.conv_fused(x, bank, lam, out_dtype=dt)
        assert out.dtype == dt and out.shape == x.shape
        e_out = (out.double().cpu() - torch.from_numpy(ref)).abs().max().item()
        print(tag, "fused out err", e_out)
        assert e_out < TOL


@pytest.mark.parametrize("shape,ks,G", [
    ((2, 1, 16, 16, 16), (9, 9, 9), 16),
    ((1, 1, 13, 9, 72), (9, 9, 9), 16),      # ragged z/x, y > 64 (two y tiles)
    ((3, 1, 8, 10, 36), (9, 5, 5), 3),
    ((1, 1, 20, 6, 132), (6, 5, 6), 5),      # even dims: pad left (k-1)//2, right k//2
    ((2, 1, 5, 4, 4), (3, 3, 3), 1),         # grid smaller than the halo
    ((1, 1, 9, 9, 12), (1, 1, 1), 2),        # pointwise
    ((1, 1, 12, 12, 12), (5, 7, 3), 7),
    ((1, 1, 7, 19, 40), (3, 3, 12), 4),      # window 15 + 11 + (8 - 5) = 29 < 32
    ((1, 1, 7, 19, 40), (3, 3, 17), 2),      # widest ky: window 15 + 16 = 31
    ((1, 1, 9, 33, 64), (9, 9, 9), 33),      # any G: the bank is combined before the convolution
    ((2, 1, 17, 17, 128), (9, 9, 9), 16),
    ((1, 1, 12, 20, 64), (11, 9, 5), 4),     # 18 x 24 = 432 halo rows: beyond the 16 register passes of 25 rows
    ((1, 1, 10, 18, 64), (9, 11, 3), 3),     # 16 x 26 = 416 halo rows
    ((1, 1, 12, 20, 64), (17, 3, 3), 4),     # 24 x 18 = 432 halo rows, a tall kernel
    ((1, 1, 9, 20, 64), (9, 7, 7), 3),       # the reference's sweep sizes (sweep_config.yml:50)
    ((1, 1, 8, 12, 32), (6, 5, 5), 3),
    ((1, 1, 8, 12, 32), (9, 6, 6), 3),       # SCENE_Net.py:30's default
])
def test_fused_linear_against_oracle_and_dense_kernel(hip_device, shape, ks, G):
    torch.manual_seed(hash((shape, ks, G)) % 2**31)
    occ = torch.rand(shape) < 0.3
    bank = _rand_bank(G, ks, 5, "cpu") * torch.logspace(-2, 0.5, G).view(G, 1, 1, 1)
    lam = (torch.rand(G) - 0.3) / G
    ref_act = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
    ref_out = torch.relu(torch.tanh((lam.double().view(1, G, 1, 1, 1) * ref_act).sum(1, keepdim=True)))
    x = occ.to(hip_device)
    assert _hip.conv_fused_supported(x, ks)
    out = _hip.conv_fused(x, bank.to(hip_device).contiguous(), lam.to(hip_device))
    assert (out.cpu().double() - ref_out).abs().max().item() < TOL
    _, dense = _hip.conv_bank(x, bank.to(hip_device).contiguous(), lam.to(hip_device), want_act=False, want_out=True)
    assert (out - dense).abs().max().item() < 2e-5      # two fixed-point roundings of different weights
    assert torch.equal(out, _hip.conv_fused(x, bank.to(hip_device).contiguous(), lam.to(hip_device)))  # deterministic


@pytest.mark.parametrize("shape,ks", [((2, 1, 24, 40, 64), (9, 9, 9)), ((1, 1, 16, 20, 72), (5, 3, 7)),
                                      ((3, 1, 9, 17, 128), (9, 5, 5)), ((1, 1, 8, 16, 64), (3, 9, 2))])
def test_fused_linear_row_packings_are_bit_identical(hip_device, shape, ks):
    """The 24-byte packing of the kernel rows (ky <= 9) and the 32-byte one accumulate the same integers."""
    torch.manual_seed(11)
    G = 5
    x = (torch.rand(shape) < 0.2).to(hip_device)
    bank = (_rand_bank(G, ks, 3, "cpu") * torch.logspace(-1, 0.3, G).view(G, 1, 1, 1)).to(hip_device).contiguous()
    lam = ((torch.rand(G) - 0.4) / G).to(hip_device)
    packed = _hip.conv_fused(x, bank, lam)
    _hip.set_option("conv_lin_no24", 1)   # (an option now: the environment is read once, never per call)
    try:
        wide = _hip.conv_fused(x, bank, lam)
    finally:
        _hip.set_option("conv_lin_no24", 0)
    assert torch.equal(packed, wide)


def test_fused_linear_rejects_what_it_does_not_serve(hip_device):
    bank = _rand_bank(2, (3, 3, 3), 1, hip_device).contiguous()
    lam = torch.tensor([0.5, 0.5], device=hip_device)
    with pytest.raises(sna.HipLibraryError, match="SN_OCC8|occupancy"):
        _hip.conv_fused(torch.rand((1, 1, 8, 8, 8), device=hip_device), bank, lam)          # not binary occupancy
    x = torch.rand((1, 1, 6, 6, 30), device=hip_device) < 0.5                                 # Y % 4 != 0
    assert not _hip.conv_fused_supported(x, (3, 3, 3))
    with pytest.raises(sna.HipLibraryError, match="sn_conv_bank"):
        _hip.conv_fused(x, bank, lam)
    assert not _hip.conv_fused_supported(torch.zeros((1, 1, 8, 8, 40), dtype=torch.bool), (3, 3, 19))  # ky > window


def test_fused_linear_full_size_equals_dense_kernel(hip_device):
    """BASELINE C2 size: the linear path and the 16-kernel contraction agree to 2e-5 everywhere."""
    from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec
    specs, names, lambdas, last = synthetic_bank_spec({"cy": 6, "cone": 5, "neg": 5})
    torch.manual_seed(0)
    model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
    apply_bank_spec(model, specs, names, lambdas, last)
    model = model.to(hip_device)
    bank, lam = model.compute_bank(hip_device), model.effective_lambdas(hip_device)
    x = torch.rand((32, 1, 64, 64, 64), device=hip_device) < 0.035
    _, dense = _hip.conv_bank(x, bank, lam, want_act=False, want_out=True)
    fused = _hip.conv_fused(x, bank, lam)
    assert (fused - dense).abs().max().item() < 2e-5
    assert fused.min().item() >= 0.0 and fused.max().item() < 1.0
