"""The z-walk kernel over pre-folded planes (conv_occ_i8z_kernel, csrc/conv_i8z.inc) behind sn_conv_bank_prepared: the same
contraction as sn_conv_bank (SceneNet.forward, core/models/SCENE_Net.py:322-339) with the per-bank work done once by
sn_conv_bank_prep.  Same integers, same head: the result must equal sn_conv_bank's bit for bit on every shape, and meet
the fp64 oracle within 1e-4.

The fp64 oracle (oracle/geneo_oracle.py: scenenet_forward, pinned to the reference's golden forwards) is met DIRECTLY on every
small shape, on tiles 0 and 31 of the full C2 batch and on one tile of C3's per-GPU share (32 x 128^3) -- no chain of
kernel-vs-kernel equalities in between (missing 2); the kernels are compared with each other bit for bit on top.

Round 4: the loud failure path (a dependency spin that gives up, a `served` launch whose bank the guard declines: NaN outputs +
the sticky device status) is tested on the product build through sn_set_option("conv_i8z_inject_fault")."""
import os
import sys

import numpy as np
import pytest
import torch

import scene_net_amd as sna
from scene_net_amd import _hip
from oracle import geneo_oracle as go

from conftest import act_err_ok

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _symmetric_bank(G, seed, scale=None):
    g = torch.Generator().manual_seed(seed)
    w = torch.rand((G, 9, 9, 9), generator=g) - 0.5
    w = w + w.flip(2)
    w = w + w.flip(3)
    if scale is None:
        scale = torch.logspace(-2, 0.3, G)
    return (w * scale.view(G, 1, 1, 1)).float().contiguous()


@pytest.fixture(params=[0, 1, 2], ids=["2rows_8waves", "1row_12waves", "1rowx2_12waves"], autouse=True)
def zwalk_variant(request):
    """every test runs on the three shapes of the walk's rounds (sn_set_option "conv_i8z_variant")"""
    default = _hip.get_option("conv_i8z_variant")
    _hip.set_option("conv_i8z_variant", request.param)
    yield request.param
    _hip.set_option("conv_i8z_variant", default)


@pytest.fixture(autouse=True)
def spin_baseline():
    """spin give-ups so far (the loud-path tests provoke some on purpose: the others assert that THEY added none)"""
    global _SPIN0
    _SPIN0 = _hip.conv_i8_spin_timeouts() if torch.cuda.is_available() else 0
    yield


_SPIN0 = 0


def _delta(before, after):
    return tuple(a - b for a, b in zip(after, before))


def _both(x, bank, lam, want_act=True, dt=torch.float32):
    prep = _hip.conv_bank_prep(bank)
    a_z, o_z = _hip.conv_bank(x, bank, lam, want_act=want_act, want_out=True, out_dtype=dt, prep=prep)
    a_r, o_r = _hip.conv_bank(x, bank, lam, want_act=want_act, want_out=True, out_dtype=dt)
    return (a_z, o_z), (a_r, o_r)


@pytest.mark.parametrize("shape,G", [((2, 1, 16, 16, 64), 16), ((1, 1, 20, 18, 64), 5), ((4, 1, 32, 32, 64), 16),
                                     ((1, 1, 64, 64, 64), 16), ((1, 1, 12, 10, 64), 33), ((1, 1, 9, 24, 128), 16),
                                     ((3, 1, 7, 5, 16), 16), ((1, 1, 1, 1, 16), 3), ((2, 1, 40, 9, 48), 16),
                                     ((1, 1, 128, 16, 32), 16)])
def test_zwalk_equals_sn_conv_bank_and_oracle(hip_device, shape, G):
    torch.manual_seed(hash((shape, G)) % 2**31)
    occ = torch.rand(shape) < 0.3
    bank = _symmetric_bank(G, G + shape[2])
    lam = (torch.rand(G) - 0.3) / G
    ref_act = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
    ref_out = torch.relu(torch.tanh((lam.double().view(1, G, 1, 1, 1) * ref_act).sum(1, keepdim=True)))
    x, b, l = occ.to(hip_device), bank.to(hip_device), lam.to(hip_device)
    c0 = _hip.conv_i8_path_counts()
    (a_z, o_z), (a_r, o_r) = _both(x, b, l)
    served, declined, routed = _delta(c0, _hip.conv_i8_path_counts())
    groups = (G + 15) // 16
    assert served == 2 * groups and declined == 0 and routed == 0, (served, declined, routed)
    assert torch.equal(a_z, a_r) and torch.equal(o_z, o_r)
    (_, only_z), (_, only_r) = _both(x, b, l, want_act=False)
    assert torch.equal(only_z, only_r) and torch.equal(only_z, o_z)
    (a_d, o_d), (a_dr, o_dr) = _both(x, b, l, dt=torch.float64)
    assert torch.equal(a_d, a_dr) and torch.equal(o_d, o_dr)
    assert act_err_ok(a_z, ref_act, TOL)
    assert (o_z.cpu().double() - ref_out).abs().max().item() < TOL
    assert (o_d.cpu() - ref_out).abs().max().item() < TOL
    assert _hip.conv_i8_spin_timeouts() == _SPIN0


def test_zwalk_full_c2_batch_and_128_cubed(hip_device):
    """BASELINE C2's batch (32 x 64^3) and one 128^3 tile (C3/C4's grid) on the bench bank: z-walk == sn_conv_bank bit for
    bit (which is itself equal to the stride-4 and four-copy kernels, tests/test_gpu_conv_fold.py)."""
    from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile
    specs, names, lambdas, last = synthetic_bank_spec()
    model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
    apply_bank_spec(model, specs, names, lambdas, last)
    model = model.to(hip_device)
    bank, lam = model.compute_bank(hip_device), model.effective_lambdas(hip_device)
    tiles = [synthetic_tile(i, 100_000)[0] for i in range(32)]
    occ = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles, device=hip_device), (64,) * 3, occ_dtype=torch.bool).occ
    c0 = _hip.conv_i8_path_counts()
    (_, o_z), (_, o_r) = _both(occ, bank, lam, want_act=False)
    assert _delta(c0, _hip.conv_i8_path_counts()) == (2, 0, 0)
    assert torch.equal(o_z, o_r)
    assert float(o_z.max()) > 0.05     # not a trivial all-zero comparison
    # the fp64 oracle inside the full batch, directly: its first and its last tile (SCENE_Net.py:322-339 restated)
    for t in (0, 31):
        ref = go.scenenet_forward(occ[t:t + 1].cpu().double(), specs, (9, 9, 9), lambdas, last, names=names)
        assert (o_z[t:t + 1].cpu().double() - ref).abs().max().item() < TOL, t
        assert float(ref.max()) > 0.05
    occ128 = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles[:2], device=hip_device), (128,) * 3,
                                occ_dtype=torch.bool).occ
    (a_z, o_z), (a_r, o_r) = _both(occ128, bank, lam, want_act=True)
    assert torch.equal(o_z, o_r) and torch.equal(a_z, a_r)
    assert _hip.conv_i8_spin_timeouts() == _SPIN0


def test_zwalk_asymmetric_bank_runs_the_unfolded_body_in_the_same_launch(hip_device):
    torch.manual_seed(5)
    occ = torch.rand((2, 1, 16, 16, 64)) < 0.4
    bank = _symmetric_bank(16, 77)
    v = bank[7, 4, 8, 3]
    bank[7, 4, 8, 3] = torch.nextafter(v, v + 1)
    lam = (torch.rand(16) - 0.3) / 16
    x, b, l = occ.to(hip_device), bank.to(hip_device), lam.to(hip_device)
    c0 = _hip.conv_i8_path_counts()
    (a_z, o_z), (a_r, o_r) = _both(x, b, l)
    assert _delta(c0, _hip.conv_i8_path_counts()) == (0, 2, 0)
    assert torch.equal(a_z, a_r) and torch.equal(o_z, o_r)
    ref = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
    assert act_err_ok(a_z, ref, TOL)


def test_zwalk_guard_routes_a_wide_bank_to_fp32(hip_device):
    """a bank whose worst-case quantisation bound exceeds the tolerance: the blob's own route flag sends the launch to the
    fp32 kernel enqueued behind it -- the same bits as the fp32 kernel called directly"""
    torch.manual_seed(6)
    occ = torch.rand((1, 1, 16, 16, 64)) < 0.5
    bank = _symmetric_bank(16, 3, scale=torch.full((16,), 40.0))
    lam = (torch.rand(16) - 0.3) / 16
    x, b, l = occ.to(hip_device), bank.to(hip_device), lam.to(hip_device)
    c0 = _hip.conv_i8_path_counts()
    prep = _hip.conv_bank_prep(b)
    a_z, o_z = _hip.conv_bank(x, b, l, want_act=True, want_out=True, prep=prep)
    assert _delta(c0, _hip.conv_i8_path_counts()) == (0, 0, 1)
    a_f, o_f = _hip.conv_bank(x.view(torch.uint8), b, l, want_act=True, want_out=True)   # u8: the fp32 kernel
    assert torch.equal(a_z, a_f) and torch.equal(o_z, o_f)
    # the same blob, tolerance off: the int8 path serves it
    _hip.set_option("conv_i8_tolerance_ppb", 0)
    try:
        a_i, o_i = _hip.conv_bank(x, b, l, want_act=True, want_out=True, prep=prep)
        a_r, o_r = _hip.conv_bank(x, b, l, want_act=True, want_out=True)
    finally:
        _hip.set_option("conv_i8_tolerance_ppb", 90000)
    assert torch.equal(a_i, a_r) and torch.equal(o_i, o_r)


def test_prep_blob_is_reusable_and_bank_specific(hip_device):
    torch.manual_seed(9)
    occ = (torch.rand((2, 1, 24, 16, 64)) < 0.2).to(hip_device)
    b1, b2 = _symmetric_bank(16, 1).to(hip_device), _symmetric_bank(16, 2).to(hip_device)
    lam = ((torch.rand(16) - 0.3) / 16).to(hip_device)
    prep = _hip.conv_bank_prep(b1)
    o1 = _hip.conv_bank(occ, b1, lam, prep=prep)[1]
    o1b = _hip.conv_bank(occ, b1, lam, prep=prep)[1]
    assert torch.equal(o1, o1b) and torch.equal(o1, _hip.conv_bank(occ, b1, lam)[1])
    _hip.conv_bank_prep(b2, prep)                      # the same memory, re-prepared for another bank
    o2 = _hip.conv_bank(occ, b2, lam, prep=prep)[1]
    assert torch.equal(o2, _hip.conv_bank(occ, b2, lam)[1]) and not torch.equal(o1, o2)
    # float / non-9^3 inputs are forwarded to sn_conv_bank untouched
    xf = occ.float()
    assert torch.equal(_hip.conv_bank(xf, b1, lam, prep=prep)[1], _hip.conv_bank(xf, b1, lam)[1])


def _bench_model(hip_device, geneo_num=None):
    from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec
    geneo_num = geneo_num or {"cy": 6, "cone": 5, "neg": 5}
    specs, names, lambdas, last = synthetic_bank_spec(geneo_num)
    model = sna.SceneNet(geneo_num, (9, 9, 9))
    apply_bank_spec(model, specs, names, lambdas, last)
    return model.to(hip_device)


@pytest.mark.parametrize("geneo_num", [{"cy": 6, "cone": 5, "neg": 5}, {"cy": 2, "cone": 1, "neg": 1},
                                       {"cy": 7, "cone": 6, "neg": 7}])
def test_bank_builder_prepares_what_the_standalone_preparation_does(hip_device, geneo_num):
    """sn_geneo_bank_prep (K2 with the preparation as its tail) == sn_geneo_bank followed by sn_conv_bank_prep: the same bank
    bits, and the same blob in every field a kernel reads (G = 16, G = 4 with twelve zero pad entries, G = 20: two groups)."""
    model = _bench_model(hip_device, geneo_num)
    bank = model.compute_bank(hip_device)
    bank_p, prep_p = model.compute_bank_prepared(hip_device)
    assert torch.equal(bank, bank_p)
    prep = _hip.conv_bank_prep(bank)
    G = bank.shape[0]
    for grp in range((G + 15) // 16):
        a = prep_p[grp * _hip.SN_CONV_PREP_BYTES:(grp + 1) * _hip.SN_CONV_PREP_BYTES].cpu().numpy()
        b = prep[grp * _hip.SN_CONV_PREP_BYTES:(grp + 1) * _hip.SN_CONV_PREP_BYTES].cpu().numpy()
        for name, lo, hi in [("digit table", 0, 12288), ("scales", 12288, 12352), ("bounds", 12352, 12480),
                             ("symmetry", 12480, 12544), ("fit", 12544, 12608), ("magic", 12608, 12612)]:
            assert np.array_equal(a[lo:hi], b[lo:hi]), (grp, name)
        sym = a[12480:12544].view(np.int32)
        assert sym.tolist() == [1] * 16          # every GENEO family is symmetric in x and y, bit for bit; pads too


@pytest.mark.parametrize("geneo_num", [{"cy": 6, "cone": 5, "neg": 5}, {"cy": 2, "cone": 1, "neg": 1},
                                       {"cy": 7, "cone": 7, "neg": 6}])
def test_bank_riding_in_the_voxelisation_launch(hip_device, geneo_num):
    """sn_voxel_occupancy_fused_bank: K2 + the preparation as extra workgroups of K1's first launch == sn_geneo_bank_prep
    (bank and every field of the blob a kernel reads, bit for bit: the bank code is compiled into voxel.hip under the
    contraction setting of bank.hip) next to == sn_voxel_occupancy_fused (occupancy, flags, descriptor); G = 16, 4, 20
    (two groups: the rider rows outnumber one row of the grid); unaligned points too."""
    from scene_net_amd.synthetic import synthetic_tile
    model = _bench_model(hip_device, geneo_num)
    bank_p, prep_p = model.compute_bank_prepared(hip_device)
    bank_p, prep_p = bank_p.clone(), prep_p.clone()
    tiles = [synthetic_tile(i, 20_000 + 1000 * i)[0] for i in range(3)]
    batch = sna.PointBatch.from_tiles(tiles, device=hip_device)
    plain = sna.voxelize_batch(batch, (64, 64, 64), occ_dtype=torch.bool)
    params, kinds, bank, prep = model.bank_rider(hip_device)
    bank.fill_(float("nan")); prep.zero_()
    rode = sna.voxelize_batch(batch, (64, 64, 64), occ_dtype=torch.bool, bank_rider=(params, kinds, bank, prep))
    assert rode.rider_done and not plain.rider_done
    assert torch.equal(rode.occ, plain.occ) and torch.equal(rode.flags, plain.flags) and torch.equal(rode.desc, plain.desc)
    assert torch.equal(bank, bank_p)
    G = bank.shape[0]
    for grp in range((G + 15) // 16):
        a = prep[grp * _hip.SN_CONV_PREP_BYTES:(grp + 1) * _hip.SN_CONV_PREP_BYTES].cpu().numpy()
        b = prep_p[grp * _hip.SN_CONV_PREP_BYTES:(grp + 1) * _hip.SN_CONV_PREP_BYTES].cpu().numpy()
        for name, lo, hi in [("digit table", 0, 12288), ("scales", 12288, 12352), ("bounds", 12352, 12480),
                             ("symmetry", 12480, 12544), ("fit", 12544, 12608), ("magic", 12608, 12612)]:
            assert np.array_equal(a[lo:hi], b[lo:hi]), (grp, name)
    # an unaligned point buffer (the other instantiation of the launch)
    pts_u = torch.empty(batch.pts.numel() + 1, dtype=torch.float64, device=hip_device)[1:].view_as(batch.pts)
    pts_u.copy_(batch.pts)
    bank.fill_(float("nan"))
    occ_u = _hip.voxel_occupancy_fused(pts_u, None, batch.offsets, (64, 64, 64), out_dtype=torch.bool,
                                       bank_rider=(params, kinds, bank, prep))[0]
    assert torch.equal(occ_u, plain.occ) and torch.equal(bank, bank_p)
    # a pipeline pass with the rider == the serial pass, eagerly and replayed from a hipGraph
    model.fused_forward = False
    with torch.no_grad():
        serial = sna.ScenePipeline(model, (64, 64, 64), overlap_bank=False)(batch)
        pipe = sna.ScenePipeline(model, (64, 64, 64), overlap_bank=True)
        assert pipe.rides()
        assert torch.equal(pipe(batch), serial)
        cap = pipe.capture(batch)
        assert torch.equal(cap.replay(), serial)
    model.fused_forward = True


def test_pipeline_with_the_bank_forked_beside_the_voxelisation(hip_device):
    """ScenePipeline(overlap_bank=True): K2 off the critical path -- riding in K1's first launch (9^3 banks; the side-stream
    fork of bank_beside otherwise) -- gives the same bits as the serial pass, for the 16-kernel contraction and for the
    forward through linearity, eagerly and replayed from a hipGraph; so does the fork itself (bank_beside, still the path of
    other kernel extents)."""
    from scene_net_amd.synthetic import synthetic_tile
    model = _bench_model(hip_device)
    tiles = [synthetic_tile(i, 50_000)[0] for i in range(4)]
    batch = sna.PointBatch.from_tiles(tiles, device=hip_device)
    for fused in (False, True):
        model.fused_forward = fused
        with torch.no_grad():
            serial = sna.ScenePipeline(model, (64, 64, 64), overlap_bank=False)(batch)
            pipe = sna.ScenePipeline(model, (64, 64, 64), overlap_bank=True)
            forked = [pipe(batch) for _ in range(3)]
            cap = pipe.capture(batch)
            replayed = cap.replay().clone()
        assert all(torch.equal(serial, f) for f in forked)
        assert torch.equal(serial, replayed)
        with torch.no_grad():   # the fork, by hand
            bank, lam, prep, join = pipe.bank_beside(hip_device)
            grids = pipe.voxelize(batch)
            join()
            by_fork = (_hip.conv_fused(grids.occ, bank, lam) if fused
                       else model.contract_prepared(grids.occ, bank, lam, prep)[1])
        assert torch.equal(serial, by_fork)
    model.fused_forward = True
    assert _hip.conv_i8_spin_timeouts() == _SPIN0


def test_all_positive_and_zero_mean_banks_agree_in_every_int8_kernel(hip_device):
    """z-walk == folded == stride-4 == four-copy bit for bit, and within 1e-4 of the fp64 oracle, on a zero-mean bank (what
    GENEO banks look like) and on an all-positive one, whose digit sums run into the tens of bits -- the case where an int32
    recombination of the three digit sums would overflow (the blob's `fit` field says so; round 3 built that one-conversion
    head in all four kernels, measured no gain and took it out again: DESIGN section 9)."""
    torch.manual_seed(21)
    occ = torch.rand((2, 1, 24, 16, 64)) < 0.35
    g = torch.Generator().manual_seed(5)
    w = torch.rand((16, 9, 9, 9), generator=g)
    w = w + w.flip(2)
    w = w + w.flip(3)
    banks = {"no fit (all weights positive: sum of Q ~ 729 x 4e6 > 2^31)": (0.004 * (w + 1.0)).float().contiguous(),
             "fit (zero-mean)": (0.01 * (w - w.mean())).float().contiguous()}
    lam = (torch.rand(16) - 0.3) / 16
    x, l = occ.to(hip_device), lam.to(hip_device)
    for name, bank in banks.items():
        b = bank.to(hip_device)
        prep = _hip.conv_bank_prep(b)
        fit = prep[12544:12544 + 64].cpu().numpy().view(np.int32)
        assert bool(fit.all()) == name.startswith("fit"), (name, fit.tolist())
        ref_act = go.conv_bank(occ.double(), bank.double().unsqueeze(1))
        ref_out = torch.relu(torch.tanh((lam.double().view(1, 16, 1, 1, 1) * ref_act).sum(1, keepdim=True)))
        outs = {"zwalk": _hip.conv_bank(x, b, l, want_act=True, want_out=True, prep=prep)}
        outs["folded"] = _hip.conv_bank(x, b, l, want_act=True, want_out=True)
        _hip.set_option("conv_i8_fold", 0)
        try:
            outs["stride-4"] = _hip.conv_bank(x, b, l, want_act=True, want_out=True)
        finally:
            _hip.set_option("conv_i8_fold", 1)
        _hip.set_option("conv_i8_legacy", 1)
        try:
            outs["four-copy"] = _hip.conv_bank(x, b, l, want_act=True, want_out=True)
        finally:
            _hip.set_option("conv_i8_legacy", 0)
        a0, o0 = outs["zwalk"]
        for k, (a, o) in outs.items():
            assert torch.equal(a, a0) and torch.equal(o, o0), (name, k)
        assert (o0.cpu().double() - ref_out).abs().max().item() < TOL, name
        assert act_err_ok(a0, ref_act, TOL), name


def test_verdict_is_learnt_without_a_sync_and_the_fallback_launch_left_out(hip_device):
    """SceneNet.contract_prepared: the first calls run with the fallback launch behind the walk and enqueue an asynchronous
    read-back of the verdict; once that has landed, calls with the SAME parameters use sn_conv_bank_prepared_served (no
    fallback launch) and give the same bits; a parameter change starts over; a bank the walk does not serve never loses its
    fallback."""
    from scene_net_amd.synthetic import synthetic_tile
    model = _bench_model(hip_device)
    model.fused_forward = False
    tiles = [synthetic_tile(i, 30_000)[0] for i in range(2)]
    occ = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles, device=hip_device), (64,) * 3, occ_dtype=torch.bool).occ
    with torch.no_grad():
        first = model(occ)
        verdict = model.__dict__["_prepared_verdict"]
        assert verdict._state == 1                                  # read-back enqueued, nothing waited for
        torch.cuda.synchronize()
        second = model(occ)                                         # learns "served" here, launches without the fallback
        assert verdict._state == 2
        third = model(occ)
        assert torch.equal(first, second) and torch.equal(first, third)
        # an in-place parameter change (what an optimiser step is) invalidates the knowledge
        model.geneos["cy_0"].geneo_params["radius"].add_(0.25)
        changed = model(occ)
        assert verdict._state == 1 and not torch.equal(changed, first)
        torch.cuda.synchronize()
    # a bank over the quantisation tolerance: verdict 1, never "served" -- and the result is the fp32 kernel's
    wide = _symmetric_bank(16, 3, scale=torch.full((16,), 40.0)).to(hip_device)
    lam = ((torch.rand(16) - 0.3) / 16).to(hip_device)
    prep = _hip.conv_bank_prep(wide)
    out = _hip.conv_bank(occ, wide, lam, prep=prep)[1]
    torch.cuda.synchronize()
    assert _hip.prep_verdicts(prep).cpu().tolist() == [1]
    assert torch.equal(out, _hip.conv_bank(occ.view(torch.uint8), wide, lam)[1])
    # what a caller who assumed wrongly gets: test_served_launch_on_a_declined_bank_is_loud (below)


def test_fused_forward_learns_its_verdict_and_drops_the_gated_launches(hip_device):
    """SceneNet.fused_served (the module's default no-grad forward on binary occupancy): the guard's verdict of sn_conv_fused_v
    lands in a caller-owned word, is learnt without a synchronisation, and from then on the gated fp32 launches are left
    out -- same bits before and after; a parameter change starts over; a tolerance the bank cannot meet is read as "not
    served" and keeps the fallback (whose result then differs from the unguarded kernel's)."""
    model = _bench_model(hip_device)
    model.fused_forward = True
    x = (torch.rand(2, 1, 64, 64, 64, device=hip_device) < 0.05)
    with torch.no_grad():
        first = model(x)
        torch.cuda.synchronize()
        verdict = model._fused_state["verdict"]
        word = _hip.conv_fused_prep_verdict(model._fused_state["blob"], (9, 9, 9))
        assert int(word.item()) == 0
        again = model(x)            # picks the read-back up: served
        assert verdict._state == 2
        served = model(x)
        assert torch.equal(first, again) and torch.equal(first, served)
        # a changed coefficient: a new key, the fallback is back until the verdict has been read again
        name = next(n for n in model.lambdas_dict if n != model.last_lambda)
        model.lambdas_dict[name].mul_(1.5)   # (an in-place op under no_grad: bumps the version the caches key on)
        changed = model(x)
        assert verdict._state in (0, 1)
        torch.cuda.synchronize()
        assert torch.equal(model(x), changed)
        # a write through .data is invisible to the version keys: invalidate_caches() is the documented way
        model.lambdas_dict[name].data.mul_(0.5)
        model.invalidate_caches()
        halved = model(x)
        assert not torch.equal(halved, changed) and model._fused_state["verdict"]._state in (0, 1)
        torch.cuda.synchronize()
        verdict = model._fused_state["verdict"]   # (a fresh state: new blob, new verdict)
        word = _hip.conv_fused_prep_verdict(model._fused_state["blob"], (9, 9, 9))
        # an impossible tolerance: verdict 1, never "served", the fp32 contraction's result
        _hip.set_option("conv_i8_tolerance_ppb", 1)
        try:
            strict = model(x)
            torch.cuda.synchronize()
            assert int(word.item()) == 1
            assert torch.equal(model(x), strict) and verdict._state == 3
            bank, lam = model.compute_bank(hip_device), model.effective_lambdas(hip_device)
            assert torch.equal(strict, _hip.conv_bank(x.view(torch.uint8), bank, lam, want_act=False, want_out=True)[1])
        finally:
            _hip.set_option("conv_i8_tolerance_ppb", 90000)


@pytest.mark.parametrize("ks", [(9, 9, 9), (9, 5, 5), (6, 5, 6), (9, 7, 7)])
def test_fused_forward_on_prepared_tables(hip_device, ks):
    """sn_conv_fused_prep + sn_conv_fused_prepared == sn_conv_fused bit for bit (f32, f64, bf16 outputs; ragged grids); the
    blob's verdict word is 0 for a bank within the tolerance and 1 -- with the fp32 contraction's result -- for a
    tolerance nothing meets."""
    torch.manual_seed(4)
    model = sna.SceneNet({"cy": 2, "cone": 2, "neg": 1}, ks).to(hip_device)
    bank, lam = model.compute_bank(hip_device), model.effective_lambdas(hip_device)
    blob = _hip.conv_fused_prep(bank, lam)
    assert blob.numel() == _hip.conv_fused_prep_bytes(ks) > 0
    for shape in [(2, 1, 64, 64, 64), (1, 1, 20, 33, 64), (3, 1, 9, 17, 128)]:
        x = torch.rand(shape, device=hip_device) < 0.06
        for dt in (torch.float32, torch.float64, torch.bfloat16):
            want = _hip.conv_fused(x, bank, lam, out_dtype=dt)
            got = _hip.conv_fused(x, bank, lam, out_dtype=dt, prep=blob)
            assert torch.equal(want, got), (ks, shape, dt)
    torch.cuda.synchronize()
    assert int(_hip.conv_fused_prep_verdict(blob, ks).item()) == 0
    _hip.set_option("conv_i8_tolerance_ppb", 1)
    try:
        strict = _hip.conv_fused_prep(bank, lam)
        x = torch.rand(2, 1, 16, 16, 64, device=hip_device) < 0.06
        got = _hip.conv_fused(x, bank, lam, prep=strict)
        torch.cuda.synchronize()
        assert int(_hip.conv_fused_prep_verdict(strict, ks).item()) == 1
        assert torch.equal(got, _hip.conv_bank(x.view(torch.uint8), bank, lam, want_act=False, want_out=True)[1])
    finally:
        _hip.set_option("conv_i8_tolerance_ppb", 90000)


# ---------------------------------------------------------------------------------------------- round 4: loud, not quiet
def _expect_latched(code):
    """the sticky status is latched with `code`, every launching entry now fails, and clearing re-arms the device"""
    torch.cuda.synchronize()
    st = _hip.device_status()
    assert st[0] == code, st
    with pytest.raises(_hip.HipLibraryError, match="latched status"):
        _hip.conv_bank_prep(torch.zeros((16, 9, 9, 9), device="cuda"))
    _hip.device_status_clear()
    assert _hip.device_status()[0] == 0


def test_spin_give_up_is_loud(hip_device):
    """A dependency that never arrives (injected: plane 0's first raw rows are never reported as landed): the walk does not
    return numbers computed from a ring slot that was not ready -- every workgroup overwrites what it owns with NaN, the
    device's sticky status is latched (code 1) and the next sn_* call fails until the status is cleared."""
    torch.manual_seed(11)
    occ = (torch.rand((2, 1, 16, 24, 64)) < 0.3).to(hip_device)
    bank = _symmetric_bank(16, 5).to(hip_device)
    lam = ((torch.rand(16) - 0.3) / 16).to(hip_device)
    prep = _hip.conv_bank_prep(bank)
    good_a, good_o = _hip.conv_bank(occ, bank, lam, want_act=True, want_out=True, prep=prep)
    t0 = _hip.conv_i8_spin_timeouts()
    assert _hip.device_status()[0] == 0
    _hip.set_option("conv_i8z_inject_fault", 1)
    try:
        # (the call itself returns: the kernel is asynchronous and nothing was latched when it was enqueued)
        act, out = _hip.conv_bank(occ, bank, lam, want_act=True, want_out=True, prep=prep)
    finally:
        _hip.set_option("conv_i8z_inject_fault", 0)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out).all()) and bool(torch.isnan(act).all())
    _expect_latched(1)
    assert _hip.conv_i8_spin_timeouts() > t0
    # re-armed: the same call is served again, same bits as before the fault
    act2, out2 = _hip.conv_bank(occ, bank, lam, want_act=True, want_out=True, prep=prep)
    torch.cuda.synchronize()
    assert torch.equal(act2, good_a) and torch.equal(out2, good_o) and _hip.device_status()[0] == 0


def test_served_launch_on_a_declined_bank_is_loud(hip_device):
    """sn_conv_bank_prepared_served leaves the fallback launch out on the caller's word that the verdict is "served".  When
    that word is wrong -- a bank over the quantisation tolerance, a bank that is not symmetric -- the outputs are NaN and
    the sticky status says so (code 2); they are never left unwritten."""
    torch.manual_seed(12)
    occ = (torch.rand((2, 1, 16, 16, 64)) < 0.3).to(hip_device)
    lam = ((torch.rand(16) - 0.3) / 16).to(hip_device)
    wide = _symmetric_bank(16, 3, scale=torch.full((16,), 40.0)).to(hip_device)
    asym = _symmetric_bank(16, 4)
    asym[3, 2, 1, 7] += 0.125
    for bank, verdict in ((wide, 1), (asym.to(hip_device), 2)):
        prep = _hip.conv_bank_prep(bank)
        assert _hip.prep_verdicts(prep).cpu().tolist() == [-1]          # fresh from the preparation: no verdict yet
        out = torch.full((2, 1, 16, 16, 64), -7.0, device=hip_device)
        rc = _hip.load().sn_conv_bank_prepared_served(occ.data_ptr(), _hip.SN_OCC8, bank.data_ptr(), lam.data_ptr(),
                                                      prep.data_ptr(), 2, 16, 16, 64, 16, 9, 9, 9, None, out.data_ptr(),
                                                      _hip.SN_F32, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0 and _hip.prep_verdicts(prep).cpu().tolist() == [verdict]
        assert bool(torch.isnan(out).all())
        _expect_latched(2)
        # with the fallback in place the same blob gives the other kernels' result
        ok = _hip.conv_bank(occ, bank, lam, prep=prep)[1]
        assert torch.equal(ok, _hip.conv_bank(occ, bank, lam)[1]) and not bool(torch.isnan(ok).any())
    assert _hip.device_status()[0] == 0


def test_fused_forward_assumed_served_on_a_declined_bank_is_loud(hip_device):
    """the same contract for the forward through linearity (sn_conv_fused_prepared with assume_served)"""
    model = _bench_model(hip_device)
    x = (torch.rand(2, 1, 32, 32, 64, device=hip_device) < 0.05)
    bank, lam = model.compute_bank(hip_device), model.effective_lambdas(hip_device)
    blob = torch.empty(_hip.conv_fused_prep_bytes((9, 9, 9)), dtype=torch.uint8, device=hip_device)
    _hip.set_option("conv_i8_tolerance_ppb", 1)      # a tolerance nothing meets: the blob's verdict is 1
    try:
        _hip.conv_fused_prep(bank, lam, blob)
        out = _hip.conv_fused(x, bank, lam, out_dtype=torch.float32, prep=blob, assume_served=True)
        torch.cuda.synchronize()
        assert bool(torch.isnan(out).all())
        _expect_latched(2)
        kept = _hip.conv_fused(x, bank, lam, out_dtype=torch.float32, prep=blob, assume_served=False)
        assert torch.equal(kept, _hip.conv_bank(x.view(torch.uint8), bank, lam, want_act=False, want_out=True)[1])
    finally:
        _hip.set_option("conv_i8_tolerance_ppb", 90000)
    assert _hip.device_status()[0] == 0


def test_a_verdict_is_never_learnt_from_a_call_the_walk_did_not_serve(hip_device):
    """parameters whose bank is over the tolerance, a first call on a grid the walk does not serve (Y % 16 != 0:
    the verdict word is not rewritten), then a 64^3 grid.  The preparation resets the word to -1, so the read-back of the
    first call teaches nothing, the second call keeps its fallback, and the result is sn_conv_bank's."""
    model = _bench_model(hip_device)
    model.fused_forward = False
    with torch.no_grad():
        served_first = model((torch.rand(1, 1, 16, 16, 64, device=hip_device) < 0.2))      # learns "served" for the bench bank
        torch.cuda.synchronize()
        model((torch.rand(1, 1, 16, 16, 64, device=hip_device) < 0.2))
        assert model.__dict__["_prepared_verdict"]._state == 2
        for layer in model.geneos.values():          # a bank far over the quantisation tolerance: sigma x 60
            layer.geneo_params["sigma"].mul_(60.0)
        odd = (torch.rand(1, 1, 16, 16, 40, device=hip_device) < 0.2)                      # Y % 16 != 0: not the walk's
        model(odd)
        torch.cuda.synchronize()
        x = (torch.rand(2, 1, 64, 64, 64, device=hip_device) < 0.05)
        got = model(x)
        torch.cuda.synchronize()
        assert model.__dict__["_prepared_verdict"]._state in (0, 1, 3)
        again = model(x)
        torch.cuda.synchronize()
        assert model.__dict__["_prepared_verdict"]._state == 3
        bank, lam = model.compute_bank(hip_device), model.effective_lambdas(hip_device)
        want = _hip.conv_bank(x, bank, lam, want_act=False, want_out=True)[1]
        assert torch.equal(got, want) and torch.equal(again, want) and not bool(torch.isnan(got).any())
    assert _hip.device_status()[0] == 0 and served_first is not None


def test_a_write_through_data_is_seen_by_the_next_forward(hip_device):
    """`p.data.fill_(..)`, `p.data = t` and an old-style `p.data.add_(..)` bump no `_version`; the model's
    parameters count their `.data` accesses instead (scene_net._TrackedParameter), and every cache keys on that count."""
    x = (torch.rand(2, 1, 32, 32, 64, device=hip_device) < 0.05)
    for fused in (True, False):
        model = _bench_model(hip_device)
        model.fused_forward = fused
        with torch.no_grad():
            first = model(x)
            torch.cuda.synchronize()
            model(x)
            p = model.geneos["cy_0"].geneo_params["radius"]
            v0 = p._version
            p.data.add_(0.5)                                         # invisible to _version ...
            assert p._version == v0
            moved = model(x)                                         # ... and still seen
            fresh = _bench_model(hip_device)
            fresh.fused_forward = fused
            fresh.geneos["cy_0"].geneo_params["radius"].add_(0.5)
            assert torch.equal(moved, fresh(x)) and not torch.equal(moved, first)
            name = next(n for n in model.lambdas_dict if n != model.last_lambda)
            model.lambdas_dict[name].data = model.lambdas_dict[name].data * 0.5      # the setter
            fresh.lambdas_dict[name].mul_(0.5)
            assert torch.equal(model(x), fresh(x))
    assert _hip.device_status()[0] == 0


def test_c3_share_32_tiles_of_128_cubed(hip_device, zwalk_variant):
    """BASELINE C3's per-GPU share -- 32 tiles of 128^3, what `bench.py --grid 128 --batch 32` runs: voxelise -> prepared
    contraction; the z-walk equals sn_conv_bank bit for bit over the whole batch, one tile meets the fp64 oracle directly,
    no spin ever gave up (SCENE_Net.py:322-339)."""
    if zwalk_variant != 2:
        pytest.skip("the shipped variant only: 32 x 128^3 twice is 0.5 GB of outputs per variant")
    from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile
    specs, names, lambdas, last = synthetic_bank_spec()
    model = sna.SceneNet({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9))
    apply_bank_spec(model, specs, names, lambdas, last)
    model = model.to(hip_device)
    bank, lam = model.compute_bank(hip_device), model.effective_lambdas(hip_device)
    tiles = [synthetic_tile(i, 120_000)[0] for i in range(32)]
    occ = sna.voxelize_batch(sna.PointBatch.from_tiles(tiles, device=hip_device), (128,) * 3, occ_dtype=torch.bool).occ
    assert tuple(occ.shape) == (32, 1, 128, 128, 128)
    t0 = _hip.conv_i8_spin_timeouts()
    c0 = _hip.conv_i8_path_counts()
    (_, o_z), (_, o_r) = _both(occ, bank, lam, want_act=False)
    assert _delta(c0, _hip.conv_i8_path_counts()) == (2, 0, 0)
    assert torch.equal(o_z, o_r)
    assert _hip.conv_i8_spin_timeouts() == t0 and _hip.device_status()[0] == 0
    t = 17
    ref = go.scenenet_forward(occ[t:t + 1].cpu().double(), specs, (9, 9, 9), lambdas, last, names=names)
    assert (o_z[t:t + 1].cpu().double() - ref).abs().max().item() < TOL
    assert float(ref.max()) > 0.05


_FIRST_LAUNCH_IN_A_CAPTURE = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from scene_net_amd import _hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
w = torch.rand((16, 9, 9, 9), generator=g) - 0.5
w = w + w.flip(2); w = w + w.flip(3)
bank = (w * torch.logspace(-2, 0.3, 16).view(16, 1, 1, 1)).float().contiguous().to(dev)
occ = (torch.rand((2, 1, 24, 16, 64), generator=g) < 0.2).to(dev)
lam = ((torch.rand(16, generator=g) - 0.3) / 16).to(dev)
prep = _hip.conv_bank_prep(bank)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):          # the library's FIRST contraction launch of this process is being captured
    out = _hip.conv_bank(occ, bank, lam, prep=prep)[1]
graph.replay(); torch.cuda.synchronize()
first = out.clone()
eager = _hip.conv_bank(occ, bank, lam, prep=prep)[1]
graph.replay(); torch.cuda.synchronize()
assert torch.equal(first, eager) and torch.equal(out, eager), "captured walk differs from the eager one"
assert _hip.device_status()[0] == 0
print("capture-first ok")
"""


def test_the_first_launch_of_a_process_may_be_inside_a_capture(hip_device, zwalk_variant):
    """The device-status words are allocated lazily; a capture must never be where that happens (hipMalloc /
    hipHostMalloc inside a capture invalidate it): a walk captured before any eager launch runs without status words,
    and replays to the eager result."""
    import subprocess
    if zwalk_variant != 0:
        pytest.skip("one fresh process is enough")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", _FIRST_LAUNCH_IN_A_CAPTURE, root], capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0 and "capture-first ok" in res.stdout, res.stdout + res.stderr


def test_launch_timing_events_time_the_walk_itself(hip_device, zwalk_variant):
    """sn_launch_timing_events: the walk's own start / stop timestamps land in the caller's events (one-shot), the result is
    untouched, and the duration is positive and no longer than the interval between two records around the call."""
    torch.manual_seed(4)
    occ = (torch.rand((4, 1, 64, 64, 64), device=hip_device) < 0.05)
    bank = _symmetric_bank(16, 3).to(hip_device)
    lam = ((torch.rand(16) - 0.3) / 16).to(hip_device)
    prep = _hip.conv_bank_prep(bank)
    ref = _hip.conv_bank(occ, bank, lam, prep=prep)[1]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    ev[0].record()
    _hip.launch_timing_events(ev[1], ev[2])
    out = _hip.conv_bank(occ, bank, lam, prep=prep)[1]
    ev[3].record()
    torch.cuda.synchronize()
    kernel_ms, around_ms = ev[1].elapsed_time(ev[2]), ev[0].elapsed_time(ev[3])
    assert torch.equal(out, ref)
    assert 0.0 < kernel_ms <= around_ms, (kernel_ms, around_ms)
    # one-shot: the next launch is an ordinary one and leaves the pair alone
    t_before = ev[1].elapsed_time(ev[2])
    _hip.conv_bank(occ, bank, lam, prep=prep)
    torch.cuda.synchronize()
    assert ev[1].elapsed_time(ev[2]) == t_before
