"""The drop-in modules end to end on the MI355X: SceneNet.forward and the fused point-cloud pipeline against
the oracle (which is pinned bit-exact to the reference's forward)."""
import os

import numpy as np
import pytest
import torch

import scene_net_amd as sna
from oracle import geneo_oracle as go
from oracle import voxel_oracle as vo
from scene_net_amd.synthetic import synthetic_tile

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _oracle_forward(model, x64):
    names = list(model.geneos.keys())
    specs = [(n.split("_")[0], {k: float(v) for k, v in model.geneos[n].geneo_params.items()}) for n in names]
    lam = [float(model.lambdas_dict[f"lambda_{n}"]) for n in names]
    last = names.index(model.last_lambda.replace("lambda_", ""))
    ks = model.kernel_size_of_bank()
    return go.scenenet_forward(x64, specs, ks, lam, last, return_bank=True, names=names)


@pytest.mark.parametrize("geneo_num,ks", [({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)),
                                          ({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9)),
                                          ({"cy": 6, "cone": 5, "neg": 5}, (9, 9, 9)),
                                          (None, None)])
def test_scenenet_forward_is_the_reference_forward(hip_device, geneo_num, ks):
    torch.manual_seed(4)
    model = sna.SceneNet(geneo_num, ks).to(hip_device)
    x = (torch.rand(2, 1, 24, 20, 28) < 0.08).double()
    ref_out, ref_act = _oracle_forward(model, x)
    out, act = model(x.to(hip_device), return_bank_activations=True)
    assert out.dtype == torch.float64 and out.device.type == "cuda" and out.shape == x.shape
    assert (out.cpu() - ref_out).abs().max().item() < TOL
    assert (act.cpu() - ref_act).abs().max().item() < TOL
    assert abs(sum(float(p) for p in model.lambdas_dict.values()) - 1.0) < 1e-6
    # second call (last lambda was re-created by the first): same answer.  Without the activations the float grid
    # goes through sn_forward_auto (binary -> int8 kernels), so "same" is the parity bar, and repeatable bit for bit
    out2 = model(x.to(hip_device))
    assert (out - out2).abs().max().item() < 2e-5 and (out2.cpu() - ref_out).abs().max().item() < TOL
    assert torch.equal(out2, model(x.to(hip_device)))
    # parameters changed in place (an optimiser step) are picked up
    with torch.no_grad():
        model.geneos["cy_0"].geneo_params["radius"].add_(0.75)
    ref3, _ = _oracle_forward(model, x)
    out3 = model(x.to(hip_device))
    assert (out3.cpu() - ref3).abs().max().item() < TOL
    assert (out3 - out).abs().max().item() > 1e-6


def test_model_on_cpu_input_on_gpu_and_fp32_input(hip_device):
    torch.manual_seed(6)
    model = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (5, 5, 5))  # parameters stay on the host
    x = (torch.rand(1, 1, 12, 12, 12) < 0.2)
    ref, _ = _oracle_forward(model, x.double())
    out = model(x.float().to(hip_device))
    assert out.dtype == torch.float32
    assert (out.double().cpu() - ref).abs().max().item() < TOL


def test_fused_pipeline_matches_reference_chain(hip_device, golden_dir):
    """points -> Voxelization -> ToFullDense -> SceneNet, HBM resident, vs the oracle chain."""
    torch.manual_seed(8)
    a = np.load(os.path.join(golden_dir, "ts40k_sample575_subset.npy"))
    tiles = [a[:, :3]] + [synthetic_tile(t, 30_000)[0] for t in range(2)]
    labels = [a[:, 3]] + [synthetic_tile(t, 30_000)[1] for t in range(2)]
    model = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9)).to(hip_device)
    pipe = sna.ScenePipeline(model, (64, 64, 64), keep_labels=[15])
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    out, grids = pipe(batch, want_gt=True)
    assert out.shape == (3, 1, 64, 64, 64) and out.dtype == torch.float32
    for b in range(3):
        vox, gt = vo.voxelization_call((tiles[b], labels[b]), [15], None, (64, 64, 64))
        x = torch.from_numpy(vo.to_full_dense(vox))[None]
        assert np.array_equal(grids.occ[b].cpu().numpy(), x[0].numpy().astype(np.float32))
        assert np.array_equal(grids.gt_occ[b].cpu().numpy(), vo.to_full_dense(gt).astype(np.float32))
        ref, _ = _oracle_forward(model, x)
        assert (out[b].double().cpu() - ref[0]).abs().max().item() < TOL


def test_c1_chain_on_the_full_reference_tile(hip_device, golden_dir):
    """BASELINE C1 end to end: data-sample/sample_575.npy verbatim (58 243 points, UTM) -> 64^3 -> 4 GENEO kernels
    (cy 2, cone 1, neg 1; 9^3) -> head, against the oracle chain (occupancy bit-exact, activations < 1e-4)."""
    torch.manual_seed(575)
    a = np.load(os.path.join(golden_dir, "ts40k_sample575_full.npz"))["tile"]
    xyz, labels = a[:, :3], a[:, 3]
    model = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9)).to(hip_device)
    pipe = sna.ScenePipeline(model, (64, 64, 64), keep_labels=[15])
    batch = sna.PointBatch.from_tiles([xyz], [labels], device=hip_device)
    out, grids = pipe(batch, want_gt=True)
    vox, gt = vo.voxelization_call((xyz, labels), [15], None, (64, 64, 64))
    x = torch.from_numpy(vo.to_full_dense(vox))[None]
    assert np.array_equal(grids.occ[0].cpu().numpy(), x[0].numpy().astype(np.float32))
    assert np.array_equal(grids.gt_occ[0].cpu().numpy(), vo.to_full_dense(gt).astype(np.float32))
    ref, act_ref = _oracle_forward(model, x)
    assert (out[0].double().cpu() - ref[0]).abs().max().item() < TOL
    # the 4-kernel contraction with its activations (int8 path on the bool grid), not only the head
    out2, act = model(grids.occ.bool(), return_bank_activations=True)
    assert (act[0].double().cpu() - act_ref[0]).abs().max().item() < TOL
    assert (out2[0].double().cpu() - ref[0]).abs().max().item() < TOL


def test_v1_module_and_wrappers(hip_device, golden_dir):
    """SCENE_Net (v1), SCENENetQuantile, SCENE_Net_Class: golden forward of the reference's v1 module."""
    import json
    F = np.load(os.path.join(golden_dir, "geneo_forward_v1.npz"))
    with open(os.path.join(golden_dir, "geneo_forward_v1_meta.json")) as f:
        meta = json.load(f)
    torch.manual_seed(0)
    model = sna.SCENE_Net({"cy": 2, "cone": 1, "neg": 1}, (9, 7, 7))
    assert list(model.state_dict().keys()) == meta["state_dict_keys"]
    names = [str(n) for n in F["names"]]
    with torch.no_grad():
        for n in names:
            for k, v in meta["geneo_params"][n].items():
                model.geneos[n].geneo_params[k].fill_(v)
        for n, v in zip(names, F["lambdas"]):
            model.lambdas_dict[f"lambda_{n}"].fill_(float(v))
    model.last_lambda = f"lambda_{names[int(F['last'])]}"
    model = model.to(hip_device)
    x = torch.from_numpy(F["x"].astype(np.float64)).to(hip_device)
    out, act = model(x, return_bank_activations=True)
    assert (act.cpu() - torch.from_numpy(F["conv"])).abs().max().item() < TOL
    assert (out.cpu() - torch.from_numpy(F["out"])).abs().max().item() < TOL
    outb = model(x.bool())  # int8 path on the same occupancy
    assert (outb.double().cpu() - torch.from_numpy(F["out"])).abs().max().item() < TOL
    # thresholded head and quantile ensemble are thin wrappers over the same forward
    torch.manual_seed(1)
    cls = sna.SCENE_Net_Class({"cy": 1, "cone": 1, "neg": 1}, plot=False).to(hip_device)
    pred = cls.gnet(x)
    assert torch.equal(cls(x), (pred >= cls.tau).to(x.dtype))
    q = sna.SCENENetQuantile({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5), device=hip_device)
    yq = q(x)
    assert yq.shape == (2, 3, 14, 12, 20) and yq.dtype == torch.float32
    assert torch.equal(yq[:, 1], q.scnets[1](x)[:, 0].float())


def test_smoke_entry(hip_device):
    import __graft_entry__
    __graft_entry__.smoke()


def test_captured_pipeline_replays_on_refilled_buffers(hip_device):
    """ScenePipeline.capture: one hipGraph for the whole pass; replay on new points in the same buffers == eager."""
    from scene_net_amd.synthetic import synthetic_tile
    torch.manual_seed(0)
    model = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9)).to(hip_device)
    pipe = sna.ScenePipeline(model, (32, 32, 32), keep_labels=[15])
    tiles, labels = zip(*[synthetic_tile(t, 20_000) for t in range(3)])
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    cap = pipe.capture(batch, want_gt=True)
    out, grids = cap.replay()
    with torch.no_grad():
        ref_out, ref_grids = pipe(batch, want_gt=True)
    assert torch.equal(out, ref_out) and torch.equal(grids.occ, ref_grids.occ) and torch.equal(grids.gt_occ, ref_grids.gt_occ)
    # refill the same device buffers with other tiles (same sizes) and replay
    tiles2, labels2 = zip(*[synthetic_tile(100 + t, 20_000) for t in range(3)])
    other = sna.PointBatch.from_tiles(tiles2, labels2, device=hip_device)
    batch.pts.copy_(other.pts)
    batch.labels.copy_(other.labels)
    out2, grids2 = cap.replay()
    with torch.no_grad():
        ref2, refg2 = pipe(other, want_gt=True)
    assert torch.equal(out2, ref2) and torch.equal(grids2.occ, refg2.occ)
    assert not torch.equal(ref2, ref_out)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_float_occupancy_grids_take_the_int8_path_on_a_device_side_check(hip_device, dt):
    """The reference feeds f64 {0., 1.} grids (ToFullDense).  sn_forward_auto checks that on the device and gates the
    int8 and the fp32 launches on the result: same answer either way, no host sync."""
    from scene_net_amd import _hip
    torch.manual_seed(3)
    model = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9)).to(hip_device)
    bank, lam = model.compute_bank(hip_device), model.effective_lambdas(hip_device)
    occ = torch.rand((2, 1, 24, 20, 32)) < 0.2
    x = occ.to(dt).to(hip_device)
    out, flag = _hip.forward_auto(x, bank, lam)
    assert flag.item() == 0 and out.dtype == dt
    ref_int8 = _hip.conv_fused(occ.to(hip_device), bank, lam, out_dtype=dt)
    assert torch.equal(out, ref_int8)                                  # the gated int8 launch produced the output
    _, ref_fp32 = _hip.conv_bank(x, bank, lam, want_act=False, want_out=True)
    assert (out - ref_fp32).abs().max().item() < 2e-5
    # one non-binary element: the flag rises and the fp32 contraction's output comes back, bit for bit
    x2 = x.clone()
    x2[1, 0, 3, 4, 5] = 0.5
    out2, flag2 = _hip.forward_auto(x2, bank, lam)
    assert flag2.item() == 1
    assert torch.equal(out2, _hip.conv_bank(x2, bank, lam, want_act=False, want_out=True)[1])
    x3 = x.clone()
    x3[0, 0, 0, 0, 0] = float("nan")
    out3, flag3 = _hip.forward_auto(x3, bank, lam)
    assert flag3.item() == 1 and torch.isnan(out3[0, 0, 0, 0, 0])
    # the module routes float grids through it
    with torch.no_grad():
        assert torch.equal(model(x), out)
    # ragged size (numel % 4 != 0, Y % 4 != 0: the int8 kernels do not serve it -> bytes through the fp32 kernel)
    occ_r = torch.rand((1, 1, 5, 7, 9)) < 0.3
    xr = occ_r.to(dt).to(hip_device)
    outr, flagr = _hip.forward_auto(xr, bank, lam)
    assert flagr.item() == 0
    assert (outr - _hip.conv_bank(xr, bank, lam, want_act=False, want_out=True)[1]).abs().max().item() < 1e-6


def _kitti_like_scan(seed, n=120_000):
    """A SemanticKITTI-shaped scan (sensor frame): ground returns thinning with range, walls, poles (label 80)."""
    rng = np.random.default_rng(seed)
    r = 2.0 + 48.0 * rng.random(n) ** 1.7
    a = rng.random(n) * 2 * np.pi
    z = -1.7 + 0.02 * r * rng.standard_normal(n)
    wall = rng.random(n) < 0.25
    z[wall] = -1.7 + rng.random(wall.sum()) * rng.choice([2.0, 4.0, 8.0], wall.sum())
    pts = np.stack([r * np.cos(a), r * np.sin(a), z], axis=1)
    labels = np.where(wall & (rng.random(n) < 0.1), 80.0, 40.0)
    return pts, labels


def test_c4_chain_size_mode_conv_head_per_point_gather(hip_device):
    """BASELINE C4 as ONE call: 4 Velodyne-shaped scans -> voxel-size mode at a 128^3 capacity (pcd_processing.py:365-367,
    semKITTI.py:453-455) -> GENEO bank conv + head -> per-point read-back, against the oracle chain on each scan's OWN grid
    (voxel_oracle -> geneo_oracle.scenenet_forward -> index by the oracle's voxel ids).  The padded part of our grid is
    zero, which is exactly the zero padding of conv3d(padding='same') on the scan's own grid."""
    torch.manual_seed(44)
    scans, labels = zip(*[_kitti_like_scan(900 + i, 60_000 + 20_000 * i) for i in range(4)])
    scans = [s * k for s, k in zip(scans, (1.0, 0.9, 0.75, 1.0))]       # different extents: different grids
    vs = (0.9, 0.9, 0.9)
    model = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9)).to(hip_device)
    model.fused_forward = False          # the 16-kernel style contraction (here 4 kernels): K3'
    batch = sna.PointBatch.from_tiles(scans, labels, device=hip_device)
    pipe = sna.ScenePipeline(model, (128, 128, 128), keep_labels=[80.0], voxel_dims=vs, per_point=True)
    with torch.no_grad():
        out, grids, per_point = pipe(batch, want_gt=True)
        model.fused_forward = True       # and through linearity (K3L): same chain
        out_l, per_point_l = pipe(batch)
    assert grids.status.cpu().tolist() == [0] * 4 and grids.counts is None      # the LDS-bitmap kernels served it
    assert per_point.shape == (1, batch.total_points)
    off = batch.offsets.cpu().numpy()
    seen = set()
    for b in range(4):
        counts, towers, gv = vo.voxel_counts(scans[b], None, vs, labels[b], [80.0])
        nx, ny, nz = (int(v) for v in gv["x_y_z"])
        seen.add((nx, ny, nz))
        assert grids.dims[b].cpu().tolist() == [nx, ny, nz] and max(nx, ny, nz) <= 128
        occ = vo.to_full_dense(vo.normalize_xyz(counts.astype(np.float64)))
        assert np.array_equal(grids.occ[b, 0].cpu().numpy()[:nz, :nx, :ny], occ > 0)
        assert np.array_equal(grids.gt_occ[b, 0].cpu().numpy()[:nz, :nx, :ny], towers > 0)
        ref, _ = _oracle_forward(model, torch.from_numpy(occ)[None, None])       # on the scan's own grid
        ref = ref[0, 0].numpy()
        got = out[b, 0].double().cpu().numpy()
        assert np.abs(got[:nz, :nx, :ny] - ref).max() < TOL
        assert np.abs(out_l[b, 0].double().cpu().numpy()[:nz, :nx, :ny] - ref).max() < TOL
        # per point: the prediction of the voxel the oracle bins the point into
        vx, vy, vz = gv["voxel_x"], gv["voxel_y"], gv["voxel_z"]
        want = ref[vz, vx, vy]
        pp = per_point[0, off[b]:off[b + 1]].double().cpu().numpy()
        assert np.abs(pp - want).max() < TOL
        assert np.array_equal(pp, got[vz, vx, vy])                               # exactly the voxel's value: same binning
        assert np.abs(per_point_l[0, off[b]:off[b + 1]].double().cpu().numpy() - want).max() < TOL
    assert len(seen) > 1
    # thresholded per-point labels (prob_to_label, voxelization.py:304-323)
    pipe_tau = sna.ScenePipeline(model, (128, 128, 128), voxel_dims=vs, per_point=True, tau=0.5)
    with torch.no_grad():
        _, lab = pipe_tau(batch)
    assert torch.equal(lab, (per_point_l >= 0.5).to(lab.dtype))


def test_single_tile_size_mode_without_a_host_round_trip_in_the_middle(hip_device, golden_dir):
    """hist_on_voxel / reg_on_voxel(voxel_dims=...): the whole chain on the device with `voxelgrid_dims` as capacity, the
    tile's own dims read back with the grid; a capacity that is too small falls back to the host-sized route.  Same
    arrays either way (and as the oracle)."""
    a = np.load(os.path.join(golden_dir, "ts40k_sample575_subset.npy"))
    xyz, labels = a[:, :3], a[:, 3]
    vs = (1.0, 1.0, 1.0)
    ref = vo.hist_on_voxel(xyz, voxel_dims=vs)
    big = sna.hist_on_voxel(xyz, (128, 128, 128), voxel_dims=vs)          # fits: cut from the padded grid
    small = sna.hist_on_voxel(xyz, (8, 8, 8), voxel_dims=vs)              # does not fit: host-sized route
    assert big.shape == ref.shape == small.shape
    assert np.array_equal(big, ref) and np.array_equal(small, ref)
    assert np.array_equal(sna.reg_on_voxel(xyz, labels, [15], (128, 128, 128), voxel_dims=vs),
                          vo.reg_on_voxel(xyz, labels, [15], voxel_dims=vs))
