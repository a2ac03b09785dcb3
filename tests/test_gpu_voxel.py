"""K1 parity on the MI355X: bbox / edge tables / atomic scatter / finalize (through the C ABI) against the
numpy oracle.  The bar is BIT-EXACT: integer counts, fp64 edges, fp64 normalised density and ratio."""
import os

import numpy as np
import pytest
import torch

import scene_net_amd as sna
from scene_net_amd import _hip
from oracle import voxel_oracle as vo
from scene_net_amd.synthetic import synthetic_tile

pytestmark = pytest.mark.gpu


def _check_tile(grids, b, xyz, labels, dims, keep):
    counts, towers, g = vo.voxel_counts(xyz, dims, None, labels, keep)
    nx, ny, nz = dims
    desc = grids.desc[b].cpu().numpy()
    assert np.array_equal(desc[:3], g["xyzmin"]) and np.array_equal(desc[3:6], g["xyzmax"])
    edges = np.concatenate(g["segments"])
    assert np.array_equal(desc[6:], edges), "edge table differs from numpy.linspace"
    if grids.counts is not None:
        assert np.array_equal(grids.counts[b].cpu().numpy(), counts)
    assert int(grids.dropped[b].item()) == 0
    if grids.towers is not None:
        assert np.array_equal(grids.towers[b].cpu().numpy(), towers)
    dens = vo.normalize_xyz(counts.astype(np.float64))
    if grids.density is not None:
        assert np.array_equal(grids.density[b, 0].cpu().numpy(), dens)
    if grids.occ is not None:
        assert np.array_equal(grids.occ[b, 0].cpu().numpy(), vo.to_full_dense(dens).astype(np.float32))
    if grids.gt_occ is not None and grids.gt is None:
        gt = vo.reg_on_voxel(xyz, labels, keep, dims)
        assert np.array_equal(grids.gt_occ[b, 0].cpu().numpy(), (gt > 0).astype(np.float32))
    if grids.gt is not None:
        gt = vo.reg_on_voxel(xyz, labels, keep, dims)
        assert np.array_equal(grids.gt[b, 0].cpu().numpy(), gt)
        if grids.gt_occ is not None:
            assert np.array_equal(grids.gt_occ[b, 0].cpu().numpy(), (gt > 0).astype(np.float32))


def test_real_ts40k_tile_utm_coordinates(hip_device, golden_dir):
    a = np.load(os.path.join(golden_dir, "ts40k_sample575_subset.npy"))
    xyz, labels = a[:, :3], a[:, 3]
    for dims in [(64, 64, 64), (32, 16, 48), (128, 128, 128)]:
        batch = sna.PointBatch.from_tiles([xyz], [labels], device=hip_device)
        g = sna.voxelize_batch(batch, dims, [15], want_density=True, want_gt=True, want_occ=True, want_gt_occ=True)
        _check_tile(g, 0, xyz, labels, dims, [15])


def test_c1_full_tile_64cubed(hip_device, golden_dir):
    """BASELINE C1's input verbatim (all 58 243 rows of sample_575.npy) at the configuration BASELINE names: bbox, every
    fp64 edge, counts, density, ratio and both occupancy planes bit-exact vs the oracle, counting and LDS-bitmap paths"""
    a = np.load(os.path.join(golden_dir, "ts40k_sample575_full.npz"))["tile"]
    xyz, labels = a[:, :3], a[:, 3]
    batch = sna.PointBatch.from_tiles([xyz], [labels], device=hip_device)
    g = sna.voxelize_batch(batch, (64, 64, 64), [15], want_density=True, want_gt=True, want_occ=True, want_gt_occ=True)
    _check_tile(g, 0, xyz, labels, (64, 64, 64), [15])
    assert int(g.counts[0].sum().item()) == 58243
    go1 = sna.voxelize_batch(batch, (64, 64, 64), [15], want_occ=True, want_gt_occ=True, occ_dtype=torch.bool)
    _check_tile(go1, 0, xyz, labels, (64, 64, 64), [15])


def test_ragged_batch_of_synthetic_tiles(hip_device):
    sizes = [20_001, 36_076, 1, 2, 3, 58_243, 117_111, 64]  # odd/even offsets exercise the 16-byte peel
    tiles, labels = zip(*[synthetic_tile(t, max(n, 3))[0:2] for t, n in enumerate(sizes)])
    tiles = [t[:n] for t, n in zip(tiles, sizes)]
    labels = [l[:n] for l, n in zip(labels, sizes)]
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    assert batch.offsets.tolist() == np.concatenate([[0], np.cumsum(sizes)]).tolist()
    g = sna.voxelize_batch(batch, (64, 64, 64), [15, 16], want_density=True, want_gt=True, want_occ=True,
                           want_gt_occ=True)
    # the occupancy-only (LDS bitmap) path on the same batch, u8 and f32, with and without the tower plane
    go1 = sna.voxelize_batch(batch, (64, 64, 64), [15, 16], want_occ=True, want_gt_occ=True, occ_dtype=torch.uint8)
    go2 = sna.voxelize_batch(batch, (64, 64, 64), want_occ=True)
    assert go1.counts is None and go1.occ.dtype == torch.uint8 and go2.occ.dtype == torch.float32
    assert go1.flags.sum().item() == 0  # every tile has an empty (z,x) row: no counting fallback
    for b in range(len(sizes)):
        _check_tile(g, b, tiles[b], labels[b], (64, 64, 64), [15, 16])
        _check_tile(go1, b, tiles[b], labels[b], (64, 64, 64), [15, 16])
        _check_tile(go2, b, tiles[b], None, (64, 64, 64), None)
        assert int(g.counts[b].sum().item()) == sizes[b]


def test_points_on_edges_and_degenerate_axes(hip_device):
    # lattice points sit exactly on voxel edges: the (e_k, e_{k+1}] convention decides the bin
    ax = np.linspace(0.0, 8.0, 17)
    lattice = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    flat = np.concatenate([lattice[:, :2], np.full((len(lattice), 1), 3.0)], 1)  # zero z-range -> cube padding
    same = np.tile(np.array([[5.44e5, 4.634e6, 150.0]]), (50, 1))  # every point identical: all edges equal
    shifted = lattice * 0.37 + np.array([5.44e5, 4.634e6, 150.0])
    tiles = [lattice, flat, same, shifted]
    batch = sna.PointBatch.from_tiles(tiles, device=hip_device)
    for dims in [(8, 8, 8), (16, 16, 16), (5, 7, 3)]:
        g = sna.voxelize_batch(batch, dims, want_density=True, want_occ=True)
        for b, t in enumerate(tiles):
            _check_tile(g, b, t, None, dims, None)


def test_points_one_ulp_around_every_edge(hip_device):
    """Adversarial for the guess-then-confirm binning: coordinates exactly on every edge of the tile's own linspace
    table and one ulp to either side, at UTM scale, through the counting path and the occupancy-only (bitmap) path."""
    rng = np.random.default_rng(77)
    dims = (64, 64, 64)
    tiles, labs = [], []
    for t in range(2):
        xyz, lab = synthetic_tile(40 + t, 30_000)
        _, _, g = vo.voxel_counts(xyz, dims, None, lab, [15.0])
        lo, hi = g["xyzmin"], g["xyzmax"]
        extra = []
        for a in range(3):
            e = g["segments"][a]
            vals = np.concatenate([e, np.nextafter(e, np.inf), np.nextafter(e, -np.inf)])
            vals = vals[(vals >= xyz[:, a].min()) & (vals <= xyz[:, a].max())]   # keep the bounding box unchanged
            pts = lo + rng.random((len(vals), 3)) * (np.minimum(hi, xyz.max(0)) - lo) * 0.999
            pts = np.clip(pts, xyz.min(0), xyz.max(0))
            pts[:, a] = vals
            extra.append(pts)
        extra = np.concatenate(extra)
        tiles.append(np.concatenate([xyz, extra]))
        labs.append(np.concatenate([lab, np.where(rng.random(len(extra)) < 0.3, 15.0, 2.0)]))
    batch = sna.PointBatch.from_tiles(tiles, labs, device=hip_device)
    full = sna.voxelize_batch(batch, dims, [15.0], want_density=True, want_gt=True, want_occ=True, want_counts=True)
    fast = sna.voxelize_batch(batch, dims, [15.0], want_occ=True, want_gt_occ=True)
    assert fast.counts is None   # the bitmap path
    for b in range(2):
        _check_tile(full, b, tiles[b], labs[b], dims, [15.0])
        _check_tile(fast, b, tiles[b], labs[b], dims, [15.0])


def test_density_column_rule(hip_device):
    # a fully occupied y-column has min > 0: its minimum cells normalise to 0 (ToFullDense -> 0)
    rng = np.random.default_rng(2)
    pts = []
    for z in range(4):
        for x in range(4):
            for y in range(4):
                k = 3 if y == 2 else int(rng.integers(0, 3))
                if y == 2:
                    k = int(rng.integers(1, 4))
                pts += [[x + 0.5, y + 0.5, z + 0.5]] * k
    pts += [[0.0, 0.0, 0.0], [4.0, 4.0, 4.0]]
    xyz = np.array(pts)
    batch = sna.PointBatch.from_tiles([xyz], device=hip_device)
    g = sna.voxelize_batch(batch, (4, 4, 4), want_density=True, want_occ=True)
    _check_tile(g, 0, xyz, None, (4, 4, 4), None)
    counts = g.counts[0].cpu().numpy()
    assert counts[:, :, 2].min() > 0
    assert (g.occ[0, 0].cpu().numpy()[:, :, 2] == (counts[:, :, 2] > counts[:, :, 2].min())).all()
    # occupancy path: no (z,x) row is empty -> the flag is raised and the gated counting kernels redo the tile
    labels = np.where(np.arange(len(xyz)) % 3 == 0, 15.0, 2.0)
    other, _ = synthetic_tile(1, 5000)
    b2 = sna.PointBatch.from_tiles([xyz, other, xyz], [labels, np.full(5000, 2.0), labels], device=hip_device)
    for dt in (torch.uint8, torch.float32):
        g2 = sna.voxelize_batch(b2, (4, 4, 8), [15], want_occ=True, want_gt_occ=True, occ_dtype=dt)
        assert g2.counts is None
        for b, (t, l) in enumerate([(xyz, labels), (other, np.full(5000, 2.0)), (xyz, labels)]):
            _check_tile(g2, b, t, l, (4, 4, 8), [15])
    g3 = sna.voxelize_batch(b2, (4, 4, 4), [15], want_occ=True, occ_dtype=torch.uint8)
    assert g3.flags.tolist()[0] == 1 and g3.flags.tolist()[2] == 1
    _check_tile(g3, 0, xyz, None, (4, 4, 4), None)
    assert not np.array_equal(g3.occ[0, 0].cpu().numpy(), (counts > 0))  # the rule matters for this tile


def test_size_mode_and_reference_signatures(hip_device, golden_dir):
    a = np.load(os.path.join(golden_dir, "ts40k_sample575_subset.npy"))
    xyz, labels = a[:, :3], a[:, 3]
    assert np.array_equal(sna.hist_on_voxel(xyz), vo.hist_on_voxel(xyz))
    assert np.array_equal(sna.hist_on_voxel(xyz, (32, 32, 32)), vo.hist_on_voxel(xyz, (32, 32, 32)))
    assert np.array_equal(sna.reg_on_voxel(xyz, labels, [15]), vo.reg_on_voxel(xyz, labels, [15]))
    assert np.array_equal(sna.reg_on_voxel(xyz, labels, 15), vo.reg_on_voxel(xyz, labels, 15))
    for vs in [(1.0, 1.0, 1.0), (0.5, 2.0, 1.5)]:
        got = sna.hist_on_voxel(xyz, voxel_dims=vs)
        ref = vo.hist_on_voxel(xyz, voxel_dims=vs)
        assert got.shape == ref.shape and np.array_equal(got, ref)
        assert np.array_equal(sna.reg_on_voxel(xyz, labels, [15], voxel_dims=vs),
                              vo.reg_on_voxel(xyz, labels, [15], voxel_dims=vs))
    vox, gt = sna.Voxelization([15], vxg_size=(64, 64, 64))((xyz, labels))
    rv, rg = vo.voxelization_call((xyz, labels), [15], None, (64, 64, 64))
    assert vox.shape == (1, 64, 64, 64) and vox.dtype == np.float64
    assert np.array_equal(vox, rv) and np.array_equal(gt, rg)
    t = sna.ToFullDense(apply=(True, True))(sna.ToTensor()((vox, gt)))
    assert np.array_equal(t[0].numpy(), vo.to_full_dense(rv)) and np.array_equal(t[1].numpy(), vo.to_full_dense(rg))


def test_full_size_c2_batch_properties(hip_device):
    """BASELINE C2: 32 tiles x 100k points, 64^3.  Every tile bit-exact against the oracle (it runs in
    milliseconds per tile) plus count conservation and run-to-run determinism of the atomics."""
    B, N = 32, 100_000
    tiles, labels = zip(*[synthetic_tile(t, N) for t in range(B)])
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    g1 = sna.voxelize_batch(batch, (64, 64, 64), [15], want_occ=True, want_gt_occ=True, want_counts=True)
    g2 = sna.voxelize_batch(batch, (64, 64, 64), [15], want_occ=True, want_gt_occ=True, want_counts=True)
    g4 = sna.voxelize_batch(batch, (64, 64, 64), [15], want_occ=True, want_gt_occ=True, occ_dtype=torch.uint8)
    assert g4.counts is None and torch.equal(g4.occ.float(), g1.occ) and torch.equal(g4.gt_occ.float(), g1.gt_occ)
    assert int(g4.flags.sum().item()) == 0 and int(g4.dropped.sum().item()) == 0
    assert torch.equal(g1.counts, g2.counts) and torch.equal(g1.towers, g2.towers) and torch.equal(g1.occ, g2.occ)
    assert g1.counts.sum(dim=(1, 2, 3)).tolist() == [N] * B
    assert int(g1.dropped.sum().item()) == 0
    for b in range(B):
        _check_tile(g1, b, tiles[b], labels[b], (64, 64, 64), [15])
    # 128^3 (C3 grid) on the same clouds: the LDS-bitmap path in 4 z-slabs (8 with the tower plane)
    g3 = sna.voxelize_batch(batch, (128, 128, 128), [15], want_occ=True, occ_dtype=torch.bool)
    g5 = sna.voxelize_batch(batch, (128, 128, 128), [15], want_occ=True, want_gt_occ=True)
    assert g3.counts is None and g5.counts is None and int(g3.flags.sum().item()) == 0
    for b in (0, 17, 31):
        _check_tile(g3, b, tiles[b], labels[b], (128, 128, 128), [15])
        _check_tile(g5, b, tiles[b], labels[b], (128, 128, 128), [15])
    # non-cubic grid whose bitmap needs 2 slabs
    g6 = sna.voxelize_batch(batch, (64, 128, 128), want_occ=True)
    assert g6.counts is None
    _check_tile(g6, 5, tiles[5], None, (64, 128, 128), None)


def test_unaligned_point_buffer(hip_device):
    xyz, _ = synthetic_tile(3, 5_001)
    big = torch.zeros(5_001 * 3 + 1, dtype=torch.float64, device=hip_device)
    view = big[1:].view(5_001, 3)  # 8-byte but not 16-byte aligned
    view.copy_(torch.from_numpy(xyz))
    assert view.data_ptr() % 16 == 8
    offsets = torch.tensor([0, 5_001], dtype=torch.int64, device=hip_device)
    batch = sna.PointBatch(view, None, offsets, (5_001,))
    g = sna.voxelize_batch(batch, (32, 32, 32), want_occ=True)
    _check_tile(g, 0, xyz, None, (32, 32, 32), None)


def test_separate_bbox_and_desc_entry_points_agree_with_prepare(hip_device):
    tiles = [synthetic_tile(t, n)[0] for t, n in enumerate([7_001, 12_000, 3])]
    batch = sna.PointBatch.from_tiles(tiles, device=hip_device)
    bbox = _hip.voxel_bbox(batch.pts, batch.offsets)
    desc = _hip.voxel_desc(bbox, (16, 32, 8), regular=True)
    desc2, bbox2 = _hip.voxel_prepare(batch.pts, batch.offsets, (16, 32, 8), want_bbox=True)
    assert torch.equal(bbox, bbox2) and torch.equal(desc, desc2)
    for b, t in enumerate(tiles):
        assert np.array_equal(bbox[b].cpu().numpy(), np.concatenate([t.min(0), t.max(0)]))


def test_vxg_to_xyz_bit_exact(golden_dir, hip_device):
    """sn_grid_to_points == the reference's vxg_to_xyz rows (golden) and the oracle at a 64^3 grid, all dtypes."""
    z = np.load(os.path.join(golden_dir, "vxg_to_xyz.npz"))
    for k in z["cases"]:
        o = z[f"{k}/origin"] if f"{k}/origin" in z else None
        vs = z[f"{k}/voxel_size"] if f"{k}/voxel_size" in z else None
        got = sna.vxg_to_xyz(torch.from_numpy(z[f"{k}/grid"]), o, vs)
        assert got.dtype == np.float64 and np.array_equal(got, z[f"{k}/rows"]), k
    rng = np.random.default_rng(5)
    origin, size = np.array([5.44e5 + 0.37, 4.634e6 - 0.11, 149.93]), np.array([0.9375, 0.46875, 0.3])
    for dt in (np.float32, np.float64, np.uint8, np.bool_):
        r = rng.random((64, 48, 32))
        g = (r < 0.2).astype(dt) if dt in (np.uint8, np.bool_) else r.astype(dt)
        got = sna.vxg_to_xyz(torch.from_numpy(g).to(hip_device), origin, size, as_tensor=True)
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), vo.vxg_to_xyz(g, origin, size)), dt
    with pytest.raises(ValueError):
        sna.vxg_to_xyz(np.zeros((4, 4)))


# ------------------------------------------------------------------ voxel-size mode for whole batches (C4)
def _velodyne_scan(seed, n=120_000):
    """A SemanticKITTI-shaped scan: a disc of ground returns thinning with range, verticals, in the sensor frame."""
    rng = np.random.default_rng(seed)
    r = 2.0 + 48.0 * rng.random(n) ** 1.7
    a = rng.random(n) * 2 * np.pi
    z = -1.7 + 0.02 * r * rng.standard_normal(n)
    wall = rng.random(n) < 0.25
    z[wall] = -1.7 + rng.random(wall.sum()) * rng.choice([2.0, 4.0, 8.0], wall.sum())
    pts = np.stack([r * np.cos(a), r * np.sin(a), z], axis=1)
    labels = np.where(wall & (rng.random(n) < 0.1), 80.0, 40.0)   # 80 = pole
    return pts, labels


def test_size_mode_batch_on_device_against_oracle(hip_device):
    """voxelize_batch(voxel_dims=...): per-tile grid extents computed on the device (no host round trip), grids
    padded to a maximum -- counts, density, ratio, dims and the per-point read-back against the oracle."""
    tiles, labels = zip(*[_velodyne_scan(50 + i, 30_000 + 7_000 * i) for i in range(3)])
    # different extents per tile: scale the scans differently
    tiles = [t * s for t, s in zip(tiles, (1.0, 0.55, 0.8))]
    vox = (1.7, 1.9, 1.6)   # the box is cubed first (regular_bounding_box), so every axis spans the largest extent
    maxd = (64, 64, 64)
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    g = sna.voxelize_batch(batch, maxd, [80.0], want_density=True, want_gt=True, want_occ=True, want_gt_occ=True,
                           voxel_dims=vox)
    assert g.status.cpu().tolist() == [0, 0, 0] and g.dropped.cpu().tolist() == [0, 0, 0]
    seen = set()
    for b in range(3):
        counts, towers, gv = vo.voxel_counts(tiles[b], None, vox, labels[b], [80.0])
        nx, ny, nz = (int(v) for v in gv["x_y_z"])
        seen.add((nx, ny, nz))
        assert g.dims[b].cpu().tolist() == [nx, ny, nz]
        assert max(nx, ny, nz) <= 64
        c = g.counts[b].cpu().numpy()
        assert np.array_equal(c[:nz, :nx, :ny], counts) and c.sum() == counts.sum()      # nothing outside the tile's part
        assert np.array_equal(g.towers[b].cpu().numpy()[:nz, :nx, :ny], towers)
        dens = vo.normalize_xyz(counts.astype(np.float64))
        d = g.density[b, 0].cpu().numpy()
        assert np.array_equal(d[:nz, :nx, :ny], dens) and np.count_nonzero(d) == np.count_nonzero(dens)
        ratio = np.zeros(counts.shape)
        ratio[counts > 0] = towers[counts > 0] / counts[counts > 0]
        assert np.array_equal(g.gt[b, 0].cpu().numpy()[:nz, :nx, :ny], ratio)
        assert np.array_equal(g.occ[b, 0].cpu().numpy()[:nz, :nx, :ny], vo.to_full_dense(dens))
        # every edge of the tile's own tables, bit for bit; +inf beyond
        e = g.desc[b].cpu().numpy()
        ex, ey, ez = e[6:6 + 65], e[6 + 65:6 + 130], e[6 + 130:6 + 195]
        for got, want, n in ((ex, gv["segments"][0], nx), (ey, gv["segments"][1], ny), (ez, gv["segments"][2], nz)):
            assert np.array_equal(got[:n + 1], want) and np.all(np.isinf(got[n + 1:]))
    assert len(seen) > 1   # the tiles really have different grids
    # per-point read-back of a grid through the same descriptor
    vals = torch.arange(3 * 64 ** 3, dtype=torch.float32, device=hip_device).reshape(3, 1, 64, 64, 64)
    got = _hip.gather_points(vals, batch.pts, batch.offsets, g.desc)[0].cpu().numpy()
    o = 0
    for b in range(3):
        gv = vo._voxelize(tiles[b], None, vox)
        flat = b * 64 ** 3 + (gv["voxel_z"] * 64 + gv["voxel_x"]) * 64 + gv["voxel_y"]
        assert np.array_equal(got[o:o + len(flat)], flat.astype(np.float32))
        o += len(flat)
    # a maximum that is too small is reported, not silently wrong
    g2 = sna.voxelize_batch(batch, (16, 16, 16), voxel_dims=vox, want_counts=True)
    assert g2.status.cpu().tolist()[0] == 1 and g2.dropped[0].item() > 0


def test_size_mode_128_cubed_velodyne_scans_full_size(hip_device):
    """BASELINE C4's shape: 120k-point scans into 128^3-capacity grids with a fixed voxel size; counts and the per-point
    gather against the oracle at full size."""
    tiles, labels = zip(*[_velodyne_scan(7 + i) for i in range(4)])
    vox = (0.8, 0.85, 0.8)   # ~100 m cube -> about 125 x 118 x 125 voxels
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    g = sna.voxelize_batch(batch, (128, 128, 128), [80.0], want_occ=True, want_gt_occ=True, want_counts=True,
                           voxel_dims=vox, occ_dtype=torch.bool)
    assert g.status.sum().item() == 0 and g.dropped.sum().item() == 0
    pred = torch.rand((4, 1, 128, 128, 128), device=hip_device)
    got = _hip.gather_points(pred, batch.pts, batch.offsets, g.desc)[0].cpu().numpy()
    o = 0
    for b in range(4):
        counts, towers, gv = vo.voxel_counts(tiles[b], None, vox, labels[b], [80.0])
        nx, ny, nz = (int(v) for v in gv["x_y_z"])
        assert g.dims[b].cpu().tolist() == [nx, ny, nz] and max(nx, ny, nz) <= 128
        assert np.array_equal(g.counts[b].cpu().numpy()[:nz, :nx, :ny], counts)
        assert g.counts[b].sum().item() == len(tiles[b])
        occ = vo.to_full_dense(vo.normalize_xyz(counts.astype(np.float64))) > 0
        assert np.array_equal(g.occ[b, 0].cpu().numpy()[:nz, :nx, :ny], occ)
        assert np.array_equal(g.gt_occ[b, 0].cpu().numpy()[:nz, :nx, :ny], towers > 0)
        want = pred[b, 0].cpu().numpy()[gv["voxel_z"], gv["voxel_x"], gv["voxel_y"]]
        assert np.array_equal(got[o:o + len(want)], want)
        o += len(want)


def test_size_mode_on_the_lds_bitmap_kernels(hip_device):
    """voxelize_batch(voxel_dims=...) with only the binary grids wanted (C4's mode) runs sn_voxel_occupancy_sized: the
    LDS-bitmap kernels on the padded per-tile tables.  Bit-identical to the counting kernels' size mode and to the oracle,
    occupancy and tower plane, incl. four 120 k-point Velodyne-shaped scans at the 128^3 capacity."""
    tiles, labels = zip(*[_velodyne_scan(50 + i, 30_000 + 7_000 * i) for i in range(3)])
    tiles = [t * s for t, s in zip(tiles, (1.0, 0.55, 0.8))]
    vox = (1.7, 1.9, 1.6)
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    slow = sna.voxelize_batch(batch, (64, 64, 64), [80.0], want_occ=True, want_gt_occ=True, want_counts=True,
                              voxel_dims=vox, occ_dtype=torch.bool)
    fast = sna.voxelize_batch(batch, (64, 64, 64), [80.0], want_occ=True, want_gt_occ=True, voxel_dims=vox,
                              occ_dtype=torch.bool)
    assert slow.counts is not None and fast.counts is None          # the two paths
    assert torch.equal(fast.occ, slow.occ) and torch.equal(fast.gt_occ, slow.gt_occ)
    assert torch.equal(fast.dims, slow.dims) and torch.equal(fast.desc, slow.desc)
    assert fast.status.cpu().tolist() == [0, 0, 0] and fast.dropped.cpu().tolist() == [0, 0, 0]
    assert int(fast.flags.sum().item()) == 0                        # every tile has an empty row inside its OWN part
    for b in range(3):
        counts, towers, gv = vo.voxel_counts(tiles[b], None, vox, labels[b], [80.0])
        nx, ny, nz = (int(v) for v in gv["x_y_z"])
        occ = fast.occ[b, 0].cpu().numpy()
        assert np.array_equal(occ[:nz, :nx, :ny], vo.to_full_dense(vo.normalize_xyz(counts.astype(np.float64))) > 0)
        assert occ.sum() == occ[:nz, :nx, :ny].sum()               # nothing outside the tile's own part
        assert np.array_equal(fast.gt_occ[b, 0].cpu().numpy()[:nz, :nx, :ny], towers > 0)
    # without the tower plane, f32 output
    f32 = sna.voxelize_batch(batch, (64, 64, 64), voxel_dims=vox)
    assert f32.occ.dtype == torch.float32 and torch.equal(f32.occ.bool(), fast.occ)
    # C4-sized: 4 scans x 120 k points, 128^3 capacity (4 z-slabs; 8 with the tower plane)
    scans, slab = zip(*[_velodyne_scan(300 + i) for i in range(4)])
    big = sna.PointBatch.from_tiles(scans, slab, device=hip_device)
    vs = (0.8, 0.8, 0.8)
    g = sna.voxelize_batch(big, (128, 128, 128), [80.0], want_occ=True, want_gt_occ=True, voxel_dims=vs,
                           occ_dtype=torch.bool)
    assert g.counts is None and g.status.cpu().tolist() == [0] * 4
    for b in range(4):
        counts, towers, gv = vo.voxel_counts(scans[b], None, vs, slab[b], [80.0])
        nx, ny, nz = (int(v) for v in gv["x_y_z"])
        assert g.dims[b].cpu().tolist() == [nx, ny, nz]
        assert np.array_equal(g.occ[b, 0].cpu().numpy()[:nz, :nx, :ny], counts > 0)
        assert np.array_equal(g.gt_occ[b, 0].cpu().numpy()[:nz, :nx, :ny], towers > 0)


def test_size_mode_full_column_takes_the_exact_fallback(hip_device):
    """a tile whose OWN part has no empty (z, x) row -- here: a y column occupied in every row -- must not be served by
    `count > 0`: the padding rows are empty but are not part of the reference's grid.  The flag goes up, the gated counting
    kernel redoes the tile with the column minima of its own part, and the result equals the counting kernels' size mode."""
    rng = np.random.default_rng(9)
    vs = (1.0, 1.0, 1.0)
    base = np.concatenate([rng.uniform(0.2, 5.8, (400, 3)), np.array([[0.0, 0.0, 0.0], [6.0, 6.0, 6.0]])])
    gv = vo.voxelgrid_compute(base, sizes=vs)                      # the grid the size mode gives this box
    ex, ey, ez = gv["segments"]
    cx, cy, cz = [(e[:-1] + e[1:]) / 2 for e in (ex, ey, ez)]
    j0 = 2                                                         # one point at the centre of every (z, x) row's cell y = j0
    col = np.stack(np.meshgrid(cx, [cy[j0]], cz, indexing="ij"), -1).reshape(-1, 3)
    col = col[(col >= 0.0).all(1) & (col <= 6.0).all(1)]           # (inside the box: it, hence the grid, is unchanged)
    edge_rows = len(cx) * len(cz) - len(col)
    pts = np.concatenate([base, col])
    g2 = vo.voxelgrid_compute(pts, sizes=vs)
    assert np.array_equal(g2["xyzmin"], gv["xyzmin"]) and np.array_equal(g2["x_y_z"], gv["x_y_z"])
    batch = sna.PointBatch.from_tiles([pts], device=hip_device)
    slow = sna.voxelize_batch(batch, (32, 32, 32), want_occ=True, want_counts=True, voxel_dims=vs, occ_dtype=torch.bool)
    fast = sna.voxelize_batch(batch, (32, 32, 32), want_occ=True, voxel_dims=vs, occ_dtype=torch.bool)
    counts, _, gv = vo.voxel_counts(pts, None, vs)
    rows_empty = int((counts.sum(axis=2) == 0).sum())              # (z, x) rows of the tile's OWN grid without a point
    if edge_rows == 0:
        assert rows_empty == 0
    assert int(fast.flags[0].item()) == (1 if rows_empty == 0 else 0)
    assert torch.equal(fast.occ, slow.occ) and torch.equal(fast.dims, slow.dims)
    nx, ny, nz = (int(v) for v in gv["x_y_z"])
    want = vo.to_full_dense(vo.normalize_xyz(counts.astype(np.float64))) > 0
    assert np.array_equal(fast.occ[0, 0].cpu().numpy()[:nz, :nx, :ny], want)
    assert rows_empty == 0 and not np.array_equal(want, counts > 0)   # the case really differs from `count > 0`


def test_one_pass_kernel_equals_the_two_kernel_form_on_many_ragged_tiles(hip_device):
    """occ_onepass_kernel (round 4: the points stay in registers between the box pass and the binning; the tile's 16
    workgroups exchange partial boxes inside the launch) against the two-kernel form it replaces (sn_set_option
    "voxel_onepass", 0): the same descriptor, occupancy, tower plane, flags and dropped counts, bit for bit -- on more tiles
    than the chip holds workgroups for at once (300 x 16 workgroups: the exchange's forward-progress argument at work), with
    one- and two-point tiles, odd offsets and a tile beyond the register budget (117 111 points); and with K2 riding.
    Reference: hist_on_voxel / reg_on_voxel, utils/voxelization.py:164-204, 244-300."""
    rng = np.random.default_rng(3)
    sizes = [int(s) for s in rng.integers(1, 4000, 296)] + [2, 1, 117_111, 60_001]   # (an empty tile is refused by PointBatch, as by the reference's numpy min())
    tiles, labels = [], []
    for t, n in enumerate(sizes):
        xyz, lab = synthetic_tile(t % 40, max(n, 3))
        tiles.append(xyz[:n]); labels.append(lab[:n])
    batch = sna.PointBatch.from_tiles(tiles, labels, device=hip_device)
    model = sna.SceneNet({"cy": 2, "cone": 1, "neg": 1}, (9, 9, 9)).to(hip_device)
    outs = {}
    for mode in (1, 0, 2):         # 2: the one-pass form with the exchange's wait at zero -- every workgroup goes alone
        _hip.set_option("voxel_onepass", 1 if mode else 0)
        _hip.set_option("voxel_onepass_spin", 0 if mode == 2 else 64)
        try:
            plain = sna.voxelize_batch(batch, (64, 64, 64), [15, 16], want_occ=True, want_gt_occ=True, occ_dtype=torch.uint8)
            head = sna.voxelize_batch(batch, (64, 64, 64), want_occ=True, occ_dtype=torch.bool)
            rider = model.bank_rider(hip_device)
            ridden = sna.voxelize_batch(batch, (64, 64, 64), want_occ=True, occ_dtype=torch.bool, bank_rider=rider)
            torch.cuda.synchronize()
            outs[mode] = (plain, head, ridden, rider[2].clone(), rider[3].clone())
        finally:
            _hip.set_option("voxel_onepass", 1)
            _hip.set_option("voxel_onepass_spin", 64)
    for a, b in zip(outs[2][:3], outs[1][:3]):
        assert torch.equal(a.occ, b.occ) and torch.equal(a.desc, b.desc) and torch.equal(a.flags, b.flags)
    assert torch.equal(outs[2][0].gt_occ, outs[1][0].gt_occ)
    (p1, h1, r1, bank1, prep1), (p0, h0, r0, bank0, prep0) = outs[1], outs[0]
    live = torch.ones(len(sizes), dtype=torch.bool, device=hip_device)
    for a, b in ((p1, p0), (h1, h0), (r1, r0)):
        assert torch.equal(a.occ[live], b.occ[live]) and torch.equal(a.dropped[live], b.dropped[live])
        assert torch.equal(a.flags[live], b.flags[live]) and torch.equal(a.desc[live], b.desc[live])
    assert torch.equal(p1.gt_occ[live], p0.gt_occ[live])
    assert torch.equal(bank1, bank0) and torch.equal(prep1, prep0) and r1.rider_done
    assert torch.equal(bank1, model.compute_bank(hip_device))
    assert _hip.device_status()[0] == 0
    for b in (5, 297, 298, 299):                                     # and against the oracle
        if sizes[b] > 1:
            _check_tile(p1, b, tiles[b], labels[b], (64, 64, 64), [15, 16])
