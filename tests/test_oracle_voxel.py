"""Voxel oracle (parity UNPINNED at the pyntcloud boundary -- see oracle/voxel_oracle.py): internal
consistency, the reference's own loop shape, invariants, edge cases."""
import os

import numpy as np
import pytest

from oracle import voxel_oracle as vo
from scene_net_amd.synthetic import synthetic_tile


@pytest.fixture(scope="module")
def real_tile(golden_dir):
    a = np.load(os.path.join(golden_dir, "ts40k_sample575_subset.npy"))
    return a[:, :3], a[:, 3]


@pytest.fixture(scope="module")
def c1_tile(golden_dir):
    """BASELINE C1's input verbatim: all of the reference's data-sample/sample_575.npy (make_golden.py tile)."""
    a = np.load(os.path.join(golden_dir, "ts40k_sample575_full.npz"))["tile"]
    assert a.shape == (58243, 4)
    return a[:, :3], a[:, 3]


def test_c1_full_tile_counts_and_groupby_shape(c1_tile):
    """the whole C1 tile at 64^3 (real UTM coordinates, |y| ~ 4.6e6): every point lands in exactly one voxel, tower
    counts match the label column, and the bincount form equals the reference's groupby/iterrows loop shape"""
    xyz, labels = c1_tile
    counts, towers, g = vo.voxel_counts(xyz, (64, 64, 64), None, labels, [15])
    assert counts.sum() == 58243 and towers.sum() == (labels == 15).sum() and (towers <= counts).all()
    assert np.array_equal(vo.hist_on_voxel(xyz), vo.hist_on_voxel_groupby(xyz))
    # the subset fixture is drawn from this tile and keeps its bbox-defining points: same grid geometry
    sub = np.load(os.path.join(os.path.dirname(__file__), "golden", "ts40k_sample575_subset.npy"))
    g2 = vo.voxelgrid_compute(sub[:, :3], n_xyz=(64, 64, 64))
    assert np.array_equal(g["xyzmin"], g2["xyzmin"]) and np.array_equal(g["xyzmax"], g2["xyzmax"])


def test_linspace_edges_are_numpy_linspace():
    rng = np.random.default_rng(0)
    for _ in range(200):
        lo = rng.uniform(-1e6, 5e6)
        hi = lo + rng.uniform(1e-3, 1e3)
        n = int(rng.integers(1, 300))
        assert np.array_equal(vo.linspace_edges(lo, hi, n), np.linspace(lo, hi, n + 1))


def test_counts_sum_and_groupby_shape(real_tile):
    xyz, labels = real_tile
    counts, towers, g = vo.voxel_counts(xyz, (64, 64, 64), None, labels, [15])
    assert counts.sum() == len(xyz)
    assert towers.sum() == (labels == 15).sum()
    assert (towers <= counts).all()
    # cube: all three extents equal
    ext = g["xyzmax"] - g["xyzmin"]
    assert np.allclose(ext, ext.max(), rtol=0, atol=1e-6)
    # bincount form == the reference's pandas groupby/iterrows form
    assert np.array_equal(vo.hist_on_voxel(xyz), vo.hist_on_voxel_groupby(xyz))


def test_interval_convention_left_open():
    # p in (e_k, e_{k+1}] -> k ; p == e_0 -> 0 (clip of -1)
    pts = np.array([[0.0, 0.0, 0.0], [4.0, 4.0, 4.0], [1.0, 2.0, 3.0], [1.0000001, 2.5, 3.999]])
    g = vo.voxelgrid_compute(pts, n_xyz=(4, 4, 4))
    assert list(g["voxel_x"]) == [0, 3, 0, 1]
    assert list(g["voxel_y"]) == [0, 3, 1, 2]
    assert list(g["voxel_z"]) == [0, 3, 2, 3]


def test_non_cubic_grid_and_layout():
    rng = np.random.default_rng(3)
    xyz = rng.uniform(0, 10, (500, 3))
    counts, _, g = vo.voxel_counts(xyz, (8, 4, 16))  # (x, y, z)
    assert counts.shape == (16, 8, 4)  # [z, x, y]
    assert counts.sum() == 500


def test_size_mode_dims():
    rng = np.random.default_rng(4)
    xyz = rng.uniform(0, 10, (1000, 3)) * np.array([1.0, 0.5, 2.0])
    g = vo.voxelgrid_compute(xyz, sizes=(0.5, 0.5, 0.5))
    assert all(int(n) >= 1 for n in g["x_y_z"])
    counts, _, _ = vo.voxel_counts(xyz, None, (0.5, 0.5, 0.5))
    assert counts.sum() == 1000


def test_reg_on_voxel_ratio(real_tile):
    xyz, labels = real_tile
    gt = vo.reg_on_voxel(xyz, labels, [15], (32, 32, 32))
    assert gt.min() >= 0 and gt.max() <= 1
    counts, towers, _ = vo.voxel_counts(xyz, (32, 32, 32), None, labels, [15])
    assert ((gt > 0) == (towers > 0)).all()


def test_voxelization_call_shapes(real_tile):
    xyz, labels = real_tile
    vox, gt = vo.voxelization_call((xyz, labels), [15], None, (16, 16, 16))
    assert vox.shape == gt.shape == (1, 16, 16, 16)
    assert vox.max() == 1.0 and vox.min() == 0.0


def test_synthetic_tile_is_deterministic_and_cubic():
    a, la = synthetic_tile(5, 10_000)
    b, lb = synthetic_tile(5, 10_000)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    assert a.shape == (10_000, 3)
    ext = a.max(0) - a.min(0)
    assert np.allclose(ext, [30, 30, 60])
    assert set(np.unique(la)) <= {1.0, 2.0, 4.0, 15.0, 16.0}


def test_empty_tile_raises():
    with pytest.raises(ValueError):
        vo.voxelgrid_compute(np.zeros((0, 3)), n_xyz=(4, 4, 4))


def test_vxg_to_xyz_matches_reference_vectors(golden_dir):
    """oracle restatement == rows dumped from the reference's vxg_to_xyz (make_golden.py dump_vxg_to_xyz)."""
    z = np.load(os.path.join(golden_dir, "vxg_to_xyz.npz"))
    for k in z["cases"]:
        o = z[f"{k}/origin"] if f"{k}/origin" in z else None
        vs = z[f"{k}/voxel_size"] if f"{k}/voxel_size" in z else None
        got = vo.vxg_to_xyz(z[f"{k}/grid"], o, vs)
        assert got.dtype == np.float64 and np.array_equal(got, z[f"{k}/rows"]), k
