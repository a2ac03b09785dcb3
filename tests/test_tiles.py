"""Tile loader (file format of core/datasets/ts40k.py) and per-point read-back."""
import os

import numpy as np
import pytest
import torch

import scene_net_amd as sna
from oracle import voxel_oracle as vo
from scene_net_amd.synthetic import synthetic_tile


@pytest.fixture()
def tile_dir(tmp_path, golden_dir):
    a = np.load(os.path.join(golden_dir, "ts40k_sample575_subset.npy"))
    d = tmp_path / "fit"
    d.mkdir()
    np.save(d / "sample_575.npy", a)
    xyz, lab = synthetic_tile(4, 5_000)
    np.save(d / "sample_001.npy", np.concatenate([xyz, lab[:, None]], 1))
    (d / "notes.txt").write_text("ignored")
    return str(tmp_path), a, (xyz, lab)


def test_listing_and_host_packing(tile_dir):
    root, real, (sx, sl) = tile_dir
    ds = sna.TS40KTiles(root, split="fit")
    assert len(ds) == 2 and list(ds.npy_files) == ["sample_001.npy", "sample_575.npy"]
    assert "2 samples" in str(ds)
    tiles, labels = ds.load_host([1, 0])
    assert np.array_equal(tiles[0], real[:, :3]) and np.array_equal(labels[0], real[:, 3])
    assert np.array_equal(tiles[1], sx) and np.array_equal(labels[1], sl)
    pts, lab, offsets, sizes = sna.pack_csr(tiles, labels)
    assert sizes == (len(real), 5_000) and offsets.tolist() == [0, len(real), len(real) + 5_000]
    assert pts.dtype == np.float64 and np.array_equal(pts[len(real):], sx) and np.array_equal(lab[:len(real)], real[:, 3])
    with pytest.raises(ValueError):
        sna.pack_csr([np.zeros((0, 3))])
    with pytest.raises(ValueError):
        sna.split_tile(np.zeros((5, 3)))


@pytest.mark.gpu
def test_load_batch_and_point_predictions(tile_dir, hip_device):
    root, real, (sx, sl) = tile_dir
    ds = sna.TS40KTiles(root, split="fit")
    batch = ds.load_batch([0, 1], device=hip_device)
    torch.cuda.synchronize()
    assert batch.sizes == (5_000, len(real)) and batch.pts.is_cuda and batch.labels is not None
    dims = (32, 16, 24)
    grids = sna.voxelize_batch(batch, dims, [15], want_density=True, want_gt=True)
    # read the grids back at the points: every point sees its own voxel
    pred = torch.cat([grids.density, grids.gt], dim=1).contiguous()  # [B,2,nz,nx,ny] f64
    per_point = sna.point_predictions(pred, batch, grids).cpu().numpy()
    assert per_point.shape == (2, batch.total_points)
    off = 0
    for xyz, lab in ((sx, sl), (real[:, :3], real[:, 3])):
        g = vo.voxelgrid_compute(xyz, n_xyz=dims)
        dens = vo.hist_on_voxel(xyz, dims)
        gt = vo.reg_on_voxel(xyz, lab, [15], dims)
        idx = (g["voxel_z"], g["voxel_x"], g["voxel_y"])
        assert np.array_equal(per_point[0, off:off + len(xyz)], dens[idx])
        assert np.array_equal(per_point[1, off:off + len(xyz)], gt[idx])
        assert (per_point[0, off:off + len(xyz)] > 0).all()  # a point's own voxel is never empty
        off += len(xyz)
    # thresholded, fp32, through the whole pipeline
    torch.manual_seed(0)
    model = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)).to(hip_device)
    pipe = sna.ScenePipeline(model, (64, 64, 64), keep_labels=[15])
    out, g2 = pipe(batch, want_gt=True)
    lab_pp = sna.point_predictions(out, batch, g2, tau=0.5)
    assert lab_pp.shape == (1, batch.total_points) and set(np.unique(lab_pp.cpu().numpy())) <= {0.0, 1.0}
    gt_pp = sna.point_predictions(g2.gt_occ.float(), batch, g2)
    assert gt_pp.sum().item() >= (batch.labels == 15).sum().item()  # every tower point sits in a tower voxel
