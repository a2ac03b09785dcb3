"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (numpy, fp64) of the reference's point-cloud -> voxel-grid step:
utils/voxelization.py:164-204 (hist_on_voxel), :244-300 (reg_on_voxel),
utils/pcd_processing.py:305-321 (normalize_xyz), :341-372 (voxelize_ply) and
core/datasets/torch_transforms.py:17-40, :44-81 (ToFullDense, Voxelization).

Parity status
-------------
* `normalize_xyz`, `to_full_dense`: PINNED against the reference's own functions
  run in the build container (tests/golden/voxel_normalize.npz).
* `voxelgrid_compute` (the bounding box / edge / binning arithmetic): **PARITY
  UNPINNED**.  That arithmetic lives in the third-party dependency
  pyntcloud == 0.1.6 (reference requirements.txt:9), class
  `pyntcloud.structures.VoxelGrid.compute`, which is neither vendored in the
  reference nor installed in this image, and the reference holds no test or
  fixture at that boundary.  The function below restates pyntcloud 0.1.6's
  published algorithm (bounding box -> cube (`regular_bounding_box=True`
  default) -> optional `size_*` extension -> `np.linspace` edges ->
  `np.searchsorted(edges, p) - 1` clipped to [0, n]) and is anchored on the
  reference's call sites (pcd_processing.py:360-368; consumers
  voxelization.py:189-197, :274-293).  It is isolated here so it can be corrected
  in one place if a real pyntcloud 0.1.6 becomes available.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np


# --------------------------------------------------------------------------- #
# pyntcloud 0.1.6 VoxelGrid.compute (restated; see header)
# --------------------------------------------------------------------------- #
def voxelgrid_compute(points: np.ndarray,
                      n_xyz: Optional[Sequence[int]] = None,
                      sizes: Optional[Sequence[Optional[float]]] = None,
                      regular_bounding_box: bool = True):
    """Returns dict(voxel_x, voxel_y, voxel_z [N] int64, x_y_z [3] int, xyzmin, xyzmax [3] f64, segments).

    Called by the reference as add_structure("voxelgrid", n_x, n_y, n_z) or
    add_structure("voxelgrid", size_x, size_y, size_z) (pcd_processing.py:360-368).
    """
    pts = np.asarray(points, dtype=np.float64)
    x_y_z = np.asarray([1, 1, 1] if n_xyz is None else list(n_xyz), dtype=np.int64)
    sizes = [None, None, None] if sizes is None else list(sizes)

    xyzmin = pts.min(0)
    xyzmax = pts.max(0)
    xyz_range = xyzmax - xyzmin  # ptp(0)

    if regular_bounding_box:
        # minimum bounding CUBE: pad the short axes symmetrically
        margin = max(xyz_range) - xyz_range
        xyzmin = xyzmin - margin / 2
        xyzmax = xyzmax + margin / 2

    for n, size in enumerate(sizes):
        if size is None:
            continue
        margin = (((xyz_range[n] // size) + 1) * size) - xyz_range[n]
        xyzmin[n] -= margin / 2
        xyzmax[n] += margin / 2
        x_y_z[n] = int((xyzmax[n] - xyzmin[n]) / size)

    segments = [np.linspace(xyzmin[i], xyzmax[i], num=int(x_y_z[i]) + 1) for i in range(3)]

    # searchsorted(side='left') - 1: p in (e_k, e_{k+1}] -> k ; p == e_0 -> -1 -> clip 0
    vox = [np.clip(np.searchsorted(segments[i], pts[:, i]) - 1, 0, x_y_z[i]) for i in range(3)]
    return dict(voxel_x=vox[0], voxel_y=vox[1], voxel_z=vox[2], x_y_z=x_y_z,
                xyzmin=xyzmin, xyzmax=xyzmax, segments=segments)


def linspace_edges(lo: float, hi: float, n: int) -> np.ndarray:
    """numpy.linspace(lo, hi, n+1) written out: e_k = k*step + lo (two roundings), e_n = hi.
    The HIP kernel builds its edge table with exactly these fp64 operations (no FMA)."""
    step = (hi - lo) / n
    e = np.arange(0, n + 1, dtype=np.float64) * step + lo
    e[-1] = hi
    return e


def _voxelize(xyz, voxelgrid_dims, voxel_dims):
    """eda.voxelize_ply (pcd_processing.py:341-372): voxel_dims overrides voxelgrid_dims;
    tuples are (x, y, z)."""
    if voxel_dims is None:
        x, y, z = voxelgrid_dims
        return voxelgrid_compute(xyz, n_xyz=(x, y, z))
    x, y, z = voxel_dims
    return voxelgrid_compute(xyz, sizes=(x, y, z))


def voxel_counts(xyz, voxelgrid_dims=(64, 64, 64), voxel_dims=None, labels=None, keep_labels=None):
    """Integer occupancy: counts[z, x, y] = #points in voxel (voxelization.py:193-200); optionally the
    per-voxel number of points whose label is in keep_labels (count_towers, voxelization.py:285-289).
    Points the clip leaves at index n (out of range; cannot occur for in-box points) are dropped, as
    the reference's `data[i] = hist` would raise IndexError there."""
    g = _voxelize(xyz, voxelgrid_dims, voxel_dims)
    nx, ny, nz = (int(v) for v in g["x_y_z"])
    vx, vy, vz = g["voxel_x"], g["voxel_y"], g["voxel_z"]
    if (vx >= nx).any() or (vy >= ny).any() or (vz >= nz).any():
        raise IndexError("voxel index out of range (reference would raise on data[i] = hist)")
    flat = (vz * nx + vx) * ny + vy
    counts = np.bincount(flat, minlength=nz * nx * ny).reshape(nz, nx, ny).astype(np.int64)
    towers = None
    if labels is not None:
        keep = np.isin(np.asarray(labels), np.array(keep_labels).reshape(-1))
        towers = np.bincount(flat[keep], minlength=nz * nx * ny).reshape(nz, nx, ny).astype(np.int64)
    return counts, towers, g


def normalize_xyz(data: np.ndarray) -> np.ndarray:
    """eda.normalize_xyz (pcd_processing.py:305-321): sklearn MinMaxScaler fitted on
    data.reshape(-1, n_last), i.e. per LAST-axis column min/max over all other axes.
    sklearn: scale = 1/(max-min) with (max-min)==0 -> 1 (to within 10 eps); X*scale + (0 - min*scale)."""
    shape = data.shape
    X = data.reshape(-1, shape[-1]).astype(np.float64)
    dmin, dmax = X.min(0), X.max(0)
    rng = dmax - dmin
    rng = np.where(rng < 10 * np.finfo(np.float64).eps, 1.0, rng)  # sklearn _handle_zeros_in_scale
    scale = 1.0 / rng
    min_ = 0.0 - dmin * scale
    out = X * scale
    out = out + min_
    return out.reshape(shape)


def hist_on_voxel(xyz, voxelgrid_dims=(64, 64, 64), voxel_dims=None) -> np.ndarray:
    """voxelization.py:164-204 -> [nz, nx, ny] float64 in [0,1]."""
    counts, _, _ = voxel_counts(xyz, voxelgrid_dims, voxel_dims)
    return normalize_xyz(counts.astype(np.float64))


def reg_on_voxel(xyz, labels, tower_label, voxelgrid_dims=(64, 64, 64), voxel_dims=None) -> np.ndarray:
    """voxelization.py:244-300 -> [nz, nx, ny] float64: tower points / points per occupied voxel, else 0."""
    counts, towers, _ = voxel_counts(xyz, voxelgrid_dims, voxel_dims, labels, tower_label)
    out = np.zeros(counts.shape, dtype=np.float64)
    occ = counts > 0
    out[occ] = towers[occ] / counts[occ]
    return out


def voxelization_call(sample, keep_labels, vox_size=None, vxg_size=None):
    """Voxelization.__call__, torch_transforms.py:74-81."""
    pts, labels = sample
    vox = hist_on_voxel(pts, voxel_dims=vox_size, voxelgrid_dims=vxg_size)
    gt = reg_on_voxel(pts, labels, keep_labels, voxel_dims=vox_size, voxelgrid_dims=vxg_size)
    return vox[None], gt[None]


def to_full_dense(t: np.ndarray) -> np.ndarray:
    """ToFullDense.densify, torch_transforms.py:33-34: (t > 0).to(t)."""
    return (t > 0).astype(t.dtype)


# --------------------------------------------------------------------------- #
# pandas-shaped restatement of the reference loop (slow; used by tests to check the
# bincount form above against the groupby/iterrows form the reference is written in)
# --------------------------------------------------------------------------- #
def hist_on_voxel_groupby(xyz, voxelgrid_dims=(64, 64, 64), voxel_dims=None) -> np.ndarray:
    import pandas as pd
    g = _voxelize(xyz, voxelgrid_dims, voxel_dims)
    nx, ny, nz = (int(v) for v in g["x_y_z"])
    data = np.zeros((nz, nx, ny))
    voxs = pd.DataFrame({"z": g["voxel_z"], "x": g["voxel_x"], "y": g["voxel_y"],
                         "points": np.ones_like(g["voxel_x"])})
    groups = voxs.groupby(["z", "x", "y"]).count()
    for i, hist in groups.iterrows():
        data[i] = hist.iloc[0]
    return normalize_xyz(data)


def vxg_to_xyz(vxg: np.ndarray, origin=None, voxel_size=None) -> np.ndarray:
    """utils/voxelization.py:328-360: rows (origin + index * voxel_size, vxg[index]) for EVERY cell, in the C order
    of np.indices(shape).reshape(3, -1).T; the reference's per-cell Python loop is vxg.reshape(-1).  (V, 4) f64.
    Pinned against the reference's own output: tests/golden/vxg_to_xyz.npz."""
    vxg = np.asarray(vxg)
    origin = np.array([0, 0, 0]) if origin is None else np.asarray(origin)
    voxel_size = np.array([1, 1, 1]) if voxel_size is None else np.asarray(voxel_size)
    idx = np.indices(vxg.shape).reshape(3, -1).T
    points = origin + idx * voxel_size
    return np.concatenate((points, vxg.reshape(-1, 1)), axis=1).astype(np.float64)
