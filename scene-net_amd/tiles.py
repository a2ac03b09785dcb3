"""Tile loader and per-point read-back -- the data formats either side of the hot path.

* `TS40KTiles` mirrors the file side of core/datasets/ts40k.py:154-225: a directory `<root>/<split>/*.npy`, each file
  one tile `(N, 4)` float64 `x, y, z, label` (ts40k.py:201-207).  Instead of one transformed sample per
  `__getitem__` it hands whole ragged batches to HBM (`PointBatch`), staging through pinned host memory so the
  H2D copy is one asynchronous transfer per array.
* `point_predictions` reads a voxel-grid prediction back at every input point (BASELINE config 4's per-point
  label gather), binning the points exactly as the voxeliser did.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _hip
from .voxelization import PointBatch, VoxelGrids


def split_tile(npy: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(N,4) -> (xyz [N,3], label [N])  (ts40k.py:207: `sample = (npy[:, 0:-1], npy[:, -1])`)."""
    if npy.ndim != 2 or npy.shape[1] < 4:
        raise ValueError(f"a TS40K tile is (N, 4) x,y,z,label (got {npy.shape})")
    return npy[:, 0:-1][:, :3], npy[:, -1]


class TS40KTiles:
    """Lists `<dataset_path>/<split>/*.npy` like TS40K.__init__ (ts40k.py:156-181) and loads batches of tiles."""

    def __init__(self, dataset_path: str, split: str = "fit") -> None:
        self.split = split
        self.dataset_path = os.path.join(dataset_path, split)
        self.npy_files = np.array(sorted(f for f in os.listdir(self.dataset_path)
                                         if os.path.isfile(os.path.join(self.dataset_path, f)) and ".npy" in f))

    def __len__(self) -> int:
        return len(self.npy_files)

    def __str__(self) -> str:
        return f"TS40K {self.split} Dataset with {len(self)} samples"

    def load_host(self, indices: Sequence[int]) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        tiles, labels = [], []
        for i in indices:
            xyz, lab = split_tile(np.load(os.path.join(self.dataset_path, self.npy_files[int(i)])))
            tiles.append(np.ascontiguousarray(xyz, dtype=np.float64))
            labels.append(np.ascontiguousarray(lab, dtype=np.float64))
        return tiles, labels

    def load_batch(self, indices: Sequence[int], device=None, stream: Optional[torch.cuda.Stream] = None) -> PointBatch:
        """Tiles `indices` as one CSR batch in HBM; the copies are asynchronous on `stream` (pinned staging)."""
        tiles, labels = self.load_host(indices)
        return batch_to_device(tiles, labels, device, stream)


def pack_csr(tiles: Sequence[np.ndarray], labels: Optional[Sequence[np.ndarray]] = None):
    """Host-side packing: (pts [total,3] f64, labels [total] f64 | None, offsets [B+1] i64, sizes)."""
    sizes = tuple(int(t.shape[0]) for t in tiles)
    if any(n == 0 for n in sizes):
        raise ValueError("zero-size array to reduction operation minimum which has no identity (empty tile)")
    total = sum(sizes)
    pts = np.empty((total, 3), dtype=np.float64)
    lab = np.empty((total,), dtype=np.float64) if labels is not None else None
    offsets = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    for b, t in enumerate(tiles):
        if t.ndim != 2 or t.shape[1] != 3:
            raise ValueError(f"each tile must be [N,3] (got {tuple(t.shape)})")
        pts[offsets[b]:offsets[b + 1]] = t
        if lab is not None:
            if labels[b].shape[0] != t.shape[0]:
                raise ValueError("labels and points disagree in length")
            lab[offsets[b]:offsets[b + 1]] = labels[b]
    return pts, lab, offsets, sizes


def batch_to_device(tiles, labels=None, device=None, stream: Optional[torch.cuda.Stream] = None) -> PointBatch:
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.type != "cuda":
        raise _hip.HipLibraryError("PointBatch lives in HBM: device must be a HIP device")
    pts, lab, offsets, sizes = pack_csr(tiles, labels)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(device))
    with ctx:
        d_pts = torch.from_numpy(pts).pin_memory().to(device, non_blocking=True)
        d_lab = torch.from_numpy(lab).pin_memory().to(device, non_blocking=True) if lab is not None else None
        d_off = torch.from_numpy(offsets).pin_memory().to(device, non_blocking=True)
    return PointBatch(d_pts, d_lab, d_off, sizes)


def point_predictions(pred: torch.Tensor, batch: PointBatch, grids: VoxelGrids, fill: float = 0.0,
                      tau: Optional[float] = None) -> torch.Tensor:
    """pred [B,C,nz,nx,ny] (f32|f64) -> [C, total_points]: every point reads the voxel it was binned into
    (same descriptor as the scatter).  With `tau`, values are thresholded like prob_to_label
    (utils/voxelization.py:304-323)."""
    out = _hip.gather_points(pred.contiguous(), batch.pts, batch.offsets, grids.desc, fill)
    if tau is not None:
        out = (out >= tau).to(out.dtype)
    return out
