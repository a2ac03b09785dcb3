"""Host-side mirror of core/datasets/torch_transforms.py: `ToTensor` (:9-13), `ToFullDense` (:17-40),
`Voxelization` (:44-81) -- same names, constructor arguments and call signatures, so
`Compose([Voxelization(...), ToTensor(), ToFullDense(apply=(True, True))])` (scripts/main.py:138-140)
works unchanged.  `Voxelization` runs the HIP voxeliser (one bbox + one scatter pass produce both the
density and the ground-truth grid; the reference voxelises the same points twice).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from . import voxelization as Vox


class ToTensor:
    def __call__(self, sample):
        sample = list(sample)
        # the reference's `np.float` is float64 (and no longer exists in numpy >= 1.24)
        return tuple([s if isinstance(s, torch.Tensor) else torch.from_numpy(s.astype(np.float64)) for s in sample])


class ToFullDense:
    """Any voxel with tower points gets belief 1; the input density becomes binary occupancy."""

    def __init__(self, apply=[True, True]) -> None:
        self.apply = apply

    def densify(self, tensor: torch.Tensor):
        return (tensor > 0).to(tensor)

    def __call__(self, sample):
        vox, gt = [self.densify(tensor) if self.apply[i] else tensor for i, tensor in enumerate(sample)]
        return vox, gt


class Voxelization:
    def __init__(self, keep_labels, vox_size: Tuple[int] = None, vxg_size: Tuple[int] = None) -> None:
        if vox_size is None and vxg_size is None:
            ValueError("Voxel size or Voxelgrid size must be provided")  # constructed, never raised (reference :65-66)
        self.vox_size = vox_size
        self.vxg_size = vxg_size
        self.keep_labels = keep_labels

    def __call__(self, sample):
        pts, labels = sample
        g = Vox._voxelize_single(pts, labels, self.keep_labels, self.vxg_size, self.vox_size,
                                 want_density=True, want_gt=True, want_occ=False)
        # vox-point-density, vox-tower-prob : [1, nz, nx, ny] float64 numpy, like the reference
        return g.density[0].cpu().numpy(), g.gt[0].cpu().numpy()
