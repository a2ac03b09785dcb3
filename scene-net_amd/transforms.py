"""Host-side mirror of core/datasets/torch_transforms.py: `ToTensor` (:9-13), `ToFullDense` (:17-40),
`Voxelization` (:44-81) -- same names, constructor arguments and call signatures, so
`Compose([Voxelization(...), ToTensor(), ToFullDense(apply=(True, True))])` (scripts/main.py:138-140)
works unchanged.  `Voxelization` runs the HIP voxeliser (one bbox + one scatter pass produce both the
density and the ground-truth grid; the reference voxelises the same points twice).
"""
from __future__ import annotations

import warnings
from typing import Tuple

import numpy as np
import torch

from . import _hip
from . import voxelization as Vox

_warned_worker = False


def _check_worker_context():
    """The reference runs this transform inside DataLoader workers (`num_workers: 8`, fork start method:
    core/lit_modules/lit_data_wrappers.py:62-72).  A HIP context does not survive a fork: in a worker forked after
    the parent touched the GPU every HIP call fails, so say what to do instead of torch's generic re-initialisation
    error.  In a worker that CAN use the device (spawned, or forked before any GPU call) the transform works but pays
    a HIP context per worker and a per-tile H2D + sync + D2H: warn once and point at the batch-side form."""
    global _warned_worker
    if torch.cuda._is_in_bad_fork():
        raise _hip.HipLibraryError(
            "Voxelization was called in a process forked after the parent initialised the GPU (a DataLoader worker "
            "with the default 'fork' start method): a HIP context cannot be used there.  Use num_workers=0, or "
            "DataLoader(..., multiprocessing_context='spawn'), or -- better -- let the workers only load the .npy "
            "tiles and voxelise the whole batch on the device with scene_net_amd.voxelize_batch / ScenePipeline "
            "(INTEGRATION.md section 1)")
    if not _warned_worker and torch.utils.data.get_worker_info() is not None:
        _warned_worker = True
        warnings.warn("scene_net_amd.Voxelization is running inside a DataLoader worker: every worker holds its own "
                      "HIP context and pays H2D + sync + D2H per tile (about 2 ms/tile).  Voxelise batch-side "
                      "instead (scene_net_amd.voxelize_batch / ScenePipeline), workers then only read files.",
                      RuntimeWarning, stacklevel=3)


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
        _check_worker_context()
        g = Vox._voxelize_single(pts, labels, self.keep_labels, self.vxg_size, self.vox_size,
                                 want_density=True, want_gt=True, want_occ=False)
        # vox-point-density, vox-tower-prob : [1, nz, nx, ny] float64 numpy, like the reference
        return g.density[0].cpu().numpy(), g.gt[0].cpu().numpy()
