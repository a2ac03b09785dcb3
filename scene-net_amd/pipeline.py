"""The fused, HBM-resident hot path: ragged point batch -> occupancy grid -> GENEO bank conv -> head.

One process per GPU; voxel tiles are independent, so a job shards its tiles across ranks in
contiguous chunks with no data-path collective.  The only collectives are the
barrier / max-over-ranks used to time a job (bench.py).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _hip
from .scene_net import SceneNet
from .voxelization import PointBatch, VoxelGrids, voxelize_batch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk [lo, hi) of `n_items` tiles owned by `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def job_time_max(local_seconds: float, device=None) -> float:
    """max over ranks of a per-rank wall time (all-reduce MAX on the default process group; identity
    when torch.distributed is not initialised)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(local_seconds)
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_sum(value: float, device=None) -> float:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class ScenePipeline:
    """model(ToFullDense(Voxelization(points))) for a whole batch, without leaving HBM.

    Equivalent reference call chain: TS40K.__getitem__ -> Compose([Voxelization, ToTensor, ToFullDense])
    (core/datasets/ts40k.py:212-213, scripts/main.py:138-140) -> collate -> SceneNet.forward
    (core/models/SCENE_Net.py:322-339).
    """

    def __init__(self, model: SceneNet, voxelgrid_dims: Sequence[int] = (64, 64, 64),
                 keep_labels: Optional[Sequence[float]] = None):
        self.model = model
        self.voxelgrid_dims = tuple(int(v) for v in voxelgrid_dims)
        self.keep_labels = keep_labels

    def voxelize(self, batch: PointBatch, want_gt: bool = False) -> VoxelGrids:
        # binary occupancy as torch.bool: 1 byte/voxel between K1 and K3, and K3 runs on the int8 matrix cores
        return voxelize_batch(batch, self.voxelgrid_dims, self.keep_labels, want_occ=True, want_gt_occ=want_gt,
                              occ_dtype=torch.bool)

    def __call__(self, batch: PointBatch, want_gt: bool = False):
        grids = self.voxelize(batch, want_gt)
        out = self.model(grids.occ)
        return (out, grids) if want_gt else out

    def capture(self, batch: PointBatch, want_gt: bool = False) -> "CapturedPipeline":
        """The whole pass over `batch` (its device buffers, as they are refilled in place later) recorded into one
        hipGraph: 7 launches replayed with one call and no dispatch gaps.  Inference only (runs under no_grad)."""
        return CapturedPipeline(self, batch, want_gt)


class CapturedPipeline:
    """A hipGraph of ScenePipeline.__call__ over one PointBatch's buffers.  `replay()` re-runs it on whatever the
    buffers hold now (same number of tiles and points per tile offsets as at capture) and returns the static outputs.
    Every launch of the C ABI goes to torch's current stream and nothing on the path synchronises or allocates outside
    torch's allocator, so the capture needs no special casing."""

    def __init__(self, pipe: ScenePipeline, batch: PointBatch, want_gt: bool = False):
        self.batch = batch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            pipe(batch, want_gt)  # warm-up off the capture stream: lazy initialisation, allocator growth
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            res = pipe(batch, want_gt)
        self.out, self.grids = res if want_gt else (res, None)

    def replay(self):
        self.graph.replay()
        return (self.out, self.grids) if self.grids is not None else self.out
