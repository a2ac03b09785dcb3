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
                 keep_labels: Optional[Sequence[float]] = None, overlap_bank: bool = True,
                 voxel_dims: Optional[Sequence[float]] = None, per_point: bool = False, tau: Optional[float] = None):
        """voxel_dims (size_x, size_y, size_z): voxel-size mode (utils/pcd_processing.py:365-367, what SemanticKITTI uses,
        core/datasets/semKITTI.py:453-455) -- `voxelgrid_dims` is then the capacity the per-scan grids are padded to.
        per_point: the call also returns every point's prediction [1, total_points] -- the voxel it was binned into,
        read back with the scatter's own binning (the per-point form of utils/voxelization.py:304-360: prob_to_label with
        `tau`, vxg_to_xyz's voxel -> coordinate walk replaced by point -> voxel)."""
        self.model = model
        self.voxelgrid_dims = tuple(int(v) for v in voxelgrid_dims)
        self.keep_labels = keep_labels
        self.voxel_dims = None if voxel_dims is None else tuple(float(v) for v in voxel_dims)
        self.per_point = bool(per_point)
        self.tau = tau
        # K2 (bank builder + the contraction's preparation) reads only the model's scalars.  overlap_bank: it leaves the
        # critical path -- as riders of K1's first launch for a 9^3 bank (rides()), else on a forked stream beside K1,
        # joined in front of K3 (bank_beside: a parallel branch when the pass is captured into a hipGraph)
        self.overlap_bank = bool(overlap_bank)
        self._side = None

    def voxelize(self, batch: PointBatch, want_gt: bool = False, bank_rider=None) -> VoxelGrids:
        # binary occupancy as torch.bool: 1 byte/voxel between K1 and K3, and K3 runs on the int8 matrix cores
        return voxelize_batch(batch, self.voxelgrid_dims, self.keep_labels, want_occ=True, want_gt_occ=want_gt,
                              occ_dtype=torch.bool, voxel_dims=self.voxel_dims, bank_rider=bank_rider)

    def rides(self, planes: int = 1) -> bool:
        """K2 can ride in K1's first launch: a 9^3 bank and the occupancy fast path (n-mode or voxel-size mode)"""
        from . import _hip as h
        return (self.overlap_bank and self.model.kernel_size_of_bank() == (9, 9, 9)
                and h.occupancy_supported(tuple(int(v) for v in self.voxelgrid_dims), planes))

    def _finish(self, out, grids, batch, want_gt):
        if self.per_point:
            from .tiles import point_predictions
            pts_pred = point_predictions(out, batch, grids, tau=self.tau)
            return (out, grids, pts_pred) if want_gt else (out, pts_pred)
        return (out, grids) if want_gt else out

    def bank_beside(self, device):
        """Starts K2 on the side stream: returns (bank, lam, prep | None, join) -- call join() on the main stream before
        the contraction.  The fork waits for everything enqueued so far (the previous pass's contraction still reads the
        model's bank / prep buffers), so K2 of pass i+1 overlaps K1 of pass i+1, not K3 of pass i."""
        model = self.model
        main = torch.cuda.current_stream(device)
        if self._side is None or self._side.device != main.device:
            self._side = torch.cuda.Stream(device=device)
        side = self._side
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if model.kernel_size_of_bank() == (9, 9, 9):
                bank, prep = model.compute_bank_prepared(device)
            else:
                bank, prep = model.compute_bank(device), None
                bank.record_stream(main)
            lam = model.effective_lambdas(device)
        return bank, lam, prep, (lambda: main.wait_stream(side))

    def __call__(self, batch: PointBatch, want_gt: bool = False):
        model = self.model
        plain = not (torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()))
        if not (self.overlap_bank and plain):
            grids = self.voxelize(batch, want_gt)
            out = model(grids.occ)
            return self._finish(out, grids, batch, want_gt)
        # inference: the module's own no-grad forward (scene_net.py), with K2 off the critical path
        with torch.no_grad():
            dev = batch.pts.device
            if self.rides(2 if want_gt else 1):
                # K2 as riders of K1's first launch ([measured] forked onto a side stream it still cost the main stream
                # 10.8 of its 12.4 serial microseconds: an event record and a wait)
                _, _, bank, prep = rider = model.bank_rider(dev)
                lam = model.effective_lambdas(dev)
                grids = self.voxelize(batch, want_gt, bank_rider=rider)
                if not grids.rider_done:   # (the voxelisation took a path without the rider: K2 as its own launch)
                    bank, prep = model.compute_bank_prepared(dev)
                x = grids.occ
                if model.fused_forward and _hip.conv_fused_supported(x, model.kernel_size_of_bank()):
                    out = model.fused_served(x, bank, lam, model.activation_dtype or torch.float32)
                else:
                    out = model.contract_prepared(x, bank, lam, prep)[1]
                return self._finish(out, grids, batch, want_gt)
            bank, lam, prep, join = self.bank_beside(dev)
            grids = self.voxelize(batch, want_gt)
            join()
            x = grids.occ
            if model.fused_forward and _hip.conv_fused_supported(x, model.kernel_size_of_bank()):
                out = model.fused_served(x, bank, lam, model.activation_dtype or torch.float32)
            else:
                out = (model.contract_prepared(x, bank, lam, prep)[1] if prep is not None and x.dtype == torch.bool
                       else _hip.conv_bank(x, bank, lam, want_act=False, want_out=True)[1])
            return self._finish(out, grids, batch, want_gt)

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
            self.result = pipe(batch, want_gt)
        res = self.result if isinstance(self.result, tuple) else (self.result,)
        self.out = res[0]
        self.grids = res[1] if want_gt else None

    def replay(self):
        self.graph.replay()
        return self.result
