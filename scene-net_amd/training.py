"""A whole training step -- voxelise (+GT) -> SceneNet.forward -> criterion -> backward -> optimiser step -- recorded once
into a hipGraph and replayed.

Eagerly the step is host bound (~50 scalar parameters, dozens of small launches: ~1 ms at BASELINE C2 against 0.34 ms
of GPU work); every launch of the C ABI goes to torch's current stream and nothing on the path synchronises or
allocates outside torch's allocator, so the step captures as it is.  Single process only (a process group's watchdog
thread may touch the device during capture); tile sizes are fixed at capture time, the point / label buffers of the
batch are refilled in place between replays.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch

from .pipeline import ScenePipeline
from .voxelization import PointBatch


def allreduce_flat_grads(params: Iterable[torch.Tensor], group=None, average: bool = True) -> int:
    """The training exchange of SURVEY 8e: ONE all-reduce per step over the flat vector of the scalar gradients
    (~50 floats: latency bound over xGMI, so one collective, not one per parameter), written back in place.
    Every rank must hold gradients for the same parameters.  Returns the number of floats exchanged (0 when there is
    no process group or it has a single rank).  Equivalent to what DistributedDataParallel's one bucket does for this
    model; use either."""
    import torch.distributed as dist
    ps = [p for p in params if p.grad is not None]
    if not ps or not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    grads = [p.grad.reshape(-1) for p in ps]
    flat = torch.cat(grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= world
    torch._foreach_copy_(grads, list(flat.split([g.numel() for g in grads])))
    return flat.numel()


class CapturedTrainingStep:
    """step = zero_grad; grids = pipe.voxelize(batch, want_gt=True); loss = criterion(model(grids.occ), grids.gt_occ,
    cvx coefficients, GENEO parameters); loss.backward(); optimizer.step().  `replay()` runs it on whatever the
    batch's device buffers hold and returns the (static) loss tensor."""

    def __init__(self, pipe: ScenePipeline, criterion: Callable, optimizer: torch.optim.Optimizer, batch: PointBatch,
                 warmup: int = 3, loss_fn: Optional[Callable] = None):
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size() > 1:
            raise RuntimeError("CapturedTrainingStep is single-process (capture and a live process group do not mix)")
        self.pipe, self.criterion, self.optimizer, self.batch = pipe, criterion, optimizer, batch
        model = pipe.model

        def step():
            optimizer.zero_grad(set_to_none=True)
            grids = pipe.voxelize(batch, want_gt=True)
            pred = model(grids.occ)
            if loss_fn is not None:
                loss = loss_fn(pred, grids)
            else:
                loss = criterion(pred, grids.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
            loss.backward()
            optimizer.step()
            return loss

        self._eager_step = step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # lazy initialisation, allocator growth, optimiser state -- off the capture
            for _ in range(max(1, warmup)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step()

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss
