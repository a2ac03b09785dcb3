"""A whole training step -- voxelise (+GT) -> SceneNet.forward -> criterion -> backward -> optimiser step -- recorded once
into a hipGraph and replayed.

Eagerly the step is host bound (~50 scalar parameters, dozens of small launches: ~1 ms at BASELINE C2 against 0.34 ms
of GPU work); every launch of the C ABI goes to torch's current stream and nothing on the path synchronises or
allocates outside torch's allocator, so the step captures as it is.  Tile sizes are fixed at capture time, the point /
label buffers of the batch are refilled in place between replays.

Under a live process group (one rank per GPU, the training config of SURVEY 8e / core.lit_modules' `pl.Trainer(gpus=-1)`)
the step is TWO graphs with the one real exchange of the path between them:
    graph 1: zero_grad, voxelise (+GT), forward, criterion, backward      (gradients land at fixed addresses)
    eager:   ONE all-reduce over the flat vector of the ~50 scalar gradients (RCCL over xGMI on a node)
    graph 2: optimizer.step()
Capture uses the relaxed (thread-local) error mode, so the group's watchdog thread may keep calling into HIP.

STATUS: REHEARSED ONLY.  The N > 1 form has run on two gloo ranks sharing one GPU (tests/test_gpu_ddp.py,
tests/test_gpu_rccl.py::test_worker_rehearsed_over_gloo_on_one_gpu); it has not yet executed over RCCL on two devices --
tests/test_gpu_rccl.py does that by itself on any box with >= 2 HIP devices.  Until such a run is on record, treat the
two-graph step under an RCCL watchdog thread as untested.
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


def voxelize_and_forward(pipe: ScenePipeline, batch: PointBatch):
    """(grids, pred) of a training step: voxelise with the ground-truth plane, then the module's forward under autograd.
    When the bank is 9 x 9 x 9 the forward's opener (bank + effective coefficients, sn_geneo_bank_lambdas) rides in the
    voxelisation's first launch (SceneNet.train_rider) instead of being a launch of its own."""
    model = pipe.model
    rides = (pipe.rides(2) and torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
             and batch.pts.is_cuda)
    if not rides:
        grids = pipe.voxelize(batch, want_gt=True)
        return grids, model(grids.occ)
    rider, bank_lam = model.train_rider(batch.pts.device)
    grids = pipe.voxelize(batch, want_gt=True, bank_rider=rider)
    return grids, model(grids.occ, bank_lam=bank_lam if grids.rider_done else None)


_SEEDS = {}


def backward_seeded(loss: torch.Tensor) -> None:
    """loss.backward() with the seed gradient taken from a cached one-element tensor: autograd otherwise builds
    ones_like(loss) with a fill launch at every step (~4 us of GPU time in a replayed graph)."""
    key = (loss.device, loss.dtype)
    one = _SEEDS.get(key)
    if one is None:
        one = _SEEDS[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    loss.backward(gradient=one)


class CapturedTrainingStep:
    """step = zero_grad; grids = pipe.voxelize(batch, want_gt=True); loss = criterion(model(grids.occ), grids.gt_occ,
    cvx coefficients, GENEO parameters); loss.backward(); [all-reduce of the gradients over `group`]; optimizer.step().
    `replay()` runs it on whatever the batch's device buffers hold and returns the (static) loss tensor of THIS rank's
    tiles.  With a process group of more than one rank every rank must build and replay the step together."""

    def __init__(self, pipe: ScenePipeline, criterion: Callable, optimizer: torch.optim.Optimizer, batch: PointBatch,
                 warmup: int = 3, loss_fn: Optional[Callable] = None, group=None):
        import torch.distributed as dist
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        self.group = group
        self.pipe, self.criterion, self.optimizer, self.batch = pipe, criterion, optimizer, batch
        model = pipe.model
        params = [p for p in model.parameters() if p.requires_grad]

        def front():
            optimizer.zero_grad(set_to_none=True)
            grids, pred = voxelize_and_forward(pipe, batch)
            if loss_fn is not None:
                loss = loss_fn(pred, grids)
            else:
                loss = criterion(pred, grids.gt_occ, model.get_cvx_coefficients(), model.get_geneo_params())
            backward_seeded(loss)
            return loss

        def exchange():
            if self.world > 1:
                allreduce_flat_grads(params, group)

        def step():
            loss = front()
            exchange()
            optimizer.step()
            return loss

        self._eager_step = step
        self._exchange = exchange
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # lazy initialisation, allocator growth, optimiser state -- off the capture
            for _ in range(max(1, warmup)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        optimizer.zero_grad(set_to_none=True)
        # a live process group's watchdog thread calls into HIP at any time: only this thread's calls are checked
        mode = "thread_local" if live else "global"
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        if self.world == 1:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.loss = step()
        else:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.loss = front()
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool(), capture_error_mode=mode):
                optimizer.step()

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        if self.graph_opt is not None:
            self._exchange()
            self.graph_opt.replay()
        # the replayed optimiser step writes the parameters through the addresses baked into the graph: no tensor's `_version`
        # moves, and the model's parameter-derived caches (packed parameters, coefficients, K3L tables, learnt verdicts) key on
        # those versions -- a validation forward between replays would run on the previous weights
        self.pipe.model.invalidate_caches()
        return self.loss
