"""The N>1 path on CPU: two gloo ranks shard a job's tiles with no data-path collective and agree on the
job time (max over ranks) and the job total (sum), exactly as bench.py does over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scene_net_amd as sna
    from scene_net_amd.pipeline import job_sum, job_time_max
    n_tiles = 37
    lo, hi = sna.shard_range(n_tiles, rank, world)
    owned = list(range(lo, hi))
    dist.barrier()
    t = job_time_max(0.25 * (rank + 1))
    total = job_sum(float(len(owned)))
    gathered = [None] * world
    dist.all_gather_object(gathered, owned)
    q.put((rank, t, total, gathered))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t, total, gathered in res:
        assert t == 0.5  # max over ranks
        assert total == 37.0  # every tile counted once
        flat = [i for chunk in gathered for i in chunk]
        assert flat == list(range(37))  # disjoint, contiguous, complete


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scene_net_amd as sna
    torch.manual_seed(3)   # same parameters on every rank, rank-dependent gradients
    params = [torch.nn.Parameter(torch.randn(())) for _ in range(7)] + [torch.nn.Parameter(torch.randn(5))]
    frozen = torch.nn.Parameter(torch.randn(()), requires_grad=False)   # no gradient: left alone
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    views = [p.grad for p in params]   # the exchange writes in place
    n = sna.allreduce_flat_grads(params + [frozen])
    q.put((rank, n, [g.clone() for g in views], frozen.grad is None))
    dist.destroy_process_group()


def test_two_rank_flat_gradient_exchange():
    """SURVEY 8e's training collective: one all-reduce over the flat scalar-gradient vector, mean over ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n, grads, frozen_untouched in res:
        assert n == 7 + 5 and frozen_untouched
        for i, g in enumerate(grads):
            assert torch.equal(g, torch.full_like(g, 1.5 * (i + 1)))   # mean of (i+1) and 2 (i+1)


def test_flat_gradient_exchange_without_group_is_a_no_op():
    import scene_net_amd as sna
    p = torch.nn.Parameter(torch.ones(()))
    p.grad = torch.full((), 2.0)
    assert sna.allreduce_flat_grads([p]) == 0 and float(p.grad) == 2.0
