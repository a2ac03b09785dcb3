"""Training config on the N>1 path: the module under DistributedDataParallel (tiles sharded per rank, gradient
all-reduce of the ~50 scalars).  Two ranks share the one GPU of the test box, so the process group is gloo here;
on a node it is nccl (= RCCL) with one rank per GPU -- the model code is the same."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scene_net_amd as sna
    from torch.nn.parallel import DistributedDataParallel as DDP
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(7)  # same initial model on every rank
    model = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)).to(dev)
    ddp = DDP(model, device_ids=[0])
    g = torch.Generator().manual_seed(100)  # the global batch, identical on every rank; each takes its shard
    x = (torch.rand(4, 1, 12, 12, 24, generator=g) < 0.2)
    y = (torch.rand(4, 1, 12, 12, 24, generator=g) < 0.1).float()
    lo, hi = sna.shard_range(4, rank, world)
    loss = ((ddp(x[lo:hi].to(dev)) - y[lo:hi].to(dev)) ** 2).mean()
    loss.backward()
    grads = {n: float(p.grad) for n, p in model.named_parameters() if p.grad is not None}
    # single-process reference of the same global step: mean over the two equal shards == mean of shard means
    torch.manual_seed(7)
    ref = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)).to(dev)
    ((ref(x.to(dev)) - y.to(dev)) ** 2).mean().backward()
    ref_grads = {n: float(p.grad) for n, p in ref.named_parameters() if p.grad is not None}
    # the same exchange without DDP: one all-reduce over the flat vector of scalar gradients
    torch.manual_seed(7)
    own = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)).to(dev)
    ((own(x[lo:hi].to(dev)) - y[lo:hi].to(dev)) ** 2).mean().backward()
    n_floats = sna.allreduce_flat_grads(own.parameters())
    flat_grads = {n: float(p.grad) for n, p in own.named_parameters() if p.grad is not None}
    assert n_floats == len(flat_grads)
    q.put((rank, grads, ref_grads, flat_grads))
    dist.destroy_process_group()


def test_ddp_gradients_are_the_global_batch_gradients():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, g0, ref0, f0), (_, g1, _, f1) = res
    assert set(g0) == set(ref0) == set(f0) and len(g0) >= 9
    for n in g0:
        assert g0[n] == g1[n] and f0[n] == f1[n], n  # all-reduced: identical on both ranks
        assert abs(float(g0[n]) - float(ref0[n])) <= 1e-5 + 1e-3 * abs(float(ref0[n])), (n, float(g0[n]), float(ref0[n]))
        assert abs(float(f0[n]) - float(ref0[n])) <= 1e-5 + 1e-3 * abs(float(ref0[n])), (n, float(f0[n]), float(ref0[n]))


def _captured_worker(rank, world, port, q):
    """Two ranks, each with its shard of four tiles: the captured training step (two hipGraphs around one all-reduce)
    must leave both ranks with the parameters a single process gets from the global batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import scene_net_amd as sna
    from scene_net_amd.synthetic import synthetic_tile
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    tiles, labels = zip(*[synthetic_tile(300 + i, 6_000) for i in range(4)])

    def build(idx):
        torch.manual_seed(11)
        model = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)).to(dev)
        batch = sna.PointBatch.from_tiles([tiles[i] for i in idx], [labels[i] for i in idx], device=dev)
        pipe = sna.ScenePipeline(model, (16, 16, 16), keep_labels=[15.0])
        crit = sna.GENEO_Tversky_Loss(targets=torch.tensor([0.0, 1.0]), weighting_scheme_path=None,
                                      save_weighting_scheme=False)
        opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        return model, batch, pipe, crit, opt

    lo, hi = sna.shard_range(4, rank, world)
    model, batch, pipe, crit, opt = build(range(lo, hi))
    step = sna.CapturedTrainingStep(pipe, crit, opt, batch, warmup=2)
    assert step.world == 2 and step.graph_opt is not None
    losses = [float(step.replay()) for _ in range(4)]
    torch.cuda.synchronize()
    got = {n: float(p) for n, p in model.named_parameters()}
    # the same six steps (2 warm-up + 4 replays) in this process alone, eagerly: per step the gradients of the two
    # shards one after the other, their mean, one optimizer step -- what the two ranks compute between them
    ref_model, b0, ref_pipe, ref_crit, ref_opt = build(range(0, 2))
    b1 = sna.PointBatch.from_tiles([tiles[i] for i in range(2, 4)], [labels[i] for i in range(2, 4)], device=dev)
    ps = [p for p in ref_model.parameters() if p.requires_grad]
    for _ in range(6):
        shard_grads = []
        for bt in (b0, b1):
            ref_opt.zero_grad(set_to_none=True)
            grids = ref_pipe.voxelize(bt, want_gt=True)
            loss = ref_crit(ref_model(grids.occ), grids.gt_occ, ref_model.get_cvx_coefficients(),
                            ref_model.get_geneo_params())
            loss.backward()
            shard_grads.append([p.grad.clone() for p in ps])
        for p, ga, gb in zip(ps, *shard_grads):
            p.grad = (ga + gb) / 2
        ref_opt.step()
    ref = {n: float(p) for n, p in ref_model.named_parameters()}
    q.put((rank, got, ref, losses))
    dist.destroy_process_group()


def test_captured_training_step_under_a_live_process_group():
    """the hipGraph-captured training step under N > 1 -- two ranks on the test box's one GPU (gloo
    there; nccl = RCCL with one rank per GPU on a node)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_captured_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, g0, ref, l0), (_, g1, _, l1) = res
    assert set(g0) == set(ref)
    moved = 0
    for n in g0:
        assert g0[n] == g1[n], n                    # same all-reduced gradients, same optimizer: identical replicas
        assert abs(g0[n] - ref[n]) <= 1e-5 + 2e-3 * abs(ref[n]), (n, g0[n], ref[n])
        moved += 1
    assert moved >= 9 and all(np.isfinite(l0)) and all(np.isfinite(l1))
