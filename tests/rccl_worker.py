"""One rank of the multi-GPU tests (tests/test_gpu_rccl.py), started by scene-net_amd/launch.py::launch_ranks:
    python tests/rccl_worker.py <out_dir> <nccl|gloo> [--one-gpu]
`nccl` (= RCCL) puts rank r on HIP device r; `gloo --one-gpu` is the rehearsal on a one-GPU box (both ranks on cuda:0).
Each rank writes <out_dir>/rank<r>.json; the test process reads and compares them.
The training exchange under test is the reference's only multi-device hook, pl.Trainer(gpus=-1) (scripts/main.py:228):
one rank per device, gradients of the ~50 scalars averaged across ranks every step."""
import json
import time
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, backend = sys.argv[1], sys.argv[2]
    one_gpu = "--one-gpu" in sys.argv
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if one_gpu else int(os.environ["LOCAL_RANK"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    import scene_net_amd as sna
    from scene_net_amd.synthetic import synthetic_tile
    res = {"rank": rank, "world": dist.get_world_size(), "backend": dist.get_backend(), "device": local}

    # ---- (a) allreduce_flat_grads == the single-process mean over the shards
    g = torch.Generator().manual_seed(100)
    x = (torch.rand(4, 1, 12, 12, 24, generator=g) < 0.2)
    y = (torch.rand(4, 1, 12, 12, 24, generator=g) < 0.1).float()
    lo, hi = sna.shard_range(4, rank, world)
    torch.manual_seed(7)
    own = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)).to(dev)
    ((own(x[lo:hi].to(dev)) - y[lo:hi].to(dev)) ** 2).mean().backward()
    n_floats = sna.allreduce_flat_grads(own.parameters())
    res["flat_floats"] = n_floats
    res["flat_grads"] = {n: float(p.grad) for n, p in own.named_parameters() if p.grad is not None}
    shard_means = []
    for r in range(world):
        a, b = sna.shard_range(4, r, world)
        torch.manual_seed(7)
        ref = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)).to(dev)
        ((ref(x[a:b].to(dev)) - y[a:b].to(dev)) ** 2).mean().backward()
        shard_means.append({n: float(p.grad) for n, p in ref.named_parameters() if p.grad is not None})
    res["ref_grads"] = {n: sum(s[n] for s in shard_means) / world for n in shard_means[0]}

    # ---- (b) the captured training step (two hipGraphs around ONE all-reduce): replicas stay bit-identical
    tiles, labels = zip(*[synthetic_tile(300 + i, 6_000) for i in range(2 * world)])
    torch.manual_seed(11)
    model = sna.SceneNet({"cy": 1, "cone": 1, "neg": 1}, (9, 5, 5)).to(dev)
    lo, hi = sna.shard_range(2 * world, rank, world)
    batch = sna.PointBatch.from_tiles([tiles[i] for i in range(lo, hi)], [labels[i] for i in range(lo, hi)], device=dev)
    pipe = sna.ScenePipeline(model, (16, 16, 16), keep_labels=[15.0])
    crit = sna.GENEO_Tversky_Loss(targets=torch.tensor([0.0, 1.0]), weighting_scheme_path=None,
                                  save_weighting_scheme=False)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    step = sna.CapturedTrainingStep(pipe, crit, opt, batch, warmup=2)
    res["captured_world"] = step.world
    res["two_graphs"] = step.graph_opt is not None
    res["losses"] = [float(step.replay()) for _ in range(4)]
    torch.cuda.synchronize()
    res["params"] = {n: float(p) for n, p in model.named_parameters()}
    # bit-level comparison across ranks, on the device: gather every rank's parameter vector
    vec = torch.stack([p.detach().reshape(()) for p in model.parameters()]).to(dev)
    gathered = [torch.empty_like(vec) for _ in range(world)]
    dist.all_gather(gathered, vec)
    res["replicas_bit_identical"] = all(torch.equal(gathered[0], t) for t in gathered)

    # ---- (c) latency of THE exchange of the training path: one all-reduce over the flat gradient vector
    flat = torch.zeros(max(n_floats, 16), dtype=torch.float32, device=dev)
    for _ in range(20):
        dist.all_reduce(flat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_rep = 200
    for _ in range(n_rep):
        dist.all_reduce(flat)
    torch.cuda.synchronize()
    res["allreduce_flat_us"] = (time.perf_counter() - t0) / n_rep * 1e6
    res["allreduce_floats"] = int(flat.numel())
    res["env"] = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "MASTER_ADDR", "HIP_VISIBLE_DEVICES")}
    res["device_name"] = torch.cuda.get_device_name(dev)
    res["device_count"] = torch.cuda.device_count()

    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
