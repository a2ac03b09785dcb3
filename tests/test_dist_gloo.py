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


# --------------------------------------------------------------------------- bench.py --gpus N: the rank launcher
def _load_launch():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_sn_launch_t", os.path.join(root, "scene-net_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, root


_RANK_SCRIPT = """
import os, sys, json
import torch, torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([float(os.environ["RANK"]) + 1.0])
dist.all_reduce(t)
with open(sys.argv[1] + "." + os.environ["RANK"], "w") as f:
    json.dump({"rank": int(os.environ["RANK"]), "local": int(os.environ["LOCAL_RANK"]),
               "world": dist.get_world_size(), "sum": float(t), "addr": os.environ["MASTER_ADDR"]}, f)
dist.destroy_process_group()
"""


def test_launcher_starts_one_rank_per_device_over_gloo(tmp_path):
    """launch_ranks sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like torch.distributed.run and the ranks can form a
    process group with them (what `bench.py --gpus N` relies on)."""
    import json
    launch, _ = _load_launch()
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    rc = launch.launch_ranks(2, str(script), [str(tmp_path / "out")], timeout_s=120)
    assert rc == 0
    got = [json.load(open(tmp_path / f"out.{r}")) for r in range(2)]
    for r, g in enumerate(got):
        assert g == {"rank": r, "local": r, "world": 2, "sum": 3.0, "addr": "127.0.0.1"}


def test_launcher_reports_a_failed_rank_and_stops_the_rest(tmp_path):
    import time as _t
    launch, _ = _load_launch()
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(5)\ntime.sleep(120)\n")
    t0 = _t.monotonic()
    rc = launch.launch_ranks(2, str(script), [], timeout_s=100)
    assert rc == 5 and _t.monotonic() - t0 < 60   # rank 0 was terminated, not waited for


def test_launcher_module_is_stdlib_only():
    """the launching parent must not import torch (or anything that could initialise the device)"""
    import subprocess
    import sys
    _, root = _load_launch()
    code = ("import importlib.util, sys; s = importlib.util.spec_from_file_location('l', sys.argv[1]); "
            "m = importlib.util.module_from_spec(s); s.loader.exec_module(m); "
            "assert 'torch' not in sys.modules and 'numpy' not in sys.modules; print(m.under_launcher({}))")
    out = subprocess.run([sys.executable, "-c", code, os.path.join(root, "scene-net_amd", "launch.py")],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "False", out.stderr


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher above it starts 2 ranks itself.  Without a HIP device a rank stops at
    the device check (exit 3, naming the device it wanted; the launcher then stops the other rank).  A --gpus /
    WORLD_SIZE mismatch is refused (exit 2)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side check of the launcher's refusal paths")
    _, root = _load_launch()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["SN_LAUNCH_VERBOSE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 3, out.stderr[-400:]
    assert "started 2 ranks" in out.stderr and "needs HIP device" in out.stderr
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr
