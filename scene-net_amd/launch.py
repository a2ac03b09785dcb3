"""One process per GPU, started from a parent that never touches the GPU.

The reference's only multi-device hook is `pl.Trainer(gpus=-1)` (scripts/main.py:228,
experiments/scenenet_ts40k/defaults_config.yml:50-51), i.e. Lightning spawning one rank per device.  Here the same
job shape is explicit: `launch_ranks(n, script, argv)` starts n fresh children of `script` with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what `python -m torch.distributed.run` would set), waits for them and
returns the worst exit code.  Stdlib only: the parent must not import torch or load the HIP library, so that the
children are started by a process that has never initialised the device.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time
from typing import Dict, Optional, Sequence


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_launcher(environ=os.environ) -> bool:
    """True inside a rank started by torchrun / torch.distributed.run / launch_ranks."""
    return "WORLD_SIZE" in environ and "RANK" in environ


def launch_ranks(n: int, script: str, argv: Sequence[str], extra_env: Optional[Dict[str, str]] = None,
                 timeout_s: Optional[float] = None, poll_s: float = 0.05) -> int:
    """Run `python script *argv` as n ranks on this node.  Children inherit stdout/stderr (rank 0 prints the result).
    If one rank exits non-zero the others are terminated (they would otherwise wait in a collective for ever).
    Returns 0 when every rank returned 0, else the first non-zero code seen (124 on timeout)."""
    if n < 1:
        raise ValueError(f"launch_ranks: n must be >= 1, got {n}")
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this host
        #   (rehearsed only: no >= 2-GPU RCCL run of this launcher is on record yet -- tests/test_gpu_rccl.py)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env))
    if os.environ.get("SN_LAUNCH_VERBOSE") == "1":
        print(f"launch_ranks: started {n} ranks of {os.path.basename(script)} (LOCAL_RANK 0..{n - 1}, port {port})",
              file=sys.stderr, flush=True)
    rc = 0
    t0 = time.monotonic()
    live = list(procs)
    try:
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 and live:
                break
            if timeout_s is not None and time.monotonic() - t0 > timeout_s:
                rc = rc or 124
                break
            if live:
                time.sleep(poll_s)
    finally:
        for p in live:   # exact PIDs we started, never a pattern
            try:
                p.send_signal(signal.SIGTERM)
            except ProcessLookupError:
                pass
        for p in live:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc
