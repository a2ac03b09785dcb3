#!/usr/bin/env python3
"""K3L floor experiments (needs `make -B EXTRA=-DSN_CONV_DEBUG`; wrong results, timing only): SN_CONV_LIN_DBG bits 1 prologue
only, 2 no MFMA loop, 4 no epilogue, 16 no deferral, 32 A table read for two steps only, 64 halo operands likewise -- one process per setting (the switch is read per call in debug builds).
CAUTION (round 4): a switch inside the software-pipelined step splits its basic block, so the debug build's own baseline is
slower than the product (65.7 against 58.5 us with two more switches in load_step24) and a variant that replaces the
switched code looks better than it is: "four ds_read2_b64 instead of eight ds_read_b64" read 55.6 against 67.8 here and
44.1 against 43.3 us when built for real (tools/debug/patches/k3l_rowcells.patch).  Trust bits 1, 2, 4, 16 (whole phases)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_net_amd as sna
from scene_net_amd import _hip
from scene_net_amd.synthetic import apply_bank_spec, synthetic_bank_spec, synthetic_tile
dev = torch.device("cuda:0")
GENEO = {"cy": 6, "cone": 5, "neg": 5}
specs, names, lambdas, last = synthetic_bank_spec(GENEO)
model = sna.SceneNet(GENEO, (9, 9, 9)); apply_bank_spec(model, specs, names, lambdas, last); model = model.to(dev)
batch = sna.PointBatch.from_tiles([synthetic_tile(t)[0] for t in range(32)], device=dev)
x = sna.voxelize_batch(batch, (64, 64, 64), occ_dtype=torch.bool).occ
bank, lam = model.compute_bank(dev), model.effective_lambdas(dev)
for dbg in ((0, 16, 0, 16, 0, 16) if '--defer' in sys.argv else (0, 1, 2, 4, 6, 16, 32, 64, 96)):
    os.environ["SN_CONV_LIN_DBG"] = str(dbg)
    for _ in range(20): _hip.conv_fused(x, bank, lam)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): _hip.conv_fused(x, bank, lam)
    b.record(); torch.cuda.synchronize()
    print(f"SN_CONV_LIN_DBG={dbg:2d}: {a.elapsed_time(b) / 50 * 1e3:7.1f} us per call (incl. the gated fallback launch)")
