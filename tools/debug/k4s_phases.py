#!/usr/bin/env python3
"""Where a job of the sparse correlation spends its time (needs `make -B EXTRA=-DSN_CONV_DEBUG`): SN_K4S_SKIP bits
1 no gather, 2 empty lists, 4 no tile writes, 8 no global loads -- wrong results, timing only.  One process per setting
(the environment is read once); run each under rocprofv3 --kernel-trace --stats for kernel times:
   SN_K4S_SKIP=3 rocprofv3 --kernel-trace --stats -d out -- python3 tools/debug/k4s_phases.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_net_amd as sna
from scene_net_amd import _hip
from scene_net_amd.synthetic import synthetic_tile
dev = torch.device("cuda:0"); B = 32
batch = sna.PointBatch.from_tiles([synthetic_tile(t)[0] for t in range(B)], device=dev)
x = sna.voxelize_batch(batch, (64, 64, 64), occ_dtype=torch.bool).occ.reshape(B, 1, 64, 64, 64)
g = torch.randn(B, 1, 64, 64, 64, device=dev); o = torch.tanh(torch.randn(B, 1, 64, 64, 64, device=dev)).clamp_min(0)
if len(sys.argv) > 1: _hip.set_option("corr_sparse_tile_bytes", int(sys.argv[1]))
def run(): return _hip.conv_corr(x, g, o, (9, 9, 9))
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
print("SN_K4S_SKIP=%s: %8.1f us per call (events around 20 calls)" % (os.environ.get("SN_K4S_SKIP", "0"), a.elapsed_time(b) / 20 * 1e3))
