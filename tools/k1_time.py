import os, sys, torch, time
sys.path.insert(0, "/root/repo")
import scene_net_amd as sna
from scene_net_amd.synthetic import synthetic_tile
dev = torch.device("cuda:0")
tiles = [synthetic_tile(i, 100_000)[0] for i in range(32)]
batch = sna.PointBatch.from_tiles(tiles, device=dev)
def run(n):
    for _ in range(n):
        g = sna.voxelize_batch(batch, (64, 64, 64), occ_dtype=torch.bool)
    return g
t = time.perf_counter()
while time.perf_counter() - t < 0.3:
    run(20); torch.cuda.synchronize()
res = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g = run(100); e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 100 * 1e3)
print("K1 stage us:", [round(x, 1) for x in res], "occ sum", int(g.occ.sum()))
