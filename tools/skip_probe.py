import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from scene_net_amd import _hip
dev = torch.device("cuda:0")
bank = (torch.rand(16, 9, 9, 9) - 0.5).to(dev).contiguous(); lam = (torch.rand(16) / 16).to(dev)
for name, x in (("zeros", torch.zeros(32, 1, 64, 64, 64, dtype=torch.bool, device=dev)),
                ("half", torch.cat([torch.zeros(32, 1, 64, 32, 64, dtype=torch.bool, device=dev), torch.rand(32, 1, 64, 32, 64, device=dev) < 0.03], 3).contiguous())):
    for opt in (0, 1):
        _hip.set_option("conv_skip_empty_tiles", opt)
        for _ in range(5): _hip.conv_bank(x, bank, lam)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): _hip.conv_bank(x, bank, lam)
        e1.record(); torch.cuda.synchronize()
        print(name, "skip", opt, f"{e0.elapsed_time(e1)/20*1000:.1f} us")
