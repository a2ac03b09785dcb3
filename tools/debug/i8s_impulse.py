"""Impulse response of the stride-4 int8 kernel: which taps (dz,dx,dy) and which output residues come out wrong."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scene_net_amd as sna
from scene_net_amd import _hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
G = 16
bank = torch.zeros(G, 9, 9, 9)
# kernel 0: weight = tap index / 729 (all distinct); kernel 1: dy only; kernel 2: dx only; kernel 3: dz only
idx = torch.arange(729).view(9, 9, 9).float()
bank[0] = (idx + 1) / 729.0
bank[1] = (torch.arange(9).view(1, 1, 9) + 1).float().expand(9, 9, 9) / 9
bank[2] = (torch.arange(9).view(1, 9, 1) + 1).float().expand(9, 9, 9) / 9
bank[3] = (torch.arange(9).view(9, 1, 1) + 1).float().expand(9, 9, 9) / 9
bank = bank.to(dev).contiguous()
for pos in [(8, 8, 21), (8, 8, 20), (8, 8, 22), (8, 8, 23)]:
    x = torch.zeros(1, 1, 16, 16, 64, dtype=torch.bool, device=dev)
    x[0, 0, pos[0], pos[1], pos[2]] = True
    act, _ = _hip.conv_bank(x, bank, None, want_act=True, want_out=False)
    ref, _ = _hip.conv_bank(x.view(torch.uint8), bank, None, want_act=True, want_out=False)
    # out[v] = sum_t W[t] x[v + t - 4] -> impulse at c puts W[t] at v = c - t + 4
    z, xx, y = pos
    patch = act[0, :, z - 4:z + 5, xx - 4:xx + 5, y - 4:y + 5].flip(1, 2, 3)   # patch[g][dz][dx][dy] should be W
    rpatch = ref[0, :, z - 4:z + 5, xx - 4:xx + 5, y - 4:y + 5].flip(1, 2, 3)
    for g in range(4):
        d = (patch[g] - rpatch[g]).abs()
        bad = (d > 1e-4).nonzero()
        print(f"impulse at {pos} kernel {g}: wrong taps {bad.shape[0]} / 729, max diff {d.max().item():.4f}")
        if g == 0 and bad.shape[0]:
            got_idx = (patch[0] * 729 - 1).round().long()
            for b in bad[:12].tolist():
                dz, dx, dy = b
                gi = got_idx[dz, dx, dy].item()
                print(f"   tap (dz,dx,dy)=({dz},{dx},{dy}) got value of tap {gi} = ({gi // 81},{(gi // 9) % 9},{gi % 9})" if 0 <= gi < 729 else f"   tap ({dz},{dx},{dy}) got {patch[0][dz,dx,dy].item():.4f}")
    outside = act[0].abs().sum().item() - act[0, :, z - 4:z + 5, xx - 4:xx + 5, y - 4:y + 5].abs().sum().item()
    print("   mass outside the patch:", outside)
