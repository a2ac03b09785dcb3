#!/usr/bin/env python3
"""Host model of the z-walk kernel's ticket protocol (csrc/conv_i8z.inc): waves claim tickets in order, every ticket's
dependency check reads monotonic per-slot counters.  Random interleavings; reports a deadlock (no wave can move) or a
counter read before its producer ran.  python tools/debug/zwalk_protocol_sim.py"""
import random, sys

def simulate(my_jobs, LZ, kH, nwaves, skip_round, seed, verbose=False):
    kTPS = 8 // kH
    PL = LZ + 8
    kZD, kZDD, kZPre = 10, 4, 14
    NPL = my_jobs * PL
    nslots = NPL + kZD
    ntickets = kTPS * nslots
    npre = min(NPL, kZPre)
    folded = [0] * 16; read = [0] * 16; landed = [[1 if sl < npre else 0 for _ in range(4)] for sl in range(16)]
    plane_landed = set((u, xr) for u in range(npre) for xr in range(4))   # ground truth
    plane_folded = {}
    rng = random.Random(seed)
    ticket = [0]
    # wave state: (phase, data)
    waves = [dict(t=None, phase='claim', pend=None, next=None) for _ in range(nwaves)]
    def claim():
        t = ticket[0]; ticket[0] += 1; return t
    for w in waves:
        w['next'] = claim()
    done = 0
    steps = 0
    while done < nwaves:
        movable = []
        for wi, w in enumerate(waves):
            if w['phase'] == 'done':
                continue
            if w['phase'] == 'claim':
                movable.append(wi)
            elif w['phase'] == 'check':
                t = w['t']; s = t // kTPS; tk = t % kTPS
                pass_ticket = kH == 2 or tk % 2 == 0
                xr = tk if kH == 2 else tk // 2
                sd = s + kZDD
                has_dma = pass_ticket and s >= kZD and sd < NPL
                has_fold = pass_ticket and s < NPL
                sigma = s - kZD
                has_round = False
                if sigma >= 0:
                    kr, o = divmod(sigma, PL)
                    has_round = o < LZ
                ok = True
                if has_round:
                    for i in range(9):
                        v = sigma + i
                        if folded[v & 15] < 4 * ((v >> 4) + 1): ok = False
                if has_fold:
                    for i in range(9):
                        v = s - 14 + i
                        if v >= 0 and read[v & 15] < kTPS * ((v >> 4) + 1): ok = False
                    if landed[s & 15][xr] < (s >> 4) + 1: ok = False
                if has_dma:
                    v = sd - 16
                    if v >= 0 and folded[v & 15] < 4 * ((v >> 4) + 1): ok = False
                if ok:
                    movable.append(wi)
            else:
                movable.append(wi)
        if not movable:
            print("DEADLOCK", dict(my_jobs=my_jobs, LZ=LZ, kH=kH, nwaves=nwaves, seed=seed))
            for wi, w in enumerate(waves):
                if w['phase'] != 'done':
                    t = w['t']; print("  wave", wi, w['phase'], "ticket", t, "slot", t // kTPS, "tk", t % kTPS, "pend", w['pend'])
            print("  folded", folded); print("  read", read); print("  landed", landed)
            return False
        wi = rng.choice(movable); w = waves[wi]
        steps += 1
        if w['phase'] == 'claim':
            t = w['next']
            if t >= ntickets:
                if w['pend'] is not None:
                    u, xr = w['pend']; landed[u & 15][xr] += 1; plane_landed.add((u, xr)); w['pend'] = None
                w['phase'] = 'done'; done += 1; continue
            w['t'] = t; w['next'] = claim()
            if w['pend'] is not None:
                u, xr = w['pend']; landed[u & 15][xr] += 1; plane_landed.add((u, xr)); w['pend'] = None
            w['phase'] = 'check'
        elif w['phase'] == 'check':
            t = w['t']; s = t // kTPS; tk = t % kTPS
            pass_ticket = kH == 2 or tk % 2 == 0
            xr = tk if kH == 2 else tk // 2
            sd = s + kZDD
            has_dma = pass_ticket and s >= kZD and sd < NPL
            has_fold = pass_ticket and s < NPL
            sigma = s - kZD
            has_round = False
            if sigma >= 0:
                kr, o = divmod(sigma, PL)
                has_round = o < LZ
            if has_dma:
                w['pend'] = (sd, xr)
            if has_fold:
                assert (s, xr) in plane_landed, ("fold before landed", s, xr)
                plane_folded[(s, xr)] = True
                folded[s & 15] += 1
            if has_round and not skip_round(rng):
                for i in range(9):
                    for x in range(4):
                        assert (sigma + i, x) in plane_folded, ("round reads unfolded plane", sigma + i, x)
                w['phase'] = 'mfma'
            else:
                read[s & 15] += 1
                w['phase'] = 'claim'
        elif w['phase'] == 'mfma':
            s = w['t'] // kTPS
            read[s & 15] += 1
            if w['pend'] is not None:
                u, xr = w['pend']; landed[u & 15][xr] += 1; plane_landed.add((u, xr)); w['pend'] = None
            w['phase'] = 'claim'
    return True

if __name__ == "__main__":
    ok = True
    for kH, nw in ((2, 8), (1, 12), (1, 16)):
        for my_jobs in (1, 2, 5):
            for LZ in (1, 3, 7, 8, 20, 64):
                for skip in (lambda r: False, lambda r: r.random() < 0.3):
                    for seed in range(6):
                        ok &= simulate(my_jobs, LZ, kH, nw, skip, seed)
    print("all interleavings completed" if ok else "FAILED")
