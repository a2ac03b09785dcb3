#!/usr/bin/env python3
"""Host model of the z-walk kernel's ticket protocol (csrc/conv_i8z.inc).

Waves claim tickets in order; a ticket = (slot s, k): [LDS-DMA of pass(es) of plane s + kZDD] + [fold pass(es) of plane s] +
[rounds of output sigma = s - kZD].  All hand-overs are monotonic counters per ring slot (folded / read / landed), checked
in one read per ticket.  The model keeps the GROUND TRUTH beside the counters -- which plane every ring row holds, which
LDS-DMA is still in flight, which rounds are reading which planes -- and asserts the four hazards the counters exist for:

  RAW  a fold reads raw rows that have landed and belong to its plane;  a round reads folded rows of ITS nine planes
  WAR  an LDS-DMA does not overwrite raw rows whose fold has not run;  a fold does not overwrite a plane a round that is
       still in flight reads (checked when the round ENDS: its planes must still be in the ring)

Round 3's model checked completion and the two RAW cases under uniformly random interleavings; it had no notion of a round
that is IN FLIGHT, so no schedule could show a WAR hazard.  This one (round 4) has, and its scheduler is adversarial: it
freezes a wave inside a round while the others run as far as the protocol lets them.  That finds the hole the kernel had:

  read[slot] is shared by the slots v and v + 16.  A ticket of slot v + 16 that has nothing to do (no round: the 8 trailing
  planes of a job or the flush slots behind the stream; no fold: s >= NPL) reported at once; with a round of slot v still in
  flight, read[v & 15] reached the value that tells the folds of planes v + 5 .. v + 13 "the readers of plane v - 11 .. v - 3
  are through" -- one report early.  Reachable at the END of a stream only (elsewhere the folds that wait for slot v hold
  every other wave), e.g. one job of 72 planes, 12 waves: the round at slot NPL - 6 frozen, three waves run through the
  flush slots.  `ordered_reads=True` is the fix the kernel carries: every ticket's check also waits for read[s - 16] to be
  complete (one more lane of the same LDS read), so a slot's reports are all in before the next owner of its counter adds.

python tools/debug/zwalk_protocol_sim.py            (all shapes, both schedulers; exits non-zero on a hazard)
python tools/debug/zwalk_protocol_sim.py --hole     (the round-3 protocol: shows the hazard above)"""
import random
import sys

RING = 16


class Hazard(Exception):
    pass


def simulate(my_jobs, LZ, kTPS, nwaves, seed, lag=3, dd=4, ordered_reads=True, skip_prob=0.0, adversarial=True,
             max_steps=2_000_000, freeze_slot=None):
    """kTPS tickets per slot: 8 (one round of one x-row per ticket; even tickets carry a pass), 4 (two rounds; every ticket
    one pass: the shipped shape), 2 (four rounds, two passes)."""
    PL = LZ + 8
    kZD = 8 + lag
    kPre = kZD + dd
    assert kPre <= RING and RING - 8 - lag >= 2
    kPPT = 2 if kTPS == 2 else 1
    NPL = my_jobs * PL
    nslots = NPL + kZD
    ntickets = kTPS * nslots
    npre = min(NPL, kPre)
    rng = random.Random(seed)
    # counters (what the kernel reads)
    folded = [0] * RING
    read = [0] * RING
    landed = [[1 if sl < npre else 0 for sl in range(RING)] for _ in range(4)]
    # ground truth
    raw = [[(sl if sl < npre else None) for sl in range(RING)] for _ in range(4)]   # raw[xr][slot] = plane | ('fly', plane) | None
    yring = [[None] * RING for _ in range(4)]                                         # yring[xr][slot] = plane
    fold_done = set()
    ticket = [0]
    waves = [dict(phase='claim', t=None, pend=None, planes=None, frozen=0) for _ in range(nwaves)]

    def shape(t):
        s, tk = divmod(t, kTPS)
        pass_ticket = kTPS <= 4 or tk % 2 == 0
        xr0 = (tk >> 1) if kTPS == 8 else tk * kPPT
        sd = s + dd
        has_dma = pass_ticket and s >= kZD and sd < NPL
        has_fold = pass_ticket and s < NPL
        sigma = s - kZD
        has_round = False
        if sigma >= 0:
            _, o = divmod(sigma, PL)
            has_round = o < LZ
        return s, tk, xr0, sd, sigma, has_dma, has_fold, has_round

    def deps_ok(t):
        s, tk, xr0, sd, sigma, has_dma, has_fold, has_round = shape(t)
        if has_round:
            for i in range(9):
                v = sigma + i
                if folded[v & 15] < 4 * ((v >> 4) + 1):
                    return False
        if has_fold:
            for i in range(9):
                v = s - RING + lag + i
                if v >= 0 and read[v & 15] < kTPS * ((v >> 4) + 1):
                    return False
            if landed[xr0][s & 15] < (s >> 4) + 1:      # (kPPT = 2: both passes are reported by one instruction)
                return False
        if has_dma:
            v = sd - RING
            if v >= 0 and folded[v & 15] < 4 * ((v >> 4) + 1):
                return False
        if ordered_reads:
            v = s - RING
            if v >= 0 and read[v & 15] < kTPS * ((v >> 4) + 1):
                return False
        return True

    def report_pend(w):
        if w['pend'] is not None:
            u, xr0 = w['pend']
            for pp in range(kPPT):
                if raw[xr0 + pp][u & 15] != ('fly', u):
                    raise Hazard(f"LDS-DMA of plane {u} pass {xr0 + pp} was overwritten in flight: {raw[xr0 + pp][u & 15]}")
                raw[xr0 + pp][u & 15] = u
                landed[xr0 + pp][u & 15] += 1
            w['pend'] = None

    done = 0
    steps = 0
    while done < nwaves:
        steps += 1
        if steps > max_steps:
            raise Hazard("no progress (step limit)")
        movable = []
        for wi, w in enumerate(waves):
            if w['phase'] == 'done':
                continue
            if w['frozen'] > 0:
                continue
            if w['phase'] == 'check' and not deps_ok(w['t']):
                continue
            movable.append(wi)
        if not movable:
            thaw = [w for w in waves if w['phase'] != 'done' and w['frozen'] > 0]
            if thaw:
                for w in thaw:
                    w['frozen'] = 0
                continue
            raise Hazard("DEADLOCK: " + "; ".join(f"wave {i} {w['phase']} ticket {w['t']} slot {w['t'] // kTPS}"
                                                  for i, w in enumerate(waves) if w['phase'] != 'done'))
        for w in waves:
            if w['frozen'] > 0:
                w['frozen'] -= 1
        wi = rng.choice(movable)
        w = waves[wi]
        if w['phase'] == 'claim':
            t = ticket[0]
            ticket[0] += 1
            if t >= ntickets:
                report_pend(w)
                w['phase'] = 'done'
                done += 1
                continue
            w['t'] = t
            report_pend(w)      # (a pass requested by an earlier ticket is reported before anything is waited for)
            w['phase'] = 'check'
        elif w['phase'] == 'check':
            s, tk, xr0, sd, sigma, has_dma, has_fold, has_round = shape(w['t'])
            if has_dma:
                for pp in range(kPPT):
                    xr = xr0 + pp
                    old = raw[xr][sd & 15]
                    if old is not None:
                        if isinstance(old, tuple):
                            raise Hazard(f"LDS-DMA of plane {sd} pass {xr} over an LDS-DMA in flight {old}")
                        if (old, xr) not in fold_done:
                            raise Hazard(f"LDS-DMA of plane {sd} pass {xr} overwrites raw rows of plane {old} before their fold")
                    raw[xr][sd & 15] = ('fly', sd)
                w['pend'] = (sd, xr0)
            if has_fold:
                for pp in range(kPPT):
                    xr = xr0 + pp
                    if raw[xr][s & 15] != s:
                        raise Hazard(f"fold of plane {s} pass {xr} reads raw rows holding {raw[xr][s & 15]}")
                    yring[xr][s & 15] = s
                    fold_done.add((s, xr))
                folded[s & 15] += kPPT
            if has_round and rng.random() >= skip_prob:
                planes = [sigma + i for i in range(9)]
                for p in planes:
                    for xr in range(4):
                        if yring[xr][p & 15] != p:
                            raise Hazard(f"round of output {sigma} (slot {s}) starts on plane {p} pass {xr}: ring holds {yring[xr][p & 15]}")
                w['planes'] = planes
                w['phase'] = 'mfma'
                if adversarial and rng.random() < 0.02:
                    w['frozen'] = rng.randrange(200, 4000)     # this round stalls; the others run as far as they are let
                if freeze_slot is not None and s == freeze_slot and tk == 0:
                    w['frozen'] = 10 ** 9                      # the directed adversary: held until nothing else can move
            else:
                read[s & 15] += 1
                w['phase'] = 'claim'
        elif w['phase'] == 'mfma':
            s = w['t'] // kTPS
            for p in w['planes']:
                for xr in range(4):
                    if yring[xr][p & 15] != p:
                        raise Hazard(f"plane {p} pass {xr} was overwritten by plane {yring[xr][p & 15]} under the round of slot {s} "
                                     f"(ticket {w['t']}) that was reading it")
            w['planes'] = None
            read[s & 15] += 1
            report_pend(w)
            w['phase'] = 'claim'
    return True


SHAPES = [(1, 1), (1, 3), (1, 8), (1, 20), (1, 64), (2, 8), (2, 26), (3, 5), (5, 1), (4, 32), (1, 128)]


def run_all(ordered_reads, seeds=range(8), verbose=True):
    bad = []
    for kTPS, nw in ((4, 12), (8, 12), (8, 8), (2, 12), (4, 16)):
        for my_jobs, LZ in SHAPES:
            for skip in (0.0, 0.3):
                for adversarial in (False, True):
                    for seed in seeds:
                        try:
                            simulate(my_jobs, LZ, kTPS, nw, seed, ordered_reads=ordered_reads, skip_prob=skip, adversarial=adversarial)
                        except Hazard as e:
                            bad.append((kTPS, nw, my_jobs, LZ, skip, adversarial, seed, str(e)))
    if verbose:
        for b in bad[:12]:
            print("HAZARD kTPS=%d waves=%d jobs=%d LZ=%d skip=%.1f adversarial=%s seed=%d: %s" % b)
        print(f"{len(bad)} hazardous runs" if bad else "no hazard, no deadlock in any run")
    return bad


if __name__ == "__main__":
    hole = "--hole" in sys.argv
    bad = run_all(ordered_reads=not hole)
    sys.exit(1 if bad and not hole else 0)
