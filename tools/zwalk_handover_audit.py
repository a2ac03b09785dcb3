#!/usr/bin/env python3
"""ISA audit of the z-walk kernel's hand-overs (csrc/conv_i8z.inc), for EVERY instantiation the library can launch.

The walk orders its workgroup with LDS counters instead of barriers; what keeps that sound below the source level is the
ORDER OF INSTRUCTIONS hipcc emits around each counter access -- the source has only `asm volatile(... ::: "memory")` between a
counter and the data it publishes.  This script cross-compiles conv_i8s.hip to ISA (no GPU), finds the hand-over points by
the `; @zw:*` comments the source's asm statements carry, and checks, per instantiation:

  R1  consumer side: inside the ticket loop no ring / raw access (ds_read_b128, ds_read2_b32, ds_write_b128, LDS-DMA, or a
      ds_read_b32 / ds_write_b32 outside the counter-and-table region) is placed between the ticket's claim and the
      `@zw:spin_exit` marker -- nothing the dependency check guards is issued before the check has been passed.
  R2  the spin reads its counter with a DS instruction (never FLAT) and waits (lgkmcnt(0)) before it compares.
  R3  HAND-OVER 1 (LDS-DMA -> fold): every `@zw:add1` (landed) is preceded by a `@zw:wait1` (s_waitcnt vmcnt(0)) with no
      LDS-DMA and no branch target between them other than the lane-0 guard of the add.
  R4  HAND-OVER 2 (fold -> rounds): every `@zw:add2` (folded) follows the fold pass's ds_write_b128 and ds_write_b32 in the
      same straight-line region, with no other DS write after it before the region ends.
  R5  HAND-OVER 3 (rounds -> fold): every `@zw:add3` (read) that ends a round is preceded by `@zw:wait3`
      (s_waitcnt lgkmcnt(0)) with no DS read between the wait and the add.
  R6  every LDS-DMA sits in the asm block that saves, sets and restores M0 (s_mov m0 / s_nop 0 / load / s_mov m0).
  R7  no FLAT memory instruction in the kernel, and no scratch access inside the ticket loop of the instantiations that are
      launched by default (a spill reload inside the loop would wait on vmcnt for the wave's LDS-DMA).

  python3 tools/zwalk_handover_audit.py [--no-build] [--verbose]      exit status 0 = every rule holds everywhere."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "build", "asm", "conv_i8s.s")
SRC = [os.path.join(ROOT, "scene-net_amd", "csrc", f) for f in
       ("conv_i8s.hip", "conv_i8_common.inc", "conv_i8s_kernel.inc", "conv_i8f.inc", "conv_i8z.inc", "conv_prep.h",
        "conv_fp32.inc", "common.h")]
# LDS carve-up of the walk (conv_i8z.inc): digit table, job table, scale / coefficients / bounds, counters, check table,
# the lane / ring-offset tables, then the rings.  Offsets below kRawBase belong to tables and counters; ring and raw data start there.
K_TABLES_END = 4 * 3 * 64 * 16 + 256 * 16 + 64 * 4 + 128 + 128 * 4 + 64 * 16 + 2 * 64 * 16   # = 20352 = 0x4f80 (incl. ltab, stab)
DEFAULT_LAUNCHED = ("Li1ELi2ELi12E",)   # sn_set_option("conv_i8z_variant") default 2: rounds of one x-row, two per ticket, 12 waves


def build():
    os.makedirs(os.path.dirname(ASM), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "scene-net_amd", "csrc"),
                           "-S", "--cuda-device-only", SRC[0], "-o", ASM], stderr=subprocess.DEVNULL)


def fresh():
    return os.path.exists(ASM) and all(os.path.getmtime(ASM) >= os.path.getmtime(f) for f in SRC)


def kernels(lines):
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w*conv_occ_i8z_kernel\w*):", l)
        if m:
            end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith(".Lfunc_end"))
            out.append((m.group(1), lines[i:end]))
    return out


def op_of(line):
    t = line.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        return None
    return t.split()[0]


def ds_offset(line):
    m = re.search(r"offset:(\d+)", line)
    return int(m.group(1)) if m else None


RING_OPS = ("ds_read_b128", "ds_read2_b32", "ds_read_b64", "ds_read2_b64", "ds_write_b128", "ds_write_b64", "global_load_lds",
            "buffer_load")
BRANCHES = ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz",
            "s_cbranch_execnz")


def is_ring_access(line):
    """an access to ring / raw data (as opposed to the tables and counters below K_TABLES_END)"""
    op = op_of(line)
    if op is None:
        return False
    off = ds_offset(line)
    in_tables = off is not None and 12288 <= off < K_TABLES_END
    if op.startswith(RING_OPS):
        # (the job table, the lane table and the ring-offset table are read with ds_read_b128 / ds_read_b64)
        return not (op in ("ds_read_b128", "ds_read_b64") and in_tables)
    if op in ("ds_read_b32", "ds_write_b32", "ds_read_b96", "ds_read2_b32"):
        return not in_tables
    return False


class CFG:
    """basic blocks of one kernel body: a block ends behind every branch and before every label"""

    def __init__(self, body):
        self.body = body
        starts = {0}
        for i, l in enumerate(body):
            if re.match(r"^\.LBB\d+_\d+:", l):
                starts.add(i)
            if op_of(l) in BRANCHES or op_of(l) == "s_endpgm":
                starts.add(i + 1)
        self.starts = sorted(x for x in starts if x < len(body))
        self.block_of = {}
        self.blocks = []
        for k, st in enumerate(self.starts):
            en = self.starts[k + 1] if k + 1 < len(self.starts) else len(body)
            self.blocks.append((st, en))
            for i in range(st, en):
                self.block_of[i] = k
        label_block = {}
        for k, (st, en) in enumerate(self.blocks):
            m = re.match(r"^(\.LBB\d+_\d+):", body[st])
            if m:
                label_block[m.group(1)] = k
        self.succ = [[] for _ in self.blocks]
        self.pred = [[] for _ in self.blocks]
        for k, (st, en) in enumerate(self.blocks):
            last = next((i for i in range(en - 1, st - 1, -1) if op_of(body[i])), None)
            op = op_of(body[last]) if last is not None else None
            tgt = body[last].split()[1] if op in BRANCHES else None
            if tgt is not None and tgt in label_block:
                self.succ[k].append(label_block[tgt])
            if op != "s_branch" and op != "s_endpgm" and k + 1 < len(self.blocks):
                self.succ[k].append(k + 1)
        for k, ss in enumerate(self.succ):
            for t in ss:
                self.pred[t].append(k)

    def walk_back(self, pos, found, forbidden, limit=4000):
        """Every backward path from instruction `pos` (exclusive) must meet a line with found(line) before one with
        forbidden(line) and before the kernel's entry.  Returns None if so, else a description of the offending line."""
        seen = set()
        work = [(self.block_of[pos], pos - 1)]
        steps = 0
        while work:
            k, i = work.pop()
            st, _ = self.blocks[k]
            hit = False
            while i >= st:
                steps += 1
                if steps > limit * 50:
                    return "search limit"
                l = self.body[i]
                if found(l):
                    hit = True
                    break
                if forbidden(l):
                    return f"`{l.strip()}` (+{i})"
                i -= 1
            if hit:
                continue
            if not self.pred[k]:
                return "reached the kernel's entry"
            for p in self.pred[k]:
                if p not in seen:
                    seen.add(p)
                    work.append((p, self.blocks[p][1] - 1))
        return None


def audit(name, body, verbose=False):
    errs = []
    n = len(body)
    cfg = CFG(body)
    mark = {k: [i for i, l in enumerate(body) if ("@zw:" + k) in l] for k in
            ("spin_exit", "add1", "add2", "add3", "wait1", "wait3", "dma")}
    claims = [i for i, l in enumerate(body) if op_of(l) == "ds_add_rtn_u32"]
    if len(mark["spin_exit"]) != 1 or len(claims) != 2:
        return [f"structure: {len(mark['spin_exit'])} spin exits, {len(claims)} claims"]
    spin_exit = mark["spin_exit"][0]
    e_blk = cfg.block_of[spin_exit]
    # ---- R1: the claim inside the loop starts an iteration (the first claim primes `next` ahead of the loop).  Forward from
    # the claim's block and from the block that guards it (lane 0 only: the other path skips the claim), NOT through the spin
    # exit: no ring access may be reachable.
    claim_blk = cfg.block_of[claims[1]]
    start = {claim_blk} | set(cfg.pred[claim_blk])
    seen, work = set(start), list(start)
    while work:
        k = work.pop()
        st, en = cfg.blocks[k]
        stop_at = spin_exit if k == e_blk else en
        for i in range(st, stop_at):
            if is_ring_access(body[i]):
                errs.append(f"R1: `{body[i].strip()}` (+{i}) can execute between a ticket's claim and its spin exit (+{spin_exit})")
        if k == e_blk:
            continue
        for t in cfg.succ[k]:
            if t not in seen:
                seen.add(t)
                work.append(t)
    pre_spin_blocks = len(seen)
    # every ring access of the kernel behind the prologue must be dominated by the spin exit: covered by the walk above for
    # the loop; the prologue (before the first claim) is ordered by barriers.
    # ---- R2
    sleeps = [i for i, l in enumerate(body) if op_of(l) == "s_sleep" and cfg.block_of[i] in seen]
    if not sleeps:
        errs.append("R2: no s_sleep between the claim and the spin exit")
    ok = False
    for r in (i for i, l in enumerate(body) if op_of(l) == "ds_read_b32" and cfg.block_of[i] in seen and (ds_offset(l) or 0) >= 16768):
        ops = [(j, op_of(body[j])) for j in range(r + 1, min(r + 7, n)) if op_of(body[j])]
        w = next((j for j, o in ops if o == "s_waitcnt"), None)
        c = next((j for j, o in ops if o.startswith("v_cmp_lt_i32")), None)
        if w is not None and c is not None and w < c and "lgkmcnt(0)" in body[w]:
            ok = True
    if not ok:
        errs.append("R2: no `ds_read_b32 counter; s_waitcnt lgkmcnt(0); v_cmp_lt_i32` sequence in the spin")
    # ---- R3: landed
    for a in mark["add1"]:
        why = cfg.walk_back(a, lambda l: "@zw:wait1" in l,
                            lambda l: (op_of(l) or "").startswith(("global_load_lds", "v_mfma")) or "@zw:spin_exit" in l)
        if why:
            errs.append(f"R3: `@zw:add1` (+{a}): a path reaches it without `@zw:wait1`: {why}")
    for w in mark["wait1"]:
        wl = next(k for k in range(w, min(w + 3, n)) if op_of(body[k]) == "s_waitcnt")
        if "vmcnt(0)" not in body[wl]:
            errs.append(f"R3: `@zw:wait1` (+{w}) is `{body[wl].strip()}`")
    # ---- R4: folded
    for a in mark["add2"]:
        for want in ("ds_write_b128", "ds_write_b32"):
            why = cfg.walk_back(a, lambda l, want=want: op_of(l) == want,
                                lambda l: "@zw:" in l and "dma" not in l)
            if why:
                errs.append(f"R4: `@zw:add2` (+{a}): a path reaches it without the fold pass's {want}: {why}")
    # ---- R5: read
    rounds_reports = 0
    for a in mark["add3"]:
        behind_mfma = cfg.walk_back(a, lambda l: (op_of(l) or "").startswith("v_mfma"), lambda l: "@zw:spin_exit" in l) is None
        if not behind_mfma:
            continue      # the report of a ticket without a round: nothing was read
        rounds_reports += 1
        why = cfg.walk_back(a, lambda l: "@zw:wait3" in l, lambda l: (op_of(l) or "").startswith(("ds_read", "v_mfma")))
        if why:
            errs.append(f"R5: `@zw:add3` (+{a}): a path reaches it without `@zw:wait3` directly above: {why}")
    if rounds_reports != 1:
        errs.append(f"R5: {rounds_reports} read reports behind MFMAs (expected 1)")
    for w in mark["wait3"]:
        wl = next(k for k in range(w, min(w + 3, n)) if op_of(body[k]) == "s_waitcnt")
        if "lgkmcnt(0)" not in body[wl]:
            errs.append(f"R5: `@zw:wait3` (+{w}) is `{body[wl].strip()}`")
    # ---- R6
    for d in mark["dma"]:
        blk = [body[k].strip() for k in range(d, min(d + 8, n))]
        ops = [b for b in blk if b and not b.startswith(";")]
        names = [b.split()[0] for b in ops]
        if names[:5] != ["s_mov_b32", "s_mov_b32", "s_nop", "global_load_lds_dwordx4", "s_mov_b32"] or \
                not ops[0].endswith(", m0") or " m0," not in ops[1] or " m0," not in ops[4]:
            errs.append(f"R6: LDS-DMA block at +{d}: {ops[:5]}")
    ndma = sum(1 for l in body if (op_of(l) or "").startswith(("global_load_lds", "buffer_load")))
    if ndma != len(mark["dma"]):
        errs.append(f"R6: {ndma} LDS-DMA / buffer loads, {len(mark['dma'])} marked blocks")
    # ---- R7
    flat = [i for i, l in enumerate(body) if (op_of(l) or "").startswith("flat_")]
    if flat:
        errs.append(f"R7: FLAT instruction at +{flat[0]}: `{body[flat[0]].strip()}`")
    # the loop = what can reach the in-loop claim again
    loop_blocks = set()
    work = [claim_blk]
    while work:
        k = work.pop()
        for p_ in cfg.pred[k]:
            if p_ not in loop_blocks and cfg.blocks[p_][0] > claims[0]:
                loop_blocks.add(p_)
                work.append(p_)
    scratch = [i for i, l in enumerate(body) if (op_of(l) or "").startswith("scratch_") and cfg.block_of[i] in loop_blocks]
    if scratch and any(t in name for t in DEFAULT_LAUNCHED) and "Lb0E" in name:
        errs.append(f"R7: scratch access inside the ticket loop of the headline instantiation at +{scratch[0]}")
    if verbose:
        print(f"  {name}: {len(cfg.blocks)} blocks, {len(loop_blocks)} in the ticket loop, {pre_spin_blocks} reachable ahead of the "
              f"spin exit (+{spin_exit}); markers " + ", ".join(f"{k}:{len(v)}" for k, v in mark.items())
              + f"; scratch in loop: {len(scratch)}")
    return errs


def main():
    if "--no-build" not in sys.argv and not fresh():
        build()
    lines = open(ASM).read().split("\n")
    ks = kernels(lines)
    if len(ks) < 12:
        print(f"expected 12 instantiations of conv_occ_i8z_kernel, found {len(ks)}")
        return 1
    bad = 0
    for name, body in ks:
        errs = audit(name, body, "--verbose" in sys.argv)
        for e in errs:
            print(f"{name}: {e}")
        bad += len(errs)
    print(f"{len(ks)} instantiations audited: " + ("every hand-over rule holds" if not bad else f"{bad} violations"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
