#!/usr/bin/env python3
"""Instruction classes per basic block of one z-walk instantiation (cross-compiled ISA; no GPU):
   python3 tools/zwalk_isa.py [mangled-name substring, default the head-only <float,1,2,12> kernel] [--dump LABEL]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "i8z_kernelIfLi1ELi2ELi12ELb0E"
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
out = os.path.join(ROOT, "build", "asm", "conv_i8s.s")
os.makedirs(os.path.dirname(out), exist_ok=True)
if "--no-build" not in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + ROOT + "/include",
                           "-I" + ROOT + "/scene-net_amd/csrc", "-S", "--cuda-device-only",
                           ROOT + "/scene-net_amd/csrc/conv_i8s.hip", "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(want) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = lines[start:end]
def cls(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_"): return "VALU"
    if op.startswith(("s_waitcnt", "s_nop")): return "WAIT"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_")): return "VMEM"
    if op.startswith("scratch_"): return "SCRATCH"
    return "OTHER"
cur, cnt, blocks, text = "entry", {}, [], {}
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((cur, cnt)); cur, cnt = m.group(1), {}
        continue
    t = l.strip()
    if not t or t.startswith((";", ".")): continue
    text.setdefault(cur, []).append(t.split(";")[0].rstrip())
    op = t.split()[0]
    cnt[cls(op)] = cnt.get(cls(op), 0) + 1
    if op.startswith(("s_cbranch", "s_branch")): cnt.setdefault("br", []).append(op[2:] + "->" + t.split()[1])
blocks.append((cur, cnt))
if dump:
    print("\n".join(text.get(dump, ["(no such block)"])))
else:
    tot = {}
    for b, c in blocks:
        br = c.pop("br", [])
        for k, v in c.items(): tot[k] = tot.get(k, 0) + v
        print(f"{b:12s}", " ".join(f"{k}={v}" for k, v in sorted(c.items())), " ".join(br))
    print("total", tot)
