#!/usr/bin/env python3
"""profiles/traffic.json from a pmc_traffic.csv (tools/pmc_summary.py over the FETCH_SIZE / WRITE_SIZE / TCC_EA0_ATOMIC_sum
passes):  python3 tools/traffic_json.py gpurun_out/<tag>/pmc_traffic.csv profiles/traffic.json <csv name for the note>.
bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction."""
import csv, json, sys

src, dst = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else src
kb = {}
for row in csv.DictReader(open(src)):
    kb.setdefault(row["kernel"], {})[row["counter"]] = float(row["avg_per_launch"])
byt = lambda k: int(round((2.0 * kb[k].get("FETCH_SIZE", 0.0) + kb[k].get("WRITE_SIZE", 0.0)) * 1024))
old = json.load(open(dst)) if len(sys.argv) > 4 and sys.argv[4] == "--keep-missing" else {}
# the voxel stage's kernels: round 4's one-pass form (occ_onepass_kernel + finalize + gated fallback) when the counters saw it,
# else the two-kernel form of rounds 1-3
vox = (["occ_onepass_kernel", "occ_finalize_kernel", "occ_fallback_kernel"] if "occ_onepass_kernel" in kb
       else ["bbox_partial_kernel", "occ_partial_kernel", "occ_finalize_kernel", "occ_fallback_kernel"])
out = {"_note": f"HBM-side bytes per launch at BASELINE C2 (32 tiles), from separate rocprofv3 --pmc passes (FETCH_SIZE; "
                f"WRITE_SIZE; TCC_EA0_ATOMIC_sum, one counter per pass) over tools/profile_path.py: {label}.  bytes = "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of a "
                "wide coalesced stream; the stride-4 kernel's halo comes in by 16-byte LDS-DMA pieces, the voxel kernels "
                "read 16 bytes per lane)."}
for k, v in old.items():
    if k != "_note":
        out[k] = v
for k in ("conv_occ_i8z_kernel", "conv_occ_i8f_kernel", "conv_occ_i8s_kernel", "conv_occ_i8_kernel", "conv_bank_kernel", "conv_lin_i8_kernel"):
    if k in kb and (kb[k].get("WRITE_SIZE", 0.0) > 0 or k not in out):
        out[k] = byt(k)
if all(k in kb for k in vox[:2]):
    out["voxel_stage_kernels"] = {k: byt(k) for k in vox if k in kb}
    out["voxel_stage"] = sum(out["voxel_stage_kernels"].values())
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_note"}, indent=1))
