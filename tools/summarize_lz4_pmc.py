#!/usr/bin/env python3
"""rocprofv3 --pmc passes of the GPU LZ4 decode kernel (tools/gpu_profile_extra.sh) -> a small text summary:
counter sums over the decode launches of the run, per block and per LZ4 sequence."""
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
tot, launches, blocks = {}, 0, 0
for sub in ("lz_pmc_sq", "lz_pmc_sq2"):
    hits = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not hits:
        continue
    seen = set()
    for r in csv.DictReader(open(max(hits, key=os.path.getmtime))):
        if "lz4_decode" not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        key = (sub, r["Dispatch_Id"])
        if sub == "lz_pmc_sq" and key not in seen:
            seen.add(key)
            launches += 1
            blocks += int(r["Grid_Size"]) // int(r["Workgroup_Size"])
lines = ["# rocprofv3 --pmc, two passes, tests/perf/trace_lz4_gpu.py 2147483648 fast:2 (five GPU-decoded passes over one image);",
         "# sums over the lz4_decode_wg launches: %d launches, %d blocks (workgroups of 8 waves)" % (launches, blocks)]
for k in sorted(tot):
    lines.append("%-22s %.4g" % (k, tot[k]))
if blocks and "SQ_INSTS_VALU" in tot:
    seqs = 156000.0   # LZ4-fast, NA12878-like block (tests/perf/lz4_stream_stats.py)
    ins = tot.get("SQ_INSTS_VALU", 0) + tot.get("SQ_INSTS_SALU", 0) + tot.get("SQ_INSTS_LDS", 0)
    lines.append("")
    lines.append("per block: %.3g VALU + %.3g SALU + %.3g LDS instructions = %.1f per LZ4 sequence (r03's kernel: 18.3)"
                 % (tot.get("SQ_INSTS_VALU", 0) / blocks, tot.get("SQ_INSTS_SALU", 0) / blocks, tot.get("SQ_INSTS_LDS", 0) / blocks, ins / blocks / seqs))
    if "SQ_WAVE_CYCLES" in tot and "SQ_WAVES" in tot:
        lines.append("wave time: SQ_WAVE_CYCLES %.3g per wave (x4 cycles); waiting (SQ_WAIT_ANY) %.0f %% of it, waiting for an issue slot (SQ_WAIT_INST_ANY) %.1f %%"
                     % (tot["SQ_WAVE_CYCLES"] / tot["SQ_WAVES"], 100 * tot.get("SQ_WAIT_ANY", 0) / tot["SQ_WAVE_CYCLES"],
                        100 * tot.get("SQ_WAIT_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"]))
    if "SQ_BUSY_CYCLES" in tot and "GRBM_GUI_ACTIVE" in tot:
        lines.append("instructions per SQ busy cycle (chip): %.2f" % (ins / max(tot["SQ_BUSY_CYCLES"], 1)))
open(os.path.join(dst, "lz4_gpu_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
