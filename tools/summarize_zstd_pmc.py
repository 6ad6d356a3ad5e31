#!/usr/bin/env python3
"""rocprofv3 --pmc passes over the GPU Zstandard decoder's four kernels (tools/zstd_pmc.sh) -> a small text summary:
counter sums per kernel over the launches of the run, per frame."""
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
tot, frames = {}, {}
for sub in ("zs_pmc_sq", "zs_pmc_sq2"):
    hits = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not hits:
        continue
    seen = set()
    for r in csv.DictReader(open(max(hits, key=os.path.getmtime))):
        name = r["Kernel_Name"]
        if "zstd_" not in name:
            continue
        k = name.split("zstd_")[1].split("<")[0]
        t = tot.setdefault(k, {})
        t[r["Counter_Name"]] = t.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        key = (sub, r["Dispatch_Id"])
        if sub == "zs_pmc_sq" and key not in seen:
            seen.add(key)
            wg = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
            waves = int(r["Grid_Size"]) // 64
            frames[k] = frames.get(k, 0) + (2 * waves if k == "chain" else wg)   # (a chain WAVE walks two frames; three waves a workgroup)
lines = ["# rocprofv3 --pmc, two passes, tests/perf/trace_lz4_gpu.py 2147483648 zstd:1 (five GPU-decoded passes over one image of 4195 frames);",
         "# sums over the launches of each kernel, per frame (1,024,000 bytes, 123 k sequences)"]
for k in ("prepare", "chain", "records", "execute"):
    t = tot.get(k)
    if not t:
        continue
    n = max(frames.get(k, 1), 1)
    ins = t.get("SQ_INSTS_VALU", 0) + t.get("SQ_INSTS_SALU", 0) + t.get("SQ_INSTS_LDS", 0)
    line = "%-8s per frame: %.3g VALU + %.3g SALU + %.3g LDS instructions" % (k, t.get("SQ_INSTS_VALU", 0) / n, t.get("SQ_INSTS_SALU", 0) / n, t.get("SQ_INSTS_LDS", 0) / n)
    if k == "chain":
        line += " = %.0f per step of the walk (16 blocks a wave: per wave-step %.0f)" % (ins / n / 123100.0, ins / n / 123100.0 * 16)
    if "SQ_WAVE_CYCLES" in t and "SQ_WAVES" in t:
        line += "; wave time %.3g cycles (x4) per wave, waiting (SQ_WAIT_ANY) %.0f %%, for an issue slot (SQ_WAIT_INST_ANY) %.1f %%" % (
            t["SQ_WAVE_CYCLES"] / t["SQ_WAVES"], 100 * t.get("SQ_WAIT_ANY", 0) / t["SQ_WAVE_CYCLES"], 100 * t.get("SQ_WAIT_INST_ANY", 0) / t["SQ_WAVE_CYCLES"])
    if "SQ_LDS_BANK_CONFLICT" in t and "SQ_LDS_IDX_ACTIVE" in t:
        line += "; LDS bank conflict cycles %.1f %% of LDS active" % (100 * t["SQ_LDS_BANK_CONFLICT"] / max(t["SQ_LDS_IDX_ACTIVE"], 1))
    if "SQ_WAIT_INST_LDS" in t and "SQ_WAVE_CYCLES" in t:
        line += "; waiting for LDS %.1f %%" % (100 * t["SQ_WAIT_INST_LDS"] / t["SQ_WAVE_CYCLES"])
    lines.append(line)
    lines.append("         raw: " + ", ".join("%s %.4g" % (c, v) for c, v in sorted(t.items())))
open(os.path.join(dst, "zstd_gpu_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
