#!/usr/bin/env python3
"""Where a small launch's time goes: reads a rocprofv3 --kernel-trace CSV and prints, per kernel
name, the median duration and the median gap to the previous kernel on the same queue.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/anat -- python3 tools/size_sweep.py --sizes 4194304 --kinds 0
  python3 tools/launch_anatomy.py gpurun_out/anat
"""
import csv
import glob
import os
import statistics
import sys


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    for f in sorted(files):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        per = {}
        prev_end = None
        for r in rows:
            name = r["Kernel_Name"].split("(")[0][-60:]
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "?")))
            d = per.setdefault(key, {"dur": [], "gap": []})
            d["dur"].append(e - s)
            if prev_end is not None:
                d["gap"].append(s - prev_end)
            prev_end = e
        print(f)
        for (name, grid), d in sorted(per.items(), key=lambda kv: -len(kv[1]["dur"])):
            if len(d["dur"]) < 5:
                continue
            gaps = [g for g in d["gap"] if g < 100000]  # ignore host-side pauses
            print("  %-60s grid %-8s n=%5d  dur median %7.2f us (p10 %7.2f)   gap-before median %6.2f us" % (
                name, grid, len(d["dur"]), statistics.median(d["dur"]) / 1e3, sorted(d["dur"])[len(d["dur"]) // 10] / 1e3,
                statistics.median(gaps) / 1e3 if gaps else float("nan")))


if __name__ == "__main__":
    main()
