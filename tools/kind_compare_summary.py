#!/usr/bin/env python3
"""Per-launch means of the rocprofv3 counters tools/kind_compare.sh collected for the two input kinds, side by side."""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    per = {}
    for d in sorted(glob.glob(os.path.join(root, "pmc_k*_s*"))):
        if not os.path.isdir(d):
            continue
        kind = int(os.path.basename(d).split("_")[1][1:])
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "flagstat_count" not in r["Kernel_Name"]:
                    continue
                per.setdefault(r["Counter_Name"], {}).setdefault(kind, []).append(float(r["Counter_Value"]))
    print("%-40s %18s %18s %9s" % ("counter (mean per K1 launch)", "uniform 0xFFFF", "NA12878-like", "NA/uni"))
    for name in sorted(per):
        a, b = per[name].get(0, []), per[name].get(1, [])
        ma = sum(a) / len(a) if a else float("nan")
        mb = sum(b) / len(b) if b else float("nan")
        print("%-40s %18.1f %18.1f %9.4f   (%d / %d launches)" % (name, ma, mb, mb / ma if ma else float("nan"), len(a), len(b)))
    for kind in (0, 1):
        for f in sorted(glob.glob(os.path.join(root, "pmc_k%d_s*.log" % kind))):
            for line in open(f):
                if "us per launch" in line:
                    print(os.path.basename(f), line.strip())


if __name__ == "__main__":
    main()
