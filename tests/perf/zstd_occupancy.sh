#!/bin/bash
# How many zstd_execute workgroups does a CU really hold?  Per-kernel durations (rocprofv3 kernel trace) of N frames in ONE
# launch of each kernel, N = 256 / 512 / 768 / 1024 on 256 CUs: with two workgroups per CU, 512 frames take what 256 take.
# ZOCC_LIBS: libraries to compare (measurement builds with another split of the roles), default the shipped one.
# usage (GPU box): bash tests/perf/zstd_occupancy.sh [outfile]
OUT=${1:-gpurun_out/zstd_occupancy.log}
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for lib in ${ZOCC_LIBS:-libflagstats_amd/libflagstats_hip.so}; do
    export FLAGSTATS_HIP_LIB=$PWD/$lib
    echo "#### $lib" >> "$OUT"
    for n in ${ZOCC_SIZES:-256 512 768 1024}; do
        d=/tmp/zocc_$n
        rm -rf "$d"
        rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tests/perf/zstd_kernel_check.py --only NONE --many "$n" > /tmp/zocc_$n.txt 2>&1 || { tail -5 /tmp/zocc_$n.txt; exit 1; }
        f=$(find "$d" -name '*kernel_stats.csv' | head -1)
        echo "== $n frames in one launch of each kernel (3 launches)" >> "$OUT"
        grep "frames of" /tmp/zocc_$n.txt >> "$OUT"
        python3 - "$f" >> "$OUT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Name"]
    if "zstd_" in nm or "lz4_" in nm:
        print("   %-14s calls %3s  avg %8.3f ms  min %8.3f ms" % (nm.split("fsk::")[1].split("<")[0], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6))
PY
    done
done
cat "$OUT"
