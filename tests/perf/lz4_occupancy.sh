#!/bin/bash
# Does a CU hold two workgroups of the LZ4 decode kernel (eight waves, 80 KB of LDS)?  N blocks in ONE launch through the product
# entry (pieces = 1), N = 256 / 512 / 1024 on 256 CUs, under rocprofv3's kernel trace: with two per CU 512 blocks take little
# more than 256.   usage (GPU box): bash tests/perf/lz4_occupancy.sh
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for mode in fast:2 hc:9; do
    for nb in 256 512 1024; do
        d=/tmp/lzocc_$nb
        rm -rf "$d"
        rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 tests/perf/zstd_pieces_sweep.py --mode "$mode" --flags "$nb*512000" --pieces 1 --reps 3 > /tmp/lzocc.txt 2>&1 || { tail -5 /tmp/lzocc.txt; exit 1; }
        echo "== $mode $nb blocks, one launch"
        grep "pieces" /tmp/lzocc.txt | grep -v simple_timer
        python3 - "$(find "$d" -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lz4_decode" in r["Name"]:
        print("   lz4_decode_wg  calls %3s  avg %8.3f ms  min %8.3f ms" % (r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6))
PY
    done
done
