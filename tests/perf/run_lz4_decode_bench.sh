#!/bin/bash
# Host-only: builds block files of NA12878-like flags with the image's liblz4 and times the product's
# decoder (pair loop on / off) against liblz4, one thread.  Usage: bash tests/perf/run_lz4_decode_bench.sh [outdir]
set -e
OUT=${1:-/tmp}
HERE=$(cd "$(dirname "$0")/../.." && pwd)
cd "$HERE"
python3 - <<PY
import sys, struct
sys.path.insert(0, "$HERE"); sys.path.insert(0, "$HERE/tools")
import oracle, blockfile_tool as bt
per = bt.BLOCK_BYTES // 2
for mode, level, name in (("fast", 1, "fast1"), ("fast", 2, "fast2"), ("hc", 1, "hc1"), ("hc", 9, "hc9")):
    with open("/tmp/na_%s.lz4" % name, "wb") as f:
        for i in range(32):
            raw = oracle.generate(oracle.GEN_NA12878, 7, 1, i * per, per).tobytes()
            c = bt.compress_block(raw, mode, level)
            f.write(struct.pack("<ii", len(raw), len(c)) + c)
with open("/tmp/u12_fast2.lz4", "wb") as f:
    for i in range(8):
        raw = oracle.generate(oracle.GEN_UNIFORM, 7, 0x0FFF, i * per, per).tobytes()
        c = bt.compress_block(raw, "fast", 2)
        f.write(struct.pack("<ii", len(raw), len(c)) + c)
PY
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -O3 -std=c++17 -o /tmp/lz4_bench_pairs tests/perf/lz4_decode_bench.cpp -ldl
$CXX -O3 -std=c++17 -DFSLZ4_NO_PAIR_LOOP -o /tmp/lz4_bench_nopairs tests/perf/lz4_decode_bench.cpp -ldl
$CXX -O3 -std=c++17 -DFSLZ4_TAIL8 -o /tmp/lz4_bench_bwide tests/perf/lz4_decode_bench.cpp -ldl
for rep in 1 2; do
  for f in na_fast1 na_fast2 na_hc1 na_hc9 u12_fast2; do
    echo -n "pair-loop    "; /tmp/lz4_bench_pairs /tmp/$f.lz4 30
    echo -n "no pair-loop "; /tmp/lz4_bench_nopairs /tmp/$f.lz4 30
    echo -n "8-byte tail  "; /tmp/lz4_bench_bwide /tmp/$f.lz4 30
  done
done
