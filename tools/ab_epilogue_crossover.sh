#!/bin/bash
# Where does the one-level epilogue start to win?  one-level (0) vs two-level (2^40) vs limits between, 192..640 MiB.
set -e
cd "$(dirname "$0")/.."
for mib in 192 256 320 384 448 512 640; do
    flags=$((mib * 524288))
    echo "== ${mib} MiB ($((mib / 8)) steps per workgroup on 256 CUs)"
    python3 tools/knob_ab.py group_max_steps 0 1099511627776 --flags $flags --rounds 16 --reps 50
done
