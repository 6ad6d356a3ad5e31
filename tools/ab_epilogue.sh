#!/bin/bash
# Interleaved A/B of K1's epilogue forms at eight array sizes:
#   group_max_steps = 0              one-level: every workgroup adds straight to out[32]
#   group_max_steps = 40             the shipped rule: per-XCD copies only for <= 40 steps per workgroup
#   group_max_steps = 1099511627776  two-level at every size (what r03 shipped)
# usage: tools/ab_epilogue.sh > profiles/rNN/ab_two_level_epilogue.log
set -e
cd "$(dirname "$0")/.."
for mib in 8 32 128 192 256 512 1024 8192; do
    flags=$((mib * 524288))
    rounds=10; reps=50
    [ $mib -ge 1024 ] && reps=20
    echo "== ${mib} MiB"
    python3 tools/knob_ab.py group_max_steps 0 40 1099511627776 --flags $flags --rounds $rounds --reps $reps
done
