#!/bin/bash
# first-contact GPU run: parity tests, smoke, bench, rocprof kernel trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/host.txt; lscpu | head -20 >> gpurun_out/host.txt; free -g >> gpurun_out/host.txt
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8 >> gpurun_out/host.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -5 gpurun_out/bench.log
