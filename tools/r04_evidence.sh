#!/bin/bash
# The round's measured evidence in one GPU call (everything lands in gpurun_out/, the summaries are copied to profiles/r04):
#   bash tools/r04_evidence.sh
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tools/ab_epilogue.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_two_level_epilogue.log
python3 tools/size_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/size_sweep.log
python3 tests/perf/bench_small_calls.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_calls.log
timeout 600 python3 tests/perf/soak_lz4_gpu.py --rounds 20 2>&1 | grep -v amdgpu.ids > gpurun_out/lz4_gpu_soak.log
timeout 600 python3 tests/perf/fuzz_lz4_gpu.py --seeds 1500 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/lz4_gpu_fuzz.log
timeout 600 python3 tests/perf/bench_gpu_lz4.py --modes fast:2,hc:9 --configs 8/1/1 --reps 2 2>&1 | grep -v amdgpu.ids > gpurun_out/lz4_whole_file_one_launch.log
FLAGSTATS_HIP_LZ4_GPU_KERNEL=1 timeout 600 python3 tests/perf/bench_gpu_lz4.py --modes fast:2,hc:9 --configs 8/4/4 --reps 2 2>&1 | grep -v amdgpu.ids > gpurun_out/lz4_r03_kernel_same_box.log
FLAGSTATS_HIP_GPU_LZ4_PROFILE=1 timeout 600 python3 tests/perf/bench_gpu_lz4.py --modes fast:2,hc:9 --configs 8/1/1 --reps 1 2>&1 | grep "profile\|image" > gpurun_out/lz4_gpu_phases.log
timeout 600 python3 tests/perf/host_pipeline_numa_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/host_pipeline_numa_ab.log
bash tools/profile_lz4_timeline.sh 1073741824 fast:2 > /dev/null 2>&1
bash tools/gpu_profile_extra.sh r04 > gpurun_out/gpu_profile_extra.log 2>&1
tail -5 gpurun_out/*.log | head -150
