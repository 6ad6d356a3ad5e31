#!/bin/bash
# The GPU Zstandard decoder's measured evidence in one GPU call (lands in gpurun_out/, summaries are copied to profiles/r04):
#   bash tools/r04_zstd_evidence.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python3 tests/perf/zstd_kernel_check.py --levels 1,3,9,19,-5 --prof 1 --many 512 2>&1 | grep -v amdgpu.ids > gpurun_out/zstd_kernel_check.log
timeout 300 python3 tests/perf/zstd_kernel_check.py --levels 1 --only na12878_1024 --many 2048 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/zstd_kernel_check.log
timeout 900 python3 tests/perf/fuzz_zstd_gpu.py --seeds 15000 2>&1 | grep -v amdgpu.ids | grep -v "^seeds" > gpurun_out/zstd_gpu_fuzz.log
timeout 900 python3 tests/perf/lz4_decoder_sweep.py --modes zstd:1,zstd:3,zstd:19 --sizes "2**26,2**27,2**28,824541892,2**30,2**31,2**32" --file-flags 824541892 2>&1 | grep -v amdgpu.ids > gpurun_out/zstd_decoder_sweep.log
FLAGSTATS_HIP_GPU_LZ4_PROFILE=1 timeout 300 python3 tests/perf/trace_lz4_gpu.py 2147483648 zstd:1 2>&1 | grep "profile\|pass" > gpurun_out/zstd_gpu_phases.log
bash tools/profile_lz4_timeline.sh 2147483648 zstd:1 > /dev/null 2>&1
cp gpurun_out/lz_timeline.txt gpurun_out/zstd_timeline.txt
rm -rf gpurun_out/zstd_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/zstd_stats -- python3 tests/perf/trace_lz4_gpu.py 2147483648 zstd:1 > gpurun_out/zstd_stats.log 2>&1
find gpurun_out/zstd_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/zstd_kernel_stats.csv \;
rm -rf gpurun_out/zstd_stats gpurun_out/lz_timeline
ZOCC_SIZES="256 512 768 1024 1536 2048" bash tests/perf/zstd_occupancy.sh gpurun_out/zstd_occupancy.log > /dev/null 2>&1
tail -4 gpurun_out/zstd_*.log | head -120; head -12 gpurun_out/zstd_kernel_stats.csv
