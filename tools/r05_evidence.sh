#!/bin/bash
# Round 5's measured evidence in one GPU call (everything lands in gpurun_out/r05/, the summaries are copied to profiles/r05):
#   gpurun -- 'bash tools/r05_evidence.sh [part ...]'      parts: sweep soak cold fuzz zstd stream (default: all)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05
mkdir -p $O
PARTS="${*:-sweep soak cold fuzz zstd stream}"
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has sweep; then
    # file mode against image mode on one box, every size; the README-size file as a file
    timeout 900 python3 tests/perf/lz4_decoder_sweep.py --modes fast:2,hc:9 --sizes "2**27,2**28,2**29,824541892,2**30,2**31,2**32" --file-flags 824541892 --reps 5 2>&1 | grep -v amdgpu.ids > $O/lz4_decoder_sweep.log
    timeout 900 python3 tests/perf/lz4_decoder_sweep.py --modes zstd:1,zstd:3,zstd:19 --sizes "2**26,2**27,2**28,824541892,2**30,2**31,2**32" --file-flags 824541892 --reps 4 2>&1 | grep -v amdgpu.ids > $O/zstd_decoder_sweep.log
    timeout 600 python3 tests/perf/lz4_decoder_sweep.py --modes fast:2,hc:9,zstd:1 --sizes "2**31" --file-flags "2**31" --reps 5 2>&1 | grep -v amdgpu.ids > $O/decoder_sweep_file_mode_4GiB.log
fi
if has soak; then
    # item 5: every round with its stages and the cgroup's throttle counters
    timeout 600 python3 tests/perf/soak_lz4_gpu.py --mode fast:2 --rounds 21 2>&1 | grep -v amdgpu.ids > $O/lz4_gpu_soak.log
    timeout 600 python3 tests/perf/soak_lz4_gpu.py --mode zstd:1 --rounds 21 2>&1 | grep -v amdgpu.ids > $O/zstd_gpu_soak.log
    timeout 600 python3 tests/perf/soak_lz4_gpu.py --mode hc:9 --rounds 21 --host-every 0 2>&1 | grep -v amdgpu.ids > $O/lz4hc9_gpu_soak_gpu_rounds_only.log
fi
if has cold; then
    # item 2: the one-shot table (fresh process per sample), page cache and evicted
    timeout 900 python3 tests/perf/cold_start.py --samples 7 --gap-s 2 2>&1 | grep -v amdgpu.ids > $O/cold_start.log
    timeout 900 python3 tests/perf/cold_start.py --samples 4 --gap-s 2 --evict --which hc9,fast,zstd,raw 2>&1 | grep -v amdgpu.ids > $O/cold_start_evicted.log
fi
if has fuzz; then
    # item 1: both fuzzers compare BYTES now
    timeout 900 python3 tests/perf/fuzz_lz4_gpu.py --seeds 1500 2>&1 | grep -v amdgpu.ids | grep -v "^seed" > $O/lz4_gpu_fuzz.log
    timeout 1100 python3 tests/perf/fuzz_zstd_gpu.py --seeds 15000 2>&1 | grep -v amdgpu.ids | grep -v "^seeds" > $O/zstd_gpu_fuzz.log
fi
if has zstd; then
    # item 7: the kernels after the whole-wave table build; PMC passes re-taken (prepare's LDS bank conflicts)
    timeout 300 python3 tests/perf/zstd_kernel_check.py --levels 1,3,9,19,-5 --prof 1 --many 512 2>&1 | grep -v amdgpu.ids > $O/zstd_kernel_check.log
    timeout 300 python3 tests/perf/zstd_kernel_check.py --levels 1 --only na12878_1024 --many 2048 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/zstd_kernel_check.log
    FLAGSTATS_HIP_GPU_LZ4_PROFILE=1 timeout 300 python3 tests/perf/trace_lz4_gpu.py 2147483648 zstd:1 2>&1 | grep "profile\|pass" > $O/zstd_gpu_phases.log
    rm -rf gpurun_out/zstd_stats
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/zstd_stats -- python3 tests/perf/trace_lz4_gpu.py 2147483648 zstd:1 > $O/zstd_stats.log 2>&1
    find gpurun_out/zstd_stats -name "*kernel_stats.csv" -exec cp {} $O/zstd_kernel_stats.csv \;
    rm -rf gpurun_out/zstd_stats
    bash tools/zstd_pmc.sh > $O/zstd_pmc_run.log 2>&1
    cp gpurun_out/summary_extra/zstd_gpu_pmc.txt $O/ 2>/dev/null
fi
if has stream; then
    # item 5: BASELINE config 2 on HEAD
    timeout 600 python3 tools/bench_host_stream.py 2>&1 | grep -v amdgpu.ids > $O/host_stream_pinned_8GiB.log
fi
tail -4 $O/*.log | cut -c1-400 | head -150
