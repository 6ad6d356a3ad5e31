#!/bin/bash
# Round 6's measured evidence (everything lands in gpurun_out/r06/, summaries are copied to profiles/r06):
#   gpurun -- 'bash tools/r06_evidence.sh [part ...]'      parts: tests tuning bench cold small decoders smallcalls soak dist distprof
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06
mkdir -p $O
PARTS="${*:-dist distprof}"
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has tests; then
    # the driver's round-end command
    timeout 1100 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
    tail -3 $O/gpu_tests.log
fi
if has cold; then
    # the one-shot table with the engine's second stream left to its helper thread (default) and waited for (r05)
    timeout 900 python3 tests/perf/cold_start.py --samples 5 --gap-s 2 2>&1 | grep -v amdgpu.ids > $O/cold_start.log || exit 1
    FLAGSTATS_HIP_EAGER_SECOND=1 timeout 900 python3 tests/perf/cold_start.py --samples 5 --gap-s 2 --which u16,hc9 2>&1 | grep -v amdgpu.ids > $O/cold_start_eager_second_stream.log || exit 1
fi
if has small; then
    # the worker pool kept with the engine against threads made and joined per call (r05), same box, alternating
    for rep in 1 2; do
        timeout 600 python3 tests/perf/small_file_phases.py 2>&1 | grep -v amdgpu.ids > $O/small_file_phases_pool_$rep.log || exit 1
        FLAGSTATS_HIP_POOL=0 timeout 600 python3 tests/perf/small_file_phases.py 2>&1 | grep -v amdgpu.ids > $O/small_file_phases_threads_per_call_$rep.log || exit 1
    done
fi
if has tuning; then
    # the measurement build: its own tests (tests/tuning) and every schedule through the parity / fuzz tests that take the library from FLAGSTATS_HIP_LIB
    FLAGSTATS_TUNING_TESTS=1 FLAGSTATS_HIP_LIB=$PWD/libflagstats_amd/libflagstats_hip_tuning.so timeout 900 python3 -m pytest tests/tuning tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_epilogue.py -x -q -m "gpu or tuning" > $O/tuning_build_tests.log 2>&1 || { tail -30 $O/tuning_build_tests.log; exit 1; }
    tail -3 $O/tuning_build_tests.log
fi
if has bench; then
    timeout 900 python3 bench.py > $O/bench_default.log 2>&1 || { tail -20 $O/bench_default.log; exit 1; }
    timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.log 2>&1 || exit 1
    tail -1 $O/bench_default.log | cut -c1-600
fi
if has decoders; then
    # the block-file entries on the final tree (the kernels are r05's; the orchestration changed: pooled readers / decoders, the index
    # pass before the lock, the engine's helper joined first): host threads against GPU decode, image and file mode
    timeout 900 python3 tests/perf/lz4_decoder_sweep.py --modes fast:2,hc:9 --sizes "2**23,2**27,824541892,2**31" --file-flags 824541892 --reps 5 2>&1 | grep -v amdgpu.ids > $O/lz4_decoder_sweep.log || { tail -n 20 $O/lz4_decoder_sweep.log; exit 1; }
    timeout 900 python3 tests/perf/lz4_decoder_sweep.py --modes zstd:1 --sizes "2**26,2**27,824541892,2**31" --file-flags 824541892 --reps 4 2>&1 | grep -v amdgpu.ids > $O/zstd_decoder_sweep.log || { tail -n 20 $O/zstd_decoder_sweep.log; exit 1; }
    cat $O/lz4_decoder_sweep.log $O/zstd_decoder_sweep.log | cut -c1-230
fi
if has smallcalls; then
    # the drop-in entry per call on HEAD (every entry now starts with the fork guard's getpid())
    timeout 600 python3 tests/perf/bench_small_calls.py 2>&1 | grep -v amdgpu.ids > $O/small_calls.log || { tail -n 20 $O/small_calls.log; exit 1; }
    cat $O/small_calls.log | cut -c1-200
fi
if has soak; then
    # the engine's helper thread, the worker pool and the index-before-lock change under concurrency: fresh processes creating
    # engines and counting at once, four caller threads mixing every entry family, the host paths from two threads
    timeout 600 python3 tests/perf/soak_concurrency.py 60 2>&1 | grep -v amdgpu.ids > $O/soak_concurrency.log || { tail -20 $O/soak_concurrency.log; exit 1; }
    timeout 600 python3 tests/perf/stress_mixed.py --seconds 60 2>&1 | grep -v amdgpu.ids > $O/stress_mixed.log || { tail -20 $O/stress_mixed.log; exit 1; }
    timeout 900 python3 tests/perf/soak_host_paths.py --rounds 300 2>&1 | grep -v amdgpu.ids > $O/soak_host_paths.log || { tail -20 $O/soak_host_paths.log; exit 1; }
    tail -3 $O/soak_concurrency.log $O/stress_mixed.log $O/soak_host_paths.log | cut -c1-300
fi
if has dist; then
    # the N > 1 step on HEAD at world size 1, every form, 8 GiB and 1 GiB shards
    timeout 1100 python3 tools/dist_step.py --sizes "2**32,2**29" --steps 200 --warmup 20 --repeat 2 2>&1 | grep -v amdgpu.ids > $O/dist_step_world1.log
fi
if has distprof; then
    # K2's and the RCCL kernel's own durations inside a step: kernel trace + stats, the rank started directly (RANK in the
    # environment: bench.py spawns nothing, the program itself sits behind `--`)
    export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
    i=0
    for size in 4294967296 536870912; do
        for form in "--no-overlap" "--overlap"; do
            i=$((i + 1))
            tag="$(echo $form | tr -d '-')_$size"
            rm -rf gpurun_out/dist_trace
            MASTER_PORT=$((29600 + i)) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dist_trace -- \
                python3 bench.py --force-dist $form --flags-per-gpu $size --steps 200 --warmup 20 --cpu-seconds 0 --parity off > $O/dist_trace_$tag.log 2>&1
            echo "trace $tag rc=$?"
            find gpurun_out/dist_trace -name "*kernel_stats.csv" -exec cp {} $O/dist_step_kernel_stats_$tag.csv \;
            python3 tools/dist_trace_gaps.py gpurun_out/dist_trace > $O/dist_step_timeline_$tag.txt 2>&1
            rm -rf gpurun_out/dist_trace
        done
    done
    unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR
fi
for f in $O/cold_start.log $O/small_file_phases.log $O/dist_step_world1.log; do if [ -f $f ]; then tail -30 $f | cut -c1-330; fi; done
