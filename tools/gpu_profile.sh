#!/bin/bash
# bench + rocprofv3 kernel trace (+ separate PMC passes) of the same bench command.
# Run as the LAST GPU action of a round so profiles/ names the shipped kernel:
#   gpurun -- 'bash tools/gpu_profile.sh'   then   python tools/summarize_profile.py gpurun_out profiles/rNN
set -x
mkdir -p gpurun_out
rm -rf gpurun_out/prof_trace gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write gpurun_out/prof_pmc_sq
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/bench.log
# what the driver runs at round end
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_args.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_driver_args.log
tail -3 gpurun_out/bench_driver_args.log
# kernel trace + stats of the SAME command (default arguments)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py > gpurun_out/prof_trace.log 2>&1
echo "trace rc=$?"
find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1 | xargs cat | head -20
# PMC passes, each on its own (no trace domains combined with --pmc)
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --parity off --probe-reps 0 > gpurun_out/prof_pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_pmc_write -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --parity off --probe-reps 0 > gpurun_out/prof_pmc_write.log 2>&1
echo "pmc write rc=$?"
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_pmc_sq -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --parity off --probe-reps 0 > gpurun_out/prof_pmc_sq.log 2>&1
echo "pmc sq rc=$?"
# the trace CSV is large (one row per launch): keep only what the summariser reads
python3 tools/summarize_profile.py gpurun_out gpurun_out/summary > gpurun_out/summary.log 2>&1
find gpurun_out/prof_trace -name "*kernel_trace.csv" -size +8M -delete
ls -R gpurun_out | head -60
du -sh gpurun_out
