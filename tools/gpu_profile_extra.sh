#!/bin/bash
# rocprofv3 evidence beyond the headline kernel (tools/gpu_profile.sh), each as kernel trace + separate PMC passes:
#   row f4   fsk::pospopcnt_count on the 8 GiB array                      -> profiles/rNN/pospopcnt_*
#   row f1   the GPU LZ4 decode kernel on a 2^31-flag NA12878-like image  -> profiles/rNN/lz4_*
#   gpurun -- 'bash tools/gpu_profile_extra.sh r04'
set -x
R=${1:-r04}
mkdir -p gpurun_out
rm -rf gpurun_out/pp_trace gpurun_out/pp_pmc_fetch gpurun_out/pp_pmc_write gpurun_out/lz_trace gpurun_out/lz_pmc_sq gpurun_out/lz_pmc_sq2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 tests/perf/profile_pospopcnt.py > gpurun_out/pospopcnt_run.log 2>&1; tail -1 gpurun_out/pospopcnt_run.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp_trace -- python3 tests/perf/profile_pospopcnt.py > gpurun_out/pp_trace.log 2>&1
echo "pp trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pp_pmc_fetch -- python3 tests/perf/profile_pospopcnt.py 4294967296 6 > gpurun_out/pp_pmc_fetch.log 2>&1
echo "pp fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pp_pmc_write -- python3 tests/perf/profile_pospopcnt.py 4294967296 6 > gpurun_out/pp_pmc_write.log 2>&1
echo "pp write rc=$?"
python3 tools/summarize_profile.py gpurun_out gpurun_out/summary_extra --kernel pospopcnt_count --sub pp_ --name pospopcnt > gpurun_out/summary_pp.log 2>&1
tail -12 gpurun_out/summary_pp.log
# the LZ4 decode kernel: five passes over one image per run
for mode in fast:2 hc:9; do
    tag=${mode%%:*}
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lz_trace_$tag -- python3 tests/perf/trace_lz4_gpu.py 2147483648 $mode > gpurun_out/lz_trace_$tag.log 2>&1
    echo "lz trace $tag rc=$?"
    find gpurun_out/lz_trace_$tag -name "*kernel_stats.csv" | head -1 | xargs cat | head -8 > gpurun_out/summary_extra/lz4_kernel_stats_$tag.csv
    cat gpurun_out/summary_extra/lz4_kernel_stats_$tag.csv
    find gpurun_out/lz_trace_$tag -name "*kernel_trace.csv" -size +8M -delete
done
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/lz_pmc_sq -- python3 tests/perf/trace_lz4_gpu.py 2147483648 fast:2 > gpurun_out/lz_pmc_sq.log 2>&1
echo "lz pmc rc=$?"
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/lz_pmc_sq2 -- python3 tests/perf/trace_lz4_gpu.py 2147483648 fast:2 > gpurun_out/lz_pmc_sq2.log 2>&1
echo "lz pmc2 rc=$?"
python3 tools/summarize_lz4_pmc.py gpurun_out gpurun_out/summary_extra > gpurun_out/summary_lz.log 2>&1
cat gpurun_out/summary_lz.log
du -sh gpurun_out
