#!/bin/bash
# rocprofv3 evidence for the GPU LZ4 decode kernel alone (the LZ4 part of tools/gpu_profile_extra.sh): kernel trace + stats for
# LZ4-fast and LZ4-HC-9 images of 2^31 flags, then two separate PMC passes (counters only: no trace domains beside them)
#   gpurun -- 'bash tools/lz4_pmc.sh'      -> gpurun_out/summary_extra/lz4_kernel_stats_{fast,hc}.csv, lz4_gpu_pmc.txt
set -x
mkdir -p gpurun_out/summary_extra
rm -rf gpurun_out/lz_trace_fast gpurun_out/lz_trace_hc gpurun_out/lz_pmc_sq gpurun_out/lz_pmc_sq2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in fast:2 hc:9; do
    tag=${mode%%:*}
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lz_trace_$tag -- python3 tests/perf/trace_lz4_gpu.py 2147483648 $mode > gpurun_out/lz_trace_$tag.log 2>&1
    echo "lz trace $tag rc=$?"
    find gpurun_out/lz_trace_$tag -name "*kernel_stats.csv" | head -1 | xargs cat | head -8 > gpurun_out/summary_extra/lz4_kernel_stats_$tag.csv
    cat gpurun_out/summary_extra/lz4_kernel_stats_$tag.csv
    rm -rf gpurun_out/lz_trace_$tag
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/lz_pmc_sq -- python3 tests/perf/trace_lz4_gpu.py 2147483648 fast:2 > gpurun_out/lz_pmc_sq.log 2>&1
echo "lz pmc rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/lz_pmc_sq2 -- python3 tests/perf/trace_lz4_gpu.py 2147483648 fast:2 > gpurun_out/lz_pmc_sq2.log 2>&1
echo "lz pmc2 rc=$?"
python3 tools/summarize_lz4_pmc.py gpurun_out gpurun_out/summary_extra > gpurun_out/summary_lz.log 2>&1
cat gpurun_out/summary_lz.log
rm -rf gpurun_out/lz_pmc_sq gpurun_out/lz_pmc_sq2
