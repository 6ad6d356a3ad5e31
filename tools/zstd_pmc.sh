#!/bin/bash
# rocprofv3 PMC passes over the GPU Zstandard decoder's kernels (counters only: no trace domains beside them)
#   gpurun -- 'bash tools/zstd_pmc.sh'
set -x
mkdir -p gpurun_out
rm -rf gpurun_out/zs_pmc_sq gpurun_out/zs_pmc_sq2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/zs_pmc_sq -- python3 tests/perf/trace_lz4_gpu.py 2147483648 zstd:1 > gpurun_out/zs_pmc_sq.log 2>&1
echo "pmc rc=$?"
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/zs_pmc_sq2 -- python3 tests/perf/trace_lz4_gpu.py 2147483648 zstd:1 > gpurun_out/zs_pmc_sq2.log 2>&1
echo "pmc2 rc=$?"
python3 tools/summarize_zstd_pmc.py gpurun_out gpurun_out/summary_extra
rm -rf gpurun_out/zs_pmc_sq gpurun_out/zs_pmc_sq2
