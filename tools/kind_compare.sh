#!/bin/bash
# NA12878-like vs uniform input through the same branch-free kernel: interleaved A/B timing with clock samples,
# then the same launches under rocprofv3 counters, one counter set per pass and per input kind
# (no trace domain is ever combined with --pmc).  Summarise with tools/kind_compare_summary.py.
#   gpurun -- 'bash tools/kind_compare.sh'
set -x
OUT=gpurun_out/kind
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 tools/kind_compare.py --flags 4294967296 > $OUT/ab_8GiB.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/kind_compare.py --flags 536870912 --reps 100 > $OUT/ab_1GiB.log 2>&1 || exit 1
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_BUSY_sum" \
           "SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_COUNT"; do
  i=$((i+1))
  for kind in 0 1; do
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_k${kind}_s$i -- python3 tools/kind_compare.py --single $kind --launches 30 > $OUT/pmc_k${kind}_s$i.log 2>&1
    echo "kind $kind set $i rc=$?"
  done
done
python3 tools/kind_compare_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/ab_8GiB.log $OUT/ab_1GiB.log $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT
