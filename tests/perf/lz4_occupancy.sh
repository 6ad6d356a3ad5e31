cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in fast:2 hc:9; do
for nb in 256 512 1024; do
  d=/tmp/lzocc_$nb; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tests/perf/zstd_pieces_sweep.py --mode $mode --flags "$nb*512000" --pieces 1 --reps 3 > /tmp/lzocc.txt 2>&1 || { tail -5 /tmp/lzocc.txt; exit 1; }
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $mode $nb blocks, one launch"; grep pieces /tmp/lzocc.txt
  grep "lz4_decode" $f | awk -F'","' '{printf "   lz4_decode_wg calls %s avg %.3f ms min %.3f ms\n", $2, $4/1e6, $6/1e6}'
done; done
