set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for n in 1000000 4194304 67108864 536870912; do
  rm -rf gpurun_out/anat_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/anat_$n -- python3 tools/size_sweep.py --sizes $n --kinds 0 > gpurun_out/anat_$n.log 2>&1 || exit 1
  python3 tools/launch_anatomy.py gpurun_out/anat_$n > gpurun_out/anat_$n.txt 2>&1
  cat gpurun_out/anat_$n.txt
  find gpurun_out/anat_$n -name "*.csv" -size +2M -delete
done
