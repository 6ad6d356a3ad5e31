#!/bin/bash
# Round 6's measured evidence (everything lands in gpurun_out/r06/, summaries are copied to profiles/r06):
#   gpurun -- 'bash tools/r06_evidence.sh [part ...]'      parts: dist distprof
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06
mkdir -p $O
PARTS="${*:-dist distprof}"
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
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
tail -40 $O/dist_step_world1.log | cut -c1-300
head -12 $O/dist_step_kernel_stats_*.csv | cut -c1-300
