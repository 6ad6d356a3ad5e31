#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the block-file pipeline (row f1): are the H2D copies back to back, and
# how long does K1 take per chunk?   gpurun -- 'bash tools/profile_blockfile.sh'
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf gpurun_out/prof_blocks
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/prof_blocks -- \
    python3 tests/perf/bench_blockfile.py --flags 2147483648 --threads 0 --chunk-mib 64 --no-serial > gpurun_out/prof_blocks.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import csv, glob, os
root = "gpurun_out/prof_blocks"
def newest(pat):
    h = glob.glob(os.path.join(root, "**", pat), recursive=True)
    return max(h, key=os.path.getmtime) if h else None
mc = newest("*memory_copy_trace.csv")
kt = newest("*kernel_trace.csv")
out = []
if mc:
    rows = [r for r in csv.DictReader(open(mc))]
    big = [r for r in rows if "HOST_TO_DEVICE" in r.get("Direction", "") and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000]
    big.sort(key=lambda r: int(r["Start_Timestamp"]))
    if big:
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in big]
        # the bench runs the file 3 times: split at gaps > 5 ms
        runs, cur = [], [big[0]]
        for a, b in zip(big, big[1:]):
            if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 5_000_000:
                runs.append(cur); cur = []
            cur.append(b)
        runs.append(cur)
        out.append("H2D chunk copies: %d, mean %.3f ms each" % (len(big), sum(dur) / len(dur) / 1e6))
        for i, run in enumerate(runs):
            span = int(run[-1]["End_Timestamp"]) - int(run[0]["Start_Timestamp"])
            busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in run)
            out.append("  pass %d: %d copies over %.2f ms, copy engine busy %.1f %% of that span" % (i, len(run), span / 1e6, 100.0 * busy / span))
if kt:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "flagstat_count" in r["Kernel_Name"]]
    if d:
        d.sort()
        out.append("K1 launches: %d, median %.1f us, max %.1f us (64 MiB chunks)" % (len(d), d[len(d) // 2] / 1e3, d[-1] / 1e3))
open("gpurun_out/prof_blocks_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find gpurun_out/prof_blocks -name "*.csv" -size +4M -delete
