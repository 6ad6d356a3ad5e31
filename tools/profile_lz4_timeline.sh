#!/bin/bash
# rocprofv3 kernel + memory-copy trace of ONE GPU-decoded LZ4 (or Zstandard: mode zstd:1) image: when does every piece land, when does its decode
# launch start and end.   gpurun -- 'bash tools/profile_lz4_timeline.sh 1073741824 fast:2'
set -x
N=${1:-1073741824}; MODE=${2:-fast:2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf gpurun_out/lz_timeline
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/lz_timeline -- python3 tests/perf/trace_lz4_gpu.py $N $MODE > gpurun_out/lz_timeline.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/lz_timeline.log
python3 - <<'PY' | tee gpurun_out/lz_timeline.txt
import csv, glob, os
root = "gpurun_out/lz_timeline"
def newest(pat):
    h = glob.glob(os.path.join(root, "**", pat), recursive=True)
    return max(h, key=os.path.getmtime) if h else None
ev = []
mc, kt = newest("*memory_copy_trace.csv"), newest("*kernel_trace.csv")
for r in csv.DictReader(open(mc)):
    if "HOST_TO_DEVICE" in r.get("Direction", "") and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", 0))
for r in csv.DictReader(open(kt)):
    if "lz4_decode" in r["Kernel_Name"]:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "decode %4d blocks" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])), int(r.get("Queue_Id", 0) or 0)))
    elif "zstd_" in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("zstd_")[1].split("<")[0].split("I")[0]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "%-8s %4d wg" % (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])), int(r.get("Queue_Id", 0) or 0)))
    elif "flagstat_count" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50000:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K1", int(r.get("Queue_Id", 0) or 0)))
ev.sort()
# the LAST pass of the run: events after the last gap > 20 ms
start = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[:i]) > 20_000_000:
        start = i
run = ev[start:]
t0 = run[0][0]
print("# one pass (the last of the run), times in ms from its first copy; queue = the hardware queue the launch ran on")
for s, e, what, q in run:
    print("%8.2f .. %8.2f  (%6.2f)  %-20s queue %d" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, what, q))
PY
