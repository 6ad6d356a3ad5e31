#!/usr/bin/env python3
"""One N > 1 step as the GPU ran it: from a rocprofv3 --kernel-trace of `bench.py --force-dist`, the K1 -> K2 -> RCCL
all-reduce kernel sequence of the steady-state steps -- each kernel's own duration, the gaps between them, and the
step period (K1 start to next K1 start).  Medians over the last 150 full-size K1 launches that have a K2 behind them.

    python3 tools/dist_trace_gaps.py <rocprofv3 output dir>
"""
import csv
import glob
import os
import statistics
import sys


def main():
    hits = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    if not hits:
        print("no kernel_trace.csv under", sys.argv[1])
        return 1
    rows = []
    for r in csv.DictReader(open(max(hits, key=os.path.getmtime))):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    k1 = [i for i, r in enumerate(rows) if "flagstat_count" in r[2]]
    if not k1:
        print("no flagstat_count launches in the trace")
        return 1
    longest = max(rows[i][1] - rows[i][0] for i in k1)
    k1 = [i for i in k1 if rows[i][1] - rows[i][0] > 0.5 * longest]
    # steps of the N > 1 form only: a K1 launch followed by its K2 (the N = 1 launches of the probes around the timed region
    # -- warm-up of the shader-clock reading, the read probe -- have no finalize behind them)
    k1 = [i for i in k1 if i + 1 < len(rows) and "finalize" in rows[i + 1][2]][-151:]
    names = {}
    per_step = []
    for a, b in zip(k1[:-1], k1[1:]):
        s0, e0 = rows[a][0], rows[a][1]
        step = {"period": rows[b][0] - s0, "K1": e0 - s0}
        others = [rows[i] for i in range(a + 1, b)]
        step["n_other"] = len(others)
        prev_end = e0
        for j, (s, e, name, q, st) in enumerate(others):
            short = "finalize" if "finalize" in name else ("rccl:" + name.split("(")[0][:60] if ("nccl" in name.lower() or "rccl" in name.lower()) else name.split("(")[0][:60])
            names[short] = names.get(short, 0) + 1
            step.setdefault("dur:" + short, 0)
            step["dur:" + short] += e - s
            step.setdefault("gap_before:" + short, s - prev_end)
            step.setdefault("start_after_K1_end:" + short, s - e0)
            step.setdefault("queue:" + short, q + "/" + st)
            prev_end = max(prev_end, e)
        step["K1_end_to_next_K1_start"] = rows[b][0] - e0
        per_step.append(step)
    keys = sorted({k for s in per_step for k in s if not k.startswith("queue:")})
    print("steady-state steps: %d (full-size K1 launches, the last of the trace); all times in us, median [p10 .. p90]" % len(per_step))
    for k in keys:
        v = sorted(s[k] for s in per_step if k in s)
        if k == "n_other":
            print("%-75s %s" % ("kernels between two K1 launches", statistics.median(v)))
            continue
        q = lambda f: v[min(len(v) - 1, int(f * (len(v) - 1)))] / 1e3   # noqa: E731
        print("%-75s %9.2f [%9.2f .. %9.2f]  (%d steps)" % (k, q(0.5), q(0.1), q(0.9), len(v)))
    for k in sorted({k for s in per_step for k in s if k.startswith("queue:")}):
        print("%-75s %s" % (k + " (queue id / stream id)", per_step[-1].get(k)))
    print("kernel names seen between K1 launches:", names)
    return 0


if __name__ == "__main__":
    sys.exit(main())
