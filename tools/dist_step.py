#!/usr/bin/env python3
"""The N > 1 step on ONE GPU (world size 1): what K2 + the all-reduce add to a step, in every form bench.py offers.

    python3 tools/dist_step.py [--sizes 2**32,2**29] [--steps 200] [--warmup 20] [--repeat 2]

Every line is a fresh `bench.py` process (this parent never touches the GPU).  Columns: wall ms per step, event-timed ms
per step, median step, the delta of the event time over the N = 1 line of the same size in us (the cost of store form + K2 +
all-reduce), `allreduce_us` (10 all-reduces of a scratch buffer between stream events, untimed part of bench.py), the form
and who issued the collective.  VERDICT r05 item 1; the budget these numbers are read against is in DESIGN.md "Multi-GPU".
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FORMS = [
    ("N=1 (K1 atomic epilogue, no collective)", []),
    ("in-line, C ABI", ["--force-dist", "--no-overlap"]),
    ("overlapped, C ABI", ["--force-dist", "--overlap"]),
    ("overlapped, C ABI, fence-free events", ["--force-dist", "--overlap", "ENV:FLAGSTATS_HIP_FENCE_FREE_EVENTS=1"]),
    ("in-line, torch", ["--force-dist", "--no-overlap", "--allreduce", "torch"]),
    ("overlapped, torch", ["--force-dist", "--overlap", "--allreduce", "torch"]),
    ("--calibrate", ["--force-dist", "--calibrate"]),
    ("--strong (calibrates by default)", ["--force-dist", "--strong"]),
]


def run(size, steps, warmup, extra):
    env = dict(os.environ)
    for x in extra:
        if x.startswith("ENV:"):
            k, v = x[4:].split("=", 1)
            env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--flags-per-gpu", str(size), "--steps", str(steps), "--warmup",
           str(warmup), "--cpu-seconds", "0"] + [x for x in extra if not x.startswith("ENV:")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        return None, (r.stdout + r.stderr)[-1500:]
    return json.loads(lines[0]), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2**32,2**29")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    for size in [int(eval(x)) for x in a.sizes.split(",")]:   # noqa: S307 -- "2**32" from our own command line
        print("# %d flags per GPU (%.3g GiB), --steps %d --warmup %d, world size 1" % (size, size * 2 / 2 ** 30, a.steps, a.warmup))
        print("%-40s %10s %10s %10s %9s %12s  %s" % ("form", "wall ms", "event ms", "median ms", "d(us)", "allreduce_us", "ran as / parity"))
        base = None
        for rep in range(a.repeat):
            for name, extra in FORMS:
                d, err = run(size, a.steps, a.warmup, extra)
                if d is None:
                    print("%-40s FAILED: %s" % (name, err.replace("\n", " | ")[-400:]))
                    continue
                ev = d["roofline"]["event_ms_per_launch"]
                if not extra:
                    base = ev
                delta = "%9.1f" % ((ev - base) * 1e3) if base is not None and extra else "%9s" % "-"
                c = d["config"]
                print("%-40s %10.5f %10.5f %10.5f %s %12s  %s | %s | %s" % (
                    name, d["ms_per_step"], ev, d["roofline"]["step_ms"]["median"], delta, c.get("allreduce_us"),
                    c.get("allreduce"), (c.get("allreduce_impl") or "")[:40], d["parity"][:9]), flush=True)
        print()


if __name__ == "__main__":
    main()
