#!/usr/bin/env python3
"""Soak of the concurrency paths (after the NULL-stream memset race of round 2): fresh processes creating
fresh engines and counting immediately (tests/consumer_multi.c: one default engine, two multi engines, two
explicit contexts on two threads), many times, each run checked against the oracle; then the in-process
multi / session tests in a loop.  Prints one line per phase."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from libflagstats_amd import _lib  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tmp = tempfile.mkdtemp(prefix="soak_")
exe = os.path.join(tmp, "consumer_multi")
libdir = os.path.join(ROOT, "libflagstats_amd")
subprocess.run(["gcc", "-O1", "-std=c11", "-pthread", "-I", os.path.join(ROOT, "include"),
                os.path.join(ROOT, "tests", "consumer_multi.c"), "-L", libdir, "-lflagstats_hip",
                "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True)
bad = 0
for r in range(rounds):
    n = [12_345_679, 1_000_003, 80_000_001, 37][r % 4]
    p = subprocess.run([exe, str(n), "2", "0", "0"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    a = ((np.arange(n, dtype=np.uint64).astype(np.uint32) * np.uint32(2654435761)) >> np.uint32(13)).astype(np.uint16)
    want = oracle.flagstat_hist(a)
    for ln in p.stdout.splitlines():
        tag, vals = ln.split()[0], np.array([int(v) for v in ln.split()[1:]], dtype=np.uint64)
        if not np.array_equal(vals, want):
            bad += 1
            print("MISMATCH round", r, "n", n, tag)
print("fresh-process consumer runs: %d, mismatches: %d" % (rounds, bad), flush=True)

import test_gpu_multi as tm  # noqa: E402
import test_gpu_session as ts  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
for r in range(max(1, rounds // 4)):
    tm.test_multi_entries_from_python(lib)
    tm.test_contexts_run_concurrently(lib)
    ts.test_sessions_of_different_threads_overlap(lib)
    lib.FLAGSTATS_hip_shutdown()          # next round starts from fresh engines again
print("in-process multi / contexts / sessions loops: %d, all exact" % max(1, rounds // 4))
sys.exit(1 if bad else 0)
