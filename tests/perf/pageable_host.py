#!/usr/bin/env python3
"""A host array in PAGEABLE memory (what numpy hands pyflagstats.flagstats, what a C caller's malloc holds) -> counters, two ways:
  runtime : FLAGSTATS_u16_x64 as shipped -- hipMemcpyAsync out of the caller's memory; the HIP runtime pins such memory as it goes
            (and remembers a buffer it has seen: repeats on ONE buffer are not what a caller with a new buffer per call sees);
  staged  : FLAGSTATS_hip_host_staged_u16 -- the block pipeline's workers copy 1 MiB slices into the engine's page-locked chunks,
            which go over PCIe behind them (what FLAGSTATS_hip_file_raw does with preads).
Every sample uses a NEW copy of the array unless --same."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="10**7,10**8,2**28,2**29")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--threads", default="0,8,16")
    args = ap.parse_args()
    import oracle
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    _lib.check(lib.FLAGSTATS_hip_set(b"staged_min_flags", 0), "set")   # ("runtime" below = hipMemcpyAsync out of the array, whatever its size)
    for n in [int(eval(x)) for x in args.sizes.split(",")]:
        base = oracle.generate(oracle.GEN_NA12878, 7, 1, 0, n)
        want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, n)

        def run(fn, fresh):
            ts = []
            keep = base
            held = []   # (the copies stay alive: a freed array's address comes back with the next one, and the runtime remembers addresses)
            for _ in range(args.reps):
                a = base.copy() if fresh else keep
                held.append(a)
                out = np.zeros(32, dtype=np.uint64)
                t0 = time.perf_counter()
                rc = fn(a, out)
                ts.append((time.perf_counter() - t0) * 1e3)
                assert rc == 0 and np.array_equal(out, want), (rc, out[:4], want[:4])
            return min(ts), sorted(ts)[len(ts) // 2]

        line = "%11d flags (%5.0f MiB): " % (n, n * 2 / 2**20)
        b, m = run(lambda a, out: lib.FLAGSTATS_u16_x64(a.ctypes.data, n, out.ctypes.data), False)
        line += "runtime, one buffer %7.2f ms (median %.2f) | " % (b, m)
        b, m = run(lambda a, out: lib.FLAGSTATS_u16_x64(a.ctypes.data, n, out.ctypes.data), True)
        line += "runtime, new buffer per call %7.2f (%.2f) = %.1f GB/s | " % (b, m, n * 2 / b / 1e6)
        for th in [int(x) for x in args.threads.split(",")]:
            b, m = run(lambda a, out: lib.FLAGSTATS_hip_host_staged_u16(a.ctypes.data, n, th, out.ctypes.data, None), True)
            line += "staged, %d threads, new buffer %7.2f (%.2f) = %.1f GB/s | " % (th, b, m, n * 2 / b / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
