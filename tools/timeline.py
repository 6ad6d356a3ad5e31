#!/usr/bin/env python3
"""Workgroup timeline of one K1 launch (tuning build only:
FLAGSTATS_HIP_LIB=libflagstats_amd/libflagstats_hip_tuning.so).

Wave 0 of every workgroup stamps the 100 MHz wall clock at entry, after its first step, after its last
step, after the final flush, after the workgroup reduction and at exit (fsk_timeline_run).  Printed per
array size: when workgroups start and end relative to the earliest entry (min / median / max over the
grid), how long each phase takes, and the mean end time per XCD -- i.e. what part of a mid-size launch
is ramp, imbalance between workgroups, flush and epilogue."""
import argparse
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib, device  # noqa: E402


def stats(x):
    x = sorted(x)
    return "%7.2f %7.2f %7.2f %7.2f" % (x[0], statistics.median(x), x[int(0.9 * (len(x) - 1))], x[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4194304,67108864,536870912,4294967296")
    ap.add_argument("--kind", type=int, default=0)
    ap.add_argument("--bpc", default="1")
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    assert lib.FLAGSTATS_hip_get(b"tuning_build"), "needs the tuning build"
    variant = args.variant if args.variant is not None else int(lib.FLAGSTATS_hip_get(b"variant"))
    run = lib.fsk_timeline_run
    run.restype = ctypes.c_uint32
    run.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32]
    sizes = [int(s) for s in args.sizes.split(",")]
    d = device.DeviceFlags(max(sizes)).generate(args.kind, seed=5, mask=0xFFFF if args.kind == 0 else 1)
    cus = lib.FLAGSTATS_hip_compute_units()
    print("us relative to the earliest workgroup entry:  min  median  p90  max      (variant %d)" % variant)
    for bpc in [int(b) for b in args.bpc.split(",")]:
        grid = cus * bpc
        for n in sizes:
            for rep in range(args.reps):
                rows = np.zeros((grid, 8), dtype=np.uint64)
                used = run(d.ptr, n, grid, variant, args.warm, rows.ctypes.data, grid)
                assert used, "fsk_timeline_run failed"
                r = rows[:used].astype(np.int64)
                t = (r[:, :6] - r[:, 0].min()) / 100.0   # us
                xcc = r[:, 6] & 0xF
                rr = int((xcc == (np.arange(used) % 8)).sum())
                print("n=%d (%.0f MiB) grid=%d bpc=%d rep %d: kernel span %.2f us  (%.3f TB/s over the span)"
                      % (n, n * 2 / 2 ** 20, used, bpc, rep, t[:, 5].max(), 2 * n / t[:, 5].max() / 1e6))
                print("   entry            %s" % stats(t[:, 0]))
                print("   first step done  %s   (since entry: %s)" % (stats(t[:, 1]), stats(t[:, 1] - t[:, 0])))
                print("   last step done   %s" % stats(t[:, 2]))
                print("   flushed          %s   (flush:   %s)" % (stats(t[:, 3]), stats(t[:, 3] - t[:, 2])))
                print("   reduced          %s   (reduce:  %s)" % (stats(t[:, 4]), stats(t[:, 4] - t[:, 3])))
                print("   exit             %s   (atomics: %s)" % (stats(t[:, 5]), stats(t[:, 5] - t[:, 4])))
                per = ["%d:%.2f/%.2f" % (x, t[xcc == x, 2].mean(), t[xcc == x, 2].max()) for x in sorted(set(xcc.tolist()))]
                print("   last-step-done per XCD (mean/max): " + "  ".join(per))
                print("   workgroups with XCC_ID == blockIdx.x %% 8: %d of %d" % (rr, used), flush=True)


if __name__ == "__main__":
    main()
