#!/usr/bin/env python3
"""A/B of one library knob on the headline launch (8 GiB by default), interleaved rounds in one process.
   python3 tools/knob_ab.py epoch_stagger 0 1 [--flags N] [--kind K]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import _lib, device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("key")
ap.add_argument("values", nargs="+", type=int)
ap.add_argument("--flags", type=int, default=2 ** 32)
ap.add_argument("--kind", type=int, default=0)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
n = args.flags
d = device.DeviceFlags(n).generate(args.kind, seed=5, mask=0xFFFF if args.kind == 0 else 1)
old = lib.FLAGSTATS_hip_get(args.key.encode())
res = {v: [] for v in args.values}
ref = None
for r in range(args.rounds):
    for v in (args.values if r % 2 == 0 else args.values[::-1]):
        _lib.check(lib.FLAGSTATS_hip_set(args.key.encode(), v), args.key)
        ms, out = device.time_device_ptr(d.ptr, n, 2, args.reps)
        res[v].append(ms / args.reps)
        if ref is None:
            ref = out.copy()
        assert (out == ref).all(), "the knob changed the counters"
lib.FLAGSTATS_hip_set(args.key.encode(), old)
print("n=%d flags, kind %d, %d rounds x %d launches, interleaved; counters identical for every value" % (n, args.kind, args.rounds, args.reps))
for v in args.values:
    t = res[v]
    print("  %s=%d   median %9.2f us %6.3f TB/s   best %9.2f   worst %9.2f" % (args.key, v, statistics.median(t) * 1e3,
          2 * n / statistics.median(t) / 1e9, min(t) * 1e3, max(t) * 1e3))
