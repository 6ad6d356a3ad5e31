#!/usr/bin/env python3
"""Measurement helper (GPU box): sweep K1 variants x grid sizes on the headline workload and
measure the read-only probe ceiling.  Interleaved rounds in ONE process, median and min.
The product library carries schedules 9 and 25; the others (and --fuse 1) need the tuning build:
FLAGSTATS_HIP_LIB=libflagstats_amd/libflagstats_hip_tuning.so (make -C libflagstats_amd/csrc tuning)."""
import argparse
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib, device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=2 ** 32)
    ap.add_argument("--variants", default="0,1,9,13,25,27")
    ap.add_argument("--bpc", default="1,2,3,4,6,8")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--kind", type=int, default=0)
    ap.add_argument("--probe", action="store_true")
    ap.add_argument("--pospopcnt", action="store_true", help="also time the positional-popcount op (row f4)")
    ap.add_argument("--fuse", default="0", help="comma list of fuse modes to compare (0 = K1+K2, 1 = K1 finalises)")
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    n = args.flags
    d = device.DeviceFlags(n).generate(args.kind, seed=2026, mask=0xFFFF if args.kind == 0 else 1)
    res = {}
    variants = [int(v) for v in args.variants.split(",")]
    bpcs = [int(v) for v in args.bpc.split(",")]
    fuses = [int(f) for f in args.fuse.split(",")]
    for r in range(args.rounds):
      for fz in fuses:
        if lib.FLAGSTATS_hip_set(b"fuse", fz) != 0:
            continue
        for v in variants:
            for b in bpcs:
                v = v % 1000 + 1000 * fz
                if lib.FLAGSTATS_hip_set(b"variant", v % 1000) != 0:
                    continue   # a schedule (or the ticket form) this build does not carry: needs the tuning build
                lib.FLAGSTATS_hip_set(b"blocks_per_cu", b)
                ms, _ = device.time_device_ptr(d.ptr, n, 1, args.reps)
                res.setdefault((v, b), []).append(ms / args.reps)
    print("variant bpc  median_ms  min_ms   TB/s(med)  TB/s(best)")
    for (v, b), t in sorted(res.items()):
        med, mn = statistics.median(t), min(t)
        print("%7d %3d  %9.4f %8.4f  %8.3f  %8.3f" % (v, b, med, mn, 2 * n / med / 1e9, 2 * n / mn / 1e9))
    if args.pospopcnt:
        import torch
        t = torch.empty(0)  # noqa: F841  (torch only for the stream-ordered device entry below)
        out = device.DeviceFlags(64)
        ts = []
        import time
        for r in range(args.rounds):
            _lib.check(lib.FLAGSTATS_hip_synchronize(), "sync")
            t0 = time.perf_counter()
            for _ in range(args.reps):
                _lib.check(lib.FLAGSTATS_hip_device_pospopcnt_u16(d.ptr, n, out.ptr, None), "pospopcnt")
            _lib.check(lib.FLAGSTATS_hip_synchronize(), "sync")
            ts.append((time.perf_counter() - t0) * 1e3 / args.reps)
        med = statistics.median(ts)
        print("pospopcnt_u16: median %.4f ms  %.3f TB/s" % (med, 2 * n / med / 1e9))
    if args.probe:
        print("read probe: nt bpc median_ms TB/s")
        for nt in (0, 1):
            for b in bpcs:
                lib.FLAGSTATS_hip_set(b"blocks_per_cu", b)
                ts = []
                for r in range(args.rounds):
                    ms = ctypes.c_float(0)
                    _lib.check(lib.FLAGSTATS_hip_read_probe(d.ptr, 2 * n, nt, 1, args.reps, ctypes.byref(ms)), "probe")
                    ts.append(ms.value / args.reps)
                med = statistics.median(ts)
                print("%d %3d %9.4f %8.3f" % (nt, b, med, 2 * n / med / 1e9))


if __name__ == "__main__":
    main()
