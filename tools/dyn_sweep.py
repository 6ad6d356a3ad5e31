#!/usr/bin/env python3
"""K1 static (variant 25) vs dynamic (variant 153) schedule, and the dynamic schedule's policy knobs, per array
size: back-to-back launches rotating over disjoint slices of an 8 GiB buffer (the 8 GiB row: one buffer),
interleaved rounds in ONE process, median and best of the rounds."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import _lib, device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="67108864,536870912,4294967296")
    ap.add_argument("--kinds", default="0")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--configs", default="25;153:75:4:32:0:3;153:90:4:32:0:3;153:50:4:32:0:3;153:75:2:32:0:3;153:75:4:32:0:4;153:75:4:32:0:2;153:75:4:32:0:0;153:75:8:16:0:3",
                    help="variant[:first_pct:div:cmax[:bpc[:lg_queues]]] separated by ';'")
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    total = 2 ** 32
    d = device.DeviceFlags(total)
    cfgs = []
    for c in args.configs.split(";"):
        f = [int(x) for x in c.split(":")]
        cfgs.append(tuple(f + [75, 4, 32, 0, 3][len(f) - 1:]))
    for kind in [int(k) for k in args.kinds.split(",")]:
        d.generate(kind, seed=5, mask=0xFFFF if kind == 0 else 1)
        for n in [int(s) for s in args.sizes.split(",")]:
            stride = (n + 7) & ~7
            slots = max(1, total // stride)
            reps = max(8, min(200, (2 ** 34) // n))
            res = {c: [] for c in cfgs}
            for r in range(args.rounds):
                for c in cfgs:
                    v, pct, div, cmax, bpc, lgq = c
                    _lib.check(lib.FLAGSTATS_hip_set(b"variant", v), "variant")
                    lib.FLAGSTATS_hip_set(b"dyn_first_pct", pct)
                    lib.FLAGSTATS_hip_set(b"dyn_div", div)
                    lib.FLAGSTATS_hip_set(b"dyn_cmax", cmax)
                    lib.FLAGSTATS_hip_set(b"blocks_per_cu", bpc)
                    lib.FLAGSTATS_hip_set(b"dyn_lg_queues", lgq)
                    lib.FLAGSTATS_hip_set(b"dyn_min_steps", 1)
                    if slots > 1:
                        ms, _ = device.time_device_rotating(d.ptr, n, stride, slots, 3, reps)
                    else:
                        ms, _ = device.time_device_ptr(d.ptr, n, 2, reps)
                    res[c].append(ms / reps)
            print("kind %d  n=%d (%.0f MiB)  %d launches per round, %d rounds, %s" % (
                kind, n, n * 2 / 2 ** 20, reps, args.rounds, "rotating slices" if slots > 1 else "one buffer"))
            for c in cfgs:
                t = res[c]
                med, mn = statistics.median(t), min(t)
                print("   variant %3d first_pct %3d div %2d cmax %5d bpc %d lgq %d   median %9.2f us %6.3f TB/s (%.1f %%)   best %9.2f us %6.3f TB/s"
                      % (c + (med * 1e3, 2 * n / med / 1e9, 2 * n / med / 1e7 / 8.0, mn * 1e3, 2 * n / mn / 1e9)), flush=True)
    lib.FLAGSTATS_hip_set(b"variant", 71)
    lib.FLAGSTATS_hip_set(b"blocks_per_cu", 0)


if __name__ == "__main__":
    main()
