#!/usr/bin/env python3
"""Hot-path latency / throughput vs array size on one MI355X (device-resident) for the sizes
BASELINE.json's configs name (1 M flags, 1 GiB, 8 GiB) and points between.

Per epilogue form (`k2` = K1 writes partials and K2 sums them, `atomic` = K1's workgroups add their totals to
out[32] themselves, one launch) three timings per array size:
  same     back-to-back launches on ONE slice (arrays <= 256 MiB may sit in the Infinity Cache between launches)
  rot      back-to-back launches rotating over disjoint slices of an 8 GiB buffer (no cache reuse; the honest
           steady-state figure for a mid-size array)
  rot1     one launch per timed region on rotating slices (r02's "rot" column: includes the event pair, the idle
           gap before the launch and the chip's ramp from idle -- the latency of ONE isolated call)
--bpc also sweeps workgroups per CU (0 = the library's own size-aware choice)."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import _lib, device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="0,1")
    ap.add_argument("--bpc", default="0")
    ap.add_argument("--epilogues", default="0,1")
    ap.add_argument("--sizes", default="1000000,4194304,16777216,67108864,268435456,536870912,1073741824,4294967296")
    ap.add_argument("--rot1", action="store_true", help="also time isolated single launches (r02's rot column)")
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    total = 2 ** 32
    d = device.DeviceFlags(total)
    names = {0: "uniform", 1: "na12878"}
    eps = [int(e) for e in args.epilogues.split(",")]
    head = "kind     bpc       flags      MiB"
    for e in eps:
        tag = "atomic" if e else "k2"
        head += "   us(%s,same) TB/s   us(%s,rot) TB/s" % (tag, tag)
        if args.rot1:
            head += "   us(%s,rot1) TB/s" % tag
    print(head)
    for kind in [int(k) for k in args.kinds.split(",")]:
        d.generate(kind, seed=5, mask=0xFFFF if kind == 0 else 1)
        for bpc in [int(b) for b in args.bpc.split(",")]:
            _lib.check(lib.FLAGSTATS_hip_set(b"blocks_per_cu", bpc), "bpc")
            for n in [int(s) for s in args.sizes.split(",")]:
                line = "%-8s %3d %11d %8.1f" % (names[kind], bpc, n, n * 2 / 2 ** 20)
                for e in eps:
                    _lib.check(lib.FLAGSTATS_hip_set(b"epilogue", e), "epilogue")
                    reps = max(5, min(200, (2 ** 33) // n))
                    same = []
                    for r in range(5):
                        ms, _ = device.time_device_ptr(d.ptr, n, 2, reps)
                        same.append(ms / reps)
                    stride = (n + 7) & ~7            # slices keep the 16-byte phase of the buffer
                    slots = max(1, total // stride)
                    rot, rot1 = [], []
                    if slots > 1:
                        for r in range(5):
                            ms, _ = device.time_device_rotating(d.ptr, n, stride, slots, 2, reps)
                            rot.append(ms / reps)
                        if args.rot1:
                            for r in range(3):
                                t = 0.0
                                k = min(slots, 64)
                                for i in range(k):
                                    ms, _ = device.time_device_ptr(d.ptr + 2 * stride * ((i * 7919) % slots), n, 0, 1)
                                    t += ms
                                rot1.append(t / k)
                    a = statistics.median(same)
                    b = statistics.median(rot) if rot else float("nan")
                    line += "   %9.2f %6.3f   %9.2f %6.3f" % (a * 1e3, 2 * n / a / 1e9, b * 1e3,
                                                             2 * n / b / 1e9 if rot else float("nan"))
                    if args.rot1:
                        c = statistics.median(rot1) if rot1 else float("nan")
                        line += "   %9.2f %6.3f" % (c * 1e3, 2 * n / c / 1e9 if rot1 else float("nan"))
                print(line, flush=True)
    lib.FLAGSTATS_hip_set(b"blocks_per_cu", 0)
    lib.FLAGSTATS_hip_set(b"epilogue", 1)


if __name__ == "__main__":
    main()
