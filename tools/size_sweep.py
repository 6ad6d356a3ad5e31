#!/usr/bin/env python3
"""K1+K2 latency / throughput vs array size on one MI355X (device-resident, back-to-back launches),
for the sizes BASELINE.json's configs name (1 M flags, 1 GiB, 8 GiB) and points between, on uniform
and NA12878-like data.  Arrays <= 256 MiB can sit in the Infinity Cache between launches; the
'rot' column rotates over enough distinct buffers to defeat that."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import _lib, device  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
total = 2 ** 32
d = device.DeviceFlags(total).generate(0, seed=5, mask=0xFFFF)
print("kind      flags        MiB   ms(same buf)  TB/s     ms(rotating)  TB/s")
for kind, name in ((0, "uniform"), (1, "na12878")):
    if kind == 1:
        d.generate(1, seed=5, mask=1)
    for n in (10 ** 6, 2 ** 22, 2 ** 24, 2 ** 26, 2 ** 28, 2 ** 29, 2 ** 30, 2 ** 32):
        reps = max(5, min(200, (2 ** 33) // n))
        same = []
        for r in range(5):
            ms, _ = device.time_device_ptr(d.ptr, n, 2, reps)
            same.append(ms / reps)
        # rotate through disjoint slices of the 8 GiB buffer
        slots = max(1, total // n)
        rot = []
        if slots > 1:
            import ctypes
            import numpy as np
            for r in range(3):
                t = 0.0
                k = min(slots, 64)
                for i in range(k):
                    ms, _ = device.time_device_ptr(d.ptr + 2 * n * ((i * 7919) % slots), n, 0, 1)
                    t += ms
                rot.append(t / k)
        a = statistics.median(same)
        b = statistics.median(rot) if rot else float("nan")
        print("%-8s %11d %8.1f   %9.4f  %7.3f   %9.4f  %7.3f" % (name, n, n * 2 / 2 ** 20, a, 2 * n / a / 1e9, b,
                                                               2 * n / b / 1e9 if rot else float("nan")))
