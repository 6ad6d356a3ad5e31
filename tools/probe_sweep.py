#!/usr/bin/env python3
"""Which read pattern does the chip serve fastest?  Sweeps the read-only probe over workgroup size,
unroll depth, grid size, traversal mode and cache policy on an 8 GiB buffer (measurement helper)."""
import ctypes
import itertools
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import _lib, device  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
n = 2 ** 32
d = device.DeviceFlags(n).generate(0, seed=1, mask=0xFFFF)
cus = lib.FLAGSTATS_hip_compute_units()
rows = []
# r03: workgroup sizes between the powers of two as well -- r01's sweep only ever had 8 / 16 / 32 / 64 KiB in flight per
# CU and missed the optimum near 24 KiB that the rolling schedule found (profiles/r03/rolling_distance_sweep.log)
sizes = (256, 512, 1024) if "--r01" in sys.argv else (128, 192, 256, 320, 384, 448, 512, 768, 1024)
combos = list(itertools.product((0, 1), (1,), sizes, (2, 4, 8, 16), (1, 2, 4)))
for rnd in range(3):
    for mode, nt, threads, unroll, bpc in combos:
        if threads * unroll * 16 * bpc > 256 * 1024:      # > 256 KiB in flight per CU: pointless
            continue
        ms = ctypes.c_float(0)
        _lib.check(lib.FLAGSTATS_hip_read_probe2(d.ptr, 2 * n, mode, unroll, threads, cus * bpc, nt, 1, 5, ctypes.byref(ms)), "probe2")
        rows.append(((mode, nt, threads, unroll, bpc), ms.value / 5))
best = {}
for k, v in rows:
    best.setdefault(k, []).append(v)
out = sorted(((statistics.median(v), k) for k, v in best.items()))
print("TB/s   ms      mode nt threads unroll blocks/CU  KiB_in_flight/CU")
for ms, k in out[:25]:
    print("%.3f  %.4f  %s  %d" % (2 * n / ms / 1e9, ms, k, k[2] * k[3] * 16 * k[4] // 1024))
print("...worst:")
for ms, k in out[-5:]:
    print("%.3f  %.4f  %s" % (2 * n / ms / 1e9, ms, k))
