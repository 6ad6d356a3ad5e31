#!/usr/bin/env python3
"""Does the cache-policy modifier of the load matter for a read-once stream?  The fastest read pattern (384 threads x 4
vectors per CU) with every combination of sc0 / sc1 / nt on global_load_dwordx4, interleaved rounds on an 8 GiB buffer."""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import _lib, device  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
n = 2 ** 32
d = device.DeviceFlags(n).generate(0, seed=1, mask=0xFFFF)
names = ["plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0", "sc0 nt"]
res = {p: [] for p in range(8)}
for r in range(9):
    for p in (range(8) if r % 2 == 0 else reversed(range(8))):
        ms = ctypes.c_float(0)
        _lib.check(lib.FLAGSTATS_hip_read_probe_policy(d.ptr, 2 * n, p, 2, 10, ctypes.byref(ms)), "probe")
        res[p].append(ms.value / 10)
print("policy        median_ms   TB/s")
for p in range(8):
    m = statistics.median(res[p])
    print("%-12s  %8.4f  %6.3f" % (names[p], m, 2 * n / m / 1e9))
