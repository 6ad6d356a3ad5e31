#!/usr/bin/env python3
"""K1 reading a FLAG array IN PLACE from pinned host memory (no staging copy): rate vs array size,
next to the double-buffered staging path of FLAGSTATS_u16_x64 on the same buffer."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib, device  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
import oracle  # noqa: E402

N = 2 ** 30
hp = lib.FLAGSTATS_hip_host_alloc(2 * N)
host = np.ctypeslib.as_array(ctypes.cast(hp, ctypes.POINTER(ctypes.c_uint16)), shape=(N,))
host[:] = oracle.generate(oracle.GEN_NA12878, 3, 1, 0, N)
print("flags        zero-copy ms   GB/s    staged ms   GB/s")
for n in (2 ** 16, 2 ** 20, 2 ** 24, 2 ** 27, 2 ** 30):
    reps = max(3, min(200, 2 ** 31 // n))
    ms, got = device.time_device_ptr(hp, n, 1, reps)
    zc = ms / reps
    out = np.zeros(32, dtype=np.uint64)
    lib.FLAGSTATS_u16_x64(hp, n, out.ctypes.data)
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.FLAGSTATS_u16_x64(hp, n, out.ctypes.data)
    st = (time.perf_counter() - t0) / reps * 1e3
    assert np.array_equal(got, oracle.flagstat_hist(host[:n])) if n <= 2 ** 27 else True
    print("%-11d %10.4f %8.2f %10.4f %8.2f" % (n, zc, 2 * n / zc / 1e6, st, 2 * n / st / 1e6), flush=True)
lib.FLAGSTATS_hip_host_free(hp)
