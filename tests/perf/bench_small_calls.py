#!/usr/bin/env python3
"""Per-call latency of the drop-in host-pointer entry FLAGSTATS_u16 at the sizes an unmodified
caller of the reference uses (benchmark/flagstats.cpp calls the kernel once per 512,000-flag block),
next to the reference's own dispatcher kernel on the same host."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
import oracle  # noqa: E402

ref = oracle.load_ref()
# `--old`: the r02 path for comparison (device staging for every size, hipStreamSynchronize)
if "--old" in sys.argv:
    lib.FLAGSTATS_hip_set(b"small_flags", 0)
    lib.FLAGSTATS_hip_set(b"poll", 0)
print("small_flags=%d poll=%d input buffer in %s" % (lib.FLAGSTATS_hip_get(b"small_flags"), lib.FLAGSTATS_hip_get(b"poll"),
      "device memory (written through the BAR)" if lib.FLAGSTATS_hip_get(b"small_in_is_device") else "pinned host memory"))
print("flags      hip_us/call  hip_Gflags/s   ref_us/call  ref_Gflags/s")
for n in (1000, 16384, 50000, 80000, 100000, 131072, 200000, 512000, 2 ** 20, 2 ** 21, 2 ** 22, 2 ** 24, 2 ** 26):
    a = oracle.generate(oracle.GEN_NA12878, 1, 1, 0, n)
    flags = np.zeros(32, dtype=np.uint32)
    reps = max(20, min(2000, 2 ** 28 // n))
    for _ in range(5):
        lib.FLAGSTATS_u16(a.ctypes.data, n, flags.ctypes.data)
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.FLAGSTATS_u16(a.ctypes.data, n, flags.ctypes.data)
    hip_us = (time.perf_counter() - t0) / reps * 1e6
    ref_us = float("nan")
    if ref is not None:
        p16 = ctypes.cast(a.ctypes.data, ctypes.POINTER(ctypes.c_uint16))
        p32 = flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        for _ in range(5):
            ref.ref_FLAGSTATS_u16(p16, n, p32)
        t0 = time.perf_counter()
        for _ in range(reps):
            ref.ref_FLAGSTATS_u16(p16, n, p32)
        ref_us = (time.perf_counter() - t0) / reps * 1e6
    print("%-9d  %10.1f  %12.2f   %10.1f  %12.2f" % (n, hip_us, n / hip_us / 1e3, ref_us, n / ref_us / 1e3))
