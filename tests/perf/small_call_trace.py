#!/usr/bin/env python3
"""2000 FLAGSTATS_u16 calls of n flags (default 1000) -- the program to put behind
`rocprofv3 --kernel-trace --stats` to see what part of the per-call latency is the kernel itself."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
a = (np.arange(n) * 7919 % 65536).astype(np.uint16)
flags = np.zeros(32, dtype=np.uint32)
for _ in range(20):
    lib.FLAGSTATS_u16(a.ctypes.data, n, flags.ctypes.data)
t0 = time.perf_counter()
for _ in range(2000):
    lib.FLAGSTATS_u16(a.ctypes.data, n, flags.ctypes.data)
print("n=%d: %.2f us per call" % (n, (time.perf_counter() - t0) / 2000 * 1e6))
