import sys, time, ctypes
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libflagstats_amd import _lib, device
lib = _lib.lib(); _lib.check(lib.FLAGSTATS_hip_init(0), "init")
import oracle
for n in (1000, 16384, 131072, 512000):
    a = oracle.generate(oracle.GEN_NA12878, 1, 1, 0, n)
    d = device.DeviceFlags(n)
    flags = np.zeros(32, dtype=np.uint32); out = np.zeros(32, dtype=np.uint64)
    def t(f, reps=2000):
        for _ in range(20): f()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        return (time.perf_counter() - t0) / reps * 1e6
    full = t(lambda: lib.FLAGSTATS_u16(a.ctypes.data, n, flags.ctypes.data))
    h2d = t(lambda: lib.FLAGSTATS_hip_memcpy_h2d(d.ptr, a.ctypes.data, n * 2))
    dev = t(lambda: lib.FLAGSTATS_hip_device_u16_sync(d.ptr, n, out.ctypes.data))
    x64 = t(lambda: lib.FLAGSTATS_u16_x64(a.ctypes.data, n, out.ctypes.data))
    print("n=%7d  FLAGSTATS_u16 %.1f us   u16_x64 %.1f   h2d memcpy(sync) %.1f   device_u16_sync %.1f" % (n, full, x64, h2d, dev))
