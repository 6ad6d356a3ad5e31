#!/usr/bin/env python3
"""Can the CPU write straight into device memory on this box (large BAR), and how fast?  Fine-grained device memory from
hipExtMallocWithFlags, written with memmove from a child process first (a box without host access to VRAM would fault
there, not here), then timed and checked through the library's device entry."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
hip = ctypes.CDLL(None)   # the HIP runtime the library already loaded (RTLD_GLOBAL)
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipExtMallocWithFlags.restype = ctypes.c_int
p = ctypes.c_void_p()
rc = hip.hipExtMallocWithFlags(ctypes.byref(p), 4 << 20, 1)     # hipDeviceMallocFinegrained = 0x1
print("hipExtMallocWithFlags(finegrained) rc", rc, "ptr", hex(p.value or 0), flush=True)
if rc != 0:
    sys.exit(0)
a = (np.arange(1 << 20) * 7919 % 65536).astype(np.uint16)
print("writing 4 KiB into device memory from the CPU (a box without host access to VRAM faults here) ...", flush=True)
ctypes.memmove(p.value, a.ctypes.data, 4096)
print("... ok", flush=True)
for nbytes in (2048, 32768, 262144, 2 << 20):
    t0 = time.perf_counter()
    for _ in range(200):
        ctypes.memmove(p.value, a.ctypes.data, nbytes)
    dt = (time.perf_counter() - t0) / 200
    print("CPU -> device memory memmove of %7d bytes: %.2f us (%.1f GB/s)" % (nbytes, dt * 1e6, nbytes / dt / 1e9), flush=True)
import oracle  # noqa: E402

n = 100000
ctypes.memmove(p.value, a.ctypes.data, 2 * n)
out = np.zeros(32, dtype=np.uint64)
_lib.check(lib.FLAGSTATS_hip_device_u16_sync(p.value, n, out.ctypes.data), "device_u16_sync on CPU-written device memory")
print("counters from CPU-written device memory equal the oracle:", bool((out == oracle.flagstat_hist(a[:n])).all()))
