#!/usr/bin/env python3
"""Row f4 under the profiler: `rocprofv3 ... -- python3 tests/perf/profile_pospopcnt.py [flags] [launches]` -- the plain
positional popcount (STORM_pospopcnt_u16, python/libalgebra.h:3496-3551) over a device-resident uint16 array, checked
against the oracle on a prefix, then `launches` back-to-back full-size launches timed with the host clock."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from libflagstats_amd import _lib, device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2 ** 32
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
d = device.DeviceFlags(n).generate(device.GEN_UNIFORM, seed=5, mask=0xFFFF)
out = device.DeviceFlags(64)
# (parity of this op: tests/test_pospopcnt.py, against goldens from the reference's own function)
for _ in range(3):
    _lib.check(lib.FLAGSTATS_hip_device_pospopcnt_u16(d.ptr, n, out.ptr, None), "pospopcnt")
_lib.check(lib.FLAGSTATS_hip_synchronize(), "sync")
t0 = time.perf_counter()
for _ in range(launches):
    _lib.check(lib.FLAGSTATS_hip_device_pospopcnt_u16(d.ptr, n, out.ptr, None), "pospopcnt")
_lib.check(lib.FLAGSTATS_hip_synchronize(), "sync")
ms = (time.perf_counter() - t0) * 1e3 / launches
print("pospopcnt_count: %d flags, %d launches back to back: %.4f ms per launch incl. the host's stream wait = %.3f TB/s = %.1f %% of 8 TB/s"
      % (n, launches, ms, 2 * n / ms / 1e9, 100 * 2 * n / ms / 1e9 / 8))
