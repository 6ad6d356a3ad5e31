#!/usr/bin/env python3
"""What the fixed cost of a small K1 launch is made of (tuning build only:
FLAGSTATS_HIP_LIB=libflagstats_amd/libflagstats_hip_tuning.so).  For each array size, back-to-back
launches of the default schedule with parts of the kernel switched off (results are wrong in those
rows -- timing only):  full | no final flush | nothing after the flush | no steps (launch + epilogue)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import _lib, device  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
assert lib.FLAGSTATS_hip_get(b"tuning_build"), "needs the tuning build"
total = 2 ** 30
d = device.DeviceFlags(total).generate(0, seed=5, mask=0xFFFF)
rows = (("full", 0), ("full, adds to 8 per-XCD copies", 8), ("no final flush", 2), ("no flush, nothing after", 6),
        ("nothing after the flush", 4), ("no steps", 1), ("no steps, no flush, nothing after", 7))
sizes = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "1000000,4194304,16777216,67108864,536870912").split(",")]
print("%-36s" % "us per launch" + "".join("%12d" % n for n in sizes))
for ep in (1, 0):
    lib.FLAGSTATS_hip_set(b"epilogue", ep)
    for name, bits in rows:
        _lib.check(lib.FLAGSTATS_hip_set(b"anatomy", bits), "anatomy")
        line = "%-36s" % (("atomic: " if ep else "k1+k2:  ") + name)
        for n in sizes:
            reps = max(5, min(200, (2 ** 33) // n))
            t = []
            for r in range(5):
                ms, _ = device.time_device_ptr(d.ptr, n, 2, reps)
                t.append(ms / reps)
            line += "%12.2f" % (statistics.median(t) * 1e3)
        print(line, flush=True)
lib.FLAGSTATS_hip_set(b"anatomy", 0)
