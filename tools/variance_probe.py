#!/usr/bin/env python3
"""Where does the 1190-1250 us spread of the 8 GiB launch come from: the allocation (which physical pages the
array landed on), the process, or time?  In ONE process: several 8 GiB buffers allocated one after the other (all
kept, so each lands on different pages), each timed in interleaved rounds with K1 and with the read-only probe."""
import ctypes
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import _lib, device  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
n = 2 ** 32
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bufs = []
for i in range(nbuf):
    d = device.DeviceFlags(n).generate(0, seed=5 + i, mask=0xFFFF)
    bufs.append(d)
    print("buffer %d at 0x%x" % (i, d.ptr), flush=True)
res = {i: [] for i in range(nbuf)}
prb = {i: [] for i in range(nbuf)}
t0 = time.time()
for r in range(8):
    for i in (range(nbuf) if r % 2 == 0 else reversed(range(nbuf))):
        ms, _ = device.time_device_ptr(bufs[i].ptr, n, 2, 20)
        res[i].append(ms / 20)
        pm = ctypes.c_float(0.0)
        _lib.check(lib.FLAGSTATS_hip_read_probe(bufs[i].ptr, 2 * n, 1, 2, 20, ctypes.byref(pm)), "probe")
        prb[i].append(pm.value / 20)
    print("round %d (t=%.1f s): " % (r, time.time() - t0) + "  ".join("%.1f/%.1f" % (res[i][-1] * 1e3, prb[i][-1] * 1e3) for i in range(nbuf)), flush=True)
for i in range(nbuf):
    print("buffer %d: K1 median %.2f us (min %.2f max %.2f)   probe median %.2f us" % (
        i, statistics.median(res[i]) * 1e3, min(res[i]) * 1e3, max(res[i]) * 1e3, statistics.median(prb[i]) * 1e3))
