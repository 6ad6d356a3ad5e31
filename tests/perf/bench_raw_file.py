#!/usr/bin/env python3
"""README-comparable raw-file run (`bench decompress -D`, benchmark/flagstats.cpp:415-468; README.md:36: 824.5 M
flags raw in 0.48 s): FLAGSTATS_hip_file_raw on a raw uint16 file of NA12878-like flags in the page cache."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib, blockfile  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
import oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 824541892
d = tempfile.mkdtemp(prefix="fsraw_", dir="/tmp")
path = os.path.join(d, "flags.bin")
a = oracle.generate(oracle.GEN_NA12878, 7, 1, 0, n)
a.tofile(path)
want = oracle.flagstat_hist(a)
for rep in range(4):
    t0 = time.perf_counter()
    got, st = blockfile.flagstat_raw_file(path)
    dt = time.perf_counter() - t0
    assert np.array_equal(got, want)
    print("raw file %d flags (%.2f GB): %.4f s  %.2f Gflags/s  %.2f GB/s" % (n, 2 * n / 1e9, dt, n / dt / 1e9, 2 * n / dt / 1e9), flush=True)
os.remove(path)
