#!/usr/bin/env python3
"""Interleaved A/B of the HOST-thread LZ4 pipeline with and without its NUMA / affinity handling (knob "numa": pinned chunk
buffers placed on the GPU's host node, decoder threads bound to that node's CPUs) on ONE box -- VERDICT r03 weak #9: r02
measured 24-27.4 Gflags/s, r03 19-23.5 'on this round's box' in the round that changed the mempolicy / affinity handling."""
import ctypes
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402
from lz4_decoder_sweep import build_image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2 ** 31
lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
_lib.check(lib.FLAGSTATS_hip_set(b"lz4_decoder", 0), "host threads")
import oracle  # noqa: E402

img = build_image(n, "fast", 2)
want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, n)
buf = np.frombuffer(img, dtype=np.uint8)
res = {0: [], 1: []}
for rnd in range(8):
    for numa in ((0, 1) if rnd % 2 == 0 else (1, 0)):
        _lib.check(lib.FLAGSTATS_hip_set(b"numa", numa), "numa")
        lib.FLAGSTATS_hip_shutdown()            # the pinned buffers are placed when they are made: new engine per setting
        _lib.check(lib.FLAGSTATS_hip_init(0), "init")
        for rep in range(2):
            out = np.zeros(32, dtype=np.uint64)
            st = _lib.BlockfileStats()
            t0 = time.perf_counter()
            _lib.check(lib.FLAGSTATS_hip_blockimage_lz4(buf.ctypes.data, buf.size, 0, out.ctypes.data, ctypes.byref(st)), "blockimage")
            dt = time.perf_counter() - t0
            assert np.array_equal(out, want) and st.gpu_decode == 0
            if rep:
                res[numa].append(n / dt / 1e9)
_lib.check(lib.FLAGSTATS_hip_set(b"numa", 1), "numa")
_lib.check(lib.FLAGSTATS_hip_set(b"lz4_decoder", 2), "decoder")
print("host-thread LZ4 pipeline, %d flags, LZ4-fast image in memory, 8 interleaved rounds on one box (node of the GPU: %d):"
      % (n, lib.FLAGSTATS_hip_get(b"numa_node")))
for numa in (0, 1):
    v = res[numa]
    print("  numa=%d: median %.1f Gflags/s (min %.1f, max %.1f)" % (numa, statistics.median(v), min(v), max(v)))
