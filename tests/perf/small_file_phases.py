#!/usr/bin/env python3
"""Where a SMALL block file's time goes on the host-thread pipeline (the product path below the GPU decoders' size rule):
FLAGSTATS_blockfile_stats of the best of 25 calls, image in memory."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import blockfile_tool as bt  # noqa: E402
import oracle  # noqa: E402
from libflagstats_amd import _lib, blockfile  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
for n in (512000, 2**21, 2**22, 2**23, 2**24):
    flags = oracle.generate(oracle.GEN_NA12878, 7, 1, 0, n)
    for mode, level, entry in (("fast", 2, blockfile.flagstat_lz4_image), ("zstd", 1, blockfile.flagstat_zstd_image)):
        img = bt.block_file_image(flags, mode=mode, level=level)
        best = None
        for _ in range(25):
            t0 = time.perf_counter()
            got, st = entry(img, 0)
            t = (time.perf_counter() - t0) * 1e3
            if best is None or t < best[0]:
                best = (t, st)
        t, st = best
        print("%9d flags %-4s %6.2f MiB %3d blocks: call %.3f ms | wall %.3f = index %.3f + buffers %.3f + pipeline %.3f (waiting for decoders %.3f, for copies %.3f) | decode cpu %.3f over %d threads, %d chunks" % (
            n, mode, len(img) / 2**20, st["n_blocks"], t, st["wall_s"] * 1e3, st["index_s"] * 1e3, (st["setup_s"] - st["index_s"]) * 1e3,
            (st["wall_s"] - st["setup_s"]) * 1e3, st["wait_decode_s"] * 1e3, st["wait_copy_s"] * 1e3, st["decode_cpu_s"] * 1e3, st["threads"], st["chunks"]), flush=True)
