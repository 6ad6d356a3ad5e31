import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools"); sys.path.insert(0, "tests/perf")
import numpy as np
import oracle
import blockfile_tool as bt
from libflagstats_amd import blockfile, _lib
lib = _lib.lib(); _lib.check(lib.FLAGSTATS_hip_init(0), "init")
lib.FLAGSTATS_hip_set(b"lz4_decoder", 0)
for n in (2**21, 2**22, 2**23, 2**24, 2**25):
    flags = oracle.generate(oracle.GEN_NA12878, 7, 1, 0, n)
    img = bt.block_file_image(flags, mode="fast", level=2)
    line = "%9d flags (%5.1f MiB file, %d blocks):" % (n, len(img) / 2**20, (2 * n + 1023999) // 1024000)
    for th in (1, 2, 4, 8, 12, 16, 0):
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); got, st = blockfile.flagstat_lz4_image(img, th); ts.append((time.perf_counter() - t0) * 1e3)
        line += "  %2d thr %.2f" % (th, min(ts))
    print(line, flush=True)
