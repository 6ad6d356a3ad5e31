#!/usr/bin/env python3
"""Where between well compressible (NA12878-like) and incompressible (uniform) flags the GPU decoders stop paying: block
images of 2^28 flags in which a share of every block is 12-bit uniform noise, both codecs, both decoders (forced)."""
import ctypes
import os
import struct
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import blockfile_tool as bt  # noqa: E402
import oracle  # noqa: E402
from libflagstats_amd import _lib  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
n = 2 ** 28
per = 512000
for codec, mode, level in (("zstd", "zstd", 1), ("lz4", "fast", 2)):
    knob = b"zstd_decoder" if codec == "zstd" else b"lz4_decoder"
    entry = lib.FLAGSTATS_hip_blockimage_zstd if codec == "zstd" else lib.FLAGSTATS_hip_blockimage_lz4
    for share in (0.0, 0.1, 0.2, 0.35, 0.5, 0.7):
        def make(i):
            f = oracle.generate(oracle.GEN_NA12878, 7, 1, i * per, per).copy()
            k = int(per * share)
            if k:
                f[:k] = oracle.generate(oracle.GEN_UNIFORM, 3, 0x0FFF, i * per, k)
            c = bt.compress_block(f.tobytes(), mode, level)
            return struct.pack("<ii", f.nbytes, len(c)) + c
        with ThreadPoolExecutor(16) as ex:
            img = b"".join(ex.map(make, range(n // per)))
        buf = np.frombuffer(img, dtype=np.uint8)
        res = {}
        for dec in (0, 1):
            _lib.check(lib.FLAGSTATS_hip_set(knob, dec), "set")
            ts = []
            for rep in range(3):
                out = np.zeros(32, dtype=np.uint64)
                st = _lib.BlockfileStats()
                t0 = time.perf_counter()
                _lib.check(entry(buf.ctypes.data, buf.size, 0, out.ctypes.data, ctypes.byref(st)), "image")
                ts.append(time.perf_counter() - t0)
            res[dec] = min(ts)
        print("%s, %2.0f %% noise: ratio %.2f (%d MiB): host threads %.1f ms, GPU decode %.1f ms" % (codec, 100 * share, 2.0 * (n // per) * per / len(img), len(img) >> 20, res[0] * 1e3, res[1] * 1e3), flush=True)
    _lib.check(lib.FLAGSTATS_hip_set(knob, 2), "set")
