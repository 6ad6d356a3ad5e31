#!/usr/bin/env python3
"""For `rocprofv3 --kernel-trace --stats -- python3 tests/perf/trace_lz4_gpu.py`: five GPU-decoded passes over one LZ4 (or, mode zstd:N, Zstandard)
block image (2^31 flags by default) through the product entry, so the kernel table shows the decode launches and K1."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402
from lz4_decoder_sweep import build_image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2 ** 31
lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
mode, level = (sys.argv[2] if len(sys.argv) > 2 else "fast:2").split(":")
zstd = mode == "zstd"
_lib.check(lib.FLAGSTATS_hip_set(b"zstd_decoder" if zstd else b"lz4_decoder", 1), "set")
entry = lib.FLAGSTATS_hip_blockimage_zstd if zstd else lib.FLAGSTATS_hip_blockimage_lz4
img = build_image(n, mode, int(level))
buf = np.frombuffer(img, dtype=np.uint8)
for rep in range(5):
    out = np.zeros(32, dtype=np.uint64)
    st = _lib.BlockfileStats()
    _lib.check(entry(buf.ctypes.data, buf.size, 0, out.ctypes.data, ctypes.byref(st)), "blockimage")
    print("pass %d: %.1f ms, %d blocks in %d pieces" % (rep, st.wall_s * 1e3, st.n_blocks, st.chunks), flush=True)
