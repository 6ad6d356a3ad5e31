#!/usr/bin/env python3
"""Soak of the GPU LZ4 (--mode fast:2 | hc:9) or Zstandard (--mode zstd:1) decode pipeline (copy stream + two decode streams + events, file and image mode): the same image
through the product entry again and again, alternating with the host-thread decoder, counters checked every time."""
import argparse
import ctypes
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402
from lz4_decoder_sweep import build_image  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=2 ** 31)
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--mode", default="fast:2")
    args = ap.parse_args()
    import oracle
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    mode, level = args.mode.split(":")
    zstd = mode == "zstd"
    knob = b"zstd_decoder" if zstd else b"lz4_decoder"
    file_entry = lib.FLAGSTATS_hip_blockfile_zstd if zstd else lib.FLAGSTATS_hip_blockfile_lz4
    image_entry = lib.FLAGSTATS_hip_blockimage_zstd if zstd else lib.FLAGSTATS_hip_blockimage_lz4
    img = build_image(args.flags, mode, int(level))
    want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, args.flags)
    buf = np.frombuffer(img, dtype=np.uint8)
    walls = {0: [], 1: []}
    with tempfile.NamedTemporaryFile(suffix=".zst" if zstd else ".lz4", dir=os.environ.get("TMPDIR", "/tmp")) as f:
        f.write(img)
        f.flush()
        for r in range(args.rounds):
            dec = 0 if r % 5 == 4 else 1
            _lib.check(lib.FLAGSTATS_hip_set(knob, dec), "set")
            out = np.zeros(32, dtype=np.uint64)
            st = _lib.BlockfileStats()
            t0 = time.perf_counter()
            if r % 2:
                _lib.check(file_entry(f.name.encode(), 0, out.ctypes.data, ctypes.byref(st)), "blockfile")
            else:
                _lib.check(image_entry(buf.ctypes.data, buf.size, 0, out.ctypes.data, ctypes.byref(st)), "blockimage")
            walls[dec].append(time.perf_counter() - t0)
            assert st.gpu_decode == dec
            assert np.array_equal(out, want), "round %d (%s, %s): counters differ from the oracle" % (r, "GPU" if dec else "host", "file" if r % 2 else "image")
    _lib.check(lib.FLAGSTATS_hip_set(knob, 2), "set")
    g = walls[1]
    print("soak: %d rounds on %d flags (%s), image and file mode alternating, all exact" % (args.rounds, args.flags, args.mode))
    print("  GPU decode, every round in order (ms; even = image, odd = file): " + " ".join("%.1f" % (w * 1e3) for w in g))
    print("  GPU decode: FIRST call %.1f ms (it allocates the large device buffers, the pinned spans, streams and events), afterwards %.1f-%.1f ms; host threads %.1f-%.1f ms"
          % (g[0] * 1e3, min(g[1:]) * 1e3, max(g[1:]) * 1e3, min(walls[0]) * 1e3, max(walls[0]) * 1e3))


if __name__ == "__main__":
    main()
