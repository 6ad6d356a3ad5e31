#!/usr/bin/env python3
"""LZ4 / Zstandard block files: host-thread decode pipeline (knob lz4_decoder / zstd_decoder = 0) against the GPU decoder
(= 1) through the product entries, over file sizes -- where the default rule (= 2: GPU from lz4_gpu_min_bytes /
zstd_gpu_min_bytes) should put its threshold.  Image mode (file already in memory) at every size, file mode (page
cache) at --file-flags.  --modes fast:2,hc:9 are LZ4, zstd:1 ... zstd:19 Zstandard."""
import argparse
import ctypes
import os
import struct
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import blockfile_tool as bt  # noqa: E402
from libflagstats_amd import _lib  # noqa: E402


def build_image(n, mode, level):
    import oracle
    per = bt.BLOCK_BYTES // 2
    nblocks = (n + per - 1) // per

    def make(i):
        f = oracle.generate(oracle.GEN_NA12878, 7, 1, i * per, min(per, n - i * per))
        comp = bt.compress_block(f.tobytes(), mode, level)
        return struct.pack("<ii", f.nbytes, len(comp)) + comp

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        return b"".join(ex.map(make, range(nblocks)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2**26,2**27,2**28,2**29,2**30,2**31,2**32", help="flags per file")
    ap.add_argument("--modes", default="fast:2")
    ap.add_argument("--file-flags", default="2**31")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--numa", type=int, default=1, help="knob numa: bind decoder / reader threads to the GPU's host NUMA node")
    args = ap.parse_args()
    import oracle
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    _lib.check(lib.FLAGSTATS_hip_set(b"numa", args.numa), "set")
    for mode_level in args.modes.split(","):
        mode, level = mode_level.split(":")
        zstd = mode == "zstd"
        knob = b"zstd_decoder" if zstd else b"lz4_decoder"
        image_entry = lib.FLAGSTATS_hip_blockimage_zstd if zstd else lib.FLAGSTATS_hip_blockimage_lz4
        file_entry = lib.FLAGSTATS_hip_blockfile_zstd if zstd else lib.FLAGSTATS_hip_blockfile_lz4
        for size in args.sizes.split(","):
            n = int(eval(size))
            img = build_image(n, mode, int(level))
            want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, n)
            buf = np.frombuffer(img, dtype=np.uint8)
            line = ("%s-%s %6.0f Mflags" % ("Zstd" if zstd else "LZ4-" + mode, level, n / 1e6)) + " (%5.0f MiB compressed, ratio %.2f) image:" % (len(img) / 2**20, 2 * n / len(img))
            best = {}
            for dec in (0, 1):
                _lib.check(lib.FLAGSTATS_hip_set(knob, dec), "set")
                ts = []
                for rep in range(args.reps):
                    out = np.zeros(32, dtype=np.uint64)
                    st = _lib.BlockfileStats()
                    t0 = time.perf_counter()
                    _lib.check(image_entry(buf.ctypes.data, buf.size, args.threads, out.ctypes.data, ctypes.byref(st)), "blockimage")
                    ts.append(time.perf_counter() - t0)
                    assert np.array_equal(out, want) and st.gpu_decode == dec
                best[dec] = min(ts)
                line += "  %s %7.1f ms = %5.1f Gflags/s" % ("GPU decode" if dec else "host threads", best[dec] * 1e3, n / best[dec] / 1e9)
            print(line + "  -> GPU/host %.2fx" % (best[0] / best[1]), flush=True)
            if n == int(eval(args.file_flags)):
                with tempfile.NamedTemporaryFile(suffix=".zst" if zstd else ".lz4", dir=os.environ.get("TMPDIR", "/tmp")) as f:
                    f.write(img)
                    f.flush()
                    os.fsync(f.fileno())   # written back before the timed reads: a file in the page cache, clean
                    line = "      the same as a FILE (page cache):"
                    for dec in (0, 1):
                        _lib.check(lib.FLAGSTATS_hip_set(knob, dec), "set")
                        ts = []
                        for rep in range(args.reps):
                            out = np.zeros(32, dtype=np.uint64)
                            st = _lib.BlockfileStats()
                            t0 = time.perf_counter()
                            _lib.check(file_entry(f.name.encode(), args.threads, out.ctypes.data, ctypes.byref(st)), "blockfile")
                            ts.append(time.perf_counter() - t0)
                            assert np.array_equal(out, want) and st.gpu_decode == dec
                        line += "  %s %7.1f ms = %5.1f Gflags/s (%d threads)" % ("GPU decode" if dec else "host threads", min(ts) * 1e3, n / min(ts) / 1e9, st.threads)
                    print(line, flush=True)
            del img, buf
        _lib.check(lib.FLAGSTATS_hip_set(knob, 2), "set")


if __name__ == "__main__":
    main()
