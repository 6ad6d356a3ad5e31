#!/usr/bin/env python3
"""Four caller threads for a while, each a different mix of entry points on the default engine -- Zstandard and LZ4 block images
through the GPU decoders and the host pipelines, host-pointer counting calls of several sizes -- every result checked against
the oracle.  The GPU decoders take the engine's lock for a whole file; the counting calls go to side engines meanwhile."""
import argparse
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402
from lz4_decoder_sweep import build_image  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--flags", type=int, default=2 ** 27)
    args = ap.parse_args()
    import oracle
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    lib.FLAGSTATS_hip_set(b"on_error", 0)
    n = args.flags
    want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, n)
    imgs = {"zstd": np.frombuffer(build_image(n, "zstd", 1), dtype=np.uint8), "lz4": np.frombuffer(build_image(n, "fast", 2), dtype=np.uint8)}
    arrays = {k: oracle.generate(oracle.GEN_UNIFORM, 3, 0x0FFF, 0, k) for k in (1000, 1 << 18, 1 << 24)}
    wants = {k: oracle.flagstat_hist(a)[:32] for k, a in arrays.items()}
    _lib.check(lib.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 1), "set")
    _lib.check(lib.FLAGSTATS_hip_set(b"lz4_gpu_min_bytes", 1), "set")
    stop = time.time() + args.seconds
    counts, errors = {}, []

    def block_files(codec, decoder):
        entry = lib.FLAGSTATS_hip_blockimage_zstd if codec == "zstd" else lib.FLAGSTATS_hip_blockimage_lz4
        buf = imgs[codec]
        k = 0
        while time.time() < stop:
            out = np.zeros(32, dtype=np.uint64)
            st = _lib.BlockfileStats()
            rc = entry(buf.ctypes.data, buf.size, 4, out.ctypes.data, ctypes.byref(st))
            if rc or not np.array_equal(out, want):
                errors.append("%s image: rc %d, counters %s" % (codec, rc, "equal" if np.array_equal(out, want) else "DIFFER"))
                return
            k += 1
        counts[codec] = k

    def counting(tag):
        k = 0
        while time.time() < stop:
            for size, a in arrays.items():
                out = np.zeros(32, dtype=np.uint32)
                rc = lib.FLAGSTATS_u16(a.ctypes.data, size, out.ctypes.data)
                if rc or not np.array_equal(out.astype(np.uint64), wants[size].astype(np.uint64)):
                    errors.append("%s: FLAGSTATS_u16(%d) rc %d" % (tag, size, rc))
                    return
                k += 1
        counts[tag] = k

    threads = [threading.Thread(target=block_files, args=("zstd", 1)), threading.Thread(target=block_files, args=("lz4", 1)),
               threading.Thread(target=counting, args=("counting-a",)), threading.Thread(target=counting, args=("counting-b",))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    _lib.check(lib.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 64 << 20), "set")
    _lib.check(lib.FLAGSTATS_hip_set(b"lz4_gpu_min_bytes", 64 << 20), "set")
    print("stress: %.0f s, four threads: %s; errors: %s" % (args.seconds, ", ".join("%s %d calls" % kv for kv in sorted(counts.items())), errors or "none"))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
