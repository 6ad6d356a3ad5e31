#!/usr/bin/env python3
"""GPU decode of ONE block-file image (default Zstandard level 1, 2^31 NA12878-like flags) through the product entry, over
the number of pieces the host side cuts the file into (env FLAGSTATS_HIP_GPU_LZ4_CHUNKS, read per call; 0 = the shipped rule).
The image is built once and cached under /tmp, so that measurement builds of the library (FLAGSTATS_HIP_LIB) see the same bytes."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402
from lz4_decoder_sweep import build_image  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", default="2**31")
    ap.add_argument("--mode", default="zstd:1")
    ap.add_argument("--pieces", default="0,3,4,5,6,8")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--pinned", type=int, default=0, help="1: the image in page-locked host memory (the runtime then copies it with the DMA engines, not with a staged shader copy)")
    ap.add_argument("--first", default="0", help="first piece as a percentage of an equal share (100: equal pieces; 0: the shipped rule), comma list")
    ap.add_argument("--last", default="0", help="last piece as a percentage of a middle one (0 / 100: like the others), comma list")
    args = ap.parse_args()
    import oracle
    n = int(eval(args.flags))
    mode, level = args.mode.split(":")
    cache = "/tmp/blockimage_%s_%s_%d.bin" % (mode, level, n)
    if os.path.exists(cache):
        buf = np.fromfile(cache, dtype=np.uint8)
    else:
        buf = np.frombuffer(build_image(n, mode, int(level)), dtype=np.uint8)
        buf.tofile(cache)
    want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, n)
    if args.pinned:
        import torch
        keep = torch.empty(buf.size, dtype=torch.uint8).pin_memory()
        keep.numpy()[:] = buf
        buf = keep.numpy()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    zstd = mode == "zstd"
    _lib.check(lib.FLAGSTATS_hip_set(b"zstd_decoder" if zstd else b"lz4_decoder", 1), "set")
    entry = lib.FLAGSTATS_hip_blockimage_zstd if zstd else lib.FLAGSTATS_hip_blockimage_lz4
    print("%s  %s-%s, %d flags, %.0f MiB%s" % (os.path.basename(_lib.LIB_PATH), mode, level, n, buf.size / 2**20, ", image page-locked" if args.pinned else ""), flush=True)
    for pc, fp, lp in [(int(x), int(y), int(z)) for x in args.pieces.split(",") for y in args.first.split(",") for z in args.last.split(",")]:
        if lp:
            os.environ["FLAGSTATS_HIP_GPU_LAST_PIECE"] = str(lp)
        else:
            os.environ.pop("FLAGSTATS_HIP_GPU_LAST_PIECE", None)
        if fp:
            os.environ["FLAGSTATS_HIP_GPU_FIRST_PIECE"] = str(fp)
        else:
            os.environ.pop("FLAGSTATS_HIP_GPU_FIRST_PIECE", None)
        if pc:
            os.environ["FLAGSTATS_HIP_GPU_LZ4_CHUNKS"] = str(pc)
        else:
            os.environ.pop("FLAGSTATS_HIP_GPU_LZ4_CHUNKS", None)
        ts = []
        for rep in range(args.reps + 1):
            out = np.zeros(32, dtype=np.uint64)
            st = _lib.BlockfileStats()
            t0 = time.perf_counter()
            _lib.check(entry(buf.ctypes.data, buf.size, 0, out.ctypes.data, ctypes.byref(st)), "blockimage")
            ts.append(time.perf_counter() - t0)
            assert np.array_equal(out, want) and st.gpu_decode == 1
        ts = ts[1:]
        print("   pieces %-8s first %3d %% (0 = rule) last %3d %%  best %6.1f ms  median %6.1f ms  = %5.1f Gflags/s  (%d pieces ran)" % (pc or "rule", fp, lp, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, n / min(ts) / 1e9, st.chunks), flush=True)


if __name__ == "__main__":
    main()
