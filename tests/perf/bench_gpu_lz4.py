#!/usr/bin/env python3
"""LZ4 block decode ON the GPU (flagstat_gpu_decode.hip: compressed image over PCIe, one wave per block)
against the product's host pipeline (threaded host decode into pinned chunks, decoded bytes over PCIe) on the same
NA12878-like block image, LZ4-fast and LZ4-HC-9.  Counters of both are checked against the oracle."""
import argparse
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
from libflagstats_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=2 ** 31)
    ap.add_argument("--modes", default="fast:2,hc:9")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--configs", default="8/1/1,8/8/4", help="ring KiB / pieces / decode streams, comma separated")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import oracle

    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    n = args.flags
    per = bt.BLOCK_BYTES // 2
    nblocks = (n + per - 1) // per
    want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, n)
    for mode_level in args.modes.split(","):
        mode, level = mode_level.split(":")

        def make(i):
            f = oracle.generate(oracle.GEN_NA12878, 7, 1, i * per, min(per, n - i * per))
            comp = bt.compress_block(f.tobytes(), mode, int(level))
            return struct.pack("<ii", f.nbytes, len(comp)) + comp

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
            img = b"".join(ex.map(make, range(nblocks)))
        print("LZ4-%s-%s image: %d flags, %d blocks, %d -> %d bytes (ratio %.2f), built in %.0f s"
              % (mode, level, n, nblocks, 2 * n, len(img), 2 * n / len(img), time.perf_counter() - t0), flush=True)
        buf = np.frombuffer(img, dtype=np.uint8)
        for ring, chunks, streams in [tuple(int(v) for v in c.split("/")) for c in args.configs.split(",")]:
            os.environ["FLAGSTATS_HIP_GPU_LZ4_RING"] = str(ring)
            os.environ["FLAGSTATS_HIP_GPU_LZ4_CHUNKS"] = str(chunks)
            os.environ["FLAGSTATS_HIP_GPU_LZ4_STREAMS"] = str(streams)
            for rep in range(args.reps):
                out = np.zeros(32, dtype=np.uint64)
                st = _lib.GpuLz4Stats()
                t0 = time.perf_counter()
                _lib.check(lib.FLAGSTATS_hip_blockimage_lz4_gpu(buf.ctypes.data, buf.size, out.ctypes.data, ctypes.byref(st)),
                           "FLAGSTATS_hip_blockimage_lz4_gpu")
                wall = time.perf_counter() - t0
                assert np.array_equal(out, want), "GPU decode: counters differ from the oracle"
                print("  GPU decode (ring %d KiB, %d pieces on %d streams) rep %d: wall %.1f ms (incl. allocations) | copies %.1f ms, decode "
                      "after the last copy %.1f ms, K1 %.2f ms | %.1f M sequences, %.2f %% of the matches behind the ring -> pipeline "
                      "%.1f ms = %.1f Gflags/s" % (st.ring_kib, st.chunks, streams, rep, wall * 1e3, st.h2d_ms, st.decode_ms, st.count_ms,
                                                   st.sequences / 1e6, 100.0 * st.far_matches / max(1, st.sequences), st.pipeline_ms,
                                                   n / st.pipeline_ms / 1e6), flush=True)
        _lib.check(lib.FLAGSTATS_hip_set(b"lz4_decoder", 0), "set lz4_decoder")   # the host-thread pipeline, whatever the size
        for rep in range(3):
            got = np.zeros(32, dtype=np.uint64)
            hs = _lib.BlockfileStats()
            _lib.check(lib.FLAGSTATS_hip_blockimage_lz4(buf.ctypes.data, buf.size, args.threads, got.ctypes.data, ctypes.byref(hs)),
                       "FLAGSTATS_hip_blockimage_lz4")
            assert np.array_equal(got, want), "host pipeline: counters differ from the oracle"
            print("  host pipeline rep %d: %.1f ms -> %.1f Gflags/s (%d decoder threads, waiting for copies %.0f %% of the time)"
                  % (rep, hs.wall_s * 1e3, n / hs.wall_s / 1e9, hs.threads, 100.0 * hs.wait_copy_s / max(hs.wall_s, 1e-9)), flush=True)
        _lib.check(lib.FLAGSTATS_hip_set(b"lz4_decoder", 2), "set lz4_decoder")


if __name__ == "__main__":
    main()
