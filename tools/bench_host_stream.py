#!/usr/bin/env python3
"""BASELINE config 2: 8 GiB NA12878-like FLAG array in PINNED HOST memory, streamed through
FLAGSTATS_u16_x64 (double-buffered hipMemcpyAsync + K1/K2 per chunk).  Roofline here is PCIe
Gen5 x16 (63 GB/s spec), not HBM.  Also times the same call on pageable memory."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib, device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=2 ** 32)
    ap.add_argument("--chunks", default="4194304,16777216,33554432,134217728")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    n = args.flags
    d = device.DeviceFlags(n).generate(device.GEN_NA12878, seed=7, mask=1)
    want = d.count()
    p = lib.FLAGSTATS_hip_host_alloc(2 * n)
    assert p
    _lib.check(lib.FLAGSTATS_hip_memcpy_d2h(p, d.ptr, 2 * n), "d2h")
    d.free()
    rows = []
    for chunk in [int(c) for c in args.chunks.split(",")]:
        lib.FLAGSTATS_hip_set(b"chunk_flags", chunk)
        best = 1e9
        for r in range(args.reps + 1):
            out = np.zeros(32, dtype=np.uint64)
            t0 = time.perf_counter()
            _lib.check(lib.FLAGSTATS_u16_x64(p, n, out.ctypes.data), "x64")
            dt = time.perf_counter() - t0
            assert np.array_equal(out, want)
            if r:
                best = min(best, dt)
        rows.append({"memory": "pinned", "chunk_flags": chunk, "s": round(best, 4), "Gflags_s": round(n / best / 1e9, 2),
                     "GB_s": round(2 * n / best / 1e9, 2)})
        print(rows[-1], flush=True)
    # pageable: a numpy copy of the first 2 GiB
    m = min(n, 2 ** 30)
    a = np.empty(m, dtype=np.uint16)
    ctypes.memmove(a.ctypes.data, p, 2 * m)
    lib.FLAGSTATS_hip_set(b"chunk_flags", 33554432)
    best = 1e9
    for r in range(3):
        out = np.zeros(32, dtype=np.uint64)
        t0 = time.perf_counter()
        _lib.check(lib.FLAGSTATS_u16_x64(a.ctypes.data, m, out.ctypes.data), "x64")
        best = min(best, time.perf_counter() - t0)
    rows.append({"memory": "pageable", "flags": m, "s": round(best, 4), "Gflags_s": round(m / best / 1e9, 2),
                 "GB_s": round(2 * m / best / 1e9, 2)})
    print(rows[-1])
    lib.FLAGSTATS_hip_host_free(p)
    print(json.dumps({"workload": "%d NA12878-like flags from host memory through FLAGSTATS_u16_x64" % n, "rows": rows}))


if __name__ == "__main__":
    main()
