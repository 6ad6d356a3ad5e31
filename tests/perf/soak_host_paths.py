#!/usr/bin/env python3
"""Soak of the host-side paths that share the default engine, mixed and from two threads at once: large pageable arrays (the staged
rule), small calls (polled result pairs, side engines), block files on host threads (ramped chunks) and on the GPU decoders.  Every
result is checked; a line every 25 rounds (a run that stops printing is a hang)."""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import blockfile_tool as bt  # noqa: E402
import oracle  # noqa: E402
from libflagstats_amd import _lib, blockfile, pyflagstats  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=300)
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    big = [oracle.generate(oracle.GEN_NA12878, 100 + i, 1, 0, n) for i, n in enumerate(((1 << 27) + 5, (1 << 28) + 12345, 3 * (1 << 27) + 1))]
    big_want = [oracle.flagstat_generated(oracle.GEN_NA12878, 100 + i, 1, 0, a.size) for i, a in enumerate(big)]
    small = oracle.generate(oracle.GEN_NA12878, 9, 1, 0, 700000)
    files = []
    for mode, level, n in (("fast", 2, 512000 * 9 + 3), ("zstd", 1, 512000 * 7), ("hc", 9, 90_000_000), ("zstd", 1, 90_000_000)):
        fl = oracle.generate(oracle.GEN_NA12878, 7, 1, 0, n)
        img = bt.block_file_image(fl, mode=mode, level=level)
        entry = blockfile.flagstat_zstd_image if mode == "zstd" else blockfile.flagstat_lz4_image
        files.append((entry, img, oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, n)))
    errors = []
    t_start = time.perf_counter()

    def small_calls():
        k = 0
        while not stop.is_set():
            lo = (k * 7919) % 600000
            m = 1 + (k * 104729) % 90000
            got = pyflagstats.counters_u64(small[lo:lo + m])
            if not np.array_equal(got, oracle.flagstat_hist(small[lo:lo + m])):
                errors.append(("small", k))
            k += 1
        counts["small"] = k

    stop = threading.Event()
    counts = {}
    th = threading.Thread(target=small_calls)
    th.start()
    staged0 = lib.FLAGSTATS_hip_get(b"staged_calls")
    worst = {}
    try:
        for r in range(args.rounds):
            i = r % 3
            t0 = time.perf_counter()
            got = pyflagstats.counters_u64(big[i])
            dt = (time.perf_counter() - t0) * 1e3
            worst["array %d" % i] = max(worst.get("array %d" % i, 0), dt)
            if not np.array_equal(got, big_want[i]):
                errors.append(("big", r))
            entry, img, want = files[r % len(files)]
            t0 = time.perf_counter()
            got, st = entry(img, 0)
            dt = (time.perf_counter() - t0) * 1e3
            worst["file %d" % (r % len(files))] = max(worst.get("file %d" % (r % len(files)), 0), dt)
            if not np.array_equal(got, want):
                errors.append(("file", r))
            if r % 25 == 24:
                print("round %d, %.0f s, %d errors, staged calls %d, worst ms %s" % (r + 1, time.perf_counter() - t_start, len(errors), lib.FLAGSTATS_hip_get(b"staged_calls") - staged0,
                                                                                 {k: round(v, 1) for k, v in sorted(worst.items())}), flush=True)
    finally:
        stop.set()
        th.join()
    print("soak: %d rounds, %d small calls beside them, %d errors" % (args.rounds, counts.get("small", 0), len(errors)), flush=True)
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
