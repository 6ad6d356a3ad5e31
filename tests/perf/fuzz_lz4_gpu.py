#!/usr/bin/env python3
"""Differential fuzz of the GPU LZ4 decoder: many seeds of the synthetic edge streams of tests/test_gpu_blockfile.py
(plus byte flips that must fail cleanly or decode like the host decoder does).  Two levels: the product entries (counters
against the oracle, accept / reject against the product's host decoder) and, per batch of 50 seeds, BOTH decode kernels
launched directly with the decoded bytes compared byte for byte against liblz4 (tests/test_gpu_decode_bytes.py)."""
import argparse
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import test_gpu_blockfile as tb  # noqa: E402
import test_gpu_decode_bytes as tdb  # noqa: E402
from libflagstats_amd import _lib, blockfile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=150)
    args = ap.parse_args()
    import oracle
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    _lib.check(lib.FLAGSTATS_hip_set(b"lz4_decoder", 1), "set")
    blocks = damaged_ok = damaged_rejected = 0
    bytes_level = [[0, 0, 0, 0], [0, 0, 0, 0]]   # per kernel: blocks byte-exact, damaged: both accept / both reject / GPU stricter than liblz4
    for seed in range(1000, 1000 + args.seeds):
        rs = np.random.RandomState(seed)
        style = ("bare", "edges", "mixed")[seed % 3]
        img = bytearray()
        want = np.zeros(32, dtype=np.uint64)
        parts = []
        for target in rs.choice([17, 100, 2000, 9000, 40000, 150000, 600000], size=5):
            comp, dec = tb._synthetic_lz4_block(rs, int(target), style)
            parts.append((comp, dec))
            img += struct.pack("<ii", len(dec), len(comp)) + comp
            want += oracle.flagstat_hist(np.frombuffer(dec[:2 * (len(dec) >> 1)], dtype=np.uint16))
            blocks += 1
        got, st = blockfile.flagstat_lz4_image(bytes(img), 2)
        assert st["gpu_decode"] == 1 and np.array_equal(got, want), ("seed", seed, style)
        # damage: flip bytes of one block's payload; the GPU decoder must agree with the host decoder on accept / reject,
        # and on the counters when both accept
        comp, dec = parts[int(rs.randint(len(parts)))]
        bad = bytearray(comp)
        for _ in range(int(rs.randint(1, 6))):
            bad[int(rs.randint(len(bad)))] ^= int(rs.randint(1, 256))
        host = blockfile.lz4_block_decode(bytes(bad), len(dec))
        one = struct.pack("<ii", len(dec), len(bad)) + bytes(bad)
        try:
            g, _ = blockfile.flagstat_lz4_image(one, 1)
            ok = True
        except _lib.FlagstatsHipError:
            ok = False
        if host is not None and len(host) == len(dec):
            assert ok, ("seed", seed, "host accepts, GPU rejects")
            assert np.array_equal(g, oracle.flagstat_hist(np.frombuffer(host[:2 * (len(host) >> 1)], dtype=np.uint16))), ("seed", seed, "damaged but valid")
            damaged_ok += 1
        else:
            assert not ok, ("seed", seed, "host rejects, GPU accepts")
            damaged_rejected += 1
        if seed % 50 == 49:
            for kernel in (0, 1):
                r = tdb.lz4_fuzz_slice(lib, kernel, seed - 49, 50)
                for k in range(4):
                    bytes_level[kernel][k] += r[k]
            print("seed %d done" % seed, flush=True)    # (a GPU run that prints nothing for minutes is taken to be hung)
    _lib.check(lib.FLAGSTATS_hip_set(b"lz4_decoder", 2), "set")
    print("fuzz: %d seeds, %d synthetic blocks exact on the GPU; damaged blocks: %d still valid (same counters as the host decoder), "
          "%d rejected by both" % (args.seeds, blocks, damaged_ok, damaged_rejected))
    for kernel, name in ((0, "workgroup kernel"), (1, "wave-per-block kernel")):
        b = bytes_level[kernel]
        print("bytes, %s: %d blocks byte-exact against liblz4; damaged payloads: %d decode to liblz4's bytes, %d rejected by both, %d refused "
              "by the GPU decoder only, none accepted that liblz4 rejects" % (name, b[0], b[1], b[2], b[3]))


if __name__ == "__main__":
    main()
