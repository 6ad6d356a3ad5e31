#!/usr/bin/env python3
"""Differential fuzz of the GPU Zstandard decoder against the image's libzstd: synthetic frames of every level and shape
must decode byte for byte; damaged frames (bit flips, truncations, spliced sections) must be rejected whenever libzstd
rejects them, and decode to the same bytes whenever both accept (the GPU decoder may be stricter: the product then takes
the libzstd host pipeline).  Many frames per launch, called through fsk_zstd_decode."""
import argparse
import ctypes
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import numpy as np  # noqa: E402

import blockfile_tool as bt  # noqa: E402
from libflagstats_amd import _lib  # noqa: E402
from zstd_kernel_check import decode_frames  # noqa: E402


def ref_decode(z, comp, n):
    dst = ctypes.create_string_buffer(max(n, 1))
    r = z.ZSTD_decompress(dst, n, bytes(comp), len(comp))
    if z.ZSTD_isError(r) or r != n:
        return None
    return dst.raw[:n]


def synthetic(rng, nrng):
    import oracle
    kind = rng.randrange(9)
    n = rng.choice([0, 1, 2, 7, 100, 4097, 70000, 131072, 131073, 262144 + 5, 400000])
    if kind == 0:
        return oracle.generate(oracle.GEN_NA12878, rng.randrange(1000), 1, 0, (n + 1) // 2).tobytes()[:n]
    if kind == 1:
        return nrng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 2:
        return bytes([rng.randrange(256)]) * n
    if kind == 3:
        return nrng.integers(0, rng.choice([2, 4, 17, 60]), n, dtype=np.uint8).tobytes()
    if kind == 4:
        a = nrng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
        return (a + bytes(rng.randrange(1, 60000)) + a[:rng.randrange(1, 30000)] + nrng.integers(0, 256, rng.randrange(1, 20000), dtype=np.uint8).tobytes() + a)[:max(n, 50000)]
    if kind == 5:
        word = bytes(nrng.integers(0, 256, rng.randrange(1, 40), dtype=np.uint8))
        return (word * (n // len(word) + 1))[:n]
    if kind == 6:
        return nrng.integers(0, 3000, (n + 1) // 2, dtype=np.uint16).tobytes()[:n]
    if kind == 7:
        # far matches: a long random stretch repeated at a distance above the 64 KiB the execution kernel keeps in LDS
        a = nrng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
        return a + nrng.integers(0, 256, 90000, dtype=np.uint8).tobytes() + a + bytes(1000) + a[100:4000]
    return oracle.generate(oracle.GEN_UNIFORM, rng.randrange(1000), 0x0FFF, 0, (n + 1) // 2).tobytes()[:n]


def compress_with_parameters(z, rng, raw):
    """ZSTD_compress2 with random advanced parameters (strategy, window / hash / chain / search logs, minimum match, target
    length, long-distance matching): frames ZSTD_compress would never write at any level"""
    z.ZSTD_createCCtx.restype = ctypes.c_void_p
    z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
    z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    z.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
    z.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    z.ZSTD_compress2.restype = ctypes.c_size_t
    cctx = z.ZSTD_createCCtx()
    # ZSTD_cParameter: compressionLevel 100, windowLog 101, hashLog 102, chainLog 103, searchLog 104, minMatch 105, targetLength 106,
    # strategy 107, enableLongDistanceMatching 160, ldmHashLog 161, ldmMinMatch 162, contentSizeFlag 200, checksumFlag 201
    picks = [(100, rng.choice([1, 3, 6, 12, 19])), (107, rng.randrange(1, 10)), (101, rng.randrange(10, 24)), (105, rng.randrange(3, 8)),
             (106, rng.choice([0, 4, 16, 64, 999])), (102, rng.randrange(6, 20)), (103, rng.randrange(6, 20)), (104, rng.randrange(1, 8)),
             (160, rng.randrange(2)), (200, rng.randrange(2)), (201, 0)]
    for key, value in picks:
        if rng.randrange(3):
            z.ZSTD_CCtx_setParameter(cctx, key, value)   # (out-of-range combinations are refused by the library: ignored here)
    bound = z.ZSTD_compressBound(len(raw))
    dst = ctypes.create_string_buffer(bound)
    n = z.ZSTD_compress2(cctx, dst, bound, raw, len(raw))
    z.ZSTD_freeCCtx(cctx)
    if z.ZSTD_isError(n):
        return None
    return dst.raw[:n]


def damage(rng, comp):
    bad = bytearray(comp)
    how = rng.randrange(4)
    if how == 0 or len(bad) < 20:
        for _ in range(rng.randrange(1, 4)):
            bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
    elif how == 1:
        del bad[rng.randrange(len(bad)):]
    elif how == 2:
        i = rng.randrange(len(bad))
        bad[i:i + rng.randrange(1, 8)] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 8)))
    else:
        i, j = sorted((rng.randrange(len(bad)), rng.randrange(len(bad))))
        bad[i:j] = bad[i:j][::-1]
    return bytes(bad) if bad else b"\0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    z = bt.zstd()
    exact = wrong = unsupported = 0
    both_ok = both_fail = strict = lenient = differ = 0
    batch = 50
    for s0 in range(args.first, args.first + args.seeds, batch):
        frames, sizes, raws, bad_frames, bad_sizes, wants = [], [], [], [], [], []
        for seed in range(s0, min(s0 + batch, args.first + args.seeds)):
            rng = random.Random(seed)
            nrng = np.random.default_rng(seed)
            raw = synthetic(rng, nrng)
            level = rng.choice([1, 1, 2, 3, 5, 7, 9, 12, 15, 19, -1, -5])
            comp = compress_with_parameters(z, rng, raw) if seed % 3 == 2 else None
            if comp is None:
                comp = bt.compress_block(raw, "zstd", level)
            frames.append(comp)
            sizes.append(len(raw))
            raws.append(raw)
            for _ in range(3):
                bad = damage(rng, comp)
                bad_frames.append(bad)
                bad_sizes.append(len(raw))
                wants.append(ref_decode(z, bad, len(raw)))
        got, st, _, _ = decode_frames(lib, frames, sizes)
        for i, raw in enumerate(raws):
            even = len(raw) & ~1
            if st[i] == 0 and got[i][:even] == raw[:even]:
                exact += 1
            elif st[i] >= 64 and st[i] != 0xFFFFFFFF:
                unsupported += 1     # valid Zstandard the decoder does not take (e.g. a 1 KiB window: hundreds of blocks): libzstd's on the host
            else:
                wrong += 1
                print("WRONG: seed %d, %d bytes, status %d" % (s0 + i, len(raw), st[i]), flush=True)
        got, st, _, _ = decode_frames(lib, bad_frames, bad_sizes)
        for i, want in enumerate(wants):
            even = bad_sizes[i] & ~1
            ok = st[i] == 0
            if want is None and not ok:
                both_fail += 1
            elif want is None:
                lenient += 1
                print("LENIENT: seed %d (damaged frame %d): libzstd rejects, the GPU decoder accepts" % (s0 + i // 3, i), flush=True)
            elif not ok:
                strict += 1
            elif got[i][:even] == want[:even]:
                both_ok += 1
            else:
                differ += 1
                print("DIFFER: seed %d (damaged frame %d)" % (s0 + i // 3, i), flush=True)
        print("seeds %d..%d done: %d exact so far" % (s0, min(s0 + batch, args.first + args.seeds) - 1, exact), flush=True)
    print("%d synthetic frames (a third of them from ZSTD_compress2 with random advanced parameters): %d exact, %d not taken (status >= 64), %d wrong | %d damaged frames: both reject %d, both accept with equal bytes %d, GPU stricter %d, GPU more lenient %d, different bytes %d" % (
        exact + wrong + unsupported, exact, unsupported, wrong, both_ok + both_fail + strict + lenient + differ, both_fail, both_ok, strict, lenient, differ))
    return 1 if wrong or lenient or differ else 0


if __name__ == "__main__":
    sys.exit(main())
