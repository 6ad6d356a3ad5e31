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
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import blockfile_tool as bt  # noqa: E402
from libflagstats_amd import _lib  # noqa: E402
from zstd_kernel_check import decode_frames  # noqa: E402
from zstd_fuzz_gen import compress_with_parameters, damage, synthetic  # noqa: E402


def ref_decode(z, comp, n):
    dst = ctypes.create_string_buffer(max(n, 1))
    r = z.ZSTD_decompress(dst, n, bytes(comp), len(comp))
    if z.ZSTD_isError(r) or r != n:
        return None
    return dst.raw[:n]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    z = bt.zstd()
    exact = wrong = unsupported = second_pass = 0
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
        # the product's second pass: frames that ran out of block slots (status 67) once more with a slot per KiB
        again = [i for i in range(len(frames)) if st[i] == 67]
        if again:
            st = st.copy()
            g2, s2, _, _ = decode_frames(lib, [frames[i] for i in again], [sizes[i] for i in again], min_blocks=max(sizes[i] for i in again) // 1024 + 16)
            for k, i in enumerate(again):
                got[i], st[i] = g2[k], s2[k]
            second_pass += len(again)
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
    print("%d synthetic frames (a third of them from ZSTD_compress2 with random advanced parameters): %d exact (%d of them in the second pass over frames of many blocks), %d not taken (status >= 64), %d wrong | %d damaged frames: both reject %d, both accept with equal bytes %d, GPU stricter %d, GPU more lenient %d, different bytes %d" % (
        exact + wrong + unsupported, exact, second_pass, unsupported, wrong, both_ok + both_fail + strict + lenient + differ, both_fail, both_ok, strict, lenient, differ))
    return 1 if wrong or lenient or differ else 0


if __name__ == "__main__":
    sys.exit(main())
