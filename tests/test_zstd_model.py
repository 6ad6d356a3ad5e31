"""The two Python restatements behind the GPU Zstandard decoder -- tests/zstd_model.py (RFC 8878) and
tests/zstd_gpu_model.py (the kernels' record / checkpoint layout and bit arithmetic) -- pinned against the image's libzstd,
the library the reference links (benchmark/flagstats.cpp:636-682), and against the .zst files the reference's own writer made."""
import ctypes
import json
import os
import random
import struct
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import blockfile_tool as bt  # noqa: E402
import zstd_gpu_model as gm  # noqa: E402
import zstd_model as zm  # noqa: E402

try:
    Z = bt.zstd()
except OSError:  # pragma: no cover
    Z = None
pytestmark = pytest.mark.skipif(Z is None, reason="no libzstd.so.1")


def ref_decode(comp, n):
    dst = ctypes.create_string_buffer(max(n, 1))
    r = Z.ZSTD_decompress(dst, n, bytes(comp), len(comp))
    return None if Z.ZSTD_isError(r) else dst.raw[:r]


def inputs():
    import oracle
    r = np.random.default_rng(5)
    a = os.urandom(20000)
    return {
        "empty": b"", "one": b"a", "abc": b"abc" * 5, "zeros": bytes(150000), "random": os.urandom(3000),
        "text": b"hello world, " * 2000,
        "na12878": oracle.generate(oracle.GEN_NA12878, 7, 1, 0, 150000).tobytes(),       # three blocks: repeat modes, treeless literals
        "u8_4": r.integers(0, 4, 140000, dtype=np.uint8).tobytes(),
        "u16": r.integers(0, 3000, 40000, dtype=np.uint16).tobytes(),
        "longruns": a + bytes(40000) + a + os.urandom(5000) + a[:15000] + bytes(70000),   # literal runs and matches above 16,383
    }


@pytest.mark.parametrize("level", [1, 3, 9, 19, -5])
def test_models_decode_what_libzstd_wrote(level):
    for name, raw in inputs().items():
        comp = bt.compress_block(raw, "zstd", level)
        assert ref_decode(comp, len(raw)) == raw
        assert zm.decode_frame(comp) == raw, (name, level)
        stage = gm.entropy_stage(comp, len(raw))
        assert gm.execute_stage(stage, len(raw)) == raw, (name, level)
        # what the execution kernel relies on: records never empty, checkpoints monotone, slots of 64
        at = 0
        for s in range(stage["nslots"]):
            o, lp, nv = stage["ck"][s]
            assert o >= at and nv <= 64
            at = o
            for j in range(nv):
                w = stage["recs"][s * 64 + j]
                assert not (w & gm.REC_REP) and ((w >> 32) & 16383) + ((w >> 46) & 16383) > 0


def test_models_decode_the_reference_written_files():
    manifest = json.load(open(os.path.join(HERE, "golden", "blockfiles", "manifest.json")))
    seen = 0
    for name, e in manifest["files"].items():
        if e.get("codec") != "zstd":
            continue
        img = open(os.path.join(HERE, "golden", "blockfiles", name), "rb").read()
        pos = 0
        while pos < len(img):
            us, cs = struct.unpack_from("<ii", img, pos)
            frame = img[pos + 8:pos + 8 + cs]
            want = ref_decode(frame, us)
            assert want is not None and len(want) == us
            assert zm.decode_frame(frame) == want and gm.decode(frame, us) == want, name
            pos += 8 + cs
            seen += 1
    assert seen >= 3


def test_damaged_frames_never_decode_to_something_else():
    """Bit flips: whatever libzstd rejects the models reject; what both accept decodes to the same bytes.  (The models may be
    stricter: the product then decodes the file with libzstd.)"""
    import oracle
    rng = random.Random(7)
    raws = [oracle.generate(oracle.GEN_NA12878, 7, 1, 0, 20000).tobytes(), b"hello world, " * 300 + os.urandom(500) + bytes(3000)]
    strict = 0
    for raw in raws:
        for level in (1, 19):
            comp = bt.compress_block(raw, "zstd", level)
            for _ in range(60):
                bad = bytearray(comp)
                for _ in range(rng.randrange(1, 4)):
                    bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
                want = ref_decode(bad, len(raw))
                if want is not None and len(want) != len(raw):
                    want = None
                try:
                    got = zm.decode_frame(bytes(bad))
                    got = got if len(got) == len(raw) else None
                except zm.ZstdError:
                    got = None
                try:
                    got2 = gm.decode(bytes(bad), len(raw))
                except gm.Fail:
                    got2 = None
                for g in (got, got2):
                    if want is None:
                        assert g is None
                    elif g is None:
                        strict += 1
                    else:
                        assert g == want
    assert strict < 40


def test_fse_table_built_cell_by_cell_equals_the_serial_spread():
    """zstd_prepare builds its FSE tables with the whole wave (fse_build_wave, flagstat_zstd_kernels.hip): cell u is visited by the
    spread at k = u * step^-1 mod S, receives slot k - (skipped visits before k), the slot's symbol comes from a max-scan over run
    starts, and a symbol's cells take its state numbers in order of u, 64 cells at a time.  Restated here and compared with the
    serial construction of RFC 8878 4.1.1 over random normalised counts of every accuracy log, with "less than one" symbols."""
    import random

    def serial(counts, log):
        size = 1 << log
        table, high = [None] * size, size - 1
        for s, c in enumerate(counts):
            if c == -1:
                table[high] = s
                high -= 1
        step, pos = (size >> 1) + (size >> 3) + 3, 0
        for s, c in enumerate(counts):
            for _ in range(max(c, 0)):
                table[pos] = s
                pos = (pos + step) & (size - 1)
                while pos > high:
                    pos = (pos + step) & (size - 1)
        assert pos == 0
        nxt = [1 if c == -1 else c for c in counts]
        out = []
        for u in range(size):
            s = table[u]
            out.append((s, nxt[s]))
            nxt[s] += 1
        return out

    def by_cell(counts, log):
        size = 1 << log
        mask = size - 1
        low = [s for s, c in enumerate(counts) if c == -1]
        high = size - 1 - len(low)
        n = [max(c, 0) for c in counts]
        slotsym, at = [0] * size, 0
        for s, k in enumerate(n):
            if k:
                slotsym[at] = s + 1
            at += k
        assert at == high + 1
        for i in range(1, size):
            slotsym[i] = max(slotsym[i], slotsym[i - 1])
        step = (size >> 1) + (size >> 3) + 3
        inv = step
        for _ in range(3):
            inv = (inv * (2 - step * inv)) & 0xFFFFFFFF
        assert (inv * step) & mask == 1
        skipk = [((size - 1 - r) * inv) & mask for r in range(len(low))]
        run, out = list(n), [None] * size
        for base in range(0, size, 64):
            seen = {}
            for u in range(base, min(base + 64, size)):
                if u > high:
                    out[u] = (low[size - 1 - u], 1)
                    continue
                k = (u * inv) & mask
                sym = slotsym[k - sum(1 for v in skipk if v < k)] - 1
                out[u] = (sym, run[sym] + seen.get(sym, 0))
                seen[sym] = seen.get(sym, 0) + 1
            for sym, c in seen.items():
                run[sym] += c
        return out

    rng = random.Random(1)
    done = 0
    while done < 1500:
        log = rng.choice([5, 6, 7, 8, 9])
        size, nsym = 1 << log, rng.randint(2, 53)
        lows = set(rng.sample(range(nsym), rng.randint(0, min(nsym - 1, size // 4))))
        others = [s for s in range(nsym) if s not in lows]
        pick = rng.sample(others, rng.randint(1, len(others)))
        rest = size - len(lows) - len(pick)
        if rest < 0:
            continue
        counts = [-1 if s in lows else (1 if s in pick else 0) for s in range(nsym)]
        for _ in range(rest):
            counts[rng.choice(pick)] += 1
        assert serial(counts, log) == by_cell(counts, log), (counts, log)
        done += 1
