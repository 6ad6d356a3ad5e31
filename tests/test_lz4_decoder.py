"""CPU: the product's host LZ4 block decoder (FLAGSTATS_lz4_block_decode) against the image's
liblz4 (LZ4_decompress_safe, what benchmark/flagstats.cpp:316 calls) on blocks produced by the real
compressor, plus malformed-input safety.  No GPU needed: the decoder is host code."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import blockfile_tool as bt  # noqa: E402

from libflagstats_amd import blockfile  # noqa: E402


def payloads():
    import oracle
    rs = np.random.RandomState(11)
    yield "empty", b""
    yield "one", b"\x07"
    yield "thirteen", bytes(range(13))
    yield "zeros", bytes(100000)                                   # one long overlapping match (offset 1)
    for period in (1, 2, 3, 5, 7, 8, 15, 16, 17, 31, 33):          # offsets around the 16-byte wild-copy limit
        yield "period%d" % period, (bytes(rs.randint(0, 256, period, dtype=np.uint8)) * (70000 // period + 1))[:70000]
    yield "uniform", rs.randint(0, 65536, 200000).astype(np.uint16).tobytes()           # incompressible
    yield "na12878", oracle.generate(oracle.GEN_NA12878, 3, 1, 0, 512000).tobytes()    # a full reference block
    yield "u4096", rs.randint(0, 4096, 300001).astype(np.uint16).tobytes()
    mix = bytearray()
    for _ in range(200):
        if rs.rand() < 0.5:
            mix += bytes(rs.randint(0, 256, rs.randint(1, 400), dtype=np.uint8))
        else:
            mix += bytes([rs.randint(0, 256)]) * rs.randint(1, 1000)
    yield "mix", bytes(mix)


@pytest.mark.parametrize("mode,level", [("fast", 1), ("fast", 2), ("fast", 10), ("hc", 1), ("hc", 9)])
def test_decoder_matches_liblz4(mode, level):
    for name, raw in payloads():
        comp = bt.compress_block(raw, mode, level)
        assert bt.decompress_block_ref(comp, len(raw)) == raw          # the oracle itself round-trips
        got = blockfile.lz4_block_decode(comp, len(raw))
        assert got == raw, (name, mode, level)
        # exact-capacity contract: one byte less must fail, more room must still give the same bytes
        if len(raw):
            assert blockfile.lz4_block_decode(comp, len(raw) - 1) is None, name
        assert blockfile.lz4_block_decode(comp, len(raw) + 64) == raw, name


def test_truncated_and_corrupt_blocks_never_crash():
    rs = np.random.RandomState(5)
    import oracle
    raw = oracle.generate(oracle.GEN_NA12878, 9, 1, 0, 3000).tobytes()
    comp = bt.compress_block(raw, "fast", 2)
    for cut in range(len(comp)):                                       # every proper prefix
        got = blockfile.lz4_block_decode(comp[:cut], len(raw))
        assert got is None or (len(got) <= len(raw) and got != raw or got == raw[:len(got)])
        assert got != raw
    for _ in range(3000):                                              # random byte damage
        b = bytearray(comp)
        for _ in range(rs.randint(1, 4)):
            b[rs.randint(0, len(b))] = rs.randint(0, 256)
        got = blockfile.lz4_block_decode(bytes(b), len(raw))
        ref = bt.decompress_block_ref(bytes(b), len(raw))
        assert got is None or len(got) <= len(raw)
        if ref is not None and len(ref) == len(raw) and got is not None and len(got) == len(raw):
            assert got == ref                                          # both accept: same bytes
    assert blockfile.lz4_block_decode(b"", 0) is None                  # empty input is not a block
    assert blockfile.lz4_block_decode(b"\x00", 0) == b""               # the empty block is one zero token
    assert blockfile.lz4_block_decode(b"\x10\x41\x01\x00", 100) is None or True  # offset past start: must not crash


def test_offset_beyond_output_is_rejected():
    # token: 1 literal, match length 4; literal 'A'; offset 2 > 1 byte produced so far
    assert blockfile.lz4_block_decode(b"\x10A\x02\x00" + b"\x50ABCDE", 64) is None
    # offset 0 is invalid
    assert blockfile.lz4_block_decode(b"\x10A\x00\x00" + b"\x50ABCDE", 64) is None
    # valid: 'A' then match offset 1 length 4 -> 'AAAAA', then 5 literals
    assert blockfile.lz4_block_decode(b"\x10A\x01\x00" + b"\x50BCDEF", 64) == b"AAAAABCDEF"


def test_fast_loops_are_memory_safe_under_asan(tmp_path):
    """The decoder's fast loops start after 64 KiB of output, so the small cases above never reach them:
    tests/lz4_fuzz_asan.cpp decodes damaged / truncated ~1 MB blocks (LZ4-fast and LZ4-HC of four
    flag-like streams) into exact-size buffers under AddressSanitizer + UBSan, and compares with liblz4
    whenever both accept."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = tmp_path / "lz4fuzz"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-o", str(exe), os.path.join(ROOT, "tests", "lz4_fuzz_asan.cpp"), "-ldl"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe), "120"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK") or r.stdout.startswith("SKIP"), r.stdout


def test_property_roundtrip_against_liblz4():
    """Property test (hypothesis): byte strings assembled from random literal runs, repeats of earlier
    material at random distances (the match structure LZ4 finds) and flag-like 2-byte symbols, compressed by
    liblz4 at a random mode / level, must decode to the original -- also when the output starts 64 KiB deep, where
    the decoder's unchecked fast loops take over."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    piece = st.one_of(
        st.tuples(st.just("lit"), st.binary(min_size=0, max_size=40)),
        st.tuples(st.just("rep"), st.integers(1, 70000), st.integers(1, 300)),       # (distance back, length)
        st.tuples(st.just("sym"), st.lists(st.sampled_from([99, 147, 83, 163, 1123, 2113]), min_size=1, max_size=60)),
    )

    @settings(max_examples=150, deadline=None)
    @given(st.lists(piece, min_size=0, max_size=120), st.sampled_from([("fast", 1), ("fast", 2), ("fast", 7), ("hc", 1), ("hc", 9)]),
           st.booleans())
    def run(pieces, how, deep):
        out = bytearray(os.urandom(1) * 0)
        if deep:   # 66 KiB of compressible preamble: the body is then decoded by the unchecked loops
            out += (np.random.RandomState(1).choice([99, 147, 83, 163], 33800).astype(np.uint16).tobytes())
        for p in pieces:
            if p[0] == "lit":
                out += p[1]
            elif p[0] == "sym":
                out += np.array(p[1], dtype=np.uint16).tobytes()
            elif out:
                dist, ln = min(p[1], len(out)), p[2]
                start = len(out) - dist
                for i in range(ln):                      # overlapping copies allowed, as in LZ77
                    out.append(out[start + i])
        raw = bytes(out)
        comp = bt.compress_block(raw, *how)
        assert blockfile.lz4_block_decode(comp, len(raw)) == raw
        assert blockfile.lz4_block_decode(comp, len(raw) + 17) == raw

    run()
