"""The two GPU block decoders are byte-exact functions, and this file checks them as such: the decode kernels are launched
directly (fsk_lz4_decode: the workgroup pipeline and the wave-per-block kernel; fsk_zstd_decode), the decoded buffer is
downloaded and compared BYTE FOR BYTE with what the image's liblz4 / libzstd decode from the same payload -- the calls
the reference makes (benchmark/flagstats.cpp:316 LZ4_decompress_safe, :96 ZSTD_decompress).  The block-file tests
(test_gpu_blockfile.py, test_gpu_zstd.py) see the decoders only through the 19 flagstat counters, which no permutation
of the flags changes and which never look at input bits 4, 5, 12-15 (libflagstats.h:118-142); here a match copied from
one flag too early, two swapped literal runs or one wrong bit anywhere fails.  The positional popcount of all 16 bits
(FLAGSTATS_hip_device_pospopcnt_u16; rule of STORM_pospopcnt_u16, python/libalgebra.h:566-574) runs on the decoded
device buffer as well: it does see bits 4, 5, 12-15."""
import os
import random
import struct

import numpy as np
import pytest

import decode_bytes_util as du
import test_gpu_blockfile as tb
from decode_bytes_util import bt

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "blockfiles")
LZ4_KERNELS = [pytest.param(0, id="workgroup_kernel"), pytest.param(1, id="r03_wave_kernel")]


def run_and_compare(hip, codec, parts, what, kernel=0, ref=None, popcnt_every=1):
    """parts: [(payload, decoded size)]; every block must decode to the reference decoder's bytes"""
    ref = ref or (du.ref_zstd if codec == "zstd" else du.ref_lz4)
    wants = [ref(p, sz) for p, sz in parts]
    for i, w in enumerate(wants):
        assert w is not None, (what, i, "the reference decoder rejects this payload")
    with du.DeviceDecode(hip, codec, [p for p, _ in parts], [sz for _, sz in parts], kernel) as dd:
        for i, w in enumerate(wants):
            du.check_decoded(dd, i, w, what)
            if popcnt_every and i % popcnt_every == 0:
                assert np.array_equal(dd.pospopcnt(i), du.pospopcnt_ref(w)), (what, i, "pospopcnt of the decoded device buffer")
    return wants


def lz4_case_image(case):
    import oracle
    kw = dict(block_bytes=bt.BLOCK_BYTES, mode="fast", level=2)
    if case == "na_ragged":
        flags = oracle.generate(oracle.GEN_NA12878, 1, 1, 0, 512000 * 3 + 12345)
    elif case == "exact_multiple":      # the reference writer's trailing empty block
        flags = oracle.generate(oracle.GEN_NA12878, 2, 0, 0, 512000 * 2)
    elif case == "uniform_incompressible":
        flags = oracle.generate(oracle.GEN_UNIFORM, 3, 0xFFFF, 0, 512000 * 2 + 77)
    elif case == "hc9":
        flags = oracle.generate(oracle.GEN_NA12878, 4, 1, 0, 512000 * 2 + 999)
        kw.update(mode="hc", level=9)
    elif case == "tiny_odd_blocks":     # odd block size: the last byte of every block is no flag
        flags = oracle.generate(oracle.GEN_UNIFORM, 5, 0x0FFF, 0, 200000)
        kw.update(block_bytes=9999)
    elif case == "full_range_hc":       # every one of the 16 input bits in use, compressible: bits 4, 5, 12-15 are decoded too
        rs = np.random.RandomState(9)
        flags = np.resize(rs.randint(0, 65536, 3000).astype(np.uint16), 512000 * 2 + 31)
        flags[rs.randint(0, flags.size, 4000)] ^= np.uint16(0xF030)
        kw.update(mode="hc", level=6)
    else:                               # many_blocks
        flags = oracle.generate(oracle.GEN_NA12878, 6, 1, 0, 512000 * 40 + 5)
    return bt.block_file_image(flags, **kw), flags, kw["block_bytes"]


@pytest.mark.parametrize("kernel", LZ4_KERNELS)
@pytest.mark.parametrize("case", ["na_ragged", "exact_multiple", "uniform_incompressible", "hc9", "tiny_odd_blocks", "full_range_hc", "many_blocks"])
def test_lz4_decode_is_byte_exact_on_liblz4_written_block_files(hip, kernel, case):
    img, flags, block_bytes = lz4_case_image(case)
    parts = du.split_image(img)
    wants = run_and_compare(hip, "lz4", parts, case, kernel, popcnt_every=7 if case == "many_blocks" else 1)
    assert b"".join(wants) == flags.tobytes()       # ... and liblz4's bytes are the flags that were written


@pytest.mark.parametrize("kernel", LZ4_KERNELS)
def test_lz4_decode_is_byte_exact_on_reference_written_files(hip, kernel):
    """every .lz4 file the reference's own `bench compress` wrote (tests/golden/blockfiles, LZ4-fast and LZ4-HC)"""
    from conftest import load_golden
    names = [n for n in load_golden("blockfiles/manifest.json")["files"] if n.endswith(".lz4")]
    assert len(names) >= 3
    for name in names:
        parts = du.split_image(open(os.path.join(GOLD, name), "rb").read())
        run_and_compare(hip, "lz4", parts, name, kernel)


@pytest.mark.parametrize("kernel", LZ4_KERNELS)
@pytest.mark.parametrize("style,seed", [("bare", 1), ("bare", 2), ("edges", 3), ("edges", 4), ("mixed", 5), ("mixed", 6)])
def test_lz4_decode_is_byte_exact_on_synthetic_edge_streams(hip, kernel, style, seed):
    """valid LZ4 that liblz4 would never write (offsets around every internal boundary, self-overlapping matches, length
    bytes): against liblz4's decode AND the bytes the generator itself produced"""
    rs = np.random.RandomState(seed)
    parts, decs = [], []
    for target in [40, 300, 5000, 70000, 200000, 1 << 20, 3000, 17]:
        comp, dec = tb._synthetic_lz4_block(rs, target, style)
        parts.append((comp, len(dec)))
        decs.append(dec)
    assert run_and_compare(hip, "lz4", parts, (style, seed), kernel) == decs


@pytest.mark.parametrize("kernel", LZ4_KERNELS)
@pytest.mark.parametrize("kind", ["zeros", "period", "runs", "repeats"])
def test_lz4_decode_is_byte_exact_on_long_matches_and_literal_runs(hip, kernel, kind):
    parts, raws = [], []
    for mode, level in [("fast", 1), ("fast", 9), ("hc", 4), ("hc", 12)]:
        for i, (n, block_bytes) in enumerate([(700_001, bt.BLOCK_BYTES), (300_000, 65536), (120_003, 4096), (1_600_000, bt.BLOCK_BYTES)]):
            flags = tb._stress_flags(kind, n, 1000 * level + i)
            got = du.split_image(bt.block_file_image(flags, block_bytes=block_bytes, mode=mode, level=level))
            parts += got
            raws.append((len(got), flags.tobytes()))
    wants = run_and_compare(hip, "lz4", parts, kind, kernel, popcnt_every=5)
    at = 0
    for nblk, raw in raws:
        assert b"".join(wants[at:at + nblk]) == raw
        at += nblk


_LZ4_SEEDS = {}      # the generator is a Python loop per sequence (~0.2 s a seed): both kernels of a run share what it made


def _lz4_fuzz_seed(seed):
    """tests/perf/fuzz_lz4_gpu.py's generator: five synthetic blocks and one damaged payload per seed"""
    if seed not in _LZ4_SEEDS:
        if len(_LZ4_SEEDS) >= 320:
            _LZ4_SEEDS.clear()
        _LZ4_SEEDS[seed] = _lz4_fuzz_seed_make(seed)
    return _LZ4_SEEDS[seed]


def _lz4_fuzz_seed_make(seed):
    rs = np.random.RandomState(seed)
    style = ("bare", "edges", "mixed")[seed % 3]
    parts, decs = [], []
    for target in rs.choice([17, 100, 2000, 9000, 40000, 150000, 600000], size=5):
        comp, dec = tb._synthetic_lz4_block(rs, int(target), style)
        parts.append((comp, len(dec)))
        decs.append(dec)
    comp, n = parts[int(rs.randint(len(parts)))]
    bad = bytearray(comp)
    for _ in range(int(rs.randint(1, 6))):
        bad[int(rs.randint(len(bad)))] ^= int(rs.randint(1, 256))
    return parts, decs, (bytes(bad), n)


def lz4_fuzz_slice(hip, kernel, first, count, batch=50):
    """returns (blocks exact, damaged: both accept / both reject / GPU stricter); raises on any difference"""
    exact = both_ok = both_bad = strict = 0
    for s0 in range(first, first + count, batch):
        parts, decs, damaged = [], [], []
        for seed in range(s0, min(s0 + batch, first + count)):
            p, d, bad = _lz4_fuzz_seed(seed)
            parts += p
            decs += d
            damaged.append(bad)
        assert run_and_compare(hip, "lz4", parts, ("fuzz seeds from", s0), kernel, popcnt_every=25) == decs
        exact += len(parts)
        # damaged payloads: never more lenient than liblz4, and the same bytes when both accept
        with du.DeviceDecode(hip, "lz4", [p for p, _ in damaged], [n for _, n in damaged], kernel) as dd:
            for i, (p, n) in enumerate(damaged):
                want = du.ref_lz4(p, n)
                if dd.status[i] == 0:
                    assert want is not None, ("seed", s0 + i, "liblz4 rejects, the GPU decoder accepts")
                    du.check_decoded(dd, i, want, ("damaged, seed", s0 + i))
                    both_ok += 1
                elif want is None:
                    both_bad += 1
                else:
                    strict += 1
    return exact, both_ok, both_bad, strict


@pytest.mark.parametrize("kernel", LZ4_KERNELS)
def test_lz4_decode_is_byte_exact_on_a_slice_of_the_fuzzer(hip, kernel):
    exact, both_ok, both_bad, strict = lz4_fuzz_slice(hip, kernel, 1000, 300)
    assert exact == 1500 and both_ok + both_bad + strict == 300 and both_bad > 50
    # (the GPU decoder may refuse a damaged payload that liblz4 still decodes -- the product's own host decoder is the
    # yardstick for that in tests/perf/fuzz_lz4_gpu.py -- but a valid stream is never refused: the 1500 blocks above)
    assert strict <= 30, strict


# ------------------------------------------------------------------------------------------------------------ Zstandard
def _zstd_parts(raws, level):
    return [(bt.compress_block(raw, "zstd", level), len(raw)) for raw in raws]


@pytest.mark.parametrize("level", [1, 5, 12, 19, -5])
def test_zstd_decode_is_byte_exact_on_synthetic_streams(hip, level):
    """raw / RLE / compressed blocks, eleven-block frames, four Huffman streams, runs above 16,383, matches at distances
    around the 32 KiB ring, the old 64 KiB one and the block size (test_gpu_zstd.synthetic_blocks)"""
    from test_gpu_zstd import synthetic_blocks
    raws = synthetic_blocks(100 + abs(level))
    assert run_and_compare(hip, "zstd", _zstd_parts(raws, level), ("synthetic", level)) == raws


def test_zstd_decode_is_byte_exact_on_reference_written_files(hip):
    from conftest import load_golden
    man = load_golden("blockfiles/manifest.json")["files"]
    names = [n for n, e in man.items() if e.get("codec") == "zstd"]
    assert len(names) >= 3
    for name in names:
        run_and_compare(hip, "zstd", du.split_image(open(os.path.join(GOLD, name), "rb").read()), name)


@pytest.mark.parametrize("level", [1, 3, 9, 19, -5])
def test_zstd_decode_is_byte_exact_on_block_files(hip, level):
    import oracle
    flags = oracle.generate(oracle.GEN_NA12878, 30 + abs(level), 1, 0, 512000 * 3 + 12345)
    parts = du.split_image(bt.block_file_image(flags, mode="zstd", level=level))
    assert b"".join(run_and_compare(hip, "zstd", parts, ("na", level))) == flags.tobytes()
    # every input bit in use, still compressible
    rs = np.random.RandomState(level + 50)
    full = np.resize(rs.randint(0, 65536, 5000).astype(np.uint16), 512000 + 77)
    full[rs.randint(0, full.size, 3000)] ^= np.uint16(0xF030)
    parts = du.split_image(bt.block_file_image(full, mode="zstd", level=level))
    assert b"".join(run_and_compare(hip, "zstd", parts, ("full range", level))) == full.tobytes()


@pytest.mark.parametrize("block_bytes", [9999, 70001, 4096000])
def test_zstd_decode_is_byte_exact_on_other_block_sizes(hip, block_bytes):
    import oracle
    flags = oracle.generate(oracle.GEN_NA12878, 51, 1, 0, 2048000 * 2 + 333)
    parts = du.split_image(bt.block_file_image(flags, block_bytes=block_bytes, mode="zstd", level=2))
    assert b"".join(run_and_compare(hip, "zstd", parts, block_bytes, popcnt_every=50)) == flags.tobytes()


def test_zstd_decode_is_byte_exact_on_frames_of_many_small_blocks(hip):
    """35 blocks in a frame: taken at once, byte-exact.  52 blocks, 391 blocks (a 1 KiB window), 1000 blocks: the first-pass entry
    answers "too many blocks" (never a wrong byte), the second-pass entry (fsk_zstd_decode_ex: a block slot per KiB) decodes them
    byte for byte."""
    import ctypes

    import oracle
    from zstd_fuzz_gen import flushed_frame
    z = bt.zstd()
    raw = oracle.generate(oracle.GEN_NA12878, 77, 1, 0, 512000).tobytes()
    run_and_compare(hip, "zstd", [(flushed_frame(z, raw, 30000), len(raw)), (flushed_frame(z, raw[:300001], 9000), 300001)], "35 blocks")
    # ZSTD_compress2 with windowLog 10: blocks of 1 KiB
    z.ZSTD_createCCtx.restype = ctypes.c_void_p
    z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
    z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    z.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    z.ZSTD_compress2.restype = ctypes.c_size_t
    cctx = z.ZSTD_createCCtx()
    z.ZSTD_CCtx_setParameter(cctx, 101, 10)          # ZSTD_c_windowLog
    small = raw[:400000]
    dst = ctypes.create_string_buffer(z.ZSTD_compressBound(len(small)))
    n = z.ZSTD_compress2(cctx, dst, len(dst), small, len(small))
    z.ZSTD_freeCCtx(cctx)
    assert not z.ZSTD_isError(n)
    cases = [("52 blocks", flushed_frame(z, raw, 20000), raw), ("1 KiB window", dst.raw[:n], small), ("1000 blocks", flushed_frame(z, raw, 1024), raw)]
    for what, frame, want in cases:
        assert du.ref_zstd(frame, len(want)) == want
        with du.DeviceDecode(hip, "zstd", [frame], [len(want)]) as dd:
            if dd.status[0] == 0:
                du.check_decoded(dd, 0, want, what)
            else:
                assert dd.status[0] == 67, (what, int(dd.status[0]))       # kZstdTooManyBlocks
        with du.DeviceDecode(hip, "zstd", [frame], [len(want)], min_blocks=len(want) // 1024 + 16) as dd:
            du.check_decoded(dd, 0, want, what + ", second pass")
            assert np.array_equal(dd.pospopcnt(0), du.pospopcnt_ref(want))


def zstd_fuzz_slice(hip, first, count, batch=50):
    """tests/perf/fuzz_zstd_gpu.py's generator (a third of the frames from ZSTD_compress2 with random advanced parameters),
    three damaged variants per frame; returns (exact, not taken, both accept, both reject, GPU stricter)"""
    from zstd_fuzz_gen import compress_with_parameters, damage, synthetic
    z = bt.zstd()
    exact = untaken = both_ok = both_bad = strict = 0
    for s0 in range(first, first + count, batch):
        parts, raws, damaged = [], [], []
        for seed in range(s0, min(s0 + batch, first + count)):
            rng = random.Random(seed)
            raw = synthetic(rng, np.random.default_rng(seed))
            level = rng.choice([1, 1, 2, 3, 5, 7, 9, 12, 15, 19, -1, -5])
            comp = compress_with_parameters(z, rng, raw) if seed % 3 == 2 else None
            if comp is None:
                comp = bt.compress_block(raw, "zstd", level)
            parts.append((comp, len(raw)))
            raws.append(raw)
            damaged += [(damage(rng, comp), len(raw)) for _ in range(3)]
        with du.DeviceDecode(hip, "zstd", [p for p, _ in parts], [n for _, n in parts]) as dd:
            for i, raw in enumerate(raws):
                assert du.ref_zstd(parts[i][0], len(raw)) == raw
                if 64 <= dd.status[i] < du.NOT_RUN:
                    untaken += 1         # valid Zstandard the decoder does not take (a window of a few KiB: hundreds of blocks)
                    continue
                du.check_decoded(dd, i, raw, ("fuzz seed", s0 + i))
                if i % 10 == 0:
                    assert np.array_equal(dd.pospopcnt(i), du.pospopcnt_ref(raw)), ("fuzz seed", s0 + i)
                exact += 1
        with du.DeviceDecode(hip, "zstd", [p for p, _ in damaged], [n for _, n in damaged]) as dd:
            for i, (p, n) in enumerate(damaged):
                want = du.ref_zstd(p, n)
                if dd.status[i] == 0:
                    assert want is not None, ("seed", s0 + i // 3, "libzstd rejects, the GPU decoder accepts")
                    du.check_decoded(dd, i, want, ("damaged, seed", s0 + i // 3))
                    both_ok += 1
                elif want is None:
                    both_bad += 1
                else:
                    strict += 1
    return exact, untaken, both_ok, both_bad, strict


def test_zstd_decode_is_byte_exact_on_a_slice_of_the_fuzzer(hip):
    exact, untaken, both_ok, both_bad, strict = zstd_fuzz_slice(hip, 0, 300)
    assert exact + untaken == 300 and exact >= 270, (exact, untaken)
    assert both_ok + both_bad + strict == 900 and both_bad > 300


def test_decoders_write_nothing_outside_their_blocks(hip):
    """Blocks of every length mod 16 packed back to back in 16-byte slots: each slot's slack (and an odd last byte) still
    holds the fill pattern after the launch -- a decoder that stores a vector too far would hit its neighbour."""
    rs = np.random.RandomState(4)
    raws = []
    for n in list(range(0, 40)) + [1023, 1024, 1025, 4095, 4097, 65535, 65537, 131071, 131073]:
        a = np.resize(rs.randint(0, 4096, max(1, n // 7 + 1)).astype(np.uint16), (n + 1) // 2).tobytes()[:n]
        raws.append(a)
    for kernel in (0, 1):
        run_and_compare(hip, "lz4", [(bt.compress_block(r, "fast", 1), len(r)) for r in raws], ("slots", kernel), kernel)
    run_and_compare(hip, "zstd", _zstd_parts(raws, 3), "slots")
