"""Zstandard block files decoded ON the GPU (libflagstats_amd/csrc/flagstat_zstd_kernels.hip; knob "zstd_decoder"): the
product entries against the oracle and against the libzstd host pipeline, on the reference-written .zst goldens, on
synthetic streams of every compressor level, on damaged and on unsupported-but-valid frames.
The reference decodes every payload with ZSTD_decompress (benchmark/flagstats.cpp:636-682)."""
import ctypes
import json
import os
import random
import struct
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
sys.path.insert(0, HERE)
import blockfile_tool as bt  # noqa: E402
from test_gpu_blockfile import expect  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(HERE, "golden", "blockfiles")


@pytest.fixture
def zgpu(hip):
    """knob zstd_decoder = 1: every .zst block file goes through the GPU decoder whatever its size, and a frame it does not
    take is an error (default 2: from zstd_gpu_min_bytes, anything it does not take goes to libzstd on the host)"""
    assert hip.FLAGSTATS_hip_get(b"zstd_decoder") == 2
    assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0
    yield hip
    assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 2) == 0


def image_of(raw_blocks, level=1):
    """block file image (benchmark/flagstats.cpp:192-226 layout) from a list of raw byte blocks"""
    out = bytearray()
    for raw in raw_blocks:
        comp = bt.compress_block(raw, "zstd", level)
        out += struct.pack("<ii", len(raw), len(comp)) + comp
    return bytes(out)


def expect_blocks(raw_blocks):
    import oracle
    out = np.zeros(32, dtype=np.uint64)
    for raw in raw_blocks:
        k = len(raw) >> 1
        out += oracle.flagstat_hist(np.frombuffer(raw[:2 * k], dtype=np.uint16))
    return out


def test_reference_written_zstd_files_on_the_gpu_decoder(zgpu):
    from libflagstats_amd import blockfile
    manifest = json.load(open(os.path.join(GOLD, "manifest.json")))
    names = [k for k, e in manifest["files"].items() if e.get("codec") == "zstd"]
    assert len(names) >= 3
    for name in names:
        e = manifest["files"][name]
        path = os.path.join(GOLD, name)
        want = np.array(e["scalar_counters"], dtype=np.uint64)
        got, st = blockfile.flagstat_zstd_file(path, threads=2)
        assert st["gpu_decode"] == 1 and st["n_flags"] == e["n_flags"]
        assert zgpu.FLAGSTATS_hip_set(b"zstd_decoder", 0) == 0
        host, st0 = blockfile.flagstat_zstd_file(path, threads=2)
        assert zgpu.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0
        assert st0["gpu_decode"] == 0 and np.array_equal(got, host), name
        assert np.array_equal(got[:len(want)], want), name
        img, _ = blockfile.flagstat_zstd_image(open(path, "rb").read(), threads=2)
        assert np.array_equal(img, host)


@pytest.mark.parametrize("level", [1, 3, 9, 19, -5])
def test_block_file_entries_with_the_gpu_zstd_decoder(zgpu, tmp_path, level):
    import oracle
    from libflagstats_amd import blockfile
    flags = oracle.generate(oracle.GEN_NA12878, 30 + abs(level), 1, 0, 512000 * 3 + 12345)
    path = tmp_path / "na.zst"
    size = bt.write_block_file(path, flags, mode="zstd", level=level)
    want, n = expect(flags, bt.BLOCK_BYTES)
    for threads in (1, 0):
        got, st = blockfile.flagstat_zstd_file(str(path), threads)
        assert np.array_equal(got, want), (level, threads)
        assert st["n_flags"] == n and st["compressed_bytes"] == size and st["gpu_decode"] == 1
    got, st = blockfile.flagstat_zstd_image(open(path, "rb").read(), 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 1
    sup_gpu, st = blockfile.flagstat_file(str(path), 2, superset=True)
    assert st["gpu_decode"] == 1
    assert zgpu.FLAGSTATS_hip_set(b"zstd_decoder", 0) == 0
    sup_host, st = blockfile.flagstat_file(str(path), 2, superset=True)
    assert zgpu.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0
    assert st["gpu_decode"] == 0 and np.array_equal(sup_gpu, sup_host)


def synthetic_blocks(seed):
    import oracle
    r = np.random.default_rng(seed)
    a = r.integers(0, 256, 40000, dtype=np.uint8).tobytes()
    return [
        b"", b"ab", bytes(1000), bytes(300000), r.integers(0, 256, 300, dtype=np.uint8).tobytes(),
        r.integers(0, 256, 200000, dtype=np.uint8).tobytes(),                      # raw blocks
        b"hello world, " * 3000,
        oracle.generate(oracle.GEN_NA12878, seed, 1, 0, 700000).tobytes(),          # eleven blocks: two passes of the entropy kernel
        r.integers(0, 4, 300000, dtype=np.uint8).tobytes(),                         # literal-heavy: four Huffman streams a block
        r.integers(0, 60, 200000, dtype=np.uint8).tobytes(),
        r.integers(0, 3000, 100000, dtype=np.uint16).tobytes(),
        a + bytes(50000) + a + r.integers(0, 256, 20000, dtype=np.uint8).tobytes() + a[:30000] + bytes(100000),   # runs above 16,383, far matches
        (r.integers(0, 256, 17000, dtype=np.uint8).tobytes() + b"x" * 17000) * 12,
        oracle.generate(oracle.GEN_UNIFORM, seed, 0x0FFF, 0, 512000).tobytes(),
    ] + [
        # a long literal run, then a long match from just beyond the 32 KiB the ring keeps: the match's source ends a few bytes
        # before the record starts (a wait for the whole source before the record's first piece never ended: kZstdStuck)
        # (levels 1 and 5 write ONE sequence of 16,383 literals and a 16,383-byte match from 33,266 ... 35,066 back)
        a[:16383] + bytes(k) + r.integers(0, 256, 16383, dtype=np.uint8).tobytes() + a[:16383] + b"tail" * 9
        for k in (2, 500, 1500, 2300, 4000)
    ] + [
        x + r.integers(0, 256, ly, dtype=np.uint8).tobytes() + x + b"tail" * 9
        for x, ly in ((a[:17000], 16500), (a[:30000], 3000))
    ] + [
        # a 3000-byte stretch repeated at distances around what the execution kernel keeps in LDS (matches up to 32,767 back read
        # the ring, farther ones the flushed output), around the ring's size (36 KiB), around the 64 KiB an earlier form of
        # the kernel kept, and around the block size
        a[:3000] + r.integers(0, 256, dist - 3000, dtype=np.uint8).tobytes() + a[:3000] + b"tail" * 50
        for dist in (32766, 32767, 32768, 32769, 35839, 36863, 36864, 36865, 65534, 65535, 65536, 65537, 68607, 69631, 69632,
                     69633, 131071, 131072, 131073)
    ]


@pytest.mark.parametrize("level", [1, 5, 12, 19, -5])
def test_gpu_zstd_decoder_on_synthetic_streams(zgpu, level):
    from libflagstats_amd import blockfile
    blocks = synthetic_blocks(100 + abs(level))
    img = image_of(blocks, level)
    got, st = blockfile.flagstat_zstd_image(img, 2)
    assert st["gpu_decode"] == 1 and st["n_blocks"] == len(blocks)
    assert np.array_equal(got, expect_blocks(blocks)), level
    # every block on its own as well (a wrong block shows up by name)
    for i, raw in enumerate(blocks):
        got, _ = blockfile.flagstat_zstd_image(image_of([raw], level), 2)
        assert np.array_equal(got, expect_blocks([raw])), (level, i)


def test_gpu_zstd_decoder_is_chosen_by_size_and_rejects_loudly(zgpu, tmp_path):
    import oracle
    from libflagstats_amd import _lib, blockfile
    hip = zgpu
    flags = oracle.generate(oracle.GEN_NA12878, 21, 1, 0, 512000 * 2 + 100)
    img = bt.block_file_image(flags, mode="zstd", level=1)
    want = expect(flags, bt.BLOCK_BYTES)[0]
    assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 2) == 0
    # the size rule: GPU for a file of at least `zstd_gpu_min_bytes`, or a smaller one that decodes to at least 2.5 x that
    assert 5 * len(img) < 2 * flags.nbytes                                   # (this file: 4.2 : 1)
    assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", flags.nbytes) == 0   # neither: host threads
    got, st = blockfile.flagstat_zstd_image(img, 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 0
    assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", len(img)) == 0       # by its compressed size
    got, st = blockfile.flagstat_zstd_image(img, 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 1
    assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 2 * flags.nbytes // 5) == 0   # smaller than the knob, but it decodes to 2.5 x it
    got, st = blockfile.flagstat_zstd_image(img, 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 1
    assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 2 * flags.nbytes // 5 + 2) == 0   # ... and just not
    got, st = blockfile.flagstat_zstd_image(img, 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 0
    assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", len(img)) == 0
    # a damaged payload: by size the file goes to libzstd, whose verdict counts; forced, the GPU decoder's code is the error
    bad = bytearray(img)
    bad[8 + 40] ^= 0x55
    bad[8 + 41] ^= 0xAA
    with pytest.raises(_lib.FlagstatsHipError):
        blockfile.flagstat_zstd_image(bytes(bad), 2)
    assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0
    with pytest.raises(_lib.FlagstatsHipError):
        blockfile.flagstat_zstd_image(bytes(bad), 2)
    assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 3) != 0
    for cut in (3, 8 + 10, len(img) - 1):
        with pytest.raises(_lib.FlagstatsHipError):
            blockfile.flagstat_zstd_image(img[:cut], 2)
    # valid Zstandard the GPU decoder does not take -- two frames in one payload: forced it is an error, by size libzstd decodes it
    raw = flags.tobytes()[:200000]
    two = bt.compress_block(raw[:100000], "zstd", 1) + bt.compress_block(raw[100000:], "zstd", 1)
    img2 = struct.pack("<ii", len(raw), len(two)) + two
    with pytest.raises(_lib.FlagstatsHipError):
        blockfile.flagstat_zstd_image(img2, 2)
    assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 2) == 0
    assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 1) == 0
    got, st = blockfile.flagstat_zstd_image(img2, 2)
    assert st["gpu_decode"] == 0 and np.array_equal(got, expect_blocks([raw]))
    got, st = blockfile.flagstat_zstd_image(img, 2)
    assert st["gpu_decode"] == 1 and np.array_equal(got, want)
    assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 64 << 20) == 0
    assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0


def test_gpu_zstd_decoder_and_libzstd_agree_on_damaged_frames(zgpu):
    """Bit flips in the payloads: the GPU decoder may be stricter than libzstd, never more lenient, and what both accept
    counts the same."""
    import oracle
    from libflagstats_amd import _lib, blockfile
    hip = zgpu
    rng = random.Random(11)
    raw = oracle.generate(oracle.GEN_NA12878, 5, 1, 0, 60000).tobytes()
    strict = agree = 0
    for level in (1, 19):
        comp = bt.compress_block(raw, "zstd", level)
        for _ in range(40):
            bad = bytearray(comp)
            for _ in range(rng.randrange(1, 4)):
                bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
            img = struct.pack("<ii", len(raw), len(bad)) + bytes(bad)
            assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 0) == 0
            try:
                host, _ = blockfile.flagstat_zstd_image(img, 1)
            except _lib.FlagstatsHipError:
                host = None
            assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0
            try:
                gpu, _ = blockfile.flagstat_zstd_image(img, 1)
            except _lib.FlagstatsHipError:
                gpu = None
            if host is None:
                assert gpu is None
            elif gpu is None:
                strict += 1
            else:
                assert np.array_equal(gpu, host)
                agree += 1
    assert agree > 10 and strict < 20


def test_gpu_zstd_decoder_goes_through_large_files_in_segments(zgpu, tmp_path, monkeypatch):
    """A file the device cannot hold at once is decoded segment after segment (forced with a 3 MiB segment on 10 frames: image,
    file, superset); every segment has its own pieces, kernels and scratch."""
    import oracle
    from libflagstats_amd import blockfile
    flags = oracle.generate(oracle.GEN_NA12878, 31, 1, 0, 512000 * 9 + 4321)
    path = tmp_path / "seg.zst"
    bt.write_block_file(path, flags, mode="zstd", level=3)
    want, n = expect(flags, bt.BLOCK_BYTES)
    one, st = blockfile.flagstat_zstd_file(str(path), 2)
    assert np.array_equal(one, want) and st["gpu_decode"] == 1
    sup_one, _ = blockfile.flagstat_file(str(path), 2, superset=True)
    for cap in (3 << 20, 1000):          # three frames per segment; one frame per segment
        monkeypatch.setenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES", str(cap))
        got, st = blockfile.flagstat_zstd_file(str(path), 3)
        assert np.array_equal(got, want) and st["gpu_decode"] == 1 and st["n_flags"] == n
        got, st = blockfile.flagstat_zstd_image(open(path, "rb").read(), 2)
        assert np.array_equal(got, want) and st["gpu_decode"] == 1
        sup, _ = blockfile.flagstat_file(str(path), 2, superset=True)
        assert np.array_equal(sup, sup_one)
        monkeypatch.delenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES")


def test_gpu_zstd_decoder_with_every_cut_of_the_pieces(zgpu, tmp_path, monkeypatch):
    """The host side cuts a file into pieces of whole frames (the first one half an equal share by default) and decodes them
    on two streams: every piece count from one to more than there are frames, with the first piece at 1 / 25 / 50 / 99 % of
    a share and with equal pieces, gives the same counters -- image and file mode, frames of ragged sizes."""
    import oracle
    from libflagstats_amd import blockfile
    rng = np.random.default_rng(77)
    blocks = []
    for i in range(13):
        k = int(rng.integers(1, 60000))
        blocks.append(oracle.generate(oracle.GEN_NA12878, 40 + i, 1, 0, k).tobytes())
    blocks.append(oracle.generate(oracle.GEN_UNIFORM, 3, 0x0FFF, 0, 512000).tobytes())
    img = image_of(blocks, level=3)
    want = expect_blocks(blocks)
    path = tmp_path / "cuts.zst"
    path.write_bytes(img)
    for pieces in (1, 2, 3, 5, 13, 14, 40):
        for first in (None, 1, 25, 50, 99, 100):
            monkeypatch.setenv("FLAGSTATS_HIP_GPU_LZ4_CHUNKS", str(pieces))
            if first is None:
                monkeypatch.delenv("FLAGSTATS_HIP_GPU_FIRST_PIECE", raising=False)
            else:
                monkeypatch.setenv("FLAGSTATS_HIP_GPU_FIRST_PIECE", str(first))
            got, st = blockfile.flagstat_zstd_image(img, 2)
            assert np.array_equal(got, want) and st["gpu_decode"] == 1, (pieces, first)
            assert 1 <= st["chunks"] <= min(pieces, len(blocks))
            if pieces in (2, 13):
                got, st = blockfile.flagstat_zstd_file(str(path), 3)
                assert np.array_equal(got, want) and st["gpu_decode"] == 1, (pieces, first)
    monkeypatch.delenv("FLAGSTATS_HIP_GPU_LZ4_CHUNKS")
    monkeypatch.delenv("FLAGSTATS_HIP_GPU_FIRST_PIECE", raising=False)
    got, st = blockfile.flagstat_zstd_image(img, 2)
    assert np.array_equal(got, want) and st["chunks"] == 2   # (the shipped rule: two pieces at least)


def test_gpu_zstd_decoder_keeps_and_releases_its_device_memory(zgpu):
    """The two large buffers AND the scratch between the kernels stay with the engine for the next file, go after eight calls of
    other entry points and after a failed call (knob lz4_gpu_keep_bytes; read-only lz4_gpu_kept_bytes)."""
    import oracle
    from libflagstats_amd import _lib, blockfile, pyflagstats
    hip = zgpu
    flags = oracle.generate(oracle.GEN_NA12878, 41, 1, 0, 512000 * 4)
    img = bt.block_file_image(flags, mode="zstd", level=1)
    want = expect(flags, bt.BLOCK_BYTES)[0]
    got, _ = blockfile.flagstat_zstd_image(img, 2)
    assert np.array_equal(got, want)
    kept = hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes")
    assert kept > len(img) + 2 * len(flags) + (16 << 20)       # compressed + decoded + scratch for two frames a stream
    got, _ = blockfile.flagstat_zstd_image(img, 2)
    assert np.array_equal(got, want) and hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == kept
    a = oracle.generate(oracle.GEN_UNIFORM, 43, 0xFFFF, 0, 5000)
    for i in range(7):
        pyflagstats.counters_u32(a)
        assert hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == kept, i
    pyflagstats.counters_u32(a)
    assert hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == 0
    bad = bytearray(img)
    bad[8] ^= 0xFF      # the first frame's magic number
    with pytest.raises(_lib.FlagstatsHipError):
        blockfile.flagstat_zstd_image(bytes(bad), 2)
    assert hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == 0
    got, _ = blockfile.flagstat_zstd_image(img, 2)
    assert np.array_equal(got, want)


def test_gpu_zstd_decoder_takes_frames_of_many_small_blocks(zgpu):
    """Frames cut into more and smaller blocks than ZSTD_compress makes (a streaming compressor flushing every 30,000 / 20,000 /
    1,024 bytes: 35 / 52 / 1000 blocks in a 1,024,000-byte frame) are Zstandard like any other.  Up to the block slots of a first
    pass (40 for such a frame) the GPU decoder takes them at once; beyond, it decodes those frames once more with a slot per KiB
    (orchestration: the second pass) -- next to ordinary frames in the same file; a frame of still smaller blocks (a flush every
    300 bytes) is not taken: forced, an error; by size, the file goes to libzstd."""
    import ctypes
    import oracle
    from libflagstats_amd import _lib, blockfile
    from zstd_fuzz_gen import flushed_frame
    hip = zgpu
    z = bt.zstd()

    raw = oracle.generate(oracle.GEN_NA12878, 77, 1, 0, 512000).tobytes()
    plain = oracle.generate(oracle.GEN_NA12878, 78, 1, 0, 300000).tobytes()
    for every, takes in ((30000, True), (20000, True), (1024, True), (300, False)):
        frame = flushed_frame(z, raw, every)
        back = ctypes.create_string_buffer(len(raw))
        assert z.ZSTD_decompress(back, len(raw), frame, len(frame)) == len(raw) and back.raw == raw
        img = image_of([plain], 1) + struct.pack("<ii", len(raw), len(frame)) + frame + image_of([plain[:100001]], 3)
        want = expect_blocks([plain, raw, plain[:100001]])
        if takes:
            got, st = blockfile.flagstat_zstd_image(img, 2)
            assert st["gpu_decode"] == 1 and np.array_equal(got, want), every
            if every == 1024:
                # ... the same with every frame in a segment of its own: each segment makes its own second pass
                os.environ["FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES"] = "1000"
                try:
                    got, st = blockfile.flagstat_zstd_image(img, 2)
                finally:
                    del os.environ["FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES"]
                assert st["gpu_decode"] == 1 and np.array_equal(got, want)
        else:
            with pytest.raises(_lib.FlagstatsHipError):
                blockfile.flagstat_zstd_image(img, 2)
            assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 2) == 0
            assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 1) == 0
            got, st = blockfile.flagstat_zstd_image(img, 2)
            assert st["gpu_decode"] == 0 and np.array_equal(got, want)
            assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 64 << 20) == 0
            assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0


@pytest.mark.parametrize("block_bytes", [9999, 70001, 4096000])
def test_gpu_zstd_decoder_on_other_block_sizes(zgpu, tmp_path, block_bytes):
    """Block files cut differently from the reference writer's 1,024,000 bytes: small odd frames (the odd last byte of a block is
    no flag), frames of one block, frames of 32 blocks (four passes of the prepare and chain kernels)."""
    import oracle
    from libflagstats_amd import blockfile
    flags = oracle.generate(oracle.GEN_NA12878, 51, 1, 0, 2048000 * 2 + 333)
    path = tmp_path / "other.zst"
    bt.write_block_file(path, flags, block_bytes=block_bytes, mode="zstd", level=2)
    want, n = expect(flags, block_bytes)
    got, st = blockfile.flagstat_zstd_file(str(path), 2)
    assert st["gpu_decode"] == 1 and st["n_flags"] == n and np.array_equal(got, want)
    got, st = blockfile.flagstat_zstd_image(open(path, "rb").read(), 2)
    assert st["gpu_decode"] == 1 and np.array_equal(got, want)


def test_gpu_zstd_decoder_is_the_default_for_large_files(hip, tmp_path):
    """With the knobs as shipped (zstd_decoder 2, zstd_gpu_min_bytes 64 MiB) a 300-frame file of NA12878-like flags (73 MiB
    compressed: two pieces, several pinned spans in file mode, the reference writer's trailing empty block) goes through the GPU
    decoder; the counters are the oracle's."""
    import oracle
    from libflagstats_amd import blockfile
    assert hip.FLAGSTATS_hip_get(b"zstd_decoder") == 2 and hip.FLAGSTATS_hip_get(b"zstd_gpu_min_bytes") == 64 << 20
    flags = oracle.generate(oracle.GEN_NA12878, 61, 1, 0, 512000 * 300)        # an exact multiple: a trailing empty block
    path = tmp_path / "big.zst"
    size = bt.write_block_file(path, flags, mode="zstd", level=1)
    assert size >= 64 << 20
    want, n = expect(flags, bt.BLOCK_BYTES)
    got, st = blockfile.flagstat_zstd_file(str(path), 0)
    assert st["gpu_decode"] == 1 and st["n_flags"] == n and st["n_blocks"] == 301 and np.array_equal(got, want)
    got, st = blockfile.flagstat_file(str(path), 4)
    assert st["gpu_decode"] == 1 and np.array_equal(got, want)
    # flags that hardly compress (12-bit uniform: Huffman-coded literals and hardly a sequence; 16-bit uniform: raw blocks) stay
    # with the host threads under the size rule -- PCIe-bound there, the GPU decoder's slow paths here -- and go through the GPU
    # decoder when it is forced
    for hi_mask, nfl in ((0x0FFF, 512000 * 100 + 3), (0xFFFF, 512000 * 70 + 1)):
        flags = oracle.generate(oracle.GEN_UNIFORM, 62, hi_mask, 0, nfl)
        img = bt.block_file_image(flags, mode="zstd", level=1)
        assert len(img) >= 64 << 20
        want = expect(flags, bt.BLOCK_BYTES)[0]
        got, st = blockfile.flagstat_zstd_image(img, 0)
        assert st["gpu_decode"] == 0 and np.array_equal(got, want)
        assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0
        try:
            got, st = blockfile.flagstat_zstd_image(img, 0)
        finally:
            assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 2) == 0
        assert st["gpu_decode"] == 1 and np.array_equal(got, want)


def test_a_frame_the_gpu_decoder_does_not_take_in_a_later_segment_sends_the_whole_file_to_libzstd(hip, monkeypatch):
    """with the decoder chosen by size, a file that goes through in several segments and meets a frame the GPU
    decoder does not take (here: two concatenated frames in one payload) in a LATER segment must still fall back to the host
    pipeline -- the segments before it must not have reached the caller's counters (they are collected and added only after
    the last segment) -- and forced, the call fails with a message and leaves the counters untouched."""
    import oracle
    from libflagstats_amd import _lib, blockfile
    raws = [oracle.generate(oracle.GEN_NA12878, 90 + i, 1, 0, 150000).tobytes() for i in range(6)]
    parts = [bt.compress_block(r, "zstd", 1) for r in raws]
    two = bt.compress_block(raws[4][:100000], "zstd", 1) + bt.compress_block(raws[4][100000:], "zstd", 1)
    parts[4] = two                                  # valid Zstandard (libzstd decodes both frames), not taken by the GPU decoder
    img = b"".join(struct.pack("<ii", len(r), len(p)) + p for r, p in zip(raws, parts))
    want = expect_blocks(raws)
    monkeypatch.setenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES", "400000")      # two frames per segment: the odd one is in the third
    assert hip.FLAGSTATS_hip_get(b"zstd_decoder") == 2
    assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 1) == 0
    try:
        got, st = blockfile.flagstat_zstd_image(img, 2)
        assert st["gpu_decode"] == 0 and np.array_equal(got, want)
        assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0
        out = np.full(32, 7, dtype=np.uint64)
        buf = (ctypes.c_char * len(img)).from_buffer_copy(img)
        rc = hip.FLAGSTATS_hip_blockimage_zstd(buf, len(img), 2, out.ctypes.data, None)
        assert rc != 0 and b"Zstandard" in hip.FLAGSTATS_hip_last_error()
        assert (out == 7).all()                      # nothing of the first two segments was added
    finally:
        assert hip.FLAGSTATS_hip_set(b"zstd_decoder", 2) == 0
        assert hip.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 64 << 20) == 0


def test_zstd_scratch_that_does_not_fit_makes_smaller_pieces(zgpu, monkeypatch):
    """the scratch between the kernels (8.5 MB per frame in flight and decode stream) is what a busy device runs out
    of first; a failed scratch allocation halves the frames per launch instead of giving the file up.  Forced here with a
    piece count of one on a file whose one-piece scratch is above the test's cap."""
    import oracle
    from libflagstats_amd import blockfile
    flags = oracle.generate(oracle.GEN_NA12878, 97, 1, 0, 512000 * 12)
    img = bt.block_file_image(flags, mode="zstd", level=1)
    want = expect(flags, bt.BLOCK_BYTES)[0]
    auto = (1 << 64) - 1
    assert zgpu.FLAGSTATS_hip_set(b"lz4_gpu_keep_bytes", 0) == 0             # nothing kept from earlier files: the scratch is asked for anew
    try:
        blockfile.flagstat_zstd_image(bt.block_file_image(flags[:1000], mode="zstd", level=1), 2)
        assert zgpu.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == 0
        monkeypatch.setenv("FLAGSTATS_HIP_GPU_SCRATCH_CAP", str(40 << 20))  # test knob: a scratch request above this "fails"
        got, st = blockfile.flagstat_zstd_image(img, 2)
        assert st["gpu_decode"] == 1 and np.array_equal(got, want)
        assert st["chunks"] >= 4                    # 13 frames at ~8.5 MB each: at most four fit under 40 MB
        monkeypatch.setenv("FLAGSTATS_HIP_GPU_SCRATCH_CAP", str(4 << 20))   # not even one frame's: the file goes to the host threads ...
        assert zgpu.FLAGSTATS_hip_set(b"zstd_decoder", 2) == 0
        assert zgpu.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 1) == 0
        got, st = blockfile.flagstat_zstd_image(img, 2)
        assert st["gpu_decode"] == 0 and np.array_equal(got, want)
    finally:
        assert zgpu.FLAGSTATS_hip_set(b"zstd_gpu_min_bytes", 64 << 20) == 0
        assert zgpu.FLAGSTATS_hip_set(b"zstd_decoder", 1) == 0
        assert zgpu.FLAGSTATS_hip_set(b"lz4_gpu_keep_bytes", auto) == 0
