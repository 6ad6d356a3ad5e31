"""GPU: block-file reader + threaded LZ4 decode pipeline (row f1) vs the oracle on the raw flags.
Files are produced by the real liblz4 in the reference's block format (tools/blockfile_tool.py)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import blockfile_tool as bt  # noqa: E402

pytestmark = pytest.mark.gpu


def expect(flags, block_bytes):
    """Reference semantics (benchmark/flagstats.cpp:323): each block contributes size >> 1 flags."""
    import oracle
    raw = np.ascontiguousarray(flags, dtype=np.uint16).tobytes()
    out = np.zeros(32, dtype=np.uint64)
    n = 0
    for pos in range(0, len(raw), block_bytes):
        chunk = raw[pos:pos + block_bytes]
        k = len(chunk) >> 1
        out += oracle.flagstat_hist(np.frombuffer(chunk[:2 * k], dtype=np.uint16))
        n += k
    return out, n


@pytest.mark.parametrize("case", ["na_ragged", "exact_multiple", "uniform_incompressible", "hc9", "tiny_odd_blocks",
                                  "three_chunks"])
def test_lz4_block_files(hip, tmp_path, case):
    """the HOST-thread pipeline (knob lz4_decoder = 0; by the default size rule files from 64 MiB go to the GPU decoder,
    whose tests follow further down)"""
    import oracle
    from libflagstats_amd import blockfile
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 0) == 0
    try:
        _lz4_block_files(hip, tmp_path, case, oracle, blockfile)
    finally:
        assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 2) == 0


def _lz4_block_files(hip, tmp_path, case, oracle, blockfile):
    kw = dict(block_bytes=bt.BLOCK_BYTES, mode="fast", level=2)
    if case == "na_ragged":
        flags = oracle.generate(oracle.GEN_NA12878, 1, 1, 0, 512000 * 3 + 12345)
    elif case == "exact_multiple":      # reference writer appends an empty block here
        flags = oracle.generate(oracle.GEN_NA12878, 2, 0, 0, 512000 * 2)
    elif case == "uniform_incompressible":
        flags = oracle.generate(oracle.GEN_UNIFORM, 3, 0xFFFF, 0, 512000 * 2 + 77)
    elif case == "hc9":
        flags = oracle.generate(oracle.GEN_NA12878, 4, 1, 0, 512000 + 999)
        kw.update(mode="hc", level=9)
    elif case == "tiny_odd_blocks":     # odd block size: the last byte of every block is dropped (N = size >> 1)
        flags = oracle.generate(oracle.GEN_UNIFORM, 5, 0x0FFF, 0, 200000)
        kw.update(block_bytes=9999)
    else:                               # > 2 chunk buffers of 64 MiB: exercises buffer recycling
        flags = oracle.generate(oracle.GEN_NA12878, 6, 1, 0, 512000 * 150 + 5)
    path = tmp_path / (case + ".lz4")
    size = bt.write_block_file(path, flags, **kw)
    want, n = expect(flags, kw["block_bytes"])
    for threads in (1, 3, 0):
        got, st = blockfile.flagstat_lz4_file(str(path), threads)
        assert np.array_equal(got, want), (case, threads)
        assert st["n_flags"] == n and st["compressed_bytes"] == size
    img = open(path, "rb").read()
    got, st = blockfile.flagstat_lz4_image(img, 2)
    assert np.array_equal(got, want)
    if case == "exact_multiple":
        assert st["n_blocks"] == 3      # 2 data blocks + the reference writer's trailing empty block
    if case == "three_chunks":
        assert st["chunks"] >= 3


def test_empty_and_damaged_files(hip, tmp_path):
    import oracle
    from libflagstats_amd import _lib, blockfile
    p = tmp_path / "empty.lz4"
    p.write_bytes(b"")
    got, st = blockfile.flagstat_lz4_file(str(p), 2)
    assert not got.any() and st["n_blocks"] == 0
    flags = oracle.generate(oracle.GEN_NA12878, 8, 1, 0, 512000 + 100)
    img = bt.block_file_image(flags)
    for cut in (3, 8 + 10, len(img) - 1):          # truncated header / payload / last byte
        with pytest.raises(_lib.FlagstatsHipError):
            blockfile.flagstat_lz4_image(img[:cut], 2)
    bad = bytearray(img)
    bad[0:4] = (512000 * 2 + 1000).to_bytes(4, "little")   # header claims more bytes than the block holds
    with pytest.raises(_lib.FlagstatsHipError):
        blockfile.flagstat_lz4_image(bytes(bad), 2)
    with pytest.raises(_lib.FlagstatsHipError):
        blockfile.flagstat_lz4_file(str(tmp_path / "missing.lz4"), 1)
    # the library is still usable after failures
    got, _ = blockfile.flagstat_lz4_image(img, 2)
    assert np.array_equal(got, expect(flags, bt.BLOCK_BYTES)[0])


@pytest.mark.parametrize("n,odd", [(0, False), (1, True), (524288, False), (524289, True), (3_000_001, True),
                                   (40_000_003, False)])
def test_raw_file(hip, tmp_path, n, odd):
    """`bench decompress -D` (benchmark/flagstats.cpp:415-468): a raw uint16 file, read by the pipeline's
    workers in 1 MiB slices straight into pinned chunks.  Sizes around the slice (2^19 flags) and chunk
    boundaries, the empty file, a trailing odd byte (ignored: `read >> 1`, :450), small chunks; the r01
    mmap form (FLAGSTATS_HIP_RAW_IO=mmap) must agree."""
    import oracle
    from libflagstats_amd import blockfile
    flags = oracle.generate(oracle.GEN_UNIFORM, 12, 0xFFFF, 0, n)
    p = tmp_path / "flags.bin"
    p.write_bytes(flags.tobytes() + (b"\x7f" if odd else b""))
    want = oracle.flagstat_hist(flags) if n else np.zeros(32, dtype=np.uint64)
    old = hip.FLAGSTATS_hip_get(b"chunk_flags")
    try:
        for chunk in (old, 3_000_000):
            hip.FLAGSTATS_hip_set(b"chunk_flags", chunk)
            got, st = blockfile.flagstat_raw_file(str(p))
            assert np.array_equal(got, want) and st["n_flags"] == flags.size, (n, chunk)
    finally:
        hip.FLAGSTATS_hip_set(b"chunk_flags", old)
    os.environ["FLAGSTATS_HIP_RAW_IO"] = "mmap"
    try:
        got, st = blockfile.flagstat_raw_file(str(p))
    finally:
        del os.environ["FLAGSTATS_HIP_RAW_IO"]
    assert np.array_equal(got, want) and st["n_flags"] == flags.size


# --------------------------------------------------------------------------- LZ4 blocks decoded on the GPU (knob "lz4_decoder")
def gpu_decode(hip, img):
    import ctypes

    from libflagstats_amd import _lib
    out = np.zeros(32, dtype=np.uint64)
    st = _lib.GpuLz4Stats()
    buf = (ctypes.c_char * max(len(img), 1)).from_buffer_copy(img if img else b"\0")
    rc = hip.FLAGSTATS_hip_blockimage_lz4_gpu(buf, len(img), out.ctypes.data, ctypes.byref(st))
    return rc, out, st


@pytest.fixture(params=[0, 1], ids=["workgroup_kernel", "r03_wave_kernel"])
def lz4_kernel(hip, request):
    """both decode kernels the library carries: the workgroup pipeline (default) and r03's one wave per block (the yardstick)"""
    assert hip.FLAGSTATS_hip_get(b"lz4_gpu_kernel") == 0
    assert hip.FLAGSTATS_hip_set(b"lz4_gpu_kernel", request.param) == 0
    yield request.param
    assert hip.FLAGSTATS_hip_set(b"lz4_gpu_kernel", 0) == 0


@pytest.mark.parametrize("case", ["na_ragged", "exact_multiple", "uniform_incompressible", "hc9", "tiny_odd_blocks", "many_blocks"])
def test_gpu_side_lz4_decode_matches_oracle(hip, lz4_kernel, case):
    """flagstat_lz4_kernels.hip (a workgroup of eight waves per block, the 64 KiB window in LDS; and r03's wave per block):
    same counters as the host pipeline's contract on liblz4-written block images -- long literal runs (incompressible),
    far matches (HC), odd block sizes, the writer's empty block."""
    import oracle
    kw = dict(block_bytes=bt.BLOCK_BYTES, mode="fast", level=2)
    if case == "na_ragged":
        flags = oracle.generate(oracle.GEN_NA12878, 1, 1, 0, 512000 * 3 + 12345)
    elif case == "exact_multiple":
        flags = oracle.generate(oracle.GEN_NA12878, 2, 0, 0, 512000 * 2)
    elif case == "uniform_incompressible":
        flags = oracle.generate(oracle.GEN_UNIFORM, 3, 0xFFFF, 0, 512000 * 2 + 77)
    elif case == "hc9":
        flags = oracle.generate(oracle.GEN_NA12878, 4, 1, 0, 512000 * 2 + 999)
        kw.update(mode="hc", level=9)
    elif case == "tiny_odd_blocks":
        flags = oracle.generate(oracle.GEN_UNIFORM, 5, 0x0FFF, 0, 200000)
        kw.update(block_bytes=9999)
    else:
        flags = oracle.generate(oracle.GEN_NA12878, 6, 1, 0, 512000 * 40 + 5)
    img = bt.block_file_image(flags, **kw)
    want, n = expect(flags, kw["block_bytes"])
    rc, got, st = gpu_decode(hip, img)
    assert rc == 0, hip.FLAGSTATS_hip_last_error()
    assert np.array_equal(got, want), case
    assert st.n_flags == n and st.bad_blocks == 0 and st.compressed_bytes == len(img)
    assert st.ring_kib == (66 if lz4_kernel == 0 else 8) and (st.far_matches == 0 or lz4_kernel == 1)


def test_gpu_side_lz4_decode_on_reference_written_files(hip):
    from conftest import GOLDEN, load_golden
    man = load_golden("blockfiles/manifest.json")["files"]
    for name, meta in man.items():
        if not name.endswith(".lz4"):
            continue
        img = open(os.path.join(GOLDEN, "blockfiles", name), "rb").read()
        rc, got, st = gpu_decode(hip, img)
        assert rc == 0, (name, hip.FLAGSTATS_hip_last_error())
        assert np.array_equal(got, np.array(meta["scalar_counters"], dtype=np.uint64)), name


def test_gpu_side_lz4_decode_rejects_damaged_blocks(hip):
    """Every index in the kernel is masked or checked: damaged payloads set a status word, the call fails loudly, nothing
    faults, and the library keeps working."""
    import oracle
    flags = oracle.generate(oracle.GEN_NA12878, 8, 1, 0, 512000 + 100)
    img = bytearray(bt.block_file_image(flags))
    rs = np.random.RandomState(3)
    for trial in range(6):
        bad = bytearray(img)
        for _ in range(40):                                   # flip bytes inside the first block's payload
            bad[8 + int(rs.randint(0, 100000))] ^= int(rs.randint(1, 256))
        rc, got, st = gpu_decode(hip, bytes(bad))      # must return (a flip may even leave a valid stream); no fault, no hang
        assert rc == 0 or b"failed to decode" in hip.FLAGSTATS_hip_last_error()
    rc, got, st = gpu_decode(hip, bytes(img))
    assert rc == 0 and np.array_equal(got, expect(flags, bt.BLOCK_BYTES)[0])


@pytest.fixture
def gpu_decoder(hip):
    """knob lz4_decoder = 1: every LZ4 block file goes through the GPU decoder whatever its size (default 2: from 1 GiB)"""
    assert hip.FLAGSTATS_hip_get(b"lz4_decoder") == 2 and hip.FLAGSTATS_hip_get(b"lz4_gpu_min_bytes") == 64 << 20
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 1) == 0
    yield hip
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 2) == 0


@pytest.mark.parametrize("case", ["na_ragged", "exact_multiple", "uniform_incompressible", "hc9", "tiny_odd_blocks", "three_spans"])
def test_block_file_entries_with_the_gpu_decoder(gpu_decoder, tmp_path, case):
    """The product entries (file, image, by-extension, superset) with the blocks decoded on the GPU: same counters as the
    host-thread pipeline and the oracle; file mode reads with 1, 3 and the default number of parallel preads, through
    more than three pinned spans in the last case."""
    import oracle
    from libflagstats_amd import blockfile
    hip = gpu_decoder
    kw = dict(block_bytes=bt.BLOCK_BYTES, mode="fast", level=2)
    if case == "na_ragged":
        flags = oracle.generate(oracle.GEN_NA12878, 11, 1, 0, 512000 * 3 + 12345)
    elif case == "exact_multiple":
        flags = oracle.generate(oracle.GEN_NA12878, 12, 0, 0, 512000 * 2)
    elif case == "uniform_incompressible":
        flags = oracle.generate(oracle.GEN_UNIFORM, 13, 0xFFFF, 0, 512000 * 2 + 77)
    elif case == "hc9":
        flags = oracle.generate(oracle.GEN_NA12878, 14, 1, 0, 512000 + 999)
        kw.update(mode="hc", level=9)
    elif case == "tiny_odd_blocks":
        flags = oracle.generate(oracle.GEN_UNIFORM, 15, 0x0FFF, 0, 200000)
        kw.update(block_bytes=9999)
    else:                               # incompressible, > 3 pinned spans of 64 MiB of FILE bytes: buffer recycling
        flags = oracle.generate(oracle.GEN_UNIFORM, 16, 0xFFFF, 0, 512000 * 280 + 5)
    path = tmp_path / (case + ".lz4")
    size = bt.write_block_file(path, flags, **kw)
    want, n = expect(flags, kw["block_bytes"])
    for threads in (1, 3, 0):
        got, st = blockfile.flagstat_lz4_file(str(path), threads)
        assert np.array_equal(got, want), (case, threads)
        assert st["n_flags"] == n and st["compressed_bytes"] == size and st["gpu_decode"] == 1
        assert st["uncompressed_bytes"] == 2 * len(flags) and st["threads"] >= 1
    img = open(path, "rb").read()
    got, st = blockfile.flagstat_lz4_image(img, 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 1 and st["threads"] == 0
    got, st = blockfile.flagstat_file(str(path), 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 1
    # superset counters (slots 0 / 16 = primary paired reads, slot 9 = pass-QC reads), against the host-thread pipeline
    sup_gpu, st = blockfile.flagstat_file(str(path), 2, superset=True)
    assert st["gpu_decode"] == 1
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 0) == 0
    sup_host, st = blockfile.flagstat_file(str(path), 2, superset=True)
    got_host, _ = blockfile.flagstat_lz4_file(str(path), 2)
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 1) == 0
    assert st["gpu_decode"] == 0 and np.array_equal(got_host, want)
    assert np.array_equal(sup_gpu, sup_host), case


def test_gpu_decoder_is_chosen_by_size_and_fails_loudly(gpu_decoder, tmp_path):
    import oracle
    from libflagstats_amd import _lib, blockfile
    hip = gpu_decoder
    flags = oracle.generate(oracle.GEN_NA12878, 21, 1, 0, 512000 * 2 + 100)
    img = bt.block_file_image(flags)
    want = expect(flags, bt.BLOCK_BYTES)[0]
    # by size (default rule 2): below the threshold the host threads decode, at it the GPU
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 2) == 0
    got, st = blockfile.flagstat_lz4_image(img, 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 0
    assert hip.FLAGSTATS_hip_set(b"lz4_gpu_min_bytes", len(img)) == 0
    got, st = blockfile.flagstat_lz4_image(img, 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 1
    # a byte below the knob the file is taken only if it decodes to 2.5 x the knob: LZ4-fast flags are 2.14 : 1
    assert 5 * (len(img) + 1) > 2 * flags.nbytes
    assert hip.FLAGSTATS_hip_set(b"lz4_gpu_min_bytes", len(img) + 1) == 0
    got, st = blockfile.flagstat_lz4_image(img, 2)
    assert np.array_equal(got, want) and st["gpu_decode"] == 0
    assert hip.FLAGSTATS_hip_set(b"lz4_gpu_min_bytes", 64 << 20) == 0
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 1) == 0
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 3) != 0
    # damaged input: truncated header / payload, a header that claims more than the block holds, a corrupted payload
    for cut in (3, 8 + 10, len(img) - 1):
        with pytest.raises(_lib.FlagstatsHipError):
            blockfile.flagstat_lz4_image(img[:cut], 2)
    bad = bytearray(img)
    bad[0:4] = (512000 * 2 + 1000).to_bytes(4, "little")
    with pytest.raises(_lib.FlagstatsHipError):
        blockfile.flagstat_lz4_image(bytes(bad), 2)
    p = tmp_path / "cut.lz4"
    p.write_bytes(img[:len(img) - 7])
    with pytest.raises(_lib.FlagstatsHipError):
        blockfile.flagstat_lz4_file(str(p), 2)
    empty = tmp_path / "empty.lz4"
    empty.write_bytes(b"")
    got, st = blockfile.flagstat_lz4_file(str(empty), 2)
    assert not got.any() and st["n_blocks"] == 0
    # ... and the library is still usable, counters untouched by the failed calls
    got, _ = blockfile.flagstat_lz4_image(img, 2)
    assert np.array_equal(got, want)


def _synthetic_lz4_block(rs, target, style):
    """A VALID LZ4 block that liblz4 would never write: sequences drawn at random with the parameters pushed to the
    decoder's edges (offsets around the 8 KiB ring / its 64-byte margin / 16 KiB / 65535, offsets 1..20 that overlap
    their own output, matches of 4, 18, 19, 20, 270+ bytes, literal runs of 0, 1, 14, 15, 16, 270+).  Returns
    (compressed, decoded)."""
    out = bytearray()
    comp = bytearray()
    edge_offs = [1, 2, 3, 4, 7, 8, 15, 16, 17, 18, 19, 20, 63, 64, 65, 8127, 8128, 8129, 8191, 8192, 8193, 16319, 16320, 16321,
                 16384, 32768, 65534, 65535]
    edge_ml = [4, 5, 15, 16, 17, 18, 19, 20, 21, 33, 64, 65, 255 + 19, 255 + 20, 600]
    edge_ll = [0, 0, 0, 0, 1, 2, 13, 14, 15, 16, 17, 64, 65, 255 + 15, 255 + 16, 700]

    def put_len(v):
        while v >= 255:
            comp.append(255)
            v -= 255
        comp.append(v)

    while len(out) < target:
        if style == "bare":            # long runs of bare short matches (the batch path), a literal now and then
            ll = 0 if rs.rand() < 0.93 or not out else int(rs.randint(1, 15))
            ml = int(rs.randint(4, 19))
        elif style == "edges":
            ll = int(edge_ll[rs.randint(len(edge_ll))])
            ml = int(edge_ml[rs.randint(len(edge_ml))])
        else:                          # mixed
            ll = int(rs.choice([0, 0, 0, int(rs.randint(1, 40))]))
            ml = int(rs.choice([int(rs.randint(4, 19)), int(rs.randint(4, 19)), int(rs.randint(19, 80))]))
        if not out and ll == 0:
            ll = 4                     # the first sequence needs something to match against
        lits = bytes(rs.randint(0, 256, ll, dtype=np.uint8)) if style != "bare" else bytes(rs.randint(0, 4, ll, dtype=np.uint8))
        have = len(out) + ll
        if rs.rand() < 0.5:
            off = int(edge_offs[rs.randint(len(edge_offs))])
        elif rs.rand() < 0.5:
            off = int(rs.randint(1, 300))
        else:
            off = int(rs.randint(1, 65536))
        off = max(1, min(off, have, 65535))
        token = (min(ll, 15) << 4) | min(ml - 4, 15)
        comp.append(token)
        if ll >= 15:
            put_len(ll - 15)
        comp += lits
        out += lits
        comp += bytes((off & 255, off >> 8))
        if ml - 4 >= 15:
            put_len(ml - 4 - 15)
        if off >= ml:                  # (what a byte-by-byte copy gives, without the Python loop)
            out += out[len(out) - off:len(out) - off + ml]
        else:                          # an overlapping match repeats with period off
            pat = bytes(out[len(out) - off:])
            out += (pat * (ml // off + 1))[:ml]
    # the block ends with a literals-only sequence; the last match must start at least 12 bytes before the end of the
    # block (LZ4 block format, "parsing restrictions": liblz4 rejects anything else)
    ll = int(rs.choice([12, 13, 14, 15, 16, 40]))
    lits = bytes(rs.randint(0, 256, ll, dtype=np.uint8))
    comp.append(min(ll, 15) << 4)
    if ll >= 15:
        put_len(ll - 15)
    comp += lits
    out += lits
    return bytes(comp), bytes(out)


@pytest.mark.parametrize("style,seed", [("bare", 1), ("bare", 2), ("edges", 3), ("edges", 4), ("mixed", 5), ("mixed", 6)])
def test_gpu_decoder_on_synthetic_edge_streams(gpu_decoder, style, seed):
    """The GPU decoder (and the host decoder beside it) on streams built to sit on its internal boundaries -- ring size
    and far-match margin, flush quarters, the 16-sequence batch, rows of 16 lanes, self-overlapping matches, length
    extension bytes, window refills -- checked against the decoded bytes the generator itself produced."""
    import struct

    import oracle
    from libflagstats_amd import blockfile
    hip = gpu_decoder
    rs = np.random.RandomState(seed)
    img = bytearray()
    want = np.zeros(32, dtype=np.uint64)
    n = 0
    sizes = [40, 300, 5000, 70000, 200000, 1 << 20, 3000, 17]
    for target in sizes:
        comp, dec = _synthetic_lz4_block(rs, target, style)
        img += struct.pack("<ii", len(dec), len(comp)) + comp
        k = len(dec) >> 1
        want += oracle.flagstat_hist(np.frombuffer(dec[:2 * k], dtype=np.uint16))
        n += k
        assert bt.decompress_block_ref(comp, len(dec)) == dec             # liblz4 (what the reference calls) accepts it
        assert blockfile.lz4_block_decode(comp, len(dec)) == dec          # ... and so does the product's host decoder
    got, st = blockfile.flagstat_lz4_image(bytes(img), 2)
    assert st["gpu_decode"] == 1 and st["n_flags"] == n
    assert np.array_equal(got, want), (style, seed)
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 0) == 0
    got_host, st = blockfile.flagstat_lz4_image(bytes(img), 2)
    assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 1) == 0
    assert st["gpu_decode"] == 0 and np.array_equal(got_host, want)


def test_gpu_decoder_goes_through_large_files_in_segments(gpu_decoder, tmp_path, monkeypatch):
    """A file whose compressed + decoded bytes do not fit the device together is decoded in segments, one after the
    other; forced here with a 3 MiB segment on 10 blocks (image, file, superset), and a segment smaller than a block."""
    import oracle
    from libflagstats_amd import blockfile
    hip = gpu_decoder
    flags = oracle.generate(oracle.GEN_NA12878, 31, 1, 0, 512000 * 9 + 4321)
    img = bt.block_file_image(flags)
    want, n = expect(flags, bt.BLOCK_BYTES)
    rc, got, st = gpu_decode(hip, img)
    assert rc == 0 and st.segments == 1 and np.array_equal(got, want)
    for cap, segs in ((3 << 20, 3), (1000, 10)):          # three blocks per segment; one block per segment
        monkeypatch.setenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES", str(cap))
        rc, got, st = gpu_decode(hip, img)
        assert rc == 0, hip.FLAGSTATS_hip_last_error()
        assert np.array_equal(got, want) and st.n_flags == n and st.segments >= segs and st.bad_blocks == 0
        path = tmp_path / "seg.lz4"
        path.write_bytes(img)
        got, bst = blockfile.flagstat_lz4_file(str(path), 3)
        assert np.array_equal(got, want) and bst["gpu_decode"] == 1 and bst["n_flags"] == n
        sup, _ = blockfile.flagstat_file(str(path), 2, superset=True)
        monkeypatch.delenv("FLAGSTATS_HIP_GPU_LZ4_SEGMENT_BYTES")
        sup_one, _ = blockfile.flagstat_file(str(path), 2, superset=True)
        assert np.array_equal(sup, sup_one)


def test_gpu_decoder_keeps_its_device_buffers_between_calls(gpu_decoder):
    """The decoder's two large device buffers are reused by the next call (knob lz4_gpu_keep_bytes; default automatic: what
    the last call needed, at most a quarter of the device) and given back when they exceed the limit, after eight calls of
    other entry points, and by a failed call; stale bytes of an earlier, larger file must not leak into a later one."""
    import oracle
    from libflagstats_amd import blockfile, pyflagstats
    hip = gpu_decoder
    auto = (1 << 64) - 1
    assert hip.FLAGSTATS_hip_get(b"lz4_gpu_keep_bytes") == auto
    big = oracle.generate(oracle.GEN_UNIFORM, 41, 0xFFFF, 0, 512000 * 6)
    got, _ = blockfile.flagstat_lz4_image(bt.block_file_image(big), 2)
    assert np.array_equal(got, expect(big, bt.BLOCK_BYTES)[0])
    kept = hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes")
    assert kept >= 2 * big.nbytes                      # compressed (incompressible: ~ raw size) + decoded
    # a smaller file with odd block sizes in the same buffers: the dropped odd bytes and the slack between blocks count nothing
    small = oracle.generate(oracle.GEN_UNIFORM, 42, 0xFFFF, 0, 300001)
    got, st = blockfile.flagstat_lz4_image(bt.block_file_image(small, block_bytes=9999), 2)
    assert st["gpu_decode"] == 1 and np.array_equal(got, expect(small, 9999)[0])
    assert hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == kept
    # the idle rule: eight calls that do not use the decoder, and the buffers are gone
    a = oracle.generate(oracle.GEN_UNIFORM, 43, 0xFFFF, 0, 5000)
    for i in range(7):
        pyflagstats.counters_u32(a)
        assert hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == kept, i
    pyflagstats.counters_u32(a)
    assert hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == 0
    # a damaged file leaves nothing behind either
    img = bytearray(bt.block_file_image(small, block_bytes=9999))
    img[len(img) // 2] ^= 0x5A
    img[len(img) // 2 + 1] ^= 0xA5
    try:
        blockfile.flagstat_lz4_image(bytes(img), 2)
    except Exception:
        assert hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == 0
    # an explicit limit: 0 = free after every call
    assert hip.FLAGSTATS_hip_set(b"lz4_gpu_keep_bytes", 0) == 0
    got, _ = blockfile.flagstat_lz4_image(bt.block_file_image(small, block_bytes=9999), 2)
    assert np.array_equal(got, expect(small, 9999)[0]) and hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == 0
    assert hip.FLAGSTATS_hip_set(b"lz4_gpu_keep_bytes", auto) == 0


def _stress_flags(kind, n, seed):
    """uint16 streams that make liblz4 emit what an NA12878-like stream hardly ever does: matches of hundreds and thousands
    of bytes (length bytes, 255-runs of them), offsets 1..3 (a match that overlaps its own output), literal runs of 15 and
    more, long incompressible stretches between compressible ones."""
    rs = np.random.RandomState(seed)
    if kind == "zeros":
        a = np.zeros(n, dtype=np.uint16)
    elif kind == "period":
        p = rs.randint(0, 4096, size=int(rs.choice([1, 2, 3, 5, 7, 64]))).astype(np.uint16)
        a = np.resize(p, n)
        a[rs.randint(0, n, size=max(1, n // 5000))] ^= 1          # a literal every few thousand flags
    elif kind == "runs":
        parts, left = [], n
        while left > 0:
            m = int(min(left, rs.choice([1, 3, 17, 40, 300, 5000, 70000])))
            parts.append(np.full(m, rs.randint(0, 4096), dtype=np.uint16) if rs.rand() < 0.7
                         else rs.randint(0, 65536, size=m).astype(np.uint16))
            left -= m
        a = np.concatenate(parts)
    elif kind == "repeats":
        base = rs.randint(0, 4096, size=3000).astype(np.uint16)
        parts, left = [], n
        while left > 0:
            o, m = int(rs.randint(0, 2900)), int(min(left, rs.randint(2, 100)))
            parts.append(base[o:o + m])
            left -= len(parts[-1])
        a = np.concatenate(parts)[:n]
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(a[:n])


@pytest.mark.parametrize("kind", ["zeros", "period", "runs", "repeats"])
@pytest.mark.parametrize("mode,level", [("fast", 1), ("fast", 9), ("hc", 4), ("hc", 12)])
def test_gpu_decoder_on_streams_with_long_matches_and_literal_runs(gpu_decoder, lz4_kernel, kind, mode, level):
    """liblz4-written blocks of highly compressible and mixed data (see _stress_flags), block sizes from 4 KiB to the
    format's 1,024,000 bytes: the GPU decoder must give the host oracle's counters (the reference-written golden
    `ragged_fast_a2.lz4`, ratio 39, once found a deadlock that NA12878-like streams never reached)."""
    from libflagstats_amd import blockfile
    for i, (n, block_bytes) in enumerate([(700_001, bt.BLOCK_BYTES), (300_000, 65536), (120_003, 4096), (1_600_000, bt.BLOCK_BYTES)]):
        flags = _stress_flags(kind, n, 1000 * level + i)
        img = bt.block_file_image(flags, block_bytes=block_bytes, mode=mode, level=level)
        want = expect(flags, block_bytes)[0]
        got, st = blockfile.flagstat_lz4_image(img, 2)
        assert st["gpu_decode"] == 1 and np.array_equal(got, want), (kind, mode, level, n, block_bytes)


def test_size_rule_leaves_incompressible_files_to_the_host_threads(hip):
    """With the decoder chosen by size, an LZ4 file whose blocks hardly compress (decoded bytes < 1.25 x the file's) stays with
    the host-thread pipeline -- it is PCIe-bound there and the GPU decoder's literal path is its slow one; forced, the GPU
    decoder takes it and gives the same counters."""
    import oracle
    from libflagstats_amd import blockfile
    flags = oracle.generate(oracle.GEN_UNIFORM, 71, 0xFFFF, 0, 512000 * 3 + 9)
    img = bt.block_file_image(flags)
    want = expect(flags, bt.BLOCK_BYTES)[0]
    assert hip.FLAGSTATS_hip_get(b"lz4_decoder") == 2
    assert hip.FLAGSTATS_hip_set(b"lz4_gpu_min_bytes", 1) == 0
    try:
        got, st = blockfile.flagstat_lz4_image(img, 2)
        assert st["gpu_decode"] == 0 and np.array_equal(got, want)
        assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 1) == 0
        got, st = blockfile.flagstat_lz4_image(img, 2)
        assert st["gpu_decode"] == 1 and np.array_equal(got, want)
    finally:
        assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 2) == 0
        assert hip.FLAGSTATS_hip_set(b"lz4_gpu_min_bytes", 64 << 20) == 0


def test_blocks_beyond_the_workgroup_kernels_limits_go_to_the_wave_kernel(gpu_decoder):
    """the workgroup kernel's records address 16 MiB of payload and 256 MiB of output per block (status 10 beyond);
    the reference's writer never makes such blocks, but they are valid files: known from the index, they send the file to the
    wave-per-block kernel instead of failing the call."""
    import oracle
    from libflagstats_amd import blockfile
    flags = oracle.generate(oracle.GEN_UNIFORM, 81, 0xFFFF, 0, (17 << 20) // 2 + 1000)         # incompressible: payload > 16 MiB
    img = bt.block_file_image(flags, block_bytes=17 << 20)
    assert max(cs for _, cs in _headers(img)) >= 1 << 24
    want = expect(flags, 17 << 20)[0]
    got, st = blockfile.flagstat_lz4_image(img, 2)
    assert st["gpu_decode"] == 1 and np.array_equal(got, want)
    rc, got2, gst = gpu_decode(gpu_decoder, img)
    assert rc == 0 and np.array_equal(got2, want) and gst.ring_kib == 8        # (8 KiB ring: the wave kernel ran)


def _headers(img):
    import struct
    pos = 0
    while pos < len(img):
        us, cs = struct.unpack_from("<ii", img, pos)
        yield us, cs
        pos += 8 + cs


def test_a_header_that_declares_more_than_its_payload_can_decode_to_is_refused(hip):
    """A damaged 60-byte file must not make the reader size buffers from a 2 GiB header: an LZ4 block grows by at most 255
    bytes per payload byte, a Zstandard frame by at most 128 KiB per 4 bytes -- both index passes refuse what lies beyond,
    loudly, on the host-thread path and on the GPU path."""
    import struct
    from libflagstats_amd import _lib, blockfile
    raw = bytes(200)
    for codec, entry, knob in (("fast", blockfile.flagstat_lz4_image, b"lz4_decoder"), ("zstd", blockfile.flagstat_zstd_image, b"zstd_decoder")):
        comp = bt.compress_block(raw, codec if codec == "zstd" else "fast", 1)
        good = struct.pack("<ii", len(raw), len(comp)) + comp
        bad = struct.pack("<ii", 0x7FFFFFF0, len(comp)) + comp
        for dec in (0, 1):
            assert hip.FLAGSTATS_hip_set(knob, dec) == 0
            try:
                got, _ = entry(good, 1)
                assert not got.any()                 # 100 zero flags count nothing
                with pytest.raises(_lib.FlagstatsHipError, match="declares more decoded bytes"):
                    entry(bad, 1)
            finally:
                assert hip.FLAGSTATS_hip_set(knob, 2) == 0


def test_file_mode_ring_of_other_shapes_and_the_hipHostMalloc_fallback(gpu_decoder, tmp_path, monkeypatch):
    """File mode reads into a ring of page-locked spans (four of 16 MiB: registered huge pages the readers touch themselves).
    Other ring shapes (two 1 MiB spans: every span boundary inside a block; eight spans) and the fallback allocator
    (FLAGSTATS_HIP_HOST_ALLOC=malloc: plain hipHostMalloc, what a system without transparent huge pages or with a runtime that
    refuses the registration gets) must give the same counters; a ring that grows is re-made."""
    import oracle
    from libflagstats_amd import blockfile
    flags = oracle.generate(oracle.GEN_NA12878, 91, 1, 0, 512000 * 21 + 7)
    path = tmp_path / "ring.lz4"
    bt.write_block_file(path, flags, mode="hc", level=4)
    want, n = expect(flags, bt.BLOCK_BYTES)
    for spans, mib, alloc in ((2, 1, None), (8, 1, None), (3, 2, "malloc"), (5, 4, "malloc"), (4, 16, None)):
        monkeypatch.setenv("FLAGSTATS_HIP_GPU_SPANS", str(spans))
        monkeypatch.setenv("FLAGSTATS_HIP_GPU_SPAN_MIB", str(mib))
        if alloc:
            monkeypatch.setenv("FLAGSTATS_HIP_HOST_ALLOC", alloc)
        else:
            monkeypatch.delenv("FLAGSTATS_HIP_HOST_ALLOC", raising=False)
        for threads in (1, 0):
            got, st = blockfile.flagstat_lz4_file(str(path), threads)
            assert st["gpu_decode"] == 1 and st["n_flags"] == n and np.array_equal(got, want), (spans, mib, alloc, threads)


@pytest.mark.parametrize("as_file", [False, True], ids=["image", "file"])
def test_a_decoded_buffer_the_device_cannot_hold_is_found_out_beside_the_copies(gpu_decoder, tmp_path, monkeypatch, as_file):
    """The decoded buffer is allocated on a thread of its own while the index goes up and the first pieces are read and copied
    (a 1.6 GB hipMalloc took 30 ms on one host); a refusal therefore arrives with copies already queued.  Forced with the test
    knob FLAGSTATS_HIP_GPU_OUT_CAP: the forced GPU decoder fails loudly and leaves the caller's counters alone, the decoder
    chosen by size hands the file to the host threads, and the next call (cap lifted) decodes on the GPU again."""
    import oracle
    from libflagstats_amd import blockfile
    hip = gpu_decoder
    flags = oracle.generate(oracle.GEN_NA12878, 123, 1, 0, 512000 * 5 + 77)
    img = bt.block_file_image(flags)
    want = expect(flags, bt.BLOCK_BYTES)[0]
    path = tmp_path / "refused.lz4"
    path.write_bytes(img)
    call = (lambda: blockfile.flagstat_lz4_file(str(path), 3)) if as_file else (lambda: blockfile.flagstat_lz4_image(img, 3))
    auto = (1 << 64) - 1
    assert hip.FLAGSTATS_hip_set(b"lz4_gpu_keep_bytes", 0) == 0      # nothing kept from earlier files: the buffer is asked for anew
    try:
        got, st = call()
        assert st["gpu_decode"] == 1 and np.array_equal(got, want) and hip.FLAGSTATS_hip_get(b"lz4_gpu_kept_bytes") == 0
        monkeypatch.setenv("FLAGSTATS_HIP_GPU_OUT_CAP", str(1 << 20))
        with pytest.raises(Exception, match="cannot hold"):
            call()
        assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 2) == 0 and hip.FLAGSTATS_hip_set(b"lz4_gpu_min_bytes", 1) == 0
        got, st = call()
        assert st["gpu_decode"] == 0 and np.array_equal(got, want)
        monkeypatch.delenv("FLAGSTATS_HIP_GPU_OUT_CAP")
        got, st = call()
        assert st["gpu_decode"] == 1 and np.array_equal(got, want)
    finally:
        assert hip.FLAGSTATS_hip_set(b"lz4_gpu_min_bytes", 64 << 20) == 0
        assert hip.FLAGSTATS_hip_set(b"lz4_decoder", 1) == 0          # (the fixture puts 2 back)
        assert hip.FLAGSTATS_hip_set(b"lz4_gpu_keep_bytes", auto) == 0


def test_image_entries_take_any_buffer_without_copying_it(hip):
    """The Python image entries hand bytes, numpy arrays and writable buffers to the C side as they are (r05: they used to copy
    the image into a ctypes array first -- 3 ms for a 30 MB image, more than the call takes); a read-only memoryview is the one
    case that is copied.  Same counters from every kind."""
    import oracle
    from libflagstats_amd import blockfile
    flags = oracle.generate(oracle.GEN_NA12878, 5, 1, 0, 512000 * 3 + 11)
    for entry, img in ((blockfile.flagstat_lz4_image, bt.block_file_image(flags)),
                       (blockfile.flagstat_zstd_image, bt.block_file_image(flags, mode="zstd", level=1))):
        want = expect(flags, bt.BLOCK_BYTES)[0]
        ba = bytearray(img)
        for image in (img, ba, np.frombuffer(img, dtype=np.uint8), memoryview(ba), memoryview(img), np.frombuffer(img, dtype=np.uint8)[:]):
            got, st = entry(image, 2)
            assert np.array_equal(got, want) and st["n_flags"] == flags.size
        got, st = entry(b"", 2)
        assert not got.any() and st["n_flags"] == 0


def test_shipped_size_rule_goes_by_what_a_file_decodes_to(hip):
    """r05: the two decoders cross at 120-200 MB of DECODED flags whatever the codec and level, which is 27-95 MiB of file
    (profiles/r05/decoder_crossover.log) -- so besides files of 64 MiB and more, the GPU decoders take smaller ones that decode to at
    least 160 MiB: a 41 MiB Zstandard file of 90 M flags goes to the GPU, a 54 MiB LZ4-fast file of 60 M flags stays on the host."""
    import oracle
    from libflagstats_amd import blockfile
    assert hip.FLAGSTATS_hip_get(b"lz4_decoder") == 2 and hip.FLAGSTATS_hip_get(b"zstd_decoder") == 2
    assert hip.FLAGSTATS_hip_get(b"lz4_gpu_min_bytes") == 64 << 20 and hip.FLAGSTATS_hip_get(b"zstd_gpu_min_bytes") == 64 << 20
    n = 90_000_000
    flags = oracle.generate(oracle.GEN_NA12878, 7, 1, 0, n)
    want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, n)
    z = bt.block_file_image(flags, mode="zstd", level=1)
    assert (16 << 20) < len(z) < (64 << 20) and flags.nbytes >= (160 << 20)
    got, st = blockfile.flagstat_zstd_image(z, 0)
    assert np.array_equal(got, want) and st["gpu_decode"] == 1
    del z
    m = 60_000_000
    l4 = bt.block_file_image(flags[:m], mode="fast", level=2)
    assert (16 << 20) < len(l4) < (64 << 20) and 2 * m < (160 << 20)
    got, st = blockfile.flagstat_lz4_image(l4, 0)
    assert np.array_equal(got, oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, m)) and st["gpu_decode"] == 0
