"""Row f1 pinned to the REFERENCE'S OWN WRITER: tests/golden/blockfiles/*.lz4 were written by the
reference's unmodified `bench compress` (benchmark/flagstats.cpp:110-190) in the build container
(tests/golden/make_blockfiles.py; manifest.json records recipe, expected counters and what the reference
binary itself printed when reading them back).

CPU part: the product's host-side index + LZ4 block decoder reproduce the recipe's bytes from those files.
GPU part (-m gpu): the decode -> H2D -> K1 pipeline, a streaming session and -- where the reference build
oracle/_ref/bench_hip travelled -- the reference's unmodified MAIN PROGRAM linked against the header shims
and libflagstats_hip.so all report FLAGSTAT_scalar's counters for them."""
import hashlib
import json
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

BF = os.path.join(GOLDEN, "blockfiles")
MANIFEST = json.load(open(os.path.join(BF, "manifest.json")))
sys.path.insert(0, GOLDEN)


def blocks_of(blob):
    """(uncompressed_size, payload) per block: int32, int32, raw LZ4 block (benchmark/flagstats.cpp:136-138)."""
    pos = 0
    while pos < len(blob):
        us, cs = struct.unpack_from("<ii", blob, pos)
        pos += 8
        yield us, blob[pos:pos + cs]
        pos += cs
    assert pos == len(blob)


LZ4_FILES = sorted(k for k, e in MANIFEST["files"].items() if e.get("codec", "lz4") == "lz4")
ZST_FILES = sorted(k for k, e in MANIFEST["files"].items() if e.get("codec") == "zstd")


@pytest.mark.parametrize("name", LZ4_FILES)
def test_host_decoder_reproduces_the_recipe_from_reference_written_files(name):
    from libflagstats_amd import blockfile
    from make_blockfiles import recipe_input
    e = MANIFEST["files"][name]
    blob = open(os.path.join(BF, name), "rb").read()
    assert len(blob) == e["bytes"]
    out = bytearray()
    sizes = []
    for us, payload in blocks_of(blob):
        dec = blockfile.lz4_block_decode(payload, us)
        assert dec is not None and len(dec) == us
        out += dec
        sizes.append(us)
    assert hashlib.sha256(bytes(out)).hexdigest() == e["input_sha256"]
    assert bytes(out) == recipe_input(e["n_flags"], e["seed"]).tobytes()      # the committed recipe still regenerates it
    assert all(s == 1024000 for s in sizes[:-1]) or len(sizes) == 1           # 512,000-flag blocks
    if e["n_flags"] % 512000 == 0:
        # the writer's extra empty block behind an exact multiple -- the one its own reader chokes on
        assert sizes[-1] == 0 and e["reference_reader_exits_on_trailing_empty_block"]


def test_zstd_fixtures_hold_the_recipe():
    """The .zst goldens (reference writer, benchmark/flagstats.cpp:192-226): same block header, one
    Zstandard frame per block, trailing block of 0 flags behind an exact multiple.  Decoded here with the
    image's libzstd -- the third-party library the product resolves at run time too."""
    import ctypes
    from make_blockfiles import recipe_input
    try:
        z = ctypes.CDLL("libzstd.so.1")
    except OSError:
        pytest.skip("no libzstd.so.1")
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    assert ZST_FILES
    for name in ZST_FILES:
        e = MANIFEST["files"][name]
        out = bytearray()
        sizes = []
        for us, payload in blocks_of(open(os.path.join(BF, name), "rb").read()):
            buf = ctypes.create_string_buffer(max(us, 1))
            r = z.ZSTD_decompress(buf, us, payload, len(payload))
            assert r == us
            out += buf.raw[:us]
            sizes.append(us)
        assert hashlib.sha256(bytes(out)).hexdigest() == e["input_sha256"]
        assert bytes(out) == recipe_input(e["n_flags"], e["seed"]).tobytes()
        if e["n_flags"] % 512000 == 0:
            assert sizes[-1] == 0


def test_zstd_entry_points_exist_and_refuse_unknown_extensions(tmp_path):
    """No GPU needed: the extension sniffing of FLAGSTATS_hip_blockfile (the reference's
    check_file_extension, benchmark/flagstats.cpp:828-839) fails loudly before anything touches a device."""
    from libflagstats_amd import _lib
    lib = _lib.lib()
    out = np.zeros(32, dtype=np.uint64)
    p = tmp_path / "flags.bin"
    p.write_bytes(b"")
    assert lib.FLAGSTATS_hip_blockfile(str(p).encode(), 1, out.ctypes.data, None) != 0
    assert b"unknown extension" in lib.FLAGSTATS_hip_last_error()
    assert lib.FLAGSTATS_hip_zstd_available() in (0, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ZST_FILES)
def test_pipeline_counts_reference_written_zstd_files(hip, name):
    from libflagstats_amd import blockfile
    assert hip.FLAGSTATS_hip_zstd_available() == 1, "libzstd.so.1 missing on the GPU box"
    e = MANIFEST["files"][name]
    want = np.array(e["scalar_counters"], dtype=np.uint64)
    path = os.path.join(BF, name)
    for threads in (1, 3):
        got, st = blockfile.flagstat_zstd_file(path, threads=threads)
        assert np.array_equal(got, want), (name, threads)
        assert st["n_flags"] == e["n_flags"] == e["reference_reader_total_flags"]
    got, _ = blockfile.flagstat_file(path, threads=2)                     # codec from the extension
    assert np.array_equal(got, want)
    got, _ = blockfile.flagstat_zstd_image(open(path, "rb").read(), threads=2)
    assert np.array_equal(got, want)
    # an LZ4 decoder on Zstandard frames must fail loudly, not count garbage
    with pytest.raises(Exception):
        blockfile.flagstat_lz4_file(path, threads=1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", LZ4_FILES)
def test_pipeline_counts_reference_written_files(hip, name):
    from libflagstats_amd import blockfile
    e = MANIFEST["files"][name]
    want = np.array(e["scalar_counters"], dtype=np.uint64)
    for threads in (1, 4):
        got, st = blockfile.flagstat_lz4_file(os.path.join(BF, name), threads=threads)
        assert np.array_equal(got, want), (name, threads)
        assert st["n_flags"] == e["n_flags"]
    got, _ = blockfile.flagstat_file(os.path.join(BF, name), threads=2)   # codec from the extension
    assert np.array_equal(got, want)
    got, _ = blockfile.flagstat_lz4_image(open(os.path.join(BF, name), "rb").read(), threads=2)
    assert np.array_equal(got, want)
    if e["reference_decompress_d"]:
        # what the reference binary printed for this very file (its dispatcher kernel) on the live slots
        import oracle
        table = {k: (int(p), int(f)) for k, p, f in e["reference_decompress_d"]}
        for slot in oracle.LIVE_SLOTS:
            assert table[oracle.SAM_FLAG_NAMES[slot % 16]][slot // 16] == int(got[slot])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MANIFEST["files"]))
def test_reference_main_program_on_the_gpu_engine(hip, name):
    """oracle/_ref/bench_hip = the reference's benchmark/flagstats.cpp, unmodified, compiled against
    include/libflagstats.h + include/libalgebra.h and linked with libflagstats_hip.so (recipe:
    oracle/Makefile `refbench`).  `decompress -d` then runs the reference's own block loop
    (:311-332) with every FLAGSTATS_get_function(N)(...) call landing on the MI355X."""
    exe = os.path.join(ROOT, "oracle", "_ref", "bench_hip")
    if not os.path.exists(exe):
        pytest.skip("reference build product oracle/_ref/bench_hip not present on this machine")
    e = MANIFEST["files"][name]
    r = subprocess.run([exe, "decompress", "-i", os.path.join(BF, name), "-d"], capture_output=True, text=True)
    if e.get("codec") == "zstd":
        # the reference's .zst reader prints no counter table (benchmark/flagstats.cpp:676-679), only the
        # number of flags it pushed through FLAGSTATS_get_function(N)(...) -- here: through the engine
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"\[ZSTD [^\]]*\] Time elapsed \d+ ms (\d+)", r.stderr)
        assert m and int(m.group(1)) == e["n_flags"], r.stderr
        return
    if e["reference_reader_exits_on_trailing_empty_block"]:
        assert r.returncode != 0          # the reference reader's own limitation, engine-independent
        return
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"^(\w+)\t(\d+)\t(\d+)$", r.stderr, flags=re.M)
    assert len(rows) == 15, r.stderr
    want = e["scalar_counters"]
    for i, (nm, p, f) in enumerate(rows):
        assert (int(p), int(f)) == (want[i], want[16 + i]), (name, nm)    # all 15 printed rows, scalar-exact
    assert ("Tot flags=%d" % e["n_flags"]) in r.stderr
