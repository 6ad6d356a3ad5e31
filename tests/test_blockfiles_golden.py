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


@pytest.mark.parametrize("name", sorted(MANIFEST["files"]))
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


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MANIFEST["files"]))
def test_pipeline_counts_reference_written_files(hip, name):
    from libflagstats_amd import blockfile
    e = MANIFEST["files"][name]
    want = np.array(e["scalar_counters"], dtype=np.uint64)
    for threads in (1, 4):
        got, st = blockfile.flagstat_lz4_file(os.path.join(BF, name), threads=threads)
        assert np.array_equal(got, want), (name, threads)
        assert st["n_flags"] == e["n_flags"]
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
