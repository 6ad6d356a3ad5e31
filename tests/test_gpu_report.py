"""GPU: row f2 end to end -- K1's counters (superset entry points: n_pair_all is COUNTED by the kernel) ->
report layer -> the samtools-flagstat text, equal to the oracle's restatement of the reference's samtools
loop (benchmark/flagstats.cpp:51-70, :577-588) and to the reference binary's own stdout on the golden files."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

U64 = np.uint64


def want_superset(oracle, flags):
    c = oracle.flagstat_hist(flags).copy()
    s = oracle.samtools_counts(flags)
    c[0], c[16] = s["n_pair_all"]
    c[9] = s["n_reads"][0]
    return c


def test_superset_slots_host_and_device(hip):
    import oracle
    from libflagstats_amd import _lib, device
    cases = [oracle.generate(oracle.GEN_NA12878, 5, 1, 0, 3_000_017),
             oracle.generate(oracle.GEN_UNIFORM, 6, 0xFFFF, 0, 2_500_003),       # full-range: every bit pattern
             np.arange(65536, dtype=np.uint16),
             np.zeros(0, dtype=np.uint16)]
    for a in cases:
        want = want_superset(oracle, a) if a.size else np.zeros(32, dtype=U64)
        out = np.zeros(32, dtype=U64)
        _lib.check(hip.FLAGSTATS_u16_x64_superset(a.ctypes.data if a.size else None, a.size, out.ctypes.data), "superset host")
        assert np.array_equal(out, want)
        # the scalar-exact entry on the same bytes leaves slots 0 / 9 / 16 untouched
        plain = np.zeros(32, dtype=U64)
        _lib.check(hip.FLAGSTATS_u16_x64(a.ctypes.data if a.size else None, a.size, plain.ctypes.data), "x64")
        assert np.array_equal(plain, oracle.flagstat_hist(a) if a.size else np.zeros(32, dtype=U64))
        if a.size:
            d = device.DeviceFlags(a.size).upload(a)
            out2 = np.zeros(32, dtype=U64)
            _lib.check(hip.FLAGSTATS_hip_device_u16_superset_sync(d.ptr, a.size, out2.ctypes.data), "superset device")
            assert np.array_equal(out2, want)
            d.free()
    # multi-chunk host streaming: pass-QC reads (slot 9 = n - fail) must add up over the chunks
    old = hip.FLAGSTATS_hip_get(b"chunk_flags")
    try:
        _lib.check(hip.FLAGSTATS_hip_set(b"chunk_flags", 400_003), "chunk")
        a = cases[1]
        out = np.zeros(32, dtype=U64)
        _lib.check(hip.FLAGSTATS_u16_x64_superset(a.ctypes.data, a.size, out.ctypes.data), "superset chunks")
        assert np.array_equal(out, want_superset(oracle, a))
    finally:
        hip.FLAGSTATS_hip_set(b"chunk_flags", old)


def test_device_arrays_to_samtools_text(hip):
    """Device-resident NA12878-like and full-range arrays -> HIP counters -> text == restated samtools loop."""
    import torch

    import oracle
    from libflagstats_amd import _lib, device
    from libflagstats_amd.report import flagstat_report, samtools_flagstat_text
    n = 50_000_021
    for kind, mask in ((device.GEN_NA12878, 1), (device.GEN_UNIFORM, 0xFFFF), (device.GEN_UNIFORM, 0x0FFF)):
        t = torch.empty(n, dtype=torch.int16, device="cuda:0")
        device.generate_torch(t, kind, seed=21, mask=mask)
        out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        import ctypes
        _lib.check(hip.FLAGSTATS_hip_device_u16_superset(t.data_ptr(), n, out.data_ptr(), ctypes.c_void_p(stream)), "superset")
        torch.cuda.synchronize()
        host = oracle.generate(kind, 21, mask, 0, n)
        want = oracle.samtools_text(oracle.samtools_counts(host))
        assert samtools_flagstat_text(out.cpu().numpy().view(U64), n) == want
    a = oracle.generate(oracle.GEN_UNIFORM, 3, 0xFFFF, 0, 1_000_001)
    assert flagstat_report(a) == oracle.samtools_text(oracle.samtools_counts(a))


def test_text_equals_reference_stdout_on_golden_blockfiles(hip):
    """Block file written by the reference -> decode + count on this engine (superset) -> the bytes the
    reference binary printed for `decompress -s` on the same file."""
    sys.path.insert(0, GOLDEN)
    from make_blockfiles import recipe_input
    from libflagstats_amd.report import flagstat_report
    man = json.load(open(os.path.join(GOLDEN, "blockfiles", "manifest.json")))
    seen = 0
    for name, e in man["files"].items():
        if e["reference_decompress_s_stdout"]:
            assert flagstat_report(recipe_input(e["n_flags"], e["seed"])) == e["reference_decompress_s_stdout"], name
            seen += 1
    assert seen >= 3


def test_file_to_text_is_the_reference_binarys_stdout(hip, tmp_path):
    """`bench decompress -i FILE -s` end to end on this engine: the reference-written golden block file is
    decoded and counted by the pipeline with SUPERSET counters (n_pair_all counted by K1; the zero flags the
    pipeline pads chunks with must not count as pass-QC reads) and the text must be byte-identical to what the
    reference binary printed for that very file.  `-S`: the same flags as a raw uint16 file."""
    sys.path.insert(0, GOLDEN)
    import oracle
    from make_blockfiles import recipe_input
    from libflagstats_amd import blockfile
    from libflagstats_amd.report import flagstat_report_file
    man = json.load(open(os.path.join(GOLDEN, "blockfiles", "manifest.json")))
    seen = 0
    for name, e in man["files"].items():
        path = os.path.join(GOLDEN, "blockfiles", name)
        flags = recipe_input(e["n_flags"], e["seed"])
        want_counts = want_superset(oracle, flags)
        for threads in (1, 5):
            got, st = blockfile.flagstat_file(path, threads, superset=True)
            assert np.array_equal(got, want_counts), (name, threads)
        text = flagstat_report_file(path)
        assert text == oracle.samtools_text(oracle.samtools_counts(flags)), name
        if e["reference_decompress_s_stdout"]:
            assert text == e["reference_decompress_s_stdout"], name
            seen += 1
        raw = tmp_path / (name + ".bin")
        raw.write_bytes(flags.tobytes() + b"\x01")          # odd trailing byte: dropped, and not a read either
        got, st = blockfile.flagstat_raw_file(str(raw), superset=True)
        assert np.array_equal(got, want_counts) and st["n_flags"] == flags.size, name
        assert flagstat_report_file(str(raw)) == text
    assert seen >= 3
