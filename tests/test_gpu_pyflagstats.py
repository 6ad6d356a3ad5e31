"""GPU: pyflagstats.flagstats (mirror of python/libflagstats.pyx) against dicts captured from
the reference's own Cython module (tests/golden/pyflagstats.json)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

# keys whose values come from the 19 live counters (libflagstats.h:118-142) + derived fields
LIVE_PASSED = ["FUNMAP", "FREAD1", "FREAD2", "FSECONDARY", "FDUP", "FSUPPLEMENTARY", "n_pair_good", "n_sgltn",
               "n_pair_map", "mapped", "paired_in_seq"]
LIVE_FAILED = ["FUNMAP", "FREAD1", "FREAD2", "FSECONDARY", "FQCFAIL", "FDUP", "FSUPPLEMENTARY", "n_pair_good",
               "n_sgltn", "n_pair_map"]
# keys the x86 SIMD kernels fill as a side effect for n >= 256; scalar leaves them 0
SUPERSET = ["FPAIRED", "FPROPER_PAIR", "FMUNMAP", "FREVERSE", "FMREVERSE"]


def test_dicts_match_reference_module(hip):
    import pyflagstats  # top-level alias, as the reference's users import it
    g = load_golden("pyflagstats.json")
    for key, want in g["dicts"].items():
        hi, n = (int(v) for v in key.split("_"))
        a = np.random.RandomState(0).randint(0, hi, n).astype(np.uint16)
        got = pyflagstats.flagstats(a)
        assert list(got.keys()) == ["n_values", "passed", "failed"]
        assert list(got["passed"].keys()) == want["passed_keys"]
        assert list(got["failed"].keys()) == want["failed_keys"]
        assert got["n_values"] == want["n_values"]
        assert type(got["passed"]["FUNMAP"]).__name__ == want["value_type"] == "uint32"
        if hi == 4096:  # raw bits 12-15 clear: the reference's SIMD path agrees on the live slots
            for k in LIVE_PASSED:
                assert int(got["passed"][k]) == want["passed"][k], (key, k)
            for k in LIVE_FAILED:
                assert int(got["failed"][k]) == want["failed"][k], (key, k)
        if n < 256:     # reference dispatches to FLAGSTAT_scalar: every key must agree
            for k, v in want["passed"].items():
                assert int(got["passed"][k]) == v, (key, k)
            for k, v in want["failed"].items():
                assert int(got["failed"][k]) == v, (key, k)
        for k in SUPERSET:  # scalar-exact contract: never written
            assert int(got["passed"][k]) == 0 and int(got["failed"][k]) == 0
        assert int(got["passed"]["FQCFAIL"]) == 0


def test_dict_values_equal_oracle_restatement(hip):
    import oracle
    import pyflagstats
    a = np.random.RandomState(5).randint(0, 65536, 123457).astype(np.uint16)
    got = pyflagstats.flagstats(a)
    want = oracle.pyflagstats_dict(oracle.flagstat_hist(a).astype(np.uint32), a.size)
    assert {k: int(v) for k, v in got["passed"].items()} == {k: int(v) for k, v in want["passed"].items()}
    assert {k: int(v) for k, v in got["failed"].items()} == {k: int(v) for k, v in want["failed"].items()}


def test_noncontiguous_input(hip, capsys):
    import pyflagstats
    g = load_golden("pyflagstats.json")["noncontig"]
    a = np.random.RandomState(0).randint(0, 4096, 2000).astype(np.uint16)
    got = pyflagstats.flagstats(a[::2])
    assert capsys.readouterr().out == g["stdout"]
    for k in LIVE_PASSED:
        assert int(got["passed"][k]) == g["dict"]["passed"][k]
    for k in LIVE_FAILED:
        assert int(got["failed"][k]) == g["dict"]["failed"][k]


def test_x64_entry(hip):
    import oracle
    import pyflagstats
    a = np.random.RandomState(6).randint(0, 65536, 70001).astype(np.uint16)
    got = pyflagstats.flagstats_x64(a)
    want = oracle.pyflagstats_dict(oracle.flagstat_hist(a), a.size)
    assert {k: int(v) for k, v in got["failed"].items()} == {k: int(v) for k, v in want["failed"].items()}
    assert type(got["passed"]["FUNMAP"]).__name__ == "uint64"
