"""CPU: the oracle against the reference's golden vectors (and the reference itself
when oracle/_ref is present).  No GPU, no product code."""
import hashlib

import numpy as np
import pytest

from conftest import case_input, inmemory_input, load_golden

IMPLS = ["flagstat_numpy", "flagstat_c", "flagstat_hist", "flagstat_mt"]


def test_single_flag_kats(oracle_mod):
    kat = load_golden("kat.json")
    for e in kat["single"]:
        x = np.array([e["x"]], dtype=np.uint16)
        want = np.zeros(32, dtype=np.uint64)
        want[e["slots"]] = 1
        assert np.array_equal(oracle_mod.flagstat_python(x), want), e
        assert np.array_equal(oracle_mod.flagstat_numpy(x), want), e
        assert np.array_equal(oracle_mod.flagstat_c(x), want), e


@pytest.mark.parametrize("K", ["4096", "65536"])
def test_exhaustive_kats(oracle_mod, K):
    kat = load_golden("kat.json")["exhaustive"][K]
    a = np.arange(int(K), dtype=np.uint32).astype(np.uint16)
    want = np.array(kat["scalar"], dtype=np.uint64)
    for name in IMPLS:
        assert np.array_equal(getattr(oracle_mod, name)(a), want), name
    if K == "4096":
        assert np.array_equal(oracle_mod.flagstat_python(a), want)
    # the one reference SIMD variant that is scalar-exact on all 32 slots
    if kat["avx512_improved3"] is not None:
        assert kat["avx512_improved3"] == kat["scalar"]
    # ... and the dispatcher's default is NOT (superset slots): documents why
    # parity is defined against FLAGSTAT_scalar
    if kat["avx512"] is not None:
        assert kat["avx512"] != kat["scalar"]
        live = list(oracle_mod.LIVE_SLOTS)
        if K == "4096":  # raw bits 12-15 clear -> live slots agree
            assert [kat["avx512"][i] for i in live] == [kat["scalar"][i] for i in live]


def test_random_cases(oracle_mod):
    g = load_golden("random_cases.json")
    assert len(g["cases"]) > 150
    for case in g["cases"]:
        a = case_input(case)
        want = np.array(case["scalar"], dtype=np.uint64)
        impls = IMPLS if case["n"] <= 131073 else ["flagstat_numpy", "flagstat_hist", "flagstat_mt"]
        for name in impls:
            got = getattr(oracle_mod, name)(a)
            assert np.array_equal(got, want), (name, case["seed"], case["n"], case["skip"])
        if case["n"] <= 257:
            assert np.array_equal(oracle_mod.flagstat_python(a), want)
        # only the 19 live slots may ever be non-zero
        dead = [i for i in range(32) if i not in oracle_mod.LIVE_SLOTS]
        assert not want[dead].any()


def test_accumulate_contract(oracle_mod):
    g = load_golden("accumulate.json")
    a = np.random.RandomState(7).randint(0, 65536, 5000).astype(np.uint16)
    b = np.random.RandomState(8).randint(0, 4096, 3000).astype(np.uint16)
    flags = np.array(g["start"], dtype=np.uint32)
    import ctypes
    lib = oracle_mod.load_c()
    p16 = ctypes.POINTER(ctypes.c_uint16)
    p32 = ctypes.POINTER(ctypes.c_uint32)
    lib.oracle_FLAGSTAT_scalar(a.ctypes.data_as(p16), a.size, flags.ctypes.data_as(p32))
    assert [int(v) for v in flags] == g["after_a"]
    lib.oracle_FLAGSTAT_scalar(b.ctypes.data_as(p16), b.size, flags.ctypes.data_as(p32))
    assert [int(v) for v in flags] == g["after_b"]


@pytest.mark.parametrize("n", ["102400", "1000000"])
def test_inmemory_harness_input(oracle_mod, n):
    g = load_golden("inmemory_mt19937.json")["cases"][n]
    a = inmemory_input(int(n))
    assert [int(v) for v in a[:16]] == g["first16"]
    assert hashlib.sha256(a.tobytes()).hexdigest() == g["sha256"]
    want = np.array(g["scalar"], dtype=np.uint64)
    assert np.array_equal(oracle_mod.flagstat_hist(a), want)
    assert np.array_equal(oracle_mod.flagstat_numpy(a), want)
    assert g["avx512_improved3"] is None or g["avx512_improved3"] == g["scalar"]


def test_against_reference_build_if_present(oracle_mod):
    """oracle/_ref = the reference's own kernels compiled from /root/reference."""
    if oracle_mod.load_ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rs = np.random.RandomState(4242)
    for n in (0, 1, 511, 512, 513, 70001):
        for hi in (4096, 65536):
            a = rs.randint(0, hi, n + 1).astype(np.uint16)[1:]
            ref = oracle_mod.ref_call("FLAGSTAT_scalar", a)
            assert np.array_equal(ref.astype(np.uint64), oracle_mod.flagstat_c(a))
            assert np.array_equal(ref.astype(np.uint64), oracle_mod.flagstat_numpy(a))
            r3 = oracle_mod.ref_call("FLAGSTAT_avx512_improved3", a)
            if r3 is not None:
                assert np.array_equal(r3, ref)


def test_generators_are_index_addressable(oracle_mod):
    o = oracle_mod
    for kind, mask in ((o.GEN_UNIFORM, 0xFFFF), (o.GEN_UNIFORM, 0x0FFF), (o.GEN_NA12878, 0), (o.GEN_NA12878, 1),
                       (o.GEN_RAMP, 0)):
        whole = o.generate(kind, 77, mask, 1000, 5000)
        part = o.generate(kind, 77, mask, 1000 + 1234, 777)
        assert np.array_equal(whole[1234:1234 + 777], part)
        if kind == o.GEN_UNIFORM:
            assert int(whole.max()) <= mask
        got = o.flagstat_generated(kind, 77, mask, 1000, 5000, threads=3)
        assert np.array_equal(got, o.flagstat_c(whole))
    ramp = o.generate(o.GEN_RAMP, 0, 0, 0, 65536)
    assert np.array_equal(ramp, np.arange(65536, dtype=np.uint32).astype(np.uint16))


def test_na12878_marginals(oracle_mod):
    """The categorical maker reproduces the README.md:178-192 marginals (within sampling noise)."""
    o = oracle_mod
    n = 4_000_000
    c = o.flagstat_generated(o.GEN_NA12878, 5, 0, 0, n)
    N = 824541892
    assert c[16:].sum() == 0 and c[8] == 0 and c[10] == 0          # no QC-fail, secondary, dup
    assert abs(c[11] / n - 5393628 / N) < 5e-4                       # supplementary
    assert abs(c[12] / n - 781085884 / N) < 1e-3                     # properly paired
    assert abs(c[14] / n - 797950890 / N) < 1e-3                     # with itself and mate mapped
    assert abs(c[13] / n - 2038885 / N) < 2e-4                       # singletons
    assert abs((c[6] + c[7]) / n - 819148264 / N) < 1e-3             # paired in sequencing
    assert abs((n - c[2]) / n - 805383403 / N) < 1e-3                # mapped
