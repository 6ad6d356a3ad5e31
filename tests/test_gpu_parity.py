"""GPU parity: the HIP path, called through the C-ABI of libflagstats_hip.so, against
the reference's golden vectors and the oracle.  Bit-exact on all 32 slots
(integer counters: no tolerance)."""
import ctypes

import numpy as np
import pytest

from conftest import case_input, inmemory_input, load_golden

pytestmark = pytest.mark.gpu

U64 = np.uint64


def capi_u16(hip, a, flags=None):
    """FLAGSTATS_u16(host pointer, uint32 n, uint32 flags[32]) -- the drop-in entry."""
    from libflagstats_amd import _lib
    if flags is None:
        flags = np.zeros(32, dtype=np.uint32)
    rc = hip.FLAGSTATS_u16(a.ctypes.data if a.size else None, a.size, flags.ctypes.data)
    _lib.check(int(rc), "FLAGSTATS_u16")
    return flags


def capi_x64(hip, a):
    from libflagstats_amd import _lib
    out = np.zeros(32, dtype=np.uint64)
    _lib.check(hip.FLAGSTATS_u16_x64(a.ctypes.data if a.size else None, a.size, out.ctypes.data), "x64")
    return out


# --------------------------------------------------------------------------- golden vectors
def test_single_flag_kats(hip):
    for e in load_golden("kat.json")["single"]:
        x = np.array([e["x"]], dtype=np.uint16)
        want = np.zeros(32, dtype=np.uint32)
        want[e["slots"]] = 1
        assert np.array_equal(capi_u16(hip, x), want), e


@pytest.mark.parametrize("K", ["4096", "65536"])
def test_exhaustive_kats(hip, K):
    want = np.array(load_golden("kat.json")["exhaustive"][K]["scalar"], dtype=np.uint32)
    a = np.arange(int(K), dtype=np.uint32).astype(np.uint16)
    assert np.array_equal(capi_u16(hip, a), want)
    assert np.array_equal(capi_x64(hip, a), want.astype(U64))


def test_golden_random_cases_host_entry(hip):
    """Every length that straddles a block boundary of any reference kernel, 12-bit and
    full-range inputs, 16-byte- and 2-byte-aligned host pointers (tests/golden/random_cases.json)."""
    cases = load_golden("random_cases.json")["cases"]
    for case in cases:
        a = case_input(case)
        want = np.array(case["scalar"], dtype=np.uint32)
        got = capi_u16(hip, a)
        assert np.array_equal(got, want), (case["seed"], case["n"], case["skip"], got, want)


def test_golden_random_cases_device_entry(hip):
    """Same fixtures through the device-resident entry, with the array placed at every
    2-byte offset of a 16-byte line (ragged head and tail inside the kernel)."""
    from libflagstats_amd import device
    cases = [c for c in load_golden("random_cases.json")["cases"] if c["n"] <= 131073]
    buf = device.DeviceFlags(131073 + 16)
    for i, case in enumerate(cases):
        a = case_input(case)
        off = i % 8
        buf.upload(a, offset=off)
        got = buf.count(offset=off, n=a.size)
        assert np.array_equal(got, np.array(case["scalar"], dtype=U64)), (case["seed"], case["n"], off)
    buf.free()


def test_accumulate_contract(hip):
    """flags[] is added to, never zeroed (libflagstats.h:118-142; pyx:19 zeroes in the caller)."""
    g = load_golden("accumulate.json")
    a = np.random.RandomState(7).randint(0, 65536, 5000).astype(np.uint16)
    b = np.random.RandomState(8).randint(0, 4096, 3000).astype(np.uint16)
    flags = np.array(g["start"], dtype=np.uint32)
    capi_u16(hip, a, flags)
    assert [int(v) for v in flags] == g["after_a"]
    capi_u16(hip, b, flags)
    assert [int(v) for v in flags] == g["after_b"]


def test_get_function_returns_callable_kernel(hip):
    """FLAGSTATS_get_function(n)(array, n, flags) as benchmark/flagstats.cpp:328-329 uses it."""
    a = np.random.RandomState(3).randint(0, 65536, 70000).astype(np.uint16)
    fn = hip.FLAGSTATS_get_function(a.size)
    flags = np.zeros(32, dtype=np.uint32)
    p16 = ctypes.POINTER(ctypes.c_uint16)
    p32 = ctypes.POINTER(ctypes.c_uint32)
    assert fn(a.ctypes.data_as(p16), a.size, flags.ctypes.data_as(p32)) == 0
    import oracle
    assert np.array_equal(flags.astype(U64), oracle.flagstat_hist(a))


@pytest.mark.parametrize("n", ["102400", "1000000"])
def test_inmemory_harness_case(hip, n):
    """benchmark/inmemory.cpp's own input (mt19937 seed 0, U[0,4095]); BASELINE config 0 at n=1M."""
    g = load_golden("inmemory_mt19937.json")["cases"][n]
    a = inmemory_input(int(n))
    assert np.array_equal(capi_u16(hip, a), np.array(g["scalar"], dtype=np.uint32))


def test_empty_and_null(hip):
    z = np.zeros(0, dtype=np.uint16)
    flags = np.full(32, 7, dtype=np.uint32)
    assert np.array_equal(capi_u16(hip, z, flags), np.full(32, 7, dtype=np.uint32))  # n == 0: no-op


# --------------------------------------------------------------------------- oracle, seeded inputs
@pytest.mark.parametrize("variant", [0, 1, 9, 13, 17, 25, 27, 29, 41, 61, 63, 65, 67, 69, 71, 75, 77, 79, 81, 89, 153])
def test_kernel_variants_agree_with_oracle(hip, variant):
    import oracle
    from libflagstats_amd import _lib, device
    if variant not in (9, 25, 71) and not hip.FLAGSTATS_hip_get(b"tuning_build"):
        # the schedules that lost the r01 sweeps are compiled only into `make TUNING=1`; the shipped
        # library must refuse them loudly instead of launching something else
        assert hip.FLAGSTATS_hip_set(b"variant", variant) != 0
        assert b"TUNING=1" in hip.FLAGSTATS_hip_last_error()
        return
    old = hip.FLAGSTATS_hip_get(b"variant")
    _lib.check(hip.FLAGSTATS_hip_set(b"variant", variant), "set variant")
    try:
        n = 40_000_003
        d = device.DeviceFlags(n + 8).generate(device.GEN_UNIFORM, seed=21 + variant, mask=0xFFFF)
        for off, cnt in ((0, n), (3, n - 11), (7, 16384 * 5 + 1)):
            want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 21 + variant, 0xFFFF, off, cnt)
            assert np.array_equal(d.count(offset=off, n=cnt), want), (variant, off, cnt)
        d.free()
    finally:
        hip.FLAGSTATS_hip_set(b"variant", old)


@pytest.mark.parametrize("bpc", [1, 2, 5, 8])
def test_any_grid_size(hip, bpc):
    """Result must not depend on the launch geometry (epoch flushes, ragged last step)."""
    import oracle
    from libflagstats_amd import _lib, device
    old = hip.FLAGSTATS_hip_get(b"blocks_per_cu")
    _lib.check(hip.FLAGSTATS_hip_set(b"blocks_per_cu", bpc), "set blocks_per_cu")
    try:
        n = 16384 * 700 + 12345
        d = device.DeviceFlags(n).generate(device.GEN_NA12878, seed=5, mask=1)
        want = oracle.flagstat_generated(oracle.GEN_NA12878, 5, 1, 0, n)
        assert np.array_equal(d.count(), want)
        d.free()
    finally:
        hip.FLAGSTATS_hip_set(b"blocks_per_cu", old)


def test_multi_epoch_single_workgroup_column(hip):
    """Force > 2^DEPTH-1 steps per workgroup so the bit-sliced planes are flushed mid-run."""
    import oracle
    from libflagstats_amd import _lib, device
    old = hip.FLAGSTATS_hip_get(b"blocks_per_cu")
    # 1 block per CU and an array of > 256 CUs * 600 steps
    _lib.check(hip.FLAGSTATS_hip_set(b"blocks_per_cu", 1), "set")
    try:
        cus = hip.FLAGSTATS_hip_compute_units()
        n = 16384 * cus * 600 + 777
        d = device.DeviceFlags(n).generate(device.GEN_UNIFORM, seed=1234, mask=0xFFFF)
        want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 1234, 0xFFFF, 0, n)
        assert np.array_equal(d.count(), want)
        d.free()
    finally:
        hip.FLAGSTATS_hip_set(b"blocks_per_cu", old)


def test_generators_match_host_twins(hip):
    """On-device input makers == oracle twins, byte for byte, incl. ragged device offsets."""
    import oracle
    from libflagstats_amd import device
    n = 300_007
    d = device.DeviceFlags(n + 16)
    for kind, mask in ((0, 0xFFFF), (0, 0x0FFF), (1, 0), (1, 1), (2, 0)):
        for off, first in ((0, 0), (3, 1001), (5, 2 ** 33 + 6)):
            d.generate(kind, seed=99, mask=mask, first_index=first, offset=off, n=n)
            got = d.download(offset=off, n=n)
            want = oracle.generate(kind, 99, mask, first, n)
            assert np.array_equal(got, want), (kind, mask, off, first)
    d.free()


def test_host_streaming_chunks_and_pinned(hip):
    """FLAGSTATS_u16_x64 double-buffers H2D chunks; tiny chunks force many ragged launches."""
    import oracle
    from libflagstats_amd import _lib
    a = np.random.RandomState(17).randint(0, 65536, 5_000_011).astype(np.uint16)
    want = oracle.flagstat_hist(a)
    old = hip.FLAGSTATS_hip_get(b"chunk_flags")
    try:
        for chunk in (1_000_003, 65536, old):
            _lib.check(hip.FLAGSTATS_hip_set(b"chunk_flags", chunk), "set chunk")
            assert np.array_equal(capi_x64(hip, a), want), chunk
            assert np.array_equal(capi_x64(hip, a[1:]), oracle.flagstat_hist(a[1:])), chunk
        # pinned host memory from the library's allocator
        p = hip.FLAGSTATS_hip_host_alloc(a.nbytes)
        assert p
        ctypes.memmove(p, a.ctypes.data, a.nbytes)
        out = np.zeros(32, dtype=np.uint64)
        _lib.check(hip.FLAGSTATS_u16_x64(p, a.size, out.ctypes.data), "x64 pinned")
        assert np.array_equal(out, want)
        hip.FLAGSTATS_hip_host_free(p)
    finally:
        hip.FLAGSTATS_hip_set(b"chunk_flags", old)


def test_large_pageable_host_arrays_go_through_the_page_locked_chunks(hip):
    """r05: a host array in PAGEABLE memory of at least `staged_min_flags` flags (default 2^27 = 256 MiB) is copied by worker
    threads into the engine's page-locked chunks instead of being handed to hipMemcpyAsync, which pins such memory as it goes
    (profiles/r05/pageable_c.log).  Same counters whichever way: ragged sizes, odd starts, tiny chunks, the superset entry, the
    32-bit entry; page-locked arrays and small ones do not take the rule."""
    import oracle
    from libflagstats_amd import _lib
    assert hip.FLAGSTATS_hip_get(b"staged_min_flags") == 1 << 27
    a = np.random.RandomState(23).randint(0, 65536, 6_000_013).astype(np.uint16)
    old = hip.FLAGSTATS_hip_get(b"chunk_flags")
    try:
        calls = hip.FLAGSTATS_hip_get(b"staged_calls")
        assert np.array_equal(capi_x64(hip, a), oracle.flagstat_hist(a))
        assert hip.FLAGSTATS_hip_get(b"staged_calls") == calls            # 12 MB: below the rule
        _lib.check(hip.FLAGSTATS_hip_set(b"staged_min_flags", 1 << 20), "set")
        for chunk in (old, 1_000_003, 70_000):
            _lib.check(hip.FLAGSTATS_hip_set(b"chunk_flags", chunk), "set chunk")
            for lo, hi in ((0, a.size), (1, a.size), (3, a.size - 5), (7, (1 << 20) + 7)):
                calls = hip.FLAGSTATS_hip_get(b"staged_calls")
                assert np.array_equal(capi_x64(hip, a[lo:hi]), oracle.flagstat_hist(a[lo:hi])), (chunk, lo, hi)
                assert hip.FLAGSTATS_hip_get(b"staged_calls") == calls + 1, (chunk, lo, hi)
        _lib.check(hip.FLAGSTATS_hip_set(b"chunk_flags", old), "set chunk")
        # the superset entry (slots 0 / 16 / 9 counted too) and the reference-shaped 32-bit entry
        out = np.zeros(32, dtype=np.uint64)
        calls = hip.FLAGSTATS_hip_get(b"staged_calls")
        _lib.check(hip.FLAGSTATS_u16_x64_superset(a.ctypes.data, a.size, out.ctypes.data), "superset")
        want = oracle.flagstat_hist(a).copy()
        pp = ((a & 0x100) == 0) & ((a & 0x800) == 0) & ((a & 1) == 1)
        fail = (a & 0x200) != 0
        want[0], want[16], want[9] = int((pp & ~fail).sum()), int((pp & fail).sum()), a.size - int(fail.sum())
        assert np.array_equal(out, want) and hip.FLAGSTATS_hip_get(b"staged_calls") == calls + 1
        o32 = np.zeros(32, dtype=np.uint32)
        assert hip.FLAGSTATS_u16(a.ctypes.data, a.size, o32.ctypes.data) == 0
        assert np.array_equal(o32.astype(np.uint64), oracle.flagstat_hist(a)) and hip.FLAGSTATS_hip_get(b"staged_calls") == calls + 2
        # page-locked memory is the runtime's fast path already: not staged
        p = hip.FLAGSTATS_hip_host_alloc(a.nbytes)
        assert p
        ctypes.memmove(p, a.ctypes.data, a.nbytes)
        out = np.zeros(32, dtype=np.uint64)
        calls = hip.FLAGSTATS_hip_get(b"staged_calls")
        _lib.check(hip.FLAGSTATS_u16_x64(p, a.size, out.ctypes.data), "x64 pinned")
        assert np.array_equal(out, oracle.flagstat_hist(a)) and hip.FLAGSTATS_hip_get(b"staged_calls") == calls
        hip.FLAGSTATS_hip_host_free(p)
        # a failed call leaves the counters alone: NULL array
        out = np.full(32, 5, dtype=np.uint64)
        assert hip.FLAGSTATS_u16_x64(None, 1 << 21, out.ctypes.data) != 0 and (out == 5).all()
    finally:
        hip.FLAGSTATS_hip_set(b"chunk_flags", old)
        hip.FLAGSTATS_hip_set(b"staged_min_flags", 1 << 27)
    # at the shipped threshold: 2^27 + 12345 flags (256 MiB) of pageable memory take the rule
    n = (1 << 27) + 12345
    big = oracle.generate(oracle.GEN_NA12878, 31, 1, 0, n)
    calls = hip.FLAGSTATS_hip_get(b"staged_calls")
    assert np.array_equal(capi_x64(hip, big), oracle.flagstat_generated(oracle.GEN_NA12878, 31, 1, 0, n))
    assert hip.FLAGSTATS_hip_get(b"staged_calls") == calls + 1


# --------------------------------------------------------------------------- full-size properties
def test_full_size_ramp_is_exact_multiple_of_kat(hip):
    """8 GiB (BASELINE metric size): 2^32 flags = 65536 repetitions of the exhaustive 0..65535
    sweep must give exactly 65536 x the reference's K=65536 vector; slot 25 = 2^31 needs the
    64-bit counters."""
    from libflagstats_amd import device
    kat = np.array(load_golden("kat.json")["exhaustive"]["65536"]["scalar"], dtype=U64)
    n = 2 ** 32
    d = device.DeviceFlags(n).generate(device.GEN_RAMP, seed=0)
    got = d.count()
    assert np.array_equal(got, kat * U64(65536))
    assert int(got[25]) == 2 ** 31
    # linearity: any split, at odd offsets, sums to the whole
    cut = 2 ** 31 + 12345
    assert np.array_equal(d.count(0, cut) + d.count(cut, n - cut), got)
    d.free()


def test_one_gib_uniform_vs_oracle(hip):
    """BASELINE config 1: 1 GiB uniform-random uint16 on one MI355X vs the oracle on identical
    bytes (regenerated chunk-wise on the host from the same counter-based generator)."""
    import oracle
    from libflagstats_amd import device
    n = 2 ** 29
    d = device.DeviceFlags(n).generate(device.GEN_UNIFORM, seed=2026, mask=0xFFFF)
    got = d.count()
    d.free()
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 2026, 0xFFFF, 0, n)
    assert np.array_equal(got, want)


def test_torch_stream_entry(hip):
    """Device entry on torch's current stream with device-side counters (what bench.py times)."""
    import torch

    import oracle
    from libflagstats_amd import device
    n = 10_000_019
    t = torch.empty(n, dtype=torch.int16, device="cuda:0")
    device.generate_torch(t, device.GEN_UNIFORM, seed=31, mask=0xFFFF)
    out = device.count_torch(t)
    device.count_torch(t, out)  # accumulates
    torch.cuda.synchronize()
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 31, 0xFFFF, 0, n)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want * U64(2))
    # store form: overwrites whatever is there, writes every slot (dead ones as 0)
    junk = torch.full((32,), 12345, dtype=torch.int64, device="cuda:0")
    device.count_torch(t, junk, store=True)
    device.count_torch(t[:0], out, store=True)       # n == 0 stores zeros
    torch.cuda.synchronize()
    assert np.array_equal(junk.cpu().numpy().view(np.uint64), want)
    assert not out.cpu().numpy().any()
