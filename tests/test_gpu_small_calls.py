"""GPU: the small-call path of the host-pointer entries (what an unmodified per-block / per-read caller of the reference
hits): input copied into a pinned buffer that K1 reads in place, counters stored to pinned host memory by the last kernel,
completion by a polled word.  Every call reuses the same pinned addresses, so stale data anywhere (GPU caches, the
result buffer, the completion word) would show up as a wrong count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def capi(hip, a, flags=None):
    flags = np.zeros(32, dtype=np.uint32) if flags is None else flags
    rc = hip.FLAGSTATS_u16(a.ctypes.data if a.size else None, a.size, flags.ctypes.data)
    assert rc == 0
    return flags


@pytest.mark.parametrize("bar", [1, 0])
@pytest.mark.parametrize("poll,small_flags", [(1, 1 << 20), (1, 131072), (1, 0), (0, 131072), (0, 0), (1, 10 ** 9)])
def test_many_small_calls_with_changing_data(hip, poll, small_flags, bar):
    """bar = 1: the input buffer is device memory written by the CPU through the PCIe BAR (where the device has a large
    BAR); bar = 0: pinned host memory read in place.  The buffer kind is fixed when an engine is created, hence the
    shutdown / re-init around each setting."""
    import oracle
    from libflagstats_amd import _lib
    old = {k: hip.FLAGSTATS_hip_get(k) for k in (b"poll", b"small_flags", b"small_bar")}
    _lib.check(hip.FLAGSTATS_hip_set(b"poll", poll), "poll")
    _lib.check(hip.FLAGSTATS_hip_set(b"small_flags", small_flags), "small_flags")
    _lib.check(hip.FLAGSTATS_hip_set(b"small_bar", bar), "small_bar")
    hip.FLAGSTATS_hip_shutdown()
    _lib.check(hip.FLAGSTATS_hip_init(0), "init")
    assert bar or not hip.FLAGSTATS_hip_get(b"small_in_is_device")
    try:
        rs = np.random.RandomState(77)
        pool = oracle.generate(oracle.GEN_UNIFORM, 123, 0xFFFF, 0, 1_200_000)
        sizes = [1, 2, 7, 8, 9, 999, 1000, 16383, 16384, 16385, 16384 * 2 + 3, 50_000, 131072, 131073, 200_000, 600_000, 1_100_000]
        for it in range(400):
            n = int(rs.choice(sizes))
            off = int(rs.randint(0, pool.size - n + 1))
            a = pool[off:off + n]
            acc = np.full(32, 5, dtype=np.uint32)            # the entry accumulates onto the caller's counters
            got = capi(hip, a, acc)
            want = oracle.flagstat_hist(a) + np.uint64(5)
            assert np.array_equal(got.astype(np.uint64), want), (it, n, off, poll, small_flags)
    finally:
        for k, v in old.items():
            hip.FLAGSTATS_hip_set(k, v)
        hip.FLAGSTATS_hip_shutdown()
        _lib.check(hip.FLAGSTATS_hip_init(0), "init")


def test_small_calls_from_several_threads(hip):
    """The polled completion word and the pinned buffers belong to the engine; concurrent callers are serialised by
    its lock and each gets its own counters."""
    import threading

    import oracle
    arrays = [oracle.generate(oracle.GEN_NA12878, 200 + i, 1, 0, 3000 + 1111 * i) for i in range(6)]
    want = [oracle.flagstat_hist(a) for a in arrays]
    errs = []

    def work(i):
        try:
            for _ in range(300):
                got = capi(hip, arrays[i])
                assert np.array_equal(got.astype(np.uint64), want[i])
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(arrays))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs


def test_superset_small_call(hip):
    import oracle
    from libflagstats_amd import _lib
    a = oracle.generate(oracle.GEN_UNIFORM, 9, 0xFFFF, 0, 12345)
    out = np.zeros(32, dtype=np.uint64)
    _lib.check(hip.FLAGSTATS_u16_x64_superset(a.ctypes.data, a.size, out.ctypes.data), "superset")
    want = oracle.flagstat_hist(a).copy()
    pp = ((a & 0x100) == 0) & ((a & 0x800) == 0) & ((a & 1) == 1)
    fail = (a & 0x200) != 0
    want[0], want[16], want[9] = int((pp & ~fail).sum()), int((pp & fail).sum()), a.size - int(fail.sum())
    assert np.array_equal(out, want)
