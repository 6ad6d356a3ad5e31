"""GPU: behaviour around the edges of the C-ABI contract: no alignment
requirement on the host pointer (the reference uses loadu everywhere, libflagstats.h:281,1064,1695),
reentrancy from several threads, shutdown and lazy re-initialisation, error reporting."""
import ctypes
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_host_pointer_at_odd_byte_address(hip):
    import oracle
    from libflagstats_amd import _lib
    n = 1_000_003
    flags = oracle.generate(oracle.GEN_UNIFORM, 8, 0xFFFF, 0, n)
    raw = bytearray(2 * n + 1)
    raw[1:] = flags.tobytes()                                # the array starts at an ODD byte address
    base = (ctypes.c_char * len(raw)).from_buffer(raw)
    out = np.zeros(32, dtype=np.uint32)
    rc = hip.FLAGSTATS_u16(ctypes.addressof(base) + 1, n, out.ctypes.data)
    _lib.check(int(rc), "FLAGSTATS_u16")
    assert np.array_equal(out.astype(np.uint64), oracle.flagstat_hist(flags))


def test_concurrent_callers(hip):
    """The reference is reentrant (no shared state but the cpuid cache); here calls from several
    threads are serialised inside the library and must each get their own, exact, counters."""
    import oracle
    arrays = [oracle.generate(oracle.GEN_UNIFORM, 100 + i, 0xFFFF, 0, 200_000 + 17 * i) for i in range(8)]
    want = [oracle.flagstat_hist(a) for a in arrays]
    got = [None] * len(arrays)
    errs = []

    def work(i):
        try:
            for _ in range(20):
                out = np.zeros(32, dtype=np.uint32)
                rc = hip.FLAGSTATS_u16(arrays[i].ctypes.data, arrays[i].size, out.ctypes.data)
                assert rc == 0
                got[i] = out
                assert np.array_equal(out.astype(np.uint64), want[i])
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(arrays))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs


def test_errors_are_loud_and_recoverable(hip):
    import oracle
    from libflagstats_amd import _lib, device
    out = np.zeros(32, dtype=np.uint64)
    # device entry: a uint16_t* must be 2-byte aligned
    d = device.DeviceFlags(1024)
    rc = hip.FLAGSTATS_hip_device_u16_sync(d.ptr + 1, 10, out.ctypes.data)
    assert rc != 0 and b"aligned" in hip.FLAGSTATS_hip_last_error()
    rc = hip.FLAGSTATS_u16_x64(None, 5, out.ctypes.data)          # NULL array with n > 0
    assert rc != 0 and b"NULL" in hip.FLAGSTATS_hip_last_error()
    assert hip.FLAGSTATS_hip_set(b"no_such_knob", 1) != 0
    assert not out.any()
    d.free()
    # still healthy afterwards
    a = oracle.generate(oracle.GEN_NA12878, 2, 1, 0, 70_001)
    _lib.check(hip.FLAGSTATS_u16_x64(a.ctypes.data, a.size, out.ctypes.data), "x64")
    assert np.array_equal(out, oracle.flagstat_hist(a))


def test_shutdown_and_lazy_reinit(hip):
    import oracle
    from libflagstats_amd import _lib
    a = oracle.generate(oracle.GEN_UNIFORM, 3, 0xFFFF, 0, 123_457)
    want = oracle.flagstat_hist(a)
    knobs = {k: hip.FLAGSTATS_hip_get(k) for k in (b"variant", b"blocks_per_cu", b"fuse", b"chunk_flags")}
    hip.FLAGSTATS_hip_shutdown()
    assert hip.FLAGSTATS_hip_device_id() == -1
    out = np.zeros(32, dtype=np.uint64)
    _lib.check(hip.FLAGSTATS_u16_x64(a.ctypes.data, a.size, out.ctypes.data), "x64 after shutdown")   # lazy re-init
    assert np.array_equal(out, want)
    assert hip.FLAGSTATS_hip_device_id() == 0
    assert {k: hip.FLAGSTATS_hip_get(k) for k in knobs} == knobs            # knobs survive a shutdown


def test_handles_opened_before_a_shutdown_stay_safe(hip):
    """FLAGSTATS_hip_shutdown with live handles.  Engines are reference-counted: an explicit context taken
    before the shutdown fails loudly afterwards (its buffers are gone) and can still be destroyed; a streaming session
    owns its streams and buffers and keeps working; nothing dereferences freed memory."""
    import oracle
    from libflagstats_amd import _lib
    a = oracle.generate(oracle.GEN_UNIFORM, 17, 0xFFFF, 0, 250_003)
    want = oracle.flagstat_hist(a)
    ctx = hip.FLAGSTATS_hip_ctx_create(0)
    assert ctx
    ses = hip.FLAGSTATS_hip_stream_open()
    assert ses
    out = np.zeros(32, dtype=np.uint64)
    _lib.check(hip.FLAGSTATS_hip_ctx_u16_x64(ctx, a.ctypes.data, a.size, out.ctypes.data), "ctx before shutdown")
    assert np.array_equal(out, want)
    hip.FLAGSTATS_hip_shutdown()
    out[:] = 0
    rc = hip.FLAGSTATS_hip_ctx_u16_x64(ctx, a.ctypes.data, a.size, out.ctypes.data)
    assert rc != 0 and b"FLAGSTATS_hip_shutdown" in hip.FLAGSTATS_hip_last_error() and not out.any()
    assert hip.FLAGSTATS_hip_ctx_device(ctx) == -1
    hip.FLAGSTATS_hip_ctx_destroy(ctx)
    for _ in range(2):
        out[:] = 0
        _lib.check(hip.FLAGSTATS_hip_stream_push(ses, a.ctypes.data, a.size), "session push after shutdown")
        _lib.check(hip.FLAGSTATS_hip_stream_finish(ses, out.ctypes.data), "session finish after shutdown")
        assert np.array_equal(out, want)
    hip.FLAGSTATS_hip_stream_close(ses)
    out[:] = 0
    _lib.check(hip.FLAGSTATS_u16_x64(a.ctypes.data, a.size, out.ctypes.data), "lazy re-init")
    assert np.array_equal(out, want)
    _lib.check(hip.FLAGSTATS_hip_init(0), "init")
    hip.FLAGSTATS_hip_set(b"on_error", 0)


def test_default_engine_callers_overlap(hip):
    """The reference API is reentrant (libflagstats.h:2980-2997): caller threads that meet on the default engine spread over
    its side engines instead of queueing behind one lock.  Four threads, results exact.  Per-block sized calls (2^18 flags =
    512 KiB, what a per-block caller of the reference makes) are latency-bound alone and must get through more than
    1.5 x together; 64 MiB calls already move ~55 GB/s from ONE thread, all the PCIe link has, so there the four together
    only must not be slower than one."""
    import threading
    import time

    import oracle

    def measure(n, reps):
        arrays = [np.random.RandomState(100 + t).randint(0, 65536, n).astype(np.uint16) for t in range(4)]
        wants = [oracle.flagstat_hist(a) for a in arrays]

        def call(a):
            out = np.zeros(32, dtype=np.uint64)
            assert hip.FLAGSTATS_u16_x64(a.ctypes.data, a.size, out.ctypes.data) == 0
            return out

        for a, w in zip(arrays, wants):           # warm: engines, staging, pinned buffers
            assert np.array_equal(call(a), w)
        t0 = time.perf_counter()
        for _ in range(reps):
            call(arrays[0])
        one = reps * n / (time.perf_counter() - t0)
        results = [None] * 4
        barrier = threading.Barrier(5)

        def worker(t):
            for _ in range(3):                    # untimed: every thread meets its side engine once
                call(arrays[t])
            barrier.wait()
            for _ in range(reps):
                results[t] = call(arrays[t])

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        for th in threads:
            th.start()
        barrier.wait()
        t0 = time.perf_counter()
        for th in threads:
            th.join()
        four = 4 * reps * n / (time.perf_counter() - t0)
        for t in range(4):
            assert np.array_equal(results[t], wants[t]), (n, t)
        print("%d flags per call: one caller thread %.2f Gflags/s, four together %.2f Gflags/s (%.2fx)" % (n, one / 1e9, four / 1e9, four / one))
        return one, four

    one, four = measure(1 << 18, 300)
    assert four > 1.5 * one, (one, four)
    one, four = measure(32 * 1024 * 1024, 6)
    assert four > 0.9 * one, (one, four)
