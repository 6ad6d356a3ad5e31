"""GPU: BASELINE.json's configurations at their full sizes, each against the oracle on the same bytes.

  config 1/metric  8 GiB uniform-random uint16 resident in HBM (the array bench.py times), one K1+K2 launch
  config 2         8 GiB NA12878-like flags in PINNED host memory, double-buffered hipMemcpyAsync chunks
                   through the drop-in's 64-bit host entry FLAGSTATS_u16_x64

The oracle side runs on all host cores (oracle_flagstat_generated regenerates the same counter-based
stream; oracle_flagstat_mt_u16 counts the very bytes handed to the library).
"""
import ctypes
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U64 = np.uint64


def test_full_8gib_uniform_device_array_vs_oracle(hip):
    """The metric's workload, whole array: 2^32 uniform-random flags, seed as in bench.py."""
    import oracle
    from libflagstats_amd import device
    n = 2 ** 32
    d = device.DeviceFlags(n).generate(device.GEN_UNIFORM, seed=2026, mask=0xFFFF)
    got = d.count()
    d.free()
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 2026, 0xFFFF, 0, n)
    assert np.array_equal(got, want)
    assert int(got[25]) > 2 ** 31 - 2 ** 20   # about half the reads fail QC: beyond any uint32-safe margin


def _fill_parallel(lib, kind, seed, mask, ptr, n, threads):
    per = (n + threads - 1) // threads

    def work(k):
        b = per * k
        c = min(per, n - b)
        if c > 0:
            lib.oracle_generate_u16(kind, seed, mask, b, c, ctypes.cast(ptr + 2 * b, ctypes.POINTER(ctypes.c_uint16)))

    ths = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


def test_8gib_na12878_pinned_host_double_buffered_vs_oracle(hip):
    """BASELINE config 2: 8 GiB of NA12878-like flags (+eps variant, so the fail-QC class is exercised)
    in memory from FLAGSTATS_hip_host_alloc, streamed by FLAGSTATS_u16_x64 in 64 MiB chunks over two
    streams; counters vs the oracle on the same buffer; the copies really overlap (>= 2 chunks in flight)."""
    import oracle
    from libflagstats_amd import _lib
    n = 2 ** 32
    cores = os.cpu_count() or 1
    p = hip.FLAGSTATS_hip_host_alloc(2 * n)
    assert p, hip.FLAGSTATS_hip_last_error()
    try:
        clib = oracle.load_c()
        _fill_parallel(clib, oracle.GEN_NA12878, 77, 1, p, n, min(cores, 64))
        host = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint16)), shape=(n,))
        want = oracle.flagstat_mt(host)
        assert np.array_equal(want, oracle.flagstat_generated(oracle.GEN_NA12878, 77, 1, 0, n))  # the buffer is what we think
        out = np.zeros(32, dtype=U64)
        _lib.check(hip.FLAGSTATS_u16_x64(p, n, out.ctypes.data), "FLAGSTATS_u16_x64(pinned 8 GiB)")
        assert np.array_equal(out, want)
        chunks = int(hip.FLAGSTATS_hip_get(b"host_chunks"))
        overlapped = int(hip.FLAGSTATS_hip_get(b"host_overlapped"))
        assert chunks == (n + int(hip.FLAGSTATS_hip_get(b"chunk_flags")) - 1) // int(hip.FLAGSTATS_hip_get(b"chunk_flags"))
        # pinned memory: nearly every chunk is handed over while its predecessor is still in flight
        assert overlapped >= chunks // 2, (chunks, overlapped)
        # a ragged, odd-offset slice of the same pinned buffer
        out2 = np.zeros(32, dtype=U64)
        off, m = 12345, 2 ** 31 + 7
        _lib.check(hip.FLAGSTATS_u16_x64(p + 2 * off, m, out2.ctypes.data), "x64 slice")
        assert np.array_equal(out2, oracle.flagstat_generated(oracle.GEN_NA12878, 77, 1, off, m))
    finally:
        hip.FLAGSTATS_hip_host_free(p)


def test_64gib_as_eight_device_shards_counted_in_one_call(hip):
    """BASELINE config 3's data volume -- 64 GiB = 8 shards of 2^32 flags, seed + rank as bench.py --gpus 8
    makes them -- resident in ONE MI355X's 288 GB and counted through the C multi-shard entry
    (FLAGSTATS_hip_multi_device_u16: every shard where it lives, host-side sum of the 8 x 256 bytes).  The 8-GPU
    placement itself is the driver's SCALE run; this pins what a box with one GPU can: 2^35 flags in one
    query, per-slot totals beyond 2^34, bit-exact vs the oracle's sum over the shards."""
    import oracle
    from libflagstats_amd import _lib, device
    n = 2 ** 32
    shards = []
    try:
        for r in range(8):
            shards.append(device.DeviceFlags(n).generate(device.GEN_UNIFORM, seed=2026 + r, mask=0xFFFF))
        ptrs = (ctypes.c_void_p * 8)(*[s.ptr for s in shards])
        ns = (ctypes.c_uint64 * 8)(*([n] * 8))
        out = np.zeros(32, dtype=U64)
        _lib.check(hip.FLAGSTATS_hip_multi_device_u16(ptrs, ns, 8, out.ctypes.data), "multi device, 64 GiB")
    finally:
        for s in shards:
            s.free()
    want = np.zeros(32, dtype=U64)
    for r in range(8):
        want += oracle.flagstat_generated(oracle.GEN_UNIFORM, 2026 + r, 0xFFFF, 0, n)
    assert np.array_equal(out, want)
    assert int(out[25]) > 2 ** 34 - 2 ** 24      # ~half of 2^35 reads fail QC
