"""GPU: size limits of the ABI.
* FLAGSTATS_u16 takes a uint32 length (libflagstats.h:3025): n_len = 2^32 - 1 is its maximum; the
  uint32 counters then sit right below their own limit (slot 25 = 2^31 - 1).
* the 64-bit device entry beyond 2^32 flags (the reference cannot express this)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_max_uint32_length_through_the_dropin_entry(hip):
    from libflagstats_amd import _lib
    kat = np.array(load_golden("kat.json")["exhaustive"]["65536"]["scalar"], dtype=np.uint64)
    n = 2 ** 32 - 1
    a = np.tile(np.arange(65536, dtype=np.uint16), 65536)[:n]       # 65536 sweeps minus the final 0xFFFF
    flags = np.zeros(32, dtype=np.uint32)
    rc = hip.FLAGSTATS_u16(a.ctypes.data, n, flags.ctypes.data)
    _lib.check(int(rc), "FLAGSTATS_u16")
    want = kat * np.uint64(65536)
    for slot in (18, 24, 25, 26):                                    # what the missing 0xFFFF would have added
        want[slot] -= np.uint64(1)
    assert int(want.max()) < 2 ** 32 and int(want[25]) == 2 ** 31 - 1
    assert np.array_equal(flags.astype(np.uint64), want)


def test_device_array_longer_than_2_pow_32(hip):
    from libflagstats_amd import device
    kat = np.array(load_golden("kat.json")["exhaustive"]["65536"]["scalar"], dtype=np.uint64)
    n = 2 ** 32 + 3 * 65536 + 5
    d = device.DeviceFlags(n).generate(device.GEN_RAMP, seed=0)
    import oracle
    want = kat * np.uint64(65536 + 3) + oracle.flagstat_c(np.arange(5, dtype=np.uint16))
    assert np.array_equal(d.count(), want)
    # a window that starts beyond 2^32 and is 2-byte aligned only
    off = 2 ** 32 + 1
    m = 3 * 65536
    got = d.count(offset=off, n=m)
    assert np.array_equal(got, kat * np.uint64(3))                   # any 65536-aligned-length window of a ramp
    d.free()
