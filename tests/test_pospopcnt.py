"""Row f4: plain positional popcount.  CPU: oracle vs the reference's golden vectors.
GPU: STORM_pospopcnt_u16 / device entry of libflagstats_hip.so vs golden + oracle."""
import numpy as np
import pytest

from conftest import load_golden


def _input(case):
    return np.random.RandomState(case["seed"]).randint(0, case["hi"], case["n"]).astype(np.uint16)


def test_oracle_matches_reference_golden(oracle_mod):
    for case in load_golden("pospopcnt.json")["cases"]:
        a = _input(case)
        want = np.array(case["counts"], dtype=np.uint64)
        assert np.array_equal(oracle_mod.pospopcnt(a), want), case["n"]
        assert np.array_equal(oracle_mod.pospopcnt_numpy(a), want), case["n"]
        ref = oracle_mod.ref_pospopcnt(a)
        if ref is not None:
            assert np.array_equal(ref.astype(np.uint64), want)


@pytest.mark.gpu
def test_gpu_host_entry_matches_golden(hip):
    from libflagstats_amd import device
    for case in load_golden("pospopcnt.json")["cases"]:
        a = _input(case)
        got = device.pospopcnt_host(a)
        assert np.array_equal(got.astype(np.uint64), np.array(case["counts"], dtype=np.uint64)), case["n"]
        # the reference-shaped symbol itself: zeroes out[] first, like the reference (python/libalgebra.h:3497)
        raw = np.full(16, 0xDEADBEEF, dtype=np.uint32)
        assert hip.STORM_pospopcnt_u16(a.ctypes.data if a.size else None, a.size, raw.ctypes.data) == 0
        assert np.array_equal(raw, got), case["n"]
        if a.size > 1:
            assert np.array_equal(device.pospopcnt_host(a[1:]).astype(np.uint64),
                                  np.array(case["counts"], dtype=np.uint64) - np.array(
                                      [(int(a[0]) >> j) & 1 for j in range(16)], dtype=np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("epilogue", [1, 0])
def test_gpu_device_entry_large_and_ragged(hip, epilogue):
    """Both finalisation forms: the count kernel adding its workgroup totals to out[16] with atomics
    (default) and partials + a finalize launch."""
    import torch

    import oracle
    from libflagstats_amd import device
    old = hip.FLAGSTATS_hip_get(b"epilogue")
    assert hip.FLAGSTATS_hip_set(b"epilogue", epilogue) == 0
    try:
        _device_entry_large_and_ragged(torch, oracle, device)
    finally:
        hip.FLAGSTATS_hip_set(b"epilogue", old)


def _device_entry_large_and_ragged(torch, oracle, device):
    n = 300_000_007                                          # > one epoch per workgroup, ragged tail
    t = torch.empty(n + 8, dtype=torch.int16, device="cuda:0")
    device.generate_torch(t, device.GEN_UNIFORM, seed=77, mask=0xFFFF)
    host = None
    for off, cnt in ((0, n), (3, n - 5), (5, 16384 * 3 + 1)):
        out = device.pospopcnt_torch(t[off:off + cnt])
        device.pospopcnt_torch(t[off:off + cnt], out)        # += : twice the counts
        torch.cuda.synchronize()
        if host is None:
            host = oracle.generate(oracle.GEN_UNIFORM, 77, 0xFFFF, 0, n + 8)
        want = oracle.pospopcnt(host[off:off + cnt])
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want * np.uint64(2)), (off, cnt)
    # exhaustive sweep: every bit is set in exactly half of 0..65535
    r = torch.empty(65536 * 4, dtype=torch.int16, device="cuda:0")
    device.generate_torch(r, device.GEN_RAMP, seed=0)
    out = device.pospopcnt_torch(r)
    torch.cuda.synchronize()
    assert out.cpu().tolist() == [32768 * 4] * 16
