"""GPU: K1's dynamic schedule (variant bit 7: guided self-scheduling through device counters, grabbed by a
scheduler wave) -- every policy corner against the oracle, and the counters' self-reset across launches.
The schedule balanced the XCDs and was not faster (profiles/r03/dyn_sweep*.log), so it is carried by the measurement
build only (flagstat_kernels_tuning.hip) and this file is NOT part of the product's test suite (tests/conftest.py leaves
tests/tuning/ out unless asked):

    make -C libflagstats_amd/csrc tuning
    FLAGSTATS_TUNING_TESTS=1 FLAGSTATS_HIP_LIB=$PWD/libflagstats_amd/libflagstats_hip_tuning.so python -m pytest tests/tuning -m tuning

(that the shipped library refuses variant 153 is checked where the product is tested: tests/test_gpu_parity.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.tuning

STEP = 16384   # flags per 32 KiB step


@pytest.fixture()
def dyn(hip):
    from libflagstats_amd import _lib
    keys = (b"variant", b"blocks_per_cu", b"epilogue", b"dyn_first_pct", b"dyn_div", b"dyn_cmax", b"dyn_min_steps", b"dyn_lg_queues")
    old = {k: hip.FLAGSTATS_hip_get(k) for k in keys}
    assert hip.FLAGSTATS_hip_get(b"tuning_build"), "load the measurement build: FLAGSTATS_HIP_LIB=.../libflagstats_hip_tuning.so"
    _lib.check(hip.FLAGSTATS_hip_set(b"variant", 153), "variant 153")
    yield hip
    for k, v in old.items():
        hip.FLAGSTATS_hip_set(k, v if k != b"blocks_per_cu" or v != 1 else 0)


@pytest.mark.parametrize("first_pct,div,cmax,lgq", [(0, 1, 1, 0), (0, 4, 32, 3), (10, 2, 3, 4), (50, 4, 32, 3), (50, 64, 65535, 1),
                                                    (90, 1, 65535, 2), (99, 8, 2, 3), (100, 4, 32, 3), (75, 4, 32, 0)])
def test_policy_corners_match_oracle(dyn, first_pct, div, cmax, lgq):
    """Any split between the static round and the dynamic chunks, any chunk-size rule: same counters.
    Sizes straddle grid multiples, a ragged head (offset 3) and a ragged tail."""
    import oracle
    from libflagstats_amd import _lib, device
    hip = dyn
    for k, v in ((b"dyn_first_pct", first_pct), (b"dyn_div", div), (b"dyn_cmax", cmax), (b"dyn_min_steps", 1), (b"dyn_lg_queues", lgq)):
        _lib.check(hip.FLAGSTATS_hip_set(k, v), k.decode())
    grid = hip.FLAGSTATS_hip_compute_units()
    sizes = [STEP * (grid * 2), STEP * (grid * 2 + 1) + 5, STEP * (grid * 7 + 13) - 3, STEP * (grid * 23 + grid // 2) + 77]
    cap = max(sizes) + 16
    d = device.DeviceFlags(cap).generate(device.GEN_UNIFORM, seed=777 + first_pct, mask=0xFFFF)
    for n in sizes:
        for off in (0, 3):
            want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 777 + first_pct, 0xFFFF, off, n)
            assert np.array_equal(d.count(offset=off, n=n), want), (first_pct, div, cmax, lgq, n, off)
    d.free()


@pytest.mark.parametrize("epilogue", [1, 0])
def test_counter_resets_itself_between_launches(dyn, epilogue):
    """300 unsynchronised launches of very different sizes on one stream and one workspace: the schedule counter
    must be back at zero for each of them (a stale counter would make workgroups skip their chunks)."""
    import torch

    import oracle
    from libflagstats_amd import _lib, device
    hip = dyn
    _lib.check(hip.FLAGSTATS_hip_set(b"epilogue", epilogue), "epilogue")
    _lib.check(hip.FLAGSTATS_hip_set(b"dyn_min_steps", 1), "dyn_min_steps")
    grid = hip.FLAGSTATS_hip_compute_units()
    n = STEP * grid * 12
    t = torch.empty(n, dtype=torch.int16, device="cuda:0")
    device.generate_torch(t, device.GEN_NA12878, seed=31, mask=1)
    host = oracle.generate(oracle.GEN_NA12878, 31, 1, 0, n)
    rs = np.random.RandomState(5)
    out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    want = np.zeros(32, dtype=np.uint64)
    for i in range(300):
        m = int(rs.choice([1, 300, STEP, STEP * grid, STEP * grid * 2 + 9, STEP * grid * 5, STEP * grid * 11 + 12345]))
        a = int(rs.randint(0, n - m + 1))
        device.count_torch(t[a:a + m], out)
        want += oracle.flagstat_hist(host[a:a + m])
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)


def test_multi_epoch_dynamic(dyn):
    """> 255 steps per workgroup with a small static round: epoch flushes happen inside dynamic chunks."""
    import oracle
    from libflagstats_amd import _lib, device
    hip = dyn
    for k, v in ((b"dyn_first_pct", 20), (b"dyn_div", 2), (b"dyn_cmax", 100), (b"dyn_min_steps", 1)):
        _lib.check(hip.FLAGSTATS_hip_set(k, v), k.decode())
    grid = hip.FLAGSTATS_hip_compute_units()
    n = STEP * grid * 600 + 4321
    d = device.DeviceFlags(n).generate(device.GEN_UNIFORM, seed=606, mask=0xFFFF)
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 606, 0xFFFF, 0, n)
    assert np.array_equal(d.count(), want)
    d.free()


@pytest.mark.parametrize("bpc", [2, 3])
def test_dynamic_with_more_workgroups_per_cu(dyn, bpc):
    import oracle
    from libflagstats_amd import _lib, device
    hip = dyn
    _lib.check(hip.FLAGSTATS_hip_set(b"blocks_per_cu", bpc), "bpc")
    _lib.check(hip.FLAGSTATS_hip_set(b"dyn_min_steps", 1), "dyn_min_steps")
    n = STEP * 256 * 31 + 999
    d = device.DeviceFlags(n).generate(device.GEN_NA12878, seed=8, mask=1)
    want = oracle.flagstat_generated(oracle.GEN_NA12878, 8, 1, 0, n)
    assert np.array_equal(d.count(), want)
    d.free()
