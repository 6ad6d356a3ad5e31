"""GPU: the two finalisation forms of the accumulate contract (libflagstats.h:118-142 `++f[...]`,
SURVEY F9) -- K1 adding its workgroup totals to out[32] with atomics (default, one launch) and
partials + K2 -- must be indistinguishable: same 32 slots, dead slots untouched, `+=` onto whatever
the caller's counters held, also when several launches on different streams target one out[32]."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LIVE = [2, 6, 7, 8, 10, 11, 12, 13, 14, 18, 22, 23, 24, 25, 26, 27, 28, 29, 30]


@pytest.fixture()
def knob(hip):
    old = hip.FLAGSTATS_hip_get(b"epilogue")
    yield lambda v: hip.FLAGSTATS_hip_set(b"epilogue", v)
    hip.FLAGSTATS_hip_set(b"epilogue", old)


@pytest.mark.parametrize("epilogue", [0, 1])
@pytest.mark.parametrize("n", [0, 1, 16383, 16384 * 3 + 5, 5_000_011, 16384 * 256 * 2])
def test_accumulates_onto_caller_counters(hip, knob, epilogue, n):
    import torch

    import oracle
    from libflagstats_amd import device
    assert knob(epilogue) == 0
    t = torch.empty(max(n, 1), dtype=torch.int16, device="cuda:0")[:n]
    device.generate_torch(t, device.GEN_UNIFORM, seed=77 + n % 13, mask=0xFFFF)
    start = torch.arange(100, 132, dtype=torch.int64, device="cuda:0")
    out = start.clone()
    device.count_torch(t, out)
    device.count_torch(t, out)                      # a second pass accumulates
    torch.cuda.synchronize()
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 77 + n % 13, 0xFFFF, 0, n).astype(np.int64)
    got = out.cpu().numpy() - start.cpu().numpy()
    assert np.array_equal(got, 2 * want), (epilogue, n)
    dead = [i for i in range(32) if i not in LIVE]
    assert not got[dead].any()                      # slots the scalar rule never writes stay untouched


@pytest.mark.parametrize("epilogue", [0, 1])
def test_superset_slots(hip, knob, epilogue):
    """Superset form: slots 0 / 16 = primary paired reads by QC class, slot 9 = pass-QC reads
    (len - fail-QC reads; with the atomic epilogue the per-workgroup terms wrap modulo 2^64 and
    only their sum is meaningful)."""
    import torch

    import oracle
    from libflagstats_amd import _lib, device
    assert knob(epilogue) == 0
    n = 7_654_321
    t = torch.empty(n, dtype=torch.int16, device="cuda:0")
    device.generate_torch(t, device.GEN_UNIFORM, seed=5, mask=0xFFFF)
    out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(hip.FLAGSTATS_hip_device_u16_superset(t.data_ptr(), n, out.data_ptr(), stream), "superset")
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 5, 0xFFFF, 0, n)
    a = t.cpu().numpy().view(np.uint16)
    pp = ((a & 0x100) == 0) & ((a & 0x800) == 0) & ((a & 1) == 1)
    fail = (a & 0x200) != 0
    want = want.copy()
    want[0] = int((pp & ~fail).sum())
    want[16] = int((pp & fail).sum())
    want[9] = n - int(fail.sum())
    assert np.array_equal(got, want)


def test_two_streams_one_counter_array(hip, knob):
    """Launches on different streams into the same out[32]: with the atomic epilogue every add is an
    atomic, so nothing is lost (the K2 form's plain += needs one stream per out[32])."""
    import torch

    import oracle
    from libflagstats_amd import device
    assert knob(1) == 0
    n = 3_000_017
    a = torch.empty(n, dtype=torch.int16, device="cuda:0")
    b = torch.empty(n, dtype=torch.int16, device="cuda:0")
    device.generate_torch(a, device.GEN_UNIFORM, seed=1, mask=0xFFFF)
    device.generate_torch(b, device.GEN_NA12878, seed=2, mask=1)
    out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    reps = 40
    for _ in range(reps):
        with torch.cuda.stream(s1):
            device.count_torch(a, out)
        with torch.cuda.stream(s2):
            device.count_torch(b, out)
    torch.cuda.synchronize()
    want = reps * (oracle.flagstat_generated(oracle.GEN_UNIFORM, 1, 0xFFFF, 0, n) +
                   oracle.flagstat_generated(oracle.GEN_NA12878, 2, 1, 0, n))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)


def test_pinned_host_counters_use_k2(hip, knob):
    """out[32] in pinned host memory (FLAGSTATS_hip_host_alloc): counters are written by K2 through
    the mapping, never by device atomics over the bus -- and are right."""
    import oracle
    from libflagstats_amd import _lib, device
    assert knob(1) == 0
    n = 2_000_003
    d = device.DeviceFlags(n).generate(device.GEN_UNIFORM, seed=31, mask=0x0FFF)
    hp = hip.FLAGSTATS_hip_host_alloc(256)
    assert hp
    try:
        host = np.ctypeslib.as_array(ctypes.cast(hp, ctypes.POINTER(ctypes.c_uint64)), shape=(32,))
        host[:] = 3
        _lib.check(hip.FLAGSTATS_hip_device_u16(d.ptr, n, hp, None), "FLAGSTATS_hip_device_u16(host counters)")
        _lib.check(hip.FLAGSTATS_hip_synchronize(), "sync")
        want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 31, 0x0FFF, 0, n)
        live = want != 0
        assert np.array_equal(host[live], want[live] + 3) and (host[~live] == 3).all()
    finally:
        hip.FLAGSTATS_hip_host_free(hp)
        d.free()


def test_flag_array_in_pinned_host_memory_is_read_in_place(hip, knob):
    """Zero-copy: a FLAG array that lives in pinned host memory (FLAGSTATS_hip_host_alloc) may be handed
    to the DEVICE-array entry; K1 then streams it over PCIe straight from host memory (no staging copy),
    with device counters and the default epilogue."""
    import torch

    import oracle
    from libflagstats_amd import _lib
    assert knob(1) == 0
    n = 3_000_001
    hp = hip.FLAGSTATS_hip_host_alloc(2 * n)
    assert hp
    try:
        host = np.ctypeslib.as_array(ctypes.cast(hp, ctypes.POINTER(ctypes.c_uint16)), shape=(n,))
        host[:] = oracle.generate(oracle.GEN_UNIFORM, 8, 0xFFFF, 0, n)
        out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(hip.FLAGSTATS_hip_device_u16(hp + 2, n - 1, out.data_ptr(), stream), "device_u16(pinned host array)")
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), oracle.flagstat_hist(host[1:]))
    finally:
        hip.FLAGSTATS_hip_host_free(hp)


@pytest.fixture()
def group_knob(hip):
    old = hip.FLAGSTATS_hip_get(b"group_min_grid")
    yield lambda v: hip.FLAGSTATS_hip_set(b"group_min_grid", v)
    hip.FLAGSTATS_hip_set(b"group_min_grid", old)


@pytest.mark.parametrize("min_grid", [0, 64, 2 ** 31])
@pytest.mark.parametrize("n", [1, 16384 * 3 + 5, 16384 * 9, 16384 * 67 + 1, 5_000_011, 16384 * 256 * 3 + 77])
def test_two_level_atomic_epilogue(hip, knob, group_knob, min_grid, n):
    """K1's adds through the workspace's per-XCD copies (grids >= group_min_grid; 0 = even a grid of one workgroup)
    or straight to out[32]: same counters, dead slots untouched, and the copies are left zero for the next launch
    (three launches in a row on one stream and workspace)."""
    import torch

    import oracle
    from libflagstats_amd import device
    assert knob(1) == 0 and group_knob(min_grid) == 0
    t = torch.empty(n, dtype=torch.int16, device="cuda:0")
    device.generate_torch(t, device.GEN_UNIFORM, seed=91 + n % 7, mask=0xFFFF)
    start = torch.arange(1000, 1032, dtype=torch.int64, device="cuda:0")
    out = start.clone()
    for _ in range(3):
        device.count_torch(t, out)
    torch.cuda.synchronize()
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 91 + n % 7, 0xFFFF, 0, n).astype(np.int64)
    got = out.cpu().numpy() - start.cpu().numpy()
    assert np.array_equal(got, 3 * want), (min_grid, n)
    assert not got[[i for i in range(32) if i not in LIVE]].any()


@pytest.fixture()
def steps_knob(hip):
    old = hip.FLAGSTATS_hip_get(b"group_max_steps")
    yield lambda v: hip.FLAGSTATS_hip_set(b"group_max_steps", v)
    hip.FLAGSTATS_hip_set(b"group_max_steps", old)


def test_two_level_epilogue_step_limit_default(hip):
    """The shipped rule (flagstat_kernels.hip fsk_launch): per-XCD copies only for grids >= 64 workgroups of <= 40 steps."""
    assert hip.FLAGSTATS_hip_get(b"group_min_grid") == 64
    assert hip.FLAGSTATS_hip_get(b"group_max_steps") == 40


@pytest.mark.parametrize("steps_per_wg, max_steps, two_level", [
    (3, 2, 0), (3, 3, 1), (3, 24, 1), (24, 24, 1), (25, 24, 0), (25, 2 ** 40, 1), (1, 0, 0)])
def test_two_level_epilogue_each_side_of_the_step_limit(hip, knob, group_knob, steps_knob, steps_per_wg, max_steps, two_level):
    """A launch on each side of group_max_steps: the form the rule picks is the one that ran (read back from the
    launcher), and both give the oracle's counters three launches in a row on one workspace."""
    import torch

    import oracle
    from libflagstats_amd import device
    grid = hip.FLAGSTATS_hip_get(b"grid")
    assert grid >= 64
    assert knob(1) == 0 and group_knob(64) == 0 and steps_knob(max_steps) == 0
    n = 16384 * grid * (steps_per_wg - 1) + 16384 * (grid - 1) + 4097   # ceil(steps / grid) == steps_per_wg, ragged last step
    t = torch.empty(n, dtype=torch.int16, device="cuda:0")
    device.generate_torch(t, device.GEN_UNIFORM, seed=5 + steps_per_wg, mask=0xFFFF)
    out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    for _ in range(3):
        device.count_torch(t, out)
        assert hip.FLAGSTATS_hip_get(b"last_k1_two_level") == two_level, (steps_per_wg, max_steps)
    torch.cuda.synchronize()
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 5 + steps_per_wg, 0xFFFF, 0, n)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), 3 * want)


def test_two_level_epilogue_two_streams_one_counter_array(hip, knob, group_knob):
    """Each stream has its own workspace (copies, tickets); both add to ONE out[32] through their group leaders."""
    import torch

    import oracle
    from libflagstats_amd import device
    assert knob(1) == 0 and group_knob(0) == 0
    n = 16384 * 256 * 2 + 333
    a = torch.empty(n, dtype=torch.int16, device="cuda:0")
    b = torch.empty(n, dtype=torch.int16, device="cuda:0")
    device.generate_torch(a, device.GEN_UNIFORM, seed=11, mask=0xFFFF)
    device.generate_torch(b, device.GEN_NA12878, seed=12, mask=1)
    out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    reps = 30
    for _ in range(reps):
        with torch.cuda.stream(s1):
            device.count_torch(a, out)
        with torch.cuda.stream(s2):
            device.count_torch(b, out)
    torch.cuda.synchronize()
    want = reps * (oracle.flagstat_generated(oracle.GEN_UNIFORM, 11, 0xFFFF, 0, n) +
                   oracle.flagstat_generated(oracle.GEN_NA12878, 12, 1, 0, n))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
