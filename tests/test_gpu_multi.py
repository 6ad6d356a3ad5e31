"""GPU: the C-level multi-GPU surface and per-device engines.

The test box has ONE MI355X, so "N devices" is exercised as N independent engines on device 0
(`devices = {0, 0}`): same code path -- private engines, one host thread per shard, host-side sum of
N x 256 bytes -- minus the second PCIe link.  RCCL is exercised at world size 1 (a real communicator,
a real ncclAllReduce of uint64[32] on the caller's stream)."""
import ctypes
import os
import subprocess
import threading

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

U64 = np.uint64


def test_c_consumer_two_contexts_and_multi_entry(hip, tmp_path):
    """A C process (no HIP, no torch): FLAGSTATS_hip_multi_u16_x64 over two engines, and two explicit
    contexts driven concurrently from two threads; all bit-exact vs the oracle."""
    import oracle
    exe = str(tmp_path / "consumer_multi")
    libdir = os.path.join(ROOT, "libflagstats_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-pthread", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "consumer_multi.c"), "-L", libdir, "-lflagstats_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True)
    for n in (0, 1, 12_345_679, 80_000_001):   # the last one: > 1 chunk per shard
        r = subprocess.run([exe, str(n), "2", "0", "0"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        a = ((np.arange(n, dtype=np.uint64).astype(np.uint32) * np.uint32(2654435761)) >> np.uint32(13)).astype(np.uint16)
        want = oracle.flagstat_hist(a) if n else np.zeros(32, dtype=U64)
        lines = dict((ln.split()[0], np.array([int(v) for v in ln.split()[1:]], dtype=U64)) for ln in r.stdout.splitlines())
        for tag in ("one", "multi", "two"):
            assert np.array_equal(lines[tag], want), (n, tag)


def test_multi_entries_from_python(hip):
    import oracle
    from libflagstats_amd import _lib, device
    a = oracle.generate(oracle.GEN_UNIFORM, 5, 0xFFFF, 0, 9_000_017)
    want = oracle.flagstat_hist(a)
    for ndev in (1, 2, 3):
        devs = (ctypes.c_int * ndev)(*([0] * ndev))
        out = np.zeros(32, dtype=U64)
        _lib.check(hip.FLAGSTATS_hip_multi_u16_x64(a.ctypes.data, a.size, devs, ndev, out.ctypes.data), "multi host")
        assert np.array_equal(out, want), ndev
    # ragged odd-aligned array, fewer flags than shards
    for m in (0, 1, 2, 5):
        out = np.zeros(32, dtype=U64)
        devs = (ctypes.c_int * 3)(0, 0, 0)
        _lib.check(hip.FLAGSTATS_hip_multi_u16_x64(a[1:].ctypes.data, m, devs, 3, out.ctypes.data), "multi tiny")
        assert np.array_equal(out, oracle.flagstat_hist(a[1:1 + m]) if m else np.zeros(32, dtype=U64))
    # device-resident shards: three slices (one empty, one at an odd offset)
    d = device.DeviceFlags(a.size).upload(a)
    cuts = [(0, 3_000_001), (3_000_001, 0), (3_000_001, a.size - 3_000_001)]
    ptrs = (ctypes.c_void_p * 3)(*[d.ptr + 2 * b for b, _ in cuts])
    ns = (ctypes.c_uint64 * 3)(*[c for _, c in cuts])
    out = np.zeros(32, dtype=U64)
    _lib.check(hip.FLAGSTATS_hip_multi_device_u16(ptrs, ns, 3, out.ctypes.data), "multi device")
    assert np.array_equal(out, want)
    # a host pointer is refused loudly by the device-shard form
    bad = (ctypes.c_void_p * 1)(a.ctypes.data)
    one = (ctypes.c_uint64 * 1)(10)
    assert hip.FLAGSTATS_hip_multi_device_u16(bad, one, 1, out.ctypes.data) != 0
    assert b"device" in hip.FLAGSTATS_hip_last_error()
    d.free()


def test_shard_range_matches_python_twin(hip):
    from libflagstats_amd.dist import shard_range
    for n in (0, 1, 7, 1000, 2 ** 35 + 3):
        for world in (1, 2, 3, 8):
            for r in range(world):
                b, e = ctypes.c_uint64(), ctypes.c_uint64()
                hip.FLAGSTATS_hip_shard_range(n, r, world, ctypes.byref(b), ctypes.byref(e))
                assert (b.value, e.value) == shard_range(n, r, world)


def test_contexts_run_concurrently(hip):
    """Two caller threads, each with its own context, overlap; results exact.  (With the round-1
    process-wide mutex the second thread could not start before the first had finished.)"""
    import oracle
    from libflagstats_amd import _lib
    arrays = [oracle.generate(oracle.GEN_NA12878, 40 + i, 1, 0, 30_000_011 + i) for i in range(2)]
    want = [oracle.flagstat_hist(a) for a in arrays]
    ctxs = [hip.FLAGSTATS_hip_ctx_create(0) for _ in range(2)]
    assert all(ctxs) and hip.FLAGSTATS_hip_ctx_device(ctxs[0]) == 0
    errs = []

    def work(i):
        try:
            for _ in range(4):
                out = np.zeros(32, dtype=U64)
                _lib.check(hip.FLAGSTATS_hip_ctx_u16_x64(ctxs[i], arrays[i].ctypes.data, arrays[i].size, out.ctypes.data), "ctx")
                assert np.array_equal(out, want[i])
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for c in ctxs:
        hip.FLAGSTATS_hip_ctx_destroy(c)
    assert not errs, errs


def test_device_entry_checks_pointers_and_keeps_current_device(hip):
    import torch

    import oracle
    from libflagstats_amd import device
    t = torch.empty(1_000_003, dtype=torch.int16, device="cuda:0")
    device.generate_torch(t, device.GEN_UNIFORM, seed=3, mask=0xFFFF)
    out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    before = torch.cuda.current_device()
    device.count_torch(t, out)
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before
    assert np.array_equal(out.cpu().numpy().view(U64), oracle.flagstat_generated(oracle.GEN_UNIFORM, 3, 0xFFFF, 0, t.numel()))
    # host memory where a device pointer is required: refused, loudly, before any launch
    h = np.zeros(64, dtype=np.uint16)
    rc = hip.FLAGSTATS_hip_device_u16(h.ctypes.data, h.size, out.data_ptr(), None)
    assert rc != 0 and b"d_array" in hip.FLAGSTATS_hip_last_error()
    hout = np.zeros(32, dtype=U64)
    rc = hip.FLAGSTATS_hip_device_u16(t.data_ptr(), 10, hout.ctypes.data, None)
    assert rc != 0 and b"d_out" in hip.FLAGSTATS_hip_last_error()
    # a side stream of the same device is accepted
    s = torch.cuda.Stream(device="cuda:0")
    with torch.cuda.stream(s):
        out2 = torch.zeros(32, dtype=torch.int64, device="cuda:0")
        device.count_torch(t, out2)
    s.synchronize()
    assert torch.equal(out, out2)


def test_rccl_allreduce_world_of_one(hip):
    """The RCCL form of the multi-GPU step through the C ABI: unique id -> communicator -> K1 + K2(store)
    + ncclAllReduce(uint64[32], sum) on one stream.  World size 1 on this box; the call sequence is the
    one every rank of bench.py --gpus N runs."""
    import torch

    import oracle
    from libflagstats_amd import _lib, device
    ident = (ctypes.c_char * 128)()
    _lib.check(hip.FLAGSTATS_hip_comm_unique_id(ident), "unique id")
    comm = hip.FLAGSTATS_hip_comm_init_rank(ident, 1, 0, 0)
    assert comm, hip.FLAGSTATS_hip_last_error()
    try:
        n = 20_000_003
        t = torch.empty(n, dtype=torch.int16, device="cuda:0")
        device.generate_torch(t, device.GEN_UNIFORM, seed=8, mask=0xFFFF)
        out = torch.full((32,), 999, dtype=torch.int64, device="cuda:0")
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):   # store form: repeated queries do not accumulate
            _lib.check(hip.FLAGSTATS_hip_device_u16_allreduce(t.data_ptr(), n, out.data_ptr(), comm, stream), "allreduce")
        torch.cuda.synchronize()
        want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 8, 0xFFFF, 0, n)
        assert np.array_equal(out.cpu().numpy().view(U64), want)
        # the overlapped form: the collective on a second stream, ordered on the device behind the kernels; a ring
        # of counter buffers, the launch stream waiting for the collective stream once per ring (as bench.py does)
        side = torch.cuda.Stream()
        side_p = ctypes.c_void_p(side.cuda_stream)
        ring = [torch.full((32,), 7, dtype=torch.int64, device="cuda:0") for _ in range(3)]
        lens = [n, n - 12345, 1_000_001, 5, n // 3, 16384 * 7, n - 1]
        for i, m in enumerate(lens):
            if i and i % len(ring) == 0:
                _lib.check(hip.FLAGSTATS_hip_stream_wait_stream(stream, side_p, 0), "stream_wait_stream")
            _lib.check(hip.FLAGSTATS_hip_device_u16_allreduce_overlapped(t.data_ptr(), m, ring[i % len(ring)].data_ptr(), comm,
                                                                         stream, side_p), "allreduce overlapped")
        torch.cuda.synchronize()
        for j in range(len(ring)):
            last = max(i for i in range(len(lens)) if i % len(ring) == j)
            want = oracle.flagstat_generated(oracle.GEN_UNIFORM, 8, 0xFFFF, 0, lens[last])
            assert np.array_equal(ring[j].cpu().numpy().view(U64), want), j
        # a stream of another kind of object is refused loudly
        assert hip.FLAGSTATS_hip_stream_wait_stream(stream, ctypes.c_void_p(0x1234), 0) != 0
    finally:
        _lib.check(hip.FLAGSTATS_hip_comm_destroy(comm), "comm destroy")


# --------------------------------------------------------------------------- two DISTINCT devices (skipped on 1-GPU boxes)
def _need_two(hip):
    if hip.FLAGSTATS_hip_device_count() < 2:
        pytest.skip("needs 2 visible GPUs (the round-end box has one; the driver's 8-GPU node runs this)")


def test_multi_host_array_over_two_devices(hip):
    """FLAGSTATS_hip_multi_u16_x64 with devices = {0, 1}: two engines on two GPUs, two PCIe links, host-side sum."""
    _need_two(hip)
    import oracle
    from libflagstats_amd import _lib
    n = 50_000_003
    a = oracle.generate(oracle.GEN_UNIFORM, 12, 0xFFFF, 0, n)
    devs = (ctypes.c_int * 2)(0, 1)
    out = np.zeros(32, dtype=U64)
    _lib.check(hip.FLAGSTATS_hip_multi_u16_x64(a.ctypes.data, n, devs, 2, out.ctypes.data), "multi_u16_x64 {0,1}")
    assert np.array_equal(out, oracle.flagstat_hist(a))


def test_multi_device_resident_shards_on_two_devices(hip):
    """FLAGSTATS_hip_multi_device_u16: shards allocated on GPU 0 and GPU 1, each counted where it lives, concurrently."""
    _need_two(hip)
    import oracle
    from libflagstats_amd import _lib
    sizes = [30_000_001, 20_000_005, 7, 16384 * 300]
    ptrs, want = [], np.zeros(32, dtype=U64)
    try:
        for i, m in enumerate(sizes):
            p = hip.FLAGSTATS_hip_device_alloc_on(i % 2, 2 * m)
            assert p, hip.FLAGSTATS_hip_last_error()
            ptrs.append(p)
            _lib.check(hip.FLAGSTATS_hip_generate_u16(p, m, 0, 40 + i, 0xFFFF, 0, None), "generate")
            want += oracle.flagstat_generated(oracle.GEN_UNIFORM, 40 + i, 0xFFFF, 0, m)
        for dev in (0, 1):
            _lib.check(hip.FLAGSTATS_hip_init(dev), "init")
            _lib.check(hip.FLAGSTATS_hip_synchronize(), "sync")
        _lib.check(hip.FLAGSTATS_hip_init(0), "init 0")
        arr = (ctypes.c_void_p * len(sizes))(*ptrs)
        ns = (ctypes.c_uint64 * len(sizes))(*sizes)
        out = np.zeros(32, dtype=U64)
        _lib.check(hip.FLAGSTATS_hip_multi_device_u16(arr, ns, len(sizes), out.ctypes.data), "multi_device_u16")
        assert np.array_equal(out, want)
    finally:
        for p in ptrs:
            hip.FLAGSTATS_hip_device_free(p)


def test_two_rccl_ranks_in_two_processes(hip):
    """bench.py --gpus 2, self-spawned: one rank per GPU, the library's own RCCL communicator (ncclCommCount == 2), in line
    and overlapped (the ring check runs on both ranks), counters = oracle sum over both shards."""
    _need_two(hip)
    import json
    import subprocess
    import sys

    from conftest import ROOT
    for extra in ((), ("--overlap",)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--flags-per-gpu", str(2 ** 28), "--steps", "10",
                            "--warmup", "3", "--cpu-seconds", "0", "--probe-reps", "3", *extra],
                           capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout + r.stderr
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert d["n_gpus"] == 2 and d["config"]["rccl_nranks"] == 2 and d["parity"].startswith("bit-exact"), d
