"""GPU: fork() at the boundary.  The reference's FLAGSTATS_u16 is a pure function that works in a
forked child (/root/reference/libflagstats.h:3024-3070); this library is an engine (HIP context, streams, helper threads)
that does not exist there.  A child forked after the library's first use must be refused by the C entries and by the
Python binding -- loudly, naming the fork and the remedy -- WITHOUT reaching the HIP runtime (a forked child that touches
the parent's HIP state is undefined and may hang the box), and the parent must go on counting correctly.  The remedy
itself -- the "spawn" start method -- is exercised too.  (Every entry family under fork, with a thread of the parent
inside the engine, runs in the host-stub build: tests/test_host_asan.py.)"""
import json
import multiprocessing
import os

import numpy as np
import pytest

import fork_worker

pytestmark = pytest.mark.gpu


def test_forked_child_is_refused_before_any_hip_call(hip):
    import oracle
    from libflagstats_amd import _lib, pyflagstats
    a = np.random.RandomState(5).randint(0, 65536, 300_001).astype(np.uint16)
    want = oracle.flagstat_hist(a)
    assert np.array_equal(pyflagstats.counters_u64(a), want)          # the parent owns the library now
    assert hip.FLAGSTATS_hip_forked() == 0
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                                      # ---- child: report through the pipe, never raise
        rep = {}
        try:
            raw = _lib._lib                                           # the handle itself (lib() would raise first)
            rep["forked"] = int(raw.FLAGSTATS_hip_forked())
            out = np.zeros(32, dtype=np.uint64)
            rep["rc_x64"] = int(raw.FLAGSTATS_u16_x64(a.ctypes.data, a.size, out.ctypes.data))
            rep["text"] = raw.FLAGSTATS_hip_last_error().decode(errors="replace")
            rep["untouched"] = not out.any()
            out32 = np.zeros(32, dtype=np.uint32)                     # (the `hip` fixture set on_error = return)
            rep["rc_u16"] = int(raw.FLAGSTATS_u16(a.ctypes.data, a.size, out32.ctypes.data))
            rep["available"] = int(raw.FLAGSTATS_hip_available())
            rep["device_alloc"] = raw.FLAGSTATS_hip_device_alloc(4096)
            raw.FLAGSTATS_hip_shutdown()                              # release-type entries: silent no-ops
            try:
                pyflagstats.flagstats(a)
                rep["py"] = "no exception"
            except Exception as e:  # noqa: BLE001
                rep["py"] = "%s: %s" % (type(e).__name__, e)
        except BaseException as e:  # noqa: BLE001
            rep["crash"] = repr(e)
        os.write(w, json.dumps(rep).encode())
        os._exit(0)
    os.close(w)
    data = b""
    while True:
        chunk = os.read(r, 65536)
        if not chunk:
            break
        data += chunk
    os.close(r)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    rep = json.loads(data)
    assert "crash" not in rep, rep
    assert rep["forked"] == 1 and rep["rc_x64"] != 0 and rep["rc_u16"] != 0 and rep["untouched"], rep
    assert "fork()ed" in rep["text"] and "spawn" in rep["text"], rep
    assert rep["available"] == 0 and not rep["device_alloc"], rep
    assert rep["py"].startswith("FlagstatsHipError") and "fork()ed" in rep["py"] and "spawn" in rep["py"], rep
    # the parent is unaffected
    assert hip.FLAGSTATS_hip_forked() == 0
    assert np.array_equal(pyflagstats.counters_u64(a), want)


def test_multiprocessing_fork_context_raises_and_spawn_context_works(hip):
    """What a Python user meets: a `multiprocessing` worker under the fork start method (the Linux default) gets
    FlagstatsHipError naming the fork; under the spawn start method it counts, bit-exact."""
    import oracle
    from libflagstats_amd import pyflagstats
    n, seed = 200_003, 17
    a = np.random.RandomState(seed).randint(0, 65536, n).astype(np.uint16)
    want = [int(v) for v in oracle.flagstat_hist(a)]
    assert [int(v) for v in pyflagstats.counters_u64(a)] == want      # the parent has used the library
    for method in ("fork", "spawn"):
        ctx = multiprocessing.get_context(method)
        here, there = ctx.Pipe(duplex=False)
        p = ctx.Process(target=fork_worker.flagstats_in_child, args=(there, seed, n))
        p.start()
        there.close()
        assert here.poll(300), "no answer from the %s child" % method
        got = here.recv()
        p.join(60)
        assert p.exitcode == 0, (method, p.exitcode)
        if method == "fork":
            assert got[0] == "error" and got[1] == "FlagstatsHipError" and "fork()ed" in got[2] and "spawn" in got[2], got
        else:
            assert got[0] == "ok" and got[1] == want, got
    assert [int(v) for v in pyflagstats.counters_u64(a)] == want
