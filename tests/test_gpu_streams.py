"""GPU: stream behaviour of the device entry: the first call on a caller's stream sets up that
stream's workspace with stream-ordered work only -- it neither waits for nor stalls other streams -- and may happen
while the stream is being captured into a HIP graph."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _while_stream_a_is_busy(fn_on_b, big, out_a, a, reps=150):
    """Queue ~45 ms of K1 launches on stream `a`, then run fn_on_b() (which queues work on another stream and returns an
    event recorded behind it); returns (ms until that event completed, ms until `a` drained, a was still busy then)."""
    import torch

    from libflagstats_amd import device
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        for _ in range(reps):
            device.count_torch(big, out_a)
        done_a = torch.cuda.Event()
        done_a.record(a)
    ev = fn_on_b()
    ev.synchronize()
    t_b = time.perf_counter() - t0
    busy = not done_a.query()
    a.synchronize()
    return t_b * 1e3, (time.perf_counter() - t0) * 1e3, busy


def test_first_call_on_a_fresh_stream_does_not_wait_for_other_streams(hip):
    """A long job (150 launches over 2 GiB, ~45 ms) runs on stream A.  A FIRST call on a brand-new stream sets its
    workspace up with an allocation plus stream-ordered zeroing on ITS stream -- no hipDeviceSynchronize, no NULL-stream
    work (r02 had both) that would wait for A to drain: it can finish while A is still busy.  (A warm call and a plain
    torch fill on other streams are timed in the same situation, for the record.)"""
    import torch

    import oracle
    from libflagstats_amd import device
    big = torch.empty(2 ** 30, dtype=torch.int16, device="cuda:0")          # 2 GiB
    device.generate_torch(big, device.GEN_UNIFORM, seed=3, mask=0xFFFF)
    small = torch.empty(3_000_001, dtype=torch.int16, device="cuda:0")
    device.generate_torch(small, device.GEN_UNIFORM, seed=4, mask=0xFFFF)
    out_a = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    out_b = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    a, warm = torch.cuda.Stream(), torch.cuda.Stream()
    for s in (a, warm):
        with torch.cuda.stream(s):
            device.count_torch(small, torch.zeros(32, dtype=torch.int64, device="cuda:0"))   # both streams known to the library
    torch.cuda.synchronize()

    def count_on(stream):
        def go():
            with torch.cuda.stream(stream):
                device.count_torch(small, out_b)
                ev = torch.cuda.Event()
                ev.record(stream)
            return ev
        return go

    def torch_fill():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(1024, device="cuda:0").add_(1)
            ev = torch.cuda.Event()
            ev.record(s)
        return ev

    # Which hardware queue a stream lands on is the runtime's business: a stream that shares A's queue waits behind A
    # whatever the library does.  So the first call is made on several fresh streams (each one IS a first call: a new
    # workspace) and at least one of them must finish while A is still busy -- with a device-wide wait inside the first
    # call (r02) none could.
    t_fill, t_a0, busy_fill = _while_stream_a_is_busy(torch_fill, big, out_a, a)
    t_warm, t_a1, busy_warm = _while_stream_a_is_busy(count_on(warm), big, out_a, a)
    firsts = []
    for _ in range(4):
        firsts.append(_while_stream_a_is_busy(count_on(torch.cuda.Stream()), big, out_a, a))
    out_b_calls = 1 + len(firsts)
    print("stream A drains in %.1f ms; on another stream meanwhile: torch fill done at %.1f ms (A busy: %s), warm count at %.1f ms "
          "(A busy: %s), FIRST counts on fresh streams at %s ms (A busy: %s)"
          % (firsts[0][1], t_fill, busy_fill, t_warm, busy_warm, ["%.1f" % f[0] for f in firsts], [f[2] for f in firsts]))
    assert any(f[2] and f[0] < 0.5 * f[1] for f in firsts), firsts
    torch.cuda.synchronize()
    want_b = oracle.flagstat_generated(oracle.GEN_UNIFORM, 4, 0xFFFF, 0, small.numel())
    assert np.array_equal(out_b.cpu().numpy().view(np.uint64), want_b * np.uint64(out_b_calls))
    want_a = oracle.flagstat_generated(oracle.GEN_UNIFORM, 3, 0xFFFF, 0, big.numel())
    assert np.array_equal(out_a.cpu().numpy().view(np.uint64), want_a * np.uint64(6 * 150))


@pytest.mark.parametrize("store", [False, True])
def test_first_use_under_stream_capture(hip, store):
    """K1 (and, in the store form, K2) captured into a HIP graph on a stream the library has never seen: the workspace
    set-up is capture-legal (allocation and zeroing in relaxed mode, beside the capture); replays give the right counters."""
    import torch

    import oracle
    from libflagstats_amd import device
    n = 16384 * 256 * 2 + 4321
    t = torch.empty(n, dtype=torch.int16, device="cuda:0")
    device.generate_torch(t, device.GEN_NA12878, seed=77, mask=1)
    out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="global"):
        device.count_torch(t, out, store=store)
    torch.cuda.synchronize()
    # a PLAIN launch on that stream before the graph has ever run: the stream's workspace (made during the capture) must
    # already be zero -- it is zeroed beside the capture, not by a node of the graph
    plain = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    with torch.cuda.stream(s):
        device.count_torch(t, plain)
    torch.cuda.synchronize()
    assert np.array_equal(plain.cpu().numpy().view(np.uint64), oracle.flagstat_generated(oracle.GEN_NA12878, 77, 1, 0, n))
    out.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    want = oracle.flagstat_generated(oracle.GEN_NA12878, 77, 1, 0, n)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want * np.uint64(1 if store else 3))
