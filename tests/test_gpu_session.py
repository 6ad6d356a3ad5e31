"""GPU: streaming sessions vs the oracle -- many blocks of ragged sizes, zero-copy acquire/commit and
push, chunk roll-over, reuse after finish, and the reference-shaped caller loop (liblz4 decode of a
block file straight into acquired pinned memory)."""
import ctypes
import os
import struct
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import blockfile_tool as bt  # noqa: E402

pytestmark = pytest.mark.gpu


def test_acquire_commit_push_and_reuse(hip):
    import oracle
    from libflagstats_amd import _lib
    from libflagstats_amd.session import StreamSession
    old = hip.FLAGSTATS_hip_get(b"chunk_flags")
    _lib.check(hip.FLAGSTATS_hip_set(b"chunk_flags", 3_000_000), "chunk")   # small chunks: many roll-overs
    try:
        rs = np.random.RandomState(3)
        with StreamSession() as s:
            for rnd in range(3):                      # the session is reusable after finish()
                want = np.zeros(32, dtype=np.uint64)
                total = 0
                for i in range(120):
                    n = int(rs.choice([0, 1, 7, 8, 9, 4097, 512000, 1_234_567]))
                    a = oracle.generate(oracle.GEN_UNIFORM, rnd * 1000 + i, 0xFFFF, 0, n)
                    if i % 2:
                        s.push(a)
                    else:
                        view = s.acquire(n + 5)       # commit fewer flags than acquired
                        view[:n] = a
                        s.commit(n)
                    want += oracle.flagstat_hist(a)
                    total += n
                assert s.pending_flags == total
                got = s.finish()
                assert np.array_equal(got, want), rnd
                assert s.pending_flags == 0
            big = oracle.generate(oracle.GEN_NA12878, 5, 1, 0, 10_000_001)   # push larger than a chunk: split inside
            s.push(big)
            assert np.array_equal(s.finish(), oracle.flagstat_hist(big))
            with pytest.raises(_lib.FlagstatsHipError):
                s.acquire(3_000_000 + 64)             # a single block may not exceed the chunk
    finally:
        hip.FLAGSTATS_hip_set(b"chunk_flags", old)


def test_reference_shaped_reader_loop(hip):
    """benchmark/flagstats.cpp:311-332 with the kernel call replaced by acquire/commit: liblz4 decodes
    every block straight into session memory; counters are read after the loop."""
    import oracle
    from libflagstats_amd.session import StreamSession
    flags = oracle.generate(oracle.GEN_NA12878, 11, 1, 0, 512000 * 9 + 4321)
    img = bt.block_file_image(flags)
    lz = bt.lz4()
    base = ctypes.cast(ctypes.c_char_p(img), ctypes.c_void_p).value
    with StreamSession() as s:
        pos = 0
        while pos < len(img):
            us, cs = struct.unpack_from("<ii", img, pos)
            pos += 8
            n = us >> 1
            p = s.acquire_ptr(n + 1)
            assert lz.LZ4_decompress_safe(base + pos, p, cs, us) == us
            s.commit(n)
            pos += cs
        got = s.finish()
    assert np.array_equal(got, oracle.flagstat_hist(flags))


def test_sessions_of_different_threads_overlap(hip):
    """Sessions own their streams, buffers and lock (no library-wide mutex on the data path): caller
    threads with a session each -- and a third thread making plain host-pointer calls meanwhile -- run
    concurrently and every one gets exactly its own counters.  ctypes drops the GIL inside the calls, so
    the threads really are inside the library at the same time (no timing assertion: exactness is the test)."""
    import threading
    import time

    import oracle
    from libflagstats_amd import pyflagstats
    from libflagstats_amd.session import StreamSession
    blocks = [oracle.generate(oracle.GEN_NA12878, 100 + k, 1, 0, 512000) for k in range(6)]
    want_block = [oracle.flagstat_hist(b) for b in blocks]
    rounds = 40
    results, spans, errors = {}, {}, []
    start = threading.Barrier(3)

    def session_worker(tid):
        try:
            with StreamSession() as s:
                start.wait()
                t0 = time.perf_counter()
                for r in range(rounds):
                    s.push(blocks[(tid + r) % len(blocks)])
                results[tid] = s.finish()
                spans[tid] = (t0, time.perf_counter())
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def host_call_worker():
        try:
            start.wait()
            t0 = time.perf_counter()
            acc = np.zeros(32, dtype=np.uint64)
            for r in range(rounds):
                acc += pyflagstats.counters_u32(blocks[r % len(blocks)]).astype(np.uint64)
            results["host"] = acc
            spans["host"] = (t0, time.perf_counter())
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ths = [threading.Thread(target=session_worker, args=(0,)), threading.Thread(target=session_worker, args=(1,)),
           threading.Thread(target=host_call_worker)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    for tid in (0, 1):
        want = sum((want_block[(tid + r) % len(blocks)] for r in range(rounds)), np.zeros(32, dtype=np.uint64))
        assert np.array_equal(results[tid], want), tid
    want = sum((want_block[r % len(blocks)] for r in range(rounds)), np.zeros(32, dtype=np.uint64))
    assert np.array_equal(results["host"], want)
