"""CPU: the product's HOST code under ThreadSanitizer.  `make -C libflagstats_amd/csrc hoststub` compiles
flagstat_engine / _capi / _multi / _blocks / _session / _text as plain C++ against tests/hoststub -- a test-only HIP stand-in
whose streams are FIFO worker threads and whose "kernels" are the oracle's scalar rule -- and tests/hoststub/tsan_driver.cpp
drives the block pipeline with 1 / 5 / 20 decoder threads (own literal-run LZ4 image + the reference-written golden
files), two concurrent sessions, four concurrent reference-API callers (polled small calls, a multi-chunk call), the
multi-device entry over {0, 1, 0}, two private contexts on two threads, and shutdown + re-creation of everything.
Every result is checked against the oracle; any data race or lock-order report fails the test."""
import os
import subprocess

from conftest import GOLDEN, ROOT


def test_host_code_is_race_free_under_tsan():
    csrc = os.path.join(ROOT, "libflagstats_amd", "csrc")
    b = subprocess.run(["make", "-C", csrc, "hoststub"], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-3000:] + b.stderr[-3000:]
    exe = os.path.join(ROOT, "tests", "hoststub", "build", "tsan_driver")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 exitcode=66")
    for k in list(env):
        if k.startswith("FLAGSTATS_HIP_"):
            del env[k]
    r = subprocess.run([exe, os.path.join(GOLDEN, "blockfiles")], capture_output=True, text=True, timeout=900, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:6000]
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
