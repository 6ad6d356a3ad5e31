"""CPU: the product's whole HOST side under AddressSanitizer + UndefinedBehaviorSanitizer (the TSan twin is
test_host_tsan.py).  `make -C libflagstats_amd/csrc hoststub SAN=address,undefined` compiles flagstat_engine / _capi / _multi /
_blocks / _session / _text / _gpu_decode as plain C++ against tests/hoststub (a test-only HIP stand-in: "device" memory is the heap,
exact-size, so ASan sees every access behind a buffer; the stand-in kernels are the oracle's scalar rule, the product's host
LZ4 decoder and the image's libzstd) and builds
  * build/asan_driver = tests/hoststub/tsan_driver.cpp, the threaded scenario of the TSan test, and
  * build/fuzz_driver = tests/hoststub/fuzz_driver.cpp: block files (LZ4 and Zstandard, made here and reference-written) whose
    headers are negative, huge, overlapping or truncated and whose payloads are damaged, raw files of every length, FLAG text
    of random bytes -- through every block-file entry, as image and as file, decoder on the host threads and "on the GPU",
    segments forced small, odd piece counts.  Every input is also a differential test: all paths agree on accept / reject and
    on the counters, an undamaged input gives the oracle's counters, a rejected one leaves the caller's counters untouched
    and sets a message (the reference's behaviour to keep: loud failure, benchmark/flagstats.cpp:105-108,256-259).
100,000 inputs (four processes of 25,000, different seeds): no sanitizer report, no failed check."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

from conftest import GOLDEN, ROOT

BUILD = os.path.join(ROOT, "tests", "hoststub", "build")


def _env():
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=67", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=68")
    for k in list(env):
        if k.startswith("FLAGSTATS_HIP_"):
            del env[k]
    return env


def _clean(r, what):
    noise = "\n".join(ln for ln in r.stderr.splitlines() if not ln.startswith("libflagstats_hip:") and not ln.startswith("fuzz_driver:"))
    assert "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr and "runtime error" not in r.stderr, (what, noise[:6000])
    assert r.returncode == 0 and "all checks passed" in r.stdout, (what, r.returncode, r.stdout[-2000:], noise[-3000:])


_built = False


def _build():
    """One ASan + UBSan build of the three drivers per test process."""
    global _built
    if _built:
        return
    csrc = os.path.join(ROOT, "libflagstats_amd", "csrc")
    b = subprocess.run(["make", "-C", csrc, "hoststub", "SAN=address,undefined"], capture_output=True, text=True, timeout=1200)
    assert b.returncode == 0, b.stdout[-3000:] + b.stderr[-3000:]
    _built = True


def test_fork_is_refused_at_every_entry_and_leaves_the_parent_intact():
    """VERDICT r05 item 2.  The reference's FLAGSTATS_u16 is a pure function and works in a forked child
    (libflagstats.h:3024-3070); the replacement is an engine (HIP context, streams, helper threads, mutexes, a polling
    protocol) that does not exist there.  tests/hoststub/fork_driver.cpp: a child forked BEFORE the first call owns the
    library and counts correctly; the parent then counts, opens a session and a context, keeps a thread inside the engine
    (its lock is held nearly all the time) and forks eight times -- each child calls every entry family and must get an
    error naming the fork from each, within its alarm (a child that waits for an inherited mutex dies by SIGALRM), with its
    counters untouched, the release-type entries as silent no-ops and the stateless helpers still working; with the default
    "on_error" the reference-shaped entry aborts in the child; afterwards the parent's worker thread, session, context and
    counters are all still right.  Under ASan + UBSan."""
    _build()
    r = subprocess.run([os.path.join(BUILD, "fork_driver")], capture_output=True, text=True, timeout=600, env=_env())
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0 and "all checks passed" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert "fork()ed" in r.stderr and "spawn" in r.stderr   # the refusals are loud and name the remedy


def test_host_code_under_address_and_undefined_behaviour_sanitizers():
    _build()
    blockfiles = os.path.join(GOLDEN, "blockfiles")
    r = subprocess.run([os.path.join(BUILD, "asan_driver"), blockfiles], capture_output=True, text=True, timeout=900, env=_env())
    _clean(r, "threaded scenario")

    def fuzz(seed):
        return subprocess.run([os.path.join(BUILD, "fuzz_driver"), "25000", blockfiles, str(seed)], capture_output=True, text=True, timeout=1500, env=_env())

    with ThreadPoolExecutor(max_workers=4) as ex:
        runs = list(ex.map(fuzz, [0x9E3779B97F4A7C15, 20261004, 5, 77777]))
    inputs = 0
    for i, r in enumerate(runs):
        _clean(r, "fuzz process %d" % i)
        inputs += int(r.stdout.split("fuzz_driver: ")[1].split(" inputs")[0])
    assert inputs >= 100000
