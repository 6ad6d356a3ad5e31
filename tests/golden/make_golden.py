#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every expected value below is produced by the reference's own code compiled
from where it lies -- never by our oracle or our kernels:

* counters      : FLAGSTAT_scalar           (libflagstats.h:170-176) through
                  oracle/_ref/libflagstats_ref.so (oracle/Makefile, ref_wrap.cpp)
* SIMD twins    : FLAGSTAT_avx512 / _avx512_improved3 on the same inputs, kept
                  so tests can document the superset-slot deviation
* inmemory case : the input of benchmark/inmemory.cpp:108-116 (mt19937 seed 0,
                  uniform_int_distribution<uint16_t>(0,4095), n = 102400),
                  regenerated with libstdc++ by a 10-line generator compiled here
* pyflagstats   : python/libflagstats.pyx built with Cython against the ROOT
                  libflagstats.h (python/'s own copy is a stale, broken snapshot,
                  SURVEY F5) and imported in this container; dicts + error texts

The fixtures are data (inputs or input recipes + expected outputs); no reference
source text is stored.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import oracle  # noqa: E402  (used only for its ctypes binding of oracle/_ref)


def ref_scalar(a):
    out = oracle.ref_call("FLAGSTAT_scalar", a)
    assert out is not None, "oracle/_ref missing: run `make -C oracle ref`"
    return [int(v) for v in out]


def ref_variant(name, a):
    out = oracle.ref_call(name, a)
    return None if out is None else [int(v) for v in out]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", name)


# lengths straddling every block boundary of every reference kernel
LENGTHS = [0, 1, 2, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257,
           511, 512, 513, 767, 768, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097,
           65535, 65536, 65537, 131071, 131072, 131073, 512000, 1048575, 1048576, 1048577]


def make_kat():
    single_vals = [0, 1, 3, 4, 11, 13, 65, 73, 77, 83, 99, 133, 141, 147, 163, 256, 355, 512,
                   516, 611, 1024, 1123, 1536, 2048, 2113, 2304, 4096, 28672, 61539, 65535]
    rs = np.random.RandomState(20261003)
    single_vals += [int(v) for v in rs.randint(0, 65536, 482)]
    singles = []
    for v in single_vals:
        c = ref_scalar(np.array([v], dtype=np.uint16))
        singles.append({"x": v, "slots": [i for i, k in enumerate(c) if k]})
        assert all(k in (0, 1) for k in c)
    exhaustive = {}
    for K in (4096, 65536):
        a = np.arange(K, dtype=np.uint32).astype(np.uint16)
        exhaustive[str(K)] = {
            "scalar": ref_scalar(a),
            "avx512": ref_variant("FLAGSTAT_avx512", a),
            "avx512_improved3": ref_variant("FLAGSTAT_avx512_improved3", a),
        }
    dump("kat.json", {
        "source": "FLAGSTAT_scalar libflagstats.h:118-176 via oracle/_ref",
        "single": singles,
        "exhaustive": exhaustive,
    })


def make_random():
    """Seeded cases: input = RandomState(seed).randint(0, hi, n).astype(uint16)[skip:]."""
    cases = []
    seed = 1000
    for hi in (4096, 65536):
        for n in LENGTHS:
            for skip in (0, 1):          # skip=1 -> 2-byte-aligned-only pointer
                seed += 1
                a = np.random.RandomState(seed).randint(0, hi, n + skip).astype(np.uint16)[skip:]
                case = {"seed": seed, "hi": hi, "n": n, "skip": skip,
                        "scalar": ref_scalar(a)}
                if n <= 64:
                    case["input"] = [int(v) for v in a]
                if n in (1024, 4097, 1048577) and skip == 0:
                    case["avx512"] = ref_variant("FLAGSTAT_avx512", a)
                    case["avx512_improved3"] = ref_variant("FLAGSTAT_avx512_improved3", a)
                cases.append(case)
    dump("random_cases.json", {
        "source": "FLAGSTAT_scalar libflagstats.h:118-176 via oracle/_ref",
        "recipe": "numpy.random.RandomState(seed).randint(0, hi, n+skip).astype(uint16)[skip:]",
        "cases": cases,
    })


def make_accumulate():
    """+= contract: second call adds onto the first's counters."""
    a = np.random.RandomState(7).randint(0, 65536, 5000).astype(np.uint16)
    b = np.random.RandomState(8).randint(0, 4096, 3000).astype(np.uint16)
    flags = np.zeros(32, dtype=np.uint32)
    flags[:] = np.arange(32) * 3 + 1   # non-zero garbage the callee must keep
    start = [int(v) for v in flags]
    oracle.ref_call("FLAGSTAT_scalar", a, flags)
    mid = [int(v) for v in flags]
    oracle.ref_call("FLAGSTAT_scalar", b, flags)
    end = [int(v) for v in flags]
    dump("accumulate.json", {
        "recipe": "a=RandomState(7).randint(0,65536,5000); b=RandomState(8).randint(0,4096,3000); "
                  "flags0=arange(32)*3+1; scalar(a) then scalar(b) into the same flags",
        "start": start, "after_a": mid, "after_b": end,
    })


def make_inmemory():
    """benchmark/inmemory.cpp:108-116 input, regenerated with libstdc++."""
    src = r"""
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
int main(int argc, char** argv) {
    size_t n = strtoull(argv[1], 0, 10);
    std::mt19937 eng; eng.seed(0);
    std::uniform_int_distribution<uint16_t> flag(0, 4096 - 1);
    for (size_t i = 0; i < n; i++) { uint16_t x = flag(eng); fwrite(&x, 2, 1, stdout); }
    return 0;
}
"""
    tmp = tempfile.mkdtemp()
    try:
        with open(os.path.join(tmp, "g.cpp"), "w") as f:
            f.write(src)
        subprocess.run(["g++", "-O2", "-o", os.path.join(tmp, "g"), os.path.join(tmp, "g.cpp")], check=True)
        out = {}
        for n in (102400, 1000000):
            raw = subprocess.run([os.path.join(tmp, "g"), str(n)], check=True, capture_output=True).stdout
            a = np.frombuffer(raw, dtype=np.uint16)
            assert a.size == n
            out[str(n)] = {
                "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
                "first16": [int(v) for v in a[:16]],
                "scalar": ref_scalar(a),
                "avx512": ref_variant("FLAGSTAT_avx512", a),
                "avx512_improved3": ref_variant("FLAGSTAT_avx512_improved3", a),
            }
    finally:
        shutil.rmtree(tmp)
    dump("inmemory_mt19937.json", {
        "source": "benchmark/inmemory.cpp:108-116 (mt19937 seed 0, uniform_int_distribution<uint16_t>(0,4095))",
        "recipe": "raw = MT19937(init_genrand(0)) 32-bit outputs; value = (raw * 4096) >> 32 (libstdc++ >= 11, Lemire; no rejection for a power-of-two range)",
        "cases": out,
    })


def make_pyflagstats():
    """Build the reference's Cython module against the ROOT header and capture dicts."""
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(os.path.join(REF, "python", "libflagstats.pyx"), tmp)
        setup = f"""
from setuptools import setup, Extension
from Cython.Build import cythonize
import numpy
setup(ext_modules=cythonize([Extension("pyflagstats", ["libflagstats.pyx"],
      include_dirs=["{REF}", "{REF}/python", numpy.get_include()],
      extra_compile_args=["-O2", "-w"])], language_level=3))
"""
        # include order: REF first so "libflagstats.h" is the root header
        with open(os.path.join(tmp, "setup.py"), "w") as f:
            f.write(setup)
        subprocess.run([sys.executable, "setup.py", "-q", "build_ext", "--inplace"], cwd=tmp, check=True,
                       capture_output=True)
        code = r"""
import json, sys, io, contextlib
import numpy as np
import pyflagstats
res = {"dicts": {}, "errors": {}}
def norm(d):
    return {"n_values": int(d["n_values"]),
            "passed": {k: int(v) for k, v in d["passed"].items()},
            "failed": {k: int(v) for k, v in d["failed"].items()},
            "passed_keys": list(d["passed"].keys()), "failed_keys": list(d["failed"].keys()),
            "value_type": type(d["passed"]["FUNMAP"]).__name__}
for hi in (4096, 65536):
    for n in (1, 100, 300, 600, 5000, 70000):
        a = np.random.RandomState(0).randint(0, hi, n).astype(np.uint16)
        res["dicts"]["%d_%d" % (hi, n)] = norm(pyflagstats.flagstats(a))
a = np.random.RandomState(0).randint(0, 4096, 2000).astype(np.uint16)
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    d = pyflagstats.flagstats(a[::2])
res["noncontig"] = {"dict": norm(d), "stdout": buf.getvalue()}
for name, arg in (("list", [1, 2, 3]), ("int32", np.zeros(4, dtype=np.int32)),
                  ("empty", np.zeros(0, dtype=np.uint16)), ("2d", np.zeros((4, 4), dtype=np.uint16))):
    try:
        pyflagstats.flagstats(arg)
        res["errors"][name] = None
    except Exception as e:
        res["errors"][name] = {"type": type(e).__name__, "msg": str(e)}
print(json.dumps(res))
"""
        out = subprocess.run([sys.executable, "-c", code], cwd=tmp, check=True, capture_output=True, text=True)
        res = json.loads(out.stdout.strip().splitlines()[-1])
    finally:
        shutil.rmtree(tmp)
    res["source"] = ("python/libflagstats.pyx:8-37 built against root libflagstats.h; "
                     "input = RandomState(0).randint(0, hi, n).astype(uint16); key = '<hi>_<n>'")
    dump("pyflagstats.json", res)


def make_pospopcnt():
    """Row f4: STORM_pospopcnt_u16 (python/libalgebra.h:3496-3551) on seeded inputs."""
    cases = []
    for seed, hi, n in ((1, 65536, 0), (2, 65536, 1), (3, 65536, 31), (4, 65536, 255), (5, 65536, 513), (6, 4096, 4097),
                        (7, 65536, 100003), (8, 65536, 1048577)):
        a = np.random.RandomState(seed).randint(0, hi, n).astype(np.uint16)
        d = oracle.ref_pospopcnt(a)
        nv = oracle.ref_pospopcnt(a, naive=True)
        assert [int(v) for v in d] == [int(v) for v in nv]
        cases.append({"seed": seed, "hi": hi, "n": n, "counts": [int(v) for v in d]})
    dump("pospopcnt.json", {"source": "STORM_pospopcnt_u16 python/libalgebra.h:3496-3551 via oracle/_ref",
                            "recipe": "numpy.random.RandomState(seed).randint(0, hi, n).astype(uint16)", "cases": cases})


if __name__ == "__main__":
    oracle.build(ref=True)
    make_pospopcnt()
    make_kat()
    make_random()
    make_accumulate()
    make_inmemory()
    make_pyflagstats()
