"""CPU: host-side logic that needs no GPU -- argument validation of the pyflagstats mirror,
dict construction, shard arithmetic, C-ABI library surface."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_validation_errors_match_reference_module():
    """Same exception types and messages as python/libflagstats.pyx:9-13,21-22 (captured from the
    reference's own module in tests/golden/pyflagstats.json).  Raised before any GPU call."""
    import pyflagstats
    errs = load_golden("pyflagstats.json")["errors"]
    args = {"list": [1, 2, 3], "int32": np.zeros(4, dtype=np.int32), "empty": np.zeros(0, dtype=np.uint16),
            "2d": np.zeros((4, 4), dtype=np.uint16)}
    types = {"ValueError": ValueError, "IndexError": IndexError}
    for name, arg in args.items():
        want = errs[name]
        with pytest.raises(types[want["type"]]) as ei:
            pyflagstats.flagstats(arg)
        assert str(ei.value) == want["msg"], name


def test_dict_layout_from_counters():
    """_as_dict restates python/libflagstats.pyx:24-35; compare with dicts the reference produced."""
    from libflagstats_amd.pyflagstats import _as_dict
    g = load_golden("pyflagstats.json")["dicts"]
    for key in ("4096_100", "65536_100", "4096_1"):  # n < 256: reference ran FLAGSTAT_scalar
        want = g[key]
        hi, n = (int(v) for v in key.split("_"))
        flags = np.zeros(32, dtype=np.uint32)
        for i, name in enumerate(want["passed_keys"][:15]):
            flags[i] = want["passed"][name]
            flags[16 + i] = want["failed"][name]
        got = _as_dict(flags, n)
        assert list(got["passed"].keys()) == want["passed_keys"]
        assert list(got["failed"].keys()) == want["failed_keys"]
        assert {k: int(v) for k, v in got["passed"].items()} == want["passed"]
        assert {k: int(v) for k, v in got["failed"].items()} == want["failed"]
        assert type(got["passed"]["mapped"]).__name__ == "uint32"


def test_shard_ranges_cover_exactly():
    from libflagstats_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1000, 2 ** 35 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
                assert e0 == b1 and b0 <= e0
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def declared_symbols():
    # the product ABI and the measurement entries (two headers, one library)
    text = open(os.path.join(ROOT, "include", "libflagstats_hip.h")).read() + open(os.path.join(ROOT, "include", "libflagstats_hip_probe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b((?:FLAGSTATS?|STORM)_[A-Za-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if n != "FLAGSTATS_func"))


def test_library_exports_every_declared_symbol():
    """The C-ABI .so loads without a GPU and exports everything include/libflagstats_hip.h and libflagstats_hip_probe.h declare;
    a reader finds the three reference symbols and their contract on the first screen of the product header."""
    from libflagstats_amd import _lib
    lib = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for name in syms:
        assert hasattr(lib, name), "missing export: " + name
        assert name in _lib.SIGNATURES, "no ctypes prototype for " + name
    assert sorted(_lib.SIGNATURES) == syms
    # header cites the reference interface each drop-in symbol replaces
    text = open(os.path.join(ROOT, "include", "libflagstats_hip.h")).read()
    for cite in ("libflagstats.h:3024", "libflagstats.h:2976", "libflagstats.h:2970"):
        assert cite in text
    first_screen = "\n".join(text.splitlines()[:60])
    for sym in ("FLAGSTATS_u16(", "FLAGSTATS_get_function(", "FLAGSTAT_hip(", "FLAGSTATS_func"):
        assert sym in first_screen, sym
    probe = open(os.path.join(ROOT, "include", "libflagstats_hip_probe.h")).read()
    for sym in ("FLAGSTATS_hip_read_probe", "FLAGSTATS_hip_time_device_u16", "FLAGSTATS_hip_sclk_under_load"):
        assert sym in probe and sym not in text, sym


def test_header_shim_consumer_compiles_and_links(tmp_path):
    """include/libflagstats.h lets an unmodified consumer of the reference API build against the .so."""
    import subprocess
    import torch
    exe = str(tmp_path / "consumer")
    libdir = os.path.join(ROOT, "libflagstats_amd")
    cmd = ["gcc", "-O1", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "consumer_shim.c"),
           "-L", libdir, "-lflagstats_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True)
    r = subprocess.run([exe, "5000"], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        # no GPU here: the call must fail loudly, not compute.  Default policy for the reference-shaped
        # entry points: message + abort() (their reference callers ignore the return value) ...
        assert r.returncode == -6 and "libflagstats_hip" in r.stderr and "aborting" in r.stderr, r.stderr
        # ... and FLAGSTATS_HIP_ON_ERROR=return hands the error to callers that check it (exit code 3)
        r = subprocess.run([exe, "5000"], capture_output=True, text=True, env=dict(os.environ, FLAGSTATS_HIP_ON_ERROR="return"))
        assert r.returncode == 3 and "libflagstats_hip" in r.stderr


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a GPU the hot path must fail loudly (non-zero + message), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from libflagstats_amd import _lib
    lib = _lib.lib()
    assert lib.FLAGSTATS_hip_available() == 0
    a = np.arange(100, dtype=np.uint16)
    flags = np.zeros(32, dtype=np.uint32)
    # importing the Python package leaves the process-wide policy alone: reference-shaped entries abort by default
    assert lib.FLAGSTATS_hip_get(b"on_error") == 1 or "FLAGSTATS_HIP_ON_ERROR" in os.environ
    old = lib.FLAGSTATS_hip_get(b"on_error")
    lib.FLAGSTATS_hip_set(b"on_error", 0)          # this test wants the return code, not an abort() of the test runner
    try:
        rc = lib.FLAGSTATS_u16(a.ctypes.data, a.size, flags.ctypes.data)
    finally:
        lib.FLAGSTATS_hip_set(b"on_error", old)
    assert rc != 0 and not flags.any()
    assert b"libflagstats_hip" in lib.FLAGSTATS_hip_last_error()
    import pyflagstats
    with pytest.raises(_lib.FlagstatsHipError):
        pyflagstats.flagstats(a)


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "libflagstats_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f
    assert "oracle" not in open(os.path.join(ROOT, "pyflagstats.py")).read()
    # measurement helpers outside tests/ must not lean on it either
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            src = open(os.path.join(ROOT, "tools", f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f


def test_shipped_kernel_file_carries_the_product_only():
    """one kernel, one body.  The shipped K1 / K2 file carries the three schedules the library ships and
    no measurement-build conditionals; the losers of the sweeps live in flagstat_kernels_tuning.hip, which only `make tuning`
    compiles; the product library therefore has no tuning launcher and refuses every other schedule."""
    csrc = os.path.join(ROOT, "libflagstats_amd", "csrc")
    text = open(os.path.join(csrc, "flagstat_kernels.hip")).read()
    assert len(text.splitlines()) < 700
    assert "#ifdef" not in text and "#ifndef" not in text and "FLAGSTAT_TUNING_VARIANTS" not in text
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = [ln for ln in mk.splitlines() if ln.startswith("SRCS")][0]
    assert "flagstat_kernels_tuning.hip" not in srcs          # only added under TUNING=1
    import ctypes
    from libflagstats_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    if not os.environ.get("FLAGSTATS_HIP_LIB"):
        assert lib.fsk_tuning_build() == 0
        assert [lib.fsk_variant_supported(v) for v in (9, 25, 71)] == [1, 1, 1]
        assert not any(lib.fsk_variant_supported(v) for v in (0, 1, 13, 17, 27, 29, 41, 57, 61, 63, 65, 67, 69, 73, 75, 77, 79, 81, 89, 153))


def test_kernel_id_is_the_device_codes_not_the_sources():
    """bench.py's kernel_source_id = sha256 over .text + .rodata of the gfx950 code object that defines fsk::flagstat_count,
    read out of the built .so (libflagstats_amd/kernel_id.py): a host-side edit no longer invalidates profiles/traffic.json."""
    from libflagstats_amd.kernel_id import _code_objects, kernel_id
    kid = kernel_id()
    assert re.fullmatch(r"[0-9a-f]{16}", kid)
    from libflagstats_amd import _lib
    cos = list(_code_objects(open(_lib.LIB_PATH, "rb").read()))
    assert len(cos) >= 5                                           # K1/K2, pospopcnt, generate, probes, LZ4, Zstandard
    import bench
    assert bench.kernel_source_id() == kid


def test_fork_is_detected_by_the_library_and_the_binding_without_a_gpu():
    """VERDICT r05 item 2, the half that needs no GPU: the first entry call claims the library for this process; a child
    fork()ed after that is told so by FLAGSTATS_hip_forked(), refused by the C entries with a text naming the fork and the
    remedy, and `_lib.lib()` raises FlagstatsHipError with the same advice.  The parent is unaffected.  (Every entry family, with
    a parent thread inside the engine during the fork: tests/test_host_asan.py; on the GPU: tests/test_gpu_fork.py.)"""
    import json
    from libflagstats_amd import _lib
    lib = _lib.lib()
    lib.FLAGSTATS_hip_available()            # any entry claims the library (no GPU here: it answers 0)
    assert lib.FLAGSTATS_hip_forked() == 0
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        rep = {}
        try:
            raw = _lib._lib
            rep["forked"] = int(raw.FLAGSTATS_hip_forked())
            out = np.zeros(32, dtype=np.uint64)
            a = np.arange(64, dtype=np.uint16)
            rep["rc"] = int(raw.FLAGSTATS_u16_x64(a.ctypes.data, a.size, out.ctypes.data))
            rep["text"] = raw.FLAGSTATS_hip_last_error().decode(errors="replace")
            try:
                _lib.lib()
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
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    rep = json.loads(data)
    assert "crash" not in rep, rep
    assert rep["forked"] == 1 and rep["rc"] != 0 and "fork()ed" in rep["text"] and "spawn" in rep["text"], rep
    assert rep["py"].startswith("FlagstatsHipError") and "spawn" in rep["py"], rep
    assert lib.FLAGSTATS_hip_forked() == 0 and _lib.lib() is lib
