"""INTEGRATION.md section B, tested: integration/apply_dispatch_patch.py applied to a scratch copy of the
reference's libflagstats.h adds the MI355X engine as the first branch of the reference's own dispatcher
(recipe: oracle/Makefile `refbench`; products oracle/_ref/dispatch_patched[_stub]).

CPU: the dispatch RULE in isolation (engine symbols stubbed): threshold, env overrides, no-GPU case.
GPU: the same patched header linked against the real libflagstats_hip.so: short inputs stay on the
reference's CPU kernels, long ones run on the MI355X, and both give FLAGSTAT_scalar's live slots."""
import os
import subprocess

import pytest

from conftest import ROOT

STUB = os.path.join(ROOT, "oracle", "_ref", "dispatch_patched_stub")
REAL = os.path.join(ROOT, "oracle", "_ref", "dispatch_patched")
LIVE = [2, 6, 7, 8, 10, 11, 12, 13, 14, 18, 22, 23, 24, 25, 26, 27, 28, 29, 30]
CPU_KERNELS = {"FLAGSTAT_scalar", "FLAGSTAT_sse4", "FLAGSTAT_avx2", "FLAGSTAT_avx512"}


def run(exe, n, **env):
    r = subprocess.run([exe, str(n), "7"], capture_output=True, text=True, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    chosen = lines[0].split()[1]
    rows = {ln.split()[0]: [int(v) for v in ln.split()[1:]] for ln in lines[1:]}
    return chosen, rows


@pytest.mark.skipif(not os.path.exists(STUB), reason="oracle/_ref/dispatch_patched_stub not built (needs /root/reference)")
def test_patched_dispatcher_rule():
    # below the threshold: the reference's own length rule (libflagstats.h:2999-3021), untouched
    for n in (0, 100, 255):
        assert run(STUB, n)[0] == "FLAGSTAT_scalar"
    small, _ = run(STUB, 1000)
    assert small in CPU_KERNELS and small != "FLAGSTAT_scalar"
    below, _ = run(STUB, (1 << 17) - 1)
    assert below in CPU_KERNELS
    # at / above it: the engine, through both entry points (slot 31 carries the stub's marker)
    for n in (1 << 17, 1_000_003):
        chosen, rows = run(STUB, n)
        assert chosen == "FLAGSTAT_hip"
        assert rows["func"][31] == 0xABCD and rows["u16"][31] == 0xABCD
        assert [rows["func"][s] for s in LIVE] == [rows["scalar"][s] for s in LIVE]
    # overrides: FLAGSTATS_BACKEND=cpu, FLAGSTATS_HIP_MIN_LEN, and "no GPU" (FLAGSTATS_hip_available() == 0)
    assert run(STUB, 1_000_003, FLAGSTATS_BACKEND="cpu")[0] in CPU_KERNELS
    assert run(STUB, 1000, FLAGSTATS_HIP_MIN_LEN="500")[0] == "FLAGSTAT_hip"
    assert run(STUB, 1_000_003, FLAGSTATS_HIP_MIN_LEN="2000000")[0] in CPU_KERNELS
    chosen, rows = run(STUB, 1_000_003, STUB_NO_GPU="1")
    assert chosen in CPU_KERNELS and rows["u16"][31] != 0xABCD


@pytest.mark.skipif(not os.path.exists(REAL), reason="oracle/_ref/dispatch_patched not built")
def test_patched_header_with_real_library_without_gpu_stays_on_cpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test below")
    chosen, rows = run(REAL, 1_000_003)
    assert chosen in CPU_KERNELS
    assert [rows["u16"][s] for s in LIVE] == [rows["scalar"][s] for s in LIVE]


@pytest.mark.gpu
def test_patched_reference_dispatch_on_hardware(hip):
    if not os.path.exists(REAL):
        pytest.skip("reference build product oracle/_ref/dispatch_patched not present on this machine")
    chosen, rows = run(REAL, 1000)
    assert chosen in CPU_KERNELS                      # short call: the host's own SIMD kernel, no PCIe round trip
    assert [rows["func"][s] for s in LIVE] == [rows["scalar"][s] for s in LIVE]
    assert run(REAL, (1 << 17) - 1)[0] in CPU_KERNELS      # just below the measured break-even (profiles/r03/small_calls.log)
    for n in (1 << 17, 5_000_001):
        chosen, rows = run(REAL, n)
        assert chosen == "FLAGSTAT_hip"
        for tag in ("func", "u16"):
            assert rows[tag] == rows["scalar"], (n, tag)      # the engine is scalar-exact on all 32 slots
    assert run(REAL, 5_000_001, FLAGSTATS_BACKEND="cpu")[0] in CPU_KERNELS
