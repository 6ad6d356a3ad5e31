"""CPU: bench.py as its own launcher (`--gpus N` without torchrun) must notice a dead or hung rank.  The parent process
never touches the GPU or imports torch, so its failure handling is testable here: a rank that exits non-zero (injected
with FLAGSTATS_BENCH_FAULT) or a deadline that starts at spawn ends the run quickly, with the other ranks stopped, a
non-zero exit code and the failing rank named -- not after some collective's own 10-30 minute timeout."""
import os
import subprocess
import sys
import time

from conftest import ROOT


def spawn(fault, *extra, timeout_env=None):
    env = dict(os.environ, FLAGSTATS_BENCH_FAULT=fault)
    if timeout_env is not None:
        env["FLAGSTATS_BENCH_SPAWN_TIMEOUT"] = str(timeout_env)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--flags-per-gpu", str(2 ** 20), "--cpu-seconds", "0", "--backend", "gloo", "--allreduce", "torch",
                        "--same-device", *extra], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    return r, time.time() - t0


def test_a_rank_that_dies_at_start_ends_the_run():
    # rank 1 dies before it does anything; rank 0 is MADE to hang at the same point (on a GPU box it would sit in the
    # rendezvous; here, without a GPU, it might otherwise die of its own in the same poll interval): the parent must stop it
    r, took = spawn("1:start,0:start:hang")
    assert r.returncode != 0 and took < 30, (r.returncode, took, r.stderr[-2000:])
    assert "rank 1 exited with code 3" in r.stderr and "injected fault: rank 1 dies at stage start" in r.stderr
    assert "injected fault: rank 0 hangs at stage start" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no result line from a failed run


def test_a_hung_rank_meets_the_deadline_that_starts_at_spawn():
    r, took = spawn("0:start:hang", timeout_env=3)
    # (rank 1 either waits for rank 0 in the rendezvous or fails on its own on a box without a GPU: both end the run)
    assert r.returncode != 0 and took < 30, (r.returncode, took, r.stderr[-2000:])
    assert "multi-rank run FAILED" in r.stderr
