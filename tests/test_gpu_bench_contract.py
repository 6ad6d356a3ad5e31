"""GPU: bench.py's output contract (the driver parses ONE JSON line from rank 0): a small end-to-end run
must print exactly one JSON object with the agreed keys, the roofline and cpu_baseline objects, a
bit-exact parity verdict, and numbers that are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*extra):
    import socket
    with socket.socket() as sk:            # a free rendezvous port for the --force-dist run
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--flags-per-gpu", str(2 ** 27), "--steps", "6",
                        "--warmup", "2", "--cpu-seconds", "0.5", "--cpu-sample", str(2 ** 22), "--cpu-dram-per-core", str(2 ** 23),
                        "--probe-reps", "3", *extra],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_single_gpu_line(hip):
    d = run_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["unit"] == "Gflags/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes_per_launch"] == 2 * 2 ** 27
    # value = flags / wall time; achieved = 2 B/flag over the event time: the two clocks must agree
    assert abs(d["value"] * 2.0 / r["achieved"] - 1.0) < 0.1
    assert abs(d["ms_per_step"] / r["event_ms_per_launch"] - 1.0) < 0.1
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    if c["kind"] == "reference":
        # the all-core neighbours: cache-resident shards (flatters the CPU) and ONE pass over contiguous DRAM-resident shards of the
        # workload -- here 16 MiB per core, 256 MiB in the default run
        a, m = c["all_cores"], c["all_cores_dram"]
        assert a["cores"] == m["cores"] >= 1 and a["value"] > 0 and m["value"] > 0 and "contiguous shards" in m["sample"]
        assert m["min"] <= m["value"] <= m["max"]
    assert d["parity"].startswith("bit-exact")


def test_multi_gpu_step_at_world_size_one_self_spawned(hip):
    """--force-dist with no launcher: bench.py starts its (one) rank itself as a fresh child process and relays its
    line.  The N > 1 step (store form + the library's own RCCL all-reduce) on one GPU: the line must say which form
    ran, how many ranks RCCL saw, and the counters must still be the oracle's.  Default form: in line."""
    d = run_bench("--force-dist", "--cpu-seconds", "0")
    assert d["n_gpus"] == 1 and d["config"]["allreduce"].split()[0] == "in-line"
    assert d["config"]["allreduce_impl"].startswith("FLAGSTATS_hip_allreduce_counters")
    assert d["config"]["rccl_nranks"] == 1
    # which RCCL carried it: the shared object that holds the bound ncclAllReduce, and its version
    lib_info = d["config"]["rccl_library"]
    assert lib_info and "rccl" in lib_info["path"].lower() and lib_info["version"] > 20000, lib_info
    assert d["parity"].startswith("bit-exact") and d["cpu_baseline"] is None


def test_overlapped_form_checked_and_calibrated(hip):
    """--calibrate: the overlapped form first has to reproduce the in-line counters in every buffer of its ring, then
    both forms are timed with stream events and the faster one runs the timed steps."""
    d = run_bench("--force-dist", "--calibrate", "--cpu-seconds", "0")
    assert d["config"]["allreduce"].split()[0] in ("in-line", "overlapped")
    assert "calibrated with stream events" in d["config"]["allreduce"]
    assert d["parity"].startswith("bit-exact")
    d = run_bench("--force-dist", "--overlap", "--cpu-seconds", "0")
    assert d["config"]["allreduce"].split()[0] == "overlapped" and d["parity"].startswith("bit-exact")


def test_strong_scaling_calibrates_by_default(hip):
    """--strong: the shard shrinks with N but K2 + the all-reduce do not, so the in-line form cannot reach the 7.5x target
    (DESIGN.md "Multi-GPU", budget table) -- a strong run times both forms and takes the faster one without being asked;
    an explicit --no-overlap / --overlap still wins."""
    d = run_bench("--force-dist", "--strong", "--cpu-seconds", "0")
    assert d["scaling"] == "strong" and d["config"]["global_flags"] == 2 ** 27
    assert "calibrated with stream events" in d["config"]["allreduce"]
    assert d["parity"].startswith("bit-exact")
    d = run_bench("--force-dist", "--strong", "--no-overlap", "--cpu-seconds", "0")
    assert d["config"]["allreduce"] == "in-line" and d["parity"].startswith("bit-exact")


def test_two_ranks_self_spawned_on_one_gpu(hip):
    """`bench.py --gpus 2` with no launcher: two fresh rank processes, rendezvous on 127.0.0.1, both on GPU 0 (test-only
    --same-device with the gloo backend and torch's all_reduce: a 1-GPU box cannot host two RCCL ranks).  Rank 0's line:
    2 ranks, weak scaling, the all-reduced counters equal the oracle's sum over both shards."""
    d = run_bench("--gpus", "2", "--same-device", "--backend", "gloo", "--allreduce", "torch", "--cpu-seconds", "0")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_flags"] == 2 * 2 ** 27
    assert d["config"]["parallelism"] == "shard2" and d["config"]["allreduce"].split()[0] == "in-line"
    assert d["parity"].startswith("bit-exact") and "all 2 shards" in d["parity"]
    # what lets a reader attribute a slow multi-GPU number: every rank's own step time, the slowest rank, the collective alone
    c = d["config"]
    assert len(c["per_rank_ms"]) == 2 and all(t > 0 for t in c["per_rank_ms"]) and c["slowest_rank"] in (0, 1)
    assert c["per_rank_ms"][c["slowest_rank"]] == max(c["per_rank_ms"]) and c["allreduce_us"] > 0
    # ... and every rank's own K1 in the N = 1 form (no K2, no collective), measured in the same run: the step minus this is what
    # the multi-GPU form costs on that rank
    assert len(c["per_rank_k1_alone_ms"]) == 2 and all(0 < k <= 1.5 * t for k, t in zip(c["per_rank_k1_alone_ms"], c["per_rank_ms"]))


def test_a_rank_that_dies_in_set_up_fails_the_run_quickly(hip):
    """A rank killed (injected fault) after the rendezvous but before the communicator: rank 0 is left inside a
    collective; the launching parent must notice the dead rank, stop rank 0 and return non-zero within 30 s."""
    import time
    env = dict(os.environ, FLAGSTATS_BENCH_FAULT="1:comm")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo",
                        "--allreduce", "torch", "--flags-per-gpu", str(2 ** 24), "--steps", "3", "--warmup", "1", "--cpu-seconds", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    took = time.time() - t0
    assert r.returncode != 0 and took < 30 + 60, (r.returncode, took, r.stderr[-2000:])   # (+ the first import torch of a fresh box)
    assert "rank 1 exited with code 3" in r.stderr and "multi-rank run FAILED" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
