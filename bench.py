#!/usr/bin/env python3
"""bench.py -- flagstat hot path on N MI355X of one node.

One "step" = one pass of the hot path over this rank's device-resident FLAG shard: N = 1: ONE launch of
K1 flagstat_count, whose workgroups add their totals to the 32 counters themselves (the ABI's +=
contract); N > 1: K1 + K2 flagstat_finalize in store form (one query per step) + the 32-counter all-reduce.  Workload = the
configuration BASELINE.json's metric is quoted on: 8 GiB of uniform-random uint16 (2^32 flags) per
GPU, generated on device by the library's counter-based generator (data: synthetic).  N > 1 is weak
scaling: every rank holds its own 8 GiB shard (seed + rank), 64 GiB at N = 8 (BASELINE config 3);
the data path of a multi-GPU step is the C ABI's FLAGSTATS_hip_device_u16_store + ONE
ncclAllReduce(uint64[32]) (FLAGSTATS_hip_allreduce_counters) per step; torch.distributed only carries
the rendezvous (unique id, barriers, max-over-ranks of the timings).

Order of a run: generate -> read-bandwidth probe (untimed; also lets the chip settle: after idle the
first ~20 launches of ANY kernel run 5-40 % slow, tools/step_times.py) -> W warm-up steps -> barrier ->
exactly K timed steps, one hipEvent per step on the launch stream -> barrier -> parity of the WHOLE
array against the oracle -> CPU baseline.

`--gpus N` without a launcher (no RANK in the environment) starts the N ranks itself: N fresh child processes of this
script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, spawned before anything in the parent has touched the GPU, rank
0's JSON line relayed.  Under `python -m torch.distributed.run --nproc-per-node N` (the driver's form) the ranks are
already there and nothing is spawned.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects: "roofline"
(HBM: algorithmic bytes = 2 B/flag over the event-timed average step on the launch stream) and
"cpu_baseline" (the reference's own dispatcher kernel, oracle/_ref, timed on this host on a bounded
sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "Gflags/s + achieved HBM GB/s vs roofline, 8 GiB uint16, 1/2/4/8 MI355X"


def kernel_source_id():
    """Identity of the K1 / K2 DEVICE code inside the shipped .so (sha256 over .text + .rodata of its gfx950 code object:
    libflagstats_amd/kernel_id.py).  profiles/traffic.json carries the id of the build its PMC figures were measured on: a
    figure measured on another build of the kernel is not reported.  (Until r05 this hashed three source files, and a
    host-side edit invalidated measurements of byte-identical device code.)"""
    try:
        from libflagstats_amd.kernel_id import kernel_id
        return kernel_id()
    except Exception as e:  # noqa: BLE001 -- an identity that cannot be read must not cost the bench line (traffic is then not reported)
        print("bench.py: kernel id not readable from the library (%r)" % (e,), file=sys.stderr)
        return "unknown"


def usable_cpus():
    """Threads the all-core leg may really run: the affinity mask, capped by the cgroup CPU quota
    (a GPU box hands a 1-GPU job a CPU *share*, e.g. 16 of 256 logical CPUs, as a CFS quota: 256
    spinning threads under a 16-CPU quota measure the throttler, not the kernel -- r01's all-core
    figures ranged 5-47 Gflags/s for that reason)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = "%d CPUs in the affinity mask" % n
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:   # cgroup v1
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        # one CPU of slack: exactly `quota` spinning threads plus the runtime's helper threads overdraw the
        # quota and the whole group is frozen for the rest of the 100 ms period (58-80 Gflags/s run to run)
        n = max(1, int(quota) - 1)
        note += ", cgroup quota %.1f CPUs (one left free)" % quota
    return n, note


def cpu_baseline(seconds: float, sample_flags: int, seed: int, dram_flags_per_core: int = 2 ** 27):
    """Time the reference's own kernel (what FLAGSTATS_get_function returns on this host,
    libflagstats.h:2976-3022) on a prefix of the rank-0 workload: 1 thread, then every core with
    pinned threads and shard-local memory (ref_dispatch_mt_bench, oracle/ref_wrap.cpp)."""
    import numpy as np

    import oracle

    a = oracle.generate(oracle.GEN_UNIFORM, seed, 0xFFFF, 0, sample_flags)
    ref = oracle.load_ref()
    lib = oracle.load_c()
    p16 = ctypes.cast(a.ctypes.data, ctypes.POINTER(ctypes.c_uint16))
    u64p = ctypes.POINTER(ctypes.c_uint64)
    if ref is not None:
        kind = "reference"
        name = ref.ref_dispatch_name(min(sample_flags, 2 ** 30)).decode()

        def run(ptr, n):
            out = np.zeros(32, dtype=np.uint64)
            ref.ref_dispatch_x64(ptr, n, out.ctypes.data_as(u64p))
            return out
    else:  # reference build absent on this box: time our C restatement instead
        kind = "port"
        name = "oracle_flagstat_hist_u16"

        def run(ptr, n):
            out = np.zeros(32, dtype=np.uint64)
            lib.oracle_flagstat_hist_u16(ptr, n, out.ctypes.data_as(u64p))
            return out

    run(p16, sample_flags)  # warm
    t0 = time.perf_counter()
    passes = 0
    while True:
        run(p16, sample_flags)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    one = sample_flags * passes / dt / 1e9

    # the reference's only SIMD variant with FLAGSTAT_scalar's exact semantics on full-range input
    # (libflagstats.h:2445-2644), 1 thread, ~3 s, for context
    exact = None
    if ref is not None and hasattr(ref, "ref_FLAGSTAT_avx512_improved3") and ref.ref_has_avx512bw():
        f32 = np.zeros(32, dtype=np.uint32)
        p32 = f32.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        m = min(sample_flags, 2 ** 30)
        ref.ref_FLAGSTAT_avx512_improved3(p16, m, p32)
        t1 = time.perf_counter()
        k = 0
        while time.perf_counter() - t1 < 3.0:
            f32[:] = 0
            ref.ref_FLAGSTAT_avx512_improved3(p16, m, p32)
            k += 1
        exact = {"kernel": "FLAGSTAT_avx512_improved3", "value": round(m * k / (time.perf_counter() - t1) / 1e9, 4),
                 "unit": "Gflags/s", "cores": 1}

    # all cores: one pinned thread per logical CPU, 8 MiB of its own (first-touched, so socket-local)
    # uniform-random flags each -- beyond the per-core caches -- timed inside C between two barriers
    allc = None
    cores, cores_note = usable_cpus()
    if ref is not None and hasattr(ref, "ref_dispatch_mt_bench"):
        ref.ref_dispatch_mt_bench.restype = ctypes.c_double
        ref.ref_dispatch_mt_bench.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, u64p]
        per = 4 * 2 ** 20
        out = np.zeros(32, dtype=np.uint64)
        probe = ref.ref_dispatch_mt_bench(per, cores, 4, seed, out.ctypes.data_as(u64p))
        reps = max(4, min(100000, int(4 * 5.0 / max(probe, 1e-6))))   # aim at ~5 s
        runs = []
        for _ in range(3):
            out[:] = 0
            dt_all = ref.ref_dispatch_mt_bench(per, cores, reps, seed, out.ctypes.data_as(u64p))
            runs.append(per * cores * reps / dt_all / 1e9)
        runs.sort()
        allc = {"value": round(runs[1], 4), "unit": "Gflags/s", "cores": cores, "min": round(runs[0], 4),
                "max": round(runs[2], 4),
                "sample": "%d flags (%.0f MiB) in %d pinned, shard-local shards x %d passes, median of 3 runs; %s"
                          % (per * cores, per * cores * 2 / 2 ** 20, cores, reps, cores_note)}

    # all cores over DRAM: the same pinned threads, each ONE pass over its contiguous shard (>= 256 MiB: far beyond the caches)
    # of the first cores x 2^27 flags of the rank-0 workload -- BASELINE.md section 4 step 3's form, and the honest neighbour of
    # an HBM-bound GPU figure (the cache-resident figure above flatters the CPU).  The shards are generated by threads pinned
    # like their readers, so a shard's pages lie where it is read.
    dram = None
    try:
        dram = _all_cores_dram(np, oracle, lib, ref, u64p, cores, cores_note, seed, dram_flags_per_core)
    except Exception as e:  # noqa: BLE001 -- a reported baseline must not cost the bench line
        print("bench.py: all-cores-over-DRAM CPU leg failed (%r)" % (e,), file=sys.stderr)

    return {
        "value": round(one, 4), "unit": "Gflags/s", "cores": 1, "kind": kind, "kernel": name,
        "sample": "first %d flags (%.0f MiB) of the rank-0 workload, %d passes in %.1f s, 1 thread"
                  % (sample_flags, sample_flags * 2 / 2 ** 20, passes, dt),
        "scalar_exact_variant": exact,
        "all_cores": allc,
        "all_cores_dram": dram,
    }


def _all_cores_dram(np, oracle, lib, ref, u64p, cores, cores_note, seed, dram_flags_per_core):
    """cpu_baseline's all-cores-over-DRAM leg (see there)."""
    dram = None
    if ref is not None and hasattr(ref, "ref_dispatch_mt_shards") and dram_flags_per_core > 0:
        import threading
        per = int(dram_flags_per_core)
        total = per * cores
        try:
            big = np.empty(total, dtype=np.uint16)
        except MemoryError:
            big = None
        if big is not None:
            allowed = sorted(os.sched_getaffinity(0))

            def fill(k):
                try:
                    os.sched_setaffinity(0, {allowed[k % len(allowed)]})   # (pid 0 = the calling thread)
                except OSError:
                    pass
                lib.oracle_generate_u16(oracle.GEN_UNIFORM, seed, 0xFFFF, k * per, per,
                                        ctypes.cast(big.ctypes.data + 2 * k * per, ctypes.POINTER(ctypes.c_uint16)))

            tg = time.perf_counter()
            ths = [threading.Thread(target=fill, args=(k,)) for k in range(cores)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            gen_s = time.perf_counter() - tg
            rounds = 3
            secs = (ctypes.c_double * rounds)()
            out = np.zeros(32, dtype=np.uint64)
            ref.ref_dispatch_mt_shards.restype = ctypes.c_int
            ref.ref_dispatch_mt_shards.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                                   ctypes.POINTER(ctypes.c_double), u64p]
            if ref.ref_dispatch_mt_shards(big.ctypes.data, total, cores, rounds, secs, out.ctypes.data_as(u64p)) == 0:
                rates = sorted(total / s_ / 1e9 for s_ in secs)
                dram = {"value": round(rates[1], 4), "unit": "Gflags/s", "cores": cores, "min": round(rates[0], 4), "max": round(rates[2], 4),
                        "GBs": round(rates[1] * 2, 1),
                        "sample": "first %d flags (%.2f GiB) of the rank-0 workload as %d contiguous shards of %.0f MiB, one pinned thread and "
                                  "ONE pass each per round, median of %d rounds (generated in %.1f s by threads pinned like the readers); %s"
                                  % (total, total * 2 / 2 ** 30, cores, per * 2 / 2 ** 20, rounds, gen_s, cores_note)}
            del big
    return dram


def quantiles(ms):
    s = sorted(ms)
    n = len(s)
    pick = lambda q: s[min(n - 1, max(0, int(round(q * (n - 1)))))]  # noqa: E731
    return {"median": round(pick(0.5), 5), "p10": round(pick(0.1), 5), "p90": round(pick(0.9), 5),
            "min": round(s[0], 5), "max": round(s[-1], 5), "first": round(ms[0], 5)}


def spawn_ranks(nproc):
    """Be our own launcher: `nproc` fresh child processes of this script, one rank each.  The parent never imports
    torch or the library (nothing here has initialised the GPU).  It relays rank 0's stdout -- the JSON line -- as it
    comes (a reader thread), keeps every rank's stderr in a file of its own, and polls ALL children: the first rank that
    exits non-zero, or a deadline that starts at spawn (env FLAGSTATS_BENCH_SPAWN_TIMEOUT, default 900 s), ends the run --
    the children started here are killed (nothing is re-executed), the failing rank's stderr tail is printed, the exit
    code is non-zero.  A rank that dies in communicator set-up therefore cannot leave the others, and this parent, inside
    a collective until somebody else's timeout."""
    import socket
    import subprocess
    import tempfile
    import threading

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    deadline = time.time() + float(os.environ.get("FLAGSTATS_BENCH_SPAWN_TIMEOUT", "900"))
    procs, errs = [], []
    for r in range(nproc):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ef = tempfile.TemporaryFile(mode="w+b")
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else ef, stderr=ef))

    def relay():
        for line in procs[0].stdout:                # rank 0's stdout, as it comes
            sys.stdout.write(line.decode(errors="replace"))
            sys.stdout.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()

    def tail(r, lines=40):
        errs[r].flush()
        errs[r].seek(0)
        text = errs[r].read().decode(errors="replace").splitlines()
        return "\n".join(text[-lines:])

    rc, why, culprit = 0, None, None
    try:
        while True:
            codes = [pr.poll() for pr in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                # (several ranks may have died inside one poll interval: all are named, the lowest one's stderr tail is shown)
                culprit, rc = bad[0], codes[bad[0]]
                why = ", ".join("rank %d exited with code %d" % (r, codes[r]) for r in bad)
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                culprit = next(r for r, c in enumerate(codes) if c is None)
                rc, why = 124, "deadline passed with rank(s) %s still running" % [r for r, c in enumerate(codes) if c is None]
                break
            time.sleep(0.2)
    except KeyboardInterrupt:
        rc, why = 130, "interrupted"
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()                           # exactly the processes started here
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        th.join(timeout=5)
    for r in range(nproc):                          # every rank's stderr, in rank order (rank > 0: its stdout too)
        text = tail(r, lines=10 ** 9)
        if text:
            sys.stderr.write(text + "\n")
    if why:
        sys.stderr.write("bench.py: multi-rank run FAILED: %s; the other ranks were stopped.\n" % why)
        if culprit is not None:
            sys.stderr.write("---- last lines of rank %d ----\n%s\n----\n" % (culprit, tail(culprit)))
        sys.stderr.flush()
        return rc if rc else 1
    return 0


def injected_fault(stage, rank):
    """TEST ONLY (tests/test_bench_spawn.py, tests/test_gpu_bench_contract.py): env FLAGSTATS_BENCH_FAULT =
    "<rank>:<stage>[:hang][,<rank>:<stage>[:hang]...]" makes each named rank exit with code 3 (or hang) when it reaches the
    stage: start | init | comm | warmup."""
    for spec in os.environ.get("FLAGSTATS_BENCH_FAULT", "").split(","):
        parts = spec.split(":")
        if not (len(parts) >= 2 and parts[0] == str(rank) and parts[1] == stage):
            continue
        if len(parts) > 2 and parts[2] == "hang":
            print("bench.py: injected fault: rank %d hangs at stage %s" % (rank, stage), file=sys.stderr, flush=True)
            while True:
                time.sleep(3600)
        print("bench.py: injected fault: rank %d dies at stage %s" % (rank, stage), file=sys.stderr, flush=True)
        os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--flags-per-gpu", type=int, default=2 ** 32, help="default 2^32 flags = 8 GiB uint16")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=2 ** 27)
    ap.add_argument("--cpu-dram-per-core", type=int, default=2 ** 27,
                    help="flags per core of the all-cores-over-DRAM leg (default 2^27 = 256 MiB per pinned thread; 0: skip)")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--probe-reps", type=int, default=40,
                    help="read-only bandwidth probe launches before the warm-up steps (0: no probe; the first timed "
                         "steps then carry the chip's after-idle transient when --warmup is small)")
    ap.add_argument("--parity", choices=("full", "off"), default="full",
                    help="full: every rank's whole shard vs the oracle (all host cores, ~1 s per 2^32 flags)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --flags-per-gpu is the TOTAL array, split into contiguous shards over the ranks "
                         "(default is weak scaling: every rank holds its own --flags-per-gpu shard)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the rendezvous; nccl = RCCL (default)")
    ap.add_argument("--allreduce", choices=("c-abi", "torch"), default="c-abi",
                    help="who issues the step's all-reduce: the library's C entry over its own RCCL communicator "
                         "(default) or torch.distributed")
    ap.add_argument("--same-device", action="store_true",
                    help="TEST ONLY: put every rank on GPU 0 (use with --backend gloo --allreduce torch) to exercise the "
                         "multi-rank launch contract on a single-GPU box; the number it prints is not a scaling result")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: run the all-reduce of step i on a side stream, overlapping K1 of step i+1 (every counter "
                         "buffer of the ring is checked on every rank during the warm-up; a mismatch falls back to in-line)")
    ap.add_argument("--calibrate", action="store_true",
                    help="N > 1: time both forms (stream events, max over ranks) before the warm-up and use the faster one")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: the all-reduce of each step on the launch stream, in line (this is the default: the "
                         "overlapped form's stream ordering has only ever run at world size 1)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 step (store + RCCL all-reduce) even at world size 1: exercises the multi-GPU "
                         "code path on a single-GPU box")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.force_dist):
        sys.exit(spawn_ranks(max(1, args.gpus)))   # no launcher: start the ranks ourselves (before any GPU call)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    injected_fault("start", rank)

    import numpy as np
    import torch
    import torch.distributed as dist

    from libflagstats_amd import _lib, device

    if not os.path.exists(_lib.LIB_PATH):
        # the in-tree extension normally travels with the snapshot; if it does not, build it here
        # (hipcc is on the GPU image): local rank 0 builds, the other ranks wait for the file
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            for _ in range(600):
                if os.path.exists(_lib.LIB_PATH):
                    break
                time.sleep(1.0)
            time.sleep(2.0)

    ndev = torch.cuda.device_count()
    if args.same_device:
        local_rank = 0
    elif ndev == 1 and local_rank > 0:
        local_rank = 0   # the launcher isolates one visible GPU per rank (HIP_VISIBLE_DEVICES)
    elif local_rank >= ndev:
        sys.exit("bench.py: local rank %d but only %d GPUs are visible: refusing to share a GPU between ranks "
                 "(--same-device is the explicit, test-only way)" % (local_rank, ndev))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(local_rank), "FLAGSTATS_hip_init")
    multi = world > 1 or args.force_dist
    comm_init_hung = False
    comm = None
    ar_impl = None
    rccl_nranks = None
    rccl_library = None   # {path, version} of the RCCL the C-ABI all-reduce is bound to (N > 1)
    if multi:
        injected_fault("init", rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        ar_impl = "torch.distributed (%s)" % args.backend
        injected_fault("comm", rank)
        if args.allreduce == "c-abi" and not args.same_device:
            # the library's own RCCL communicator: rank 0 makes the 128-byte id, the rendezvous ships it
            try:
                ident = torch.zeros(128, dtype=torch.uint8)
                if rank == 0:
                    # a failure here must not keep rank 0 out of the broadcast the other ranks are waiting in: it
                    # ships an all-zero id instead, and every rank then skips the C-ABI communicator
                    buf = (ctypes.c_char * 128)()
                    if lib.FLAGSTATS_hip_comm_unique_id(buf) == 0:
                        ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
                    else:
                        print("bench.py: FLAGSTATS_hip_comm_unique_id failed: %s"
                              % lib.FLAGSTATS_hip_last_error().decode(errors="replace"), file=sys.stderr)
                if args.backend == "nccl":
                    ident = ident.to(dev)
                dist.broadcast(ident, src=0)
                raw = bytes(ident.cpu().numpy().tobytes())
                if not any(raw):
                    raise _lib.FlagstatsHipError("rank 0 could not make an RCCL unique id")
                # ncclCommInitRank is itself a collective: if it never returns on some rank (first time this path
                # meets a real multi-GPU node), do not hang the run -- give it a bounded time on a helper thread,
                # then let every rank agree (below) to carry the all-reduce with torch.distributed instead
                import threading
                box = {}

                def init_comm():
                    box["comm"] = lib.FLAGSTATS_hip_comm_init_rank(raw, world, rank, local_rank)
                    box["err"] = lib.FLAGSTATS_hip_last_error().decode(errors="replace")

                th = threading.Thread(target=init_comm, daemon=True)
                th.start()
                th.join(timeout=float(os.environ.get("FLAGSTATS_BENCH_COMM_TIMEOUT", "120")))
                if th.is_alive():
                    comm_init_hung = True
                    raise _lib.FlagstatsHipError("FLAGSTATS_hip_comm_init_rank did not return within the time limit")
                comm = box.get("comm")
                if not comm:
                    raise _lib.FlagstatsHipError(box.get("err", "communicator creation failed"))
                ar_impl = "FLAGSTATS_hip_allreduce_counters (C ABI, ncclAllReduce uint64[32])"
                rccl_nranks = int(lib.FLAGSTATS_hip_comm_count(comm))
                buf, ver = ctypes.create_string_buffer(1024), ctypes.c_int(-1)
                if lib.FLAGSTATS_hip_comm_library(buf, len(buf), ctypes.byref(ver)) == 0:
                    rccl_library = {"path": buf.value.decode(errors="replace"), "version": int(ver.value)}
                if rccl_nranks != world:
                    raise _lib.FlagstatsHipError("ncclCommCount says %d ranks, the launcher %d" % (rccl_nranks, world))
            except Exception as e:  # noqa: BLE001 -- a scaling run must not die on the communicator; say so instead
                print("bench.py: C-ABI RCCL communicator unavailable (%r); using torch.distributed all_reduce" % (e,),
                      file=sys.stderr)
                if comm and not comm_init_hung:
                    lib.FLAGSTATS_hip_comm_destroy(comm)
                comm = None
                rccl_nranks = None
                rccl_library = None
                ar_impl = "torch.distributed (%s)" % args.backend
            # every rank must take the same path
            flag = torch.tensor([1 if comm else 0], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and comm:
                lib.FLAGSTATS_hip_comm_destroy(comm)
                comm = None
                ar_impl = "torch.distributed (%s)" % args.backend

    n = args.flags_per_gpu
    first = 0
    if args.strong:
        from libflagstats_amd.dist import shard_range
        b, e = shard_range(args.flags_per_gpu, rank, world)
        n, first = e - b, b
    flags = torch.empty(n, dtype=torch.int16, device=dev)           # this rank's shard, resident in HBM
    shard_seed = args.seed if args.strong else args.seed + rank
    device.generate_torch(flags, device.GEN_UNIFORM, seed=shard_seed, mask=0xFFFF, first_index=first)
    counters = torch.zeros(32, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    # Strong scaling shrinks the shard with N (8 GiB total: 1 GiB shards at N = 8, K1 ~150 us) while K2 + the all-reduce
    # do not shrink: in line they cost more than the 3-4 us per step that 7.5x leaves (DESIGN.md "Multi-GPU", budget table),
    # so --strong times both forms first and takes the faster one unless a form was asked for explicitly.
    if args.strong and multi and not (args.overlap or args.no_overlap):
        args.calibrate = True
    overlap = multi and (args.overlap or args.calibrate) and not args.no_overlap   # weak scaling default: in line
    main_stream = torch.cuda.current_stream(dev)
    comm_stream = torch.cuda.Stream(device=dev) if overlap else None
    RING = 8                   # counter buffers in flight between the launch stream and the all-reduce stream
    bufs = [counters] + [torch.zeros(32, dtype=torch.int64, device=dev) for _ in range(RING - 1)]
    reduced = [None] * RING    # event: the all-reduce that last used bufs[k] has finished
    state = {"i": 0, "overlap": overlap}

    def allreduce(buf, stream):
        if comm:
            _lib.check(lib.FLAGSTATS_hip_allreduce_counters(buf.data_ptr(), comm, ctypes.c_void_p(stream.cuda_stream)),
                       "FLAGSTATS_hip_allreduce_counters")
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    def step():
        # N = 1: counters accumulate across steps (the ABI's += contract, as the reference's
        # bench accumulates across blocks, benchmark/flagstats.cpp:304,328-329), so a step is
        # exactly one K1 launch (its workgroups add their totals to the counters).  N > 1: a step is one whole query: count (K2 stores), all-reduce of the
        # 32 counters.  The all-reduce of query i runs on a side stream while K1 of query i+1 streams
        # its shard (a ring of counter buffers), so the collective's latency is off the critical path;
        # every query's all-reduce still completes inside the timed region (drain() below).
        if not state["overlap"]:
            device.count_torch(flags, counters, store=multi)   # K1 + K2 on torch's current stream
            if multi:
                allreduce(counters, main_stream)  # the path's only exchange: 256 B over xGMI
            return
        k = state["i"] % RING
        state["i"] += 1
        buf = bufs[k]
        if comm:
            # the library orders the collective behind the kernels with device-scope events (no system fence);
            # once per ring the launch stream waits (on the device) for the all-reduce stream to catch up, which
            # makes every counter buffer of the ring reusable
            if k == 0 and state["i"] > 1:
                _lib.check(lib.FLAGSTATS_hip_stream_wait_stream(ctypes.c_void_p(main_stream.cuda_stream),
                                                                ctypes.c_void_p(comm_stream.cuda_stream), local_rank),
                           "FLAGSTATS_hip_stream_wait_stream")
            _lib.check(lib.FLAGSTATS_hip_device_u16_allreduce_overlapped(
                flags.data_ptr(), n, buf.data_ptr(), comm, ctypes.c_void_p(main_stream.cuda_stream),
                ctypes.c_void_p(comm_stream.cuda_stream)), "FLAGSTATS_hip_device_u16_allreduce_overlapped")
            return
        # torch.distributed carries the collective: torch events order the two streams
        if reduced[k] is not None and not reduced[k].query():
            # K2 may overwrite buf only after the all-reduce that last used it (RING steps ago) has finished.
            # The HOST waits (it runs steps ahead of the GPU, so practically never); a wait_event on the launch
            # stream instead stalls the queue on this runtime (profiles/r02/dist_step_overhead.log)
            reduced[k].synchronize()
        device.count_torch(flags, buf, store=True)
        counted = torch.cuda.Event()
        counted.record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(counted)
            allreduce(buf, comm_stream)
            ev = torch.cuda.Event()
            ev.record(comm_stream)
        reduced[k] = ev

    def drain():
        if state["overlap"]:
            main_stream.wait_stream(comm_stream)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # read-only probe with K1's load pattern ("fraction of a measured read-only
    # probe kernel"): what this chip delivers to ANY kernel reading this buffer, measured in this run
    probe_gbs = None
    if args.probe_reps > 0 and n >= 2 ** 22:
        ms = ctypes.c_float(0.0)
        _lib.check(lib.FLAGSTATS_hip_read_probe(flags.data_ptr(), 2 * n, 1, args.probe_reps, args.probe_reps,
                                                ctypes.byref(ms)), "FLAGSTATS_hip_read_probe")
        probe_gbs = 2.0 * n * args.probe_reps / (ms.value * 1e-3) / 1e9

    # N > 1, untimed: which form of the step is faster HERE?  Overlapping hides the collective's latency but lets its
    # kernel share CUs with the next K1 (whose grid ends with its slowest workgroup); in line exposes the latency
    # instead.  On one GPU in line wins by ~12 us per step; on an 8-GPU node the all-reduce is slower and the answer
    # may flip, so both are run for a few steps and every rank takes the form with the smaller max-over-ranks time.
    calib_note = ""

    def ring_is_right():
        """Overlapped form only: one in-line step gives the reference counters, then 2 x RING overlapped steps; every
        buffer of the ring must hold exactly those counters on every rank (the ordering between the launch stream,
        the all-reduce stream and the ring re-use is the part that has never met a second GPU)."""
        state["overlap"] = False
        step()
        torch.cuda.synchronize()
        ref = counters.clone()
        state["overlap"] = True
        state["i"] = 0
        for b in bufs:
            b.fill_(-1)
        for _ in range(2 * RING):
            step()
        drain()
        torch.cuda.synchronize()
        ok = all(bool(torch.equal(b, ref)) for b in bufs)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        state["i"] = 0
        return int(flag.item()) == 1

    if multi and overlap:
        if not ring_is_right():
            print("bench.py: overlapped all-reduce gave wrong counters in some ring buffer on some rank; using the in-line form",
                  file=sys.stderr)
            overlap = False
            state["overlap"] = False
            calib_note = " (overlapped form failed its ring check)"
    if multi and overlap and args.calibrate:
        times = [0.0, 0.0]
        state["overlap"] = False
        for _ in range(20):           # the chip settles first (the first launches after idle run long)
            step()
        drain()
        for seg in range(6):          # in line / overlapped alternately, 3 segments of 15 steps each
            form = seg & 1
            state["overlap"] = bool(form)
            step()
            drain()
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(15):
                step()
            drain()               # the launch stream waits for the collectives: e1 is behind all of them
            e1.record()
            e1.synchronize()
            times[form] += e0.elapsed_time(e1) * 1e-3
        tt = torch.tensor(times, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        state["overlap"] = bool(float(tt[1]) < float(tt[0]))
        calib_note = " (calibrated with stream events: in-line %.1f us/step, overlapped %.1f us/step)" % (
            float(tt[0]) / 45 * 1e6, float(tt[1]) / 45 * 1e6)
        state["i"] = 0
    elif multi:
        state["overlap"] = overlap

    for _ in range(args.warmup):
        step()
    drain()
    injected_fault("warmup", rank)
    # N > 1, untimed: what the collective alone costs here (stream events around 10 all-reduces of a scratch uint64[32] on
    # the launch stream; the slowest rank's figure) -- so that a first 8-GPU number can be read: step = K1 + K2 + this
    allreduce_us = None
    if multi:
        scratch = torch.zeros(32, dtype=torch.int64, device=dev)
        allreduce(scratch, main_stream)
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(10):
            allreduce(scratch, main_stream)
        a1.record()
        a1.synchronize()
        t = torch.tensor([a0.elapsed_time(a1) * 100.0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allreduce_us = round(float(t[0]), 2)
    # N > 1, untimed: every rank's OWN K1 (the N = 1 form: one launch, atomic epilogue, no K2, no collective) over its shard, 20
    # launches between stream events -- so that a scaling number can be taken apart without a second run: step - K1 alone = what
    # K2 + the collective + waiting for the slowest rank cost; the spread of K1 alone over the ranks = chip-to-chip variation
    per_rank_k1_ms = None
    if multi:
        k1_scratch = torch.zeros(32, dtype=torch.int64, device=dev)
        for _ in range(3):
            device.count_torch(flags, k1_scratch)
        k0, k1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        for _ in range(20):
            device.count_torch(flags, k1_scratch)
        k1e.record()
        k1e.synchronize()
        mine = torch.zeros(world, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        mine[rank] = k0.elapsed_time(k1e) / 20.0
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank_k1_ms = [round(float(x), 5) for x in mine.cpu()]
    if not multi:
        counters.zero_()      # stream-ordered behind the warm-up steps: ONE sync gap before the timed region, not two
    # the per-step events exist (torch creates the HIP event at the first record) BEFORE the barrier: the GPU idles
    # between the synchronize and the first timed launch, and every extra 100 us of that gap shows in the first steps
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for e in evs:
        e.record()
    barrier()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        if i + 1 < args.steps:
            evs[i + 1].record()
    drain()
    evs[args.steps].record()
    barrier()
    wall = time.perf_counter() - t0
    ev_ms = evs[0].elapsed_time(evs[args.steps])
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]

    per_rank_ms = None
    if multi:
        # every rank's own event-timed mean step, gathered: a slow first 8-GPU number can then be laid at a rank's door
        mine = torch.zeros(world, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        mine[rank] = ev_ms / args.steps
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(x), 5) for x in mine.cpu()]
        tmax = torch.tensor([wall, ev_ms], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tmax[0]), float(tmax[1])

    # shader clock under K1's load, measured after the timed region (same thermal state, a few ms later)
    sclk = None
    if rank == 0:
        try:
            mhz = ctypes.c_double(0.0)
            _lib.check(lib.FLAGSTATS_hip_sclk_under_load(flags.data_ptr(), n, max(3, min(20, int(30.0 / max(ev_ms / args.steps, 1e-3)))),
                                                         ctypes.byref(mhz)), "FLAGSTATS_hip_sclk_under_load")
            sclk = round(mhz.value, 1)
        except Exception as e:  # noqa: BLE001 -- a missing clock reading must not cost the bench line
            print("bench.py: no shader clock reading (%r)" % (e,), file=sys.stderr)
    ms_per_step = wall * 1e3 / args.steps
    total_flags = args.flags_per_gpu if args.strong else n * world
    value = total_flags * args.steps / wall / 1e9          # whole-job Gflags/s
    ev_ms_per_step = ev_ms / args.steps
    achieved = 2.0 * n / (ev_ms_per_step * 1e-3) / 1e9     # GB/s per GPU, algorithmic bytes

    result = None
    if rank == 0:
        last = bufs[(state["i"] - 1) % RING] if state["overlap"] else counters
        got = last.cpu().numpy().view(np.uint64)
        passes = 1 if multi else args.steps   # N = 1 accumulated `steps` identical passes
        assert not (got % np.uint64(passes)).any(), "accumulated counters are not a multiple of the step count"
        got = got // np.uint64(passes)
        parity = "not checked"
        if args.parity == "full":
            # checker: the oracle regenerates every rank's whole shard (counter-based generator) on all
            # host cores and counts it with FLAGSTAT_scalar's rule; N > 1: the all-reduced counters must
            # equal the sum over the ranks' shards
            import oracle
            tp = time.perf_counter()
            want = np.zeros(32, dtype=np.uint64)
            for r in range(world):
                if args.strong:
                    from libflagstats_amd.dist import shard_range
                    b, e = shard_range(args.flags_per_gpu, r, world)
                    want += oracle.flagstat_generated(oracle.GEN_UNIFORM, args.seed, 0xFFFF, b, e - b)
                else:
                    want += oracle.flagstat_generated(oracle.GEN_UNIFORM, args.seed + r, 0xFFFF, 0, n)
            ok = np.array_equal(got, want)
            parity = ("bit-exact vs oracle on the whole workload (%d flags, all %d shards; oracle took %.1f s)"
                      % (total_flags, world, time.perf_counter() - tp)) if ok else "MISMATCH vs oracle"
            if not ok:
                print("PARITY MISMATCH", got, want, file=sys.stderr)
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            cpu = cpu_baseline(args.cpu_seconds, min(args.cpu_sample, n), args.seed, args.cpu_dram_per_core)
        traffic = None
        traffic_note = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath))
                if t.get("flags_per_launch") == n and t.get("kernel_source_id") == kernel_source_id():
                    traffic = t.get("hbm_bytes_per_launch")
                    traffic_note = t.get("source")
            except Exception:
                traffic = None
        result = {
            "metric": METRIC, "value": round(value, 3), "unit": "Gflags/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "%.3g GiB uniform-random uint16 FLAG array (%d flags) per GPU, device-resident, %s"
                                   % (n * 2 / 2 ** 30, n,
                                      "K1 flagstat_count + K2 flagstat_finalize (store form) + RCCL all-reduce uint64[32]" if multi
                                      else ("K1 flagstat_count accumulating into the 32 counters (atomic epilogue, one launch)"
                                            if lib.FLAGSTATS_hip_get(b"epilogue") else "K1 flagstat_count + K2 flagstat_finalize")),
                       "flags_per_gpu": n, "global_flags": total_flags, "parallelism": "shard%d" % world,
                       "allreduce": (("overlapped" if state["overlap"] else "in-line") + calib_note) if multi else None,
                       "allreduce_impl": ar_impl,
                       "rccl_nranks": rccl_nranks,
                       "rccl_library": rccl_library if comm else None,
                       "per_rank_ms": per_rank_ms,
                       "per_rank_k1_alone_ms": per_rank_k1_ms,
                       "slowest_rank": (max(range(world), key=lambda r: per_rank_ms[r]) if per_rank_ms else None),
                       "allreduce_us": allreduce_us,
                       "kernel_variant": int(lib.FLAGSTATS_hip_get(b"variant")),
                       "grid_blocks": int(lib.FLAGSTATS_hip_get(b"grid")),
                       "kernel_source_id": kernel_source_id()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": 2 * n,
                         "event_ms_per_launch": round(ev_ms_per_step, 5),
                         "step_ms": quantiles(step_ms),
                         "sclk_mhz": sclk,
                         "read_probe_GBs": round(probe_gbs, 1) if probe_gbs else None,
                         "frac_of_read_probe": round(achieved / probe_gbs, 4) if probe_gbs else None},
            "cpu_baseline": cpu,
            "parity": parity,
            "counters_fail_qc_reads": int(got[25]),
        }
        if cpu:
            result["gpu_over_cpu_1thread"] = round(value / cpu["value"], 1) if cpu["value"] else None
        print(json.dumps(result), flush=True)

    if multi:
        dist.barrier()
        if comm:
            lib.FLAGSTATS_hip_comm_destroy(comm)
        dist.destroy_process_group()
        if comm_init_hung:
            sys.stdout.flush()
            os._exit(0)   # a helper thread is still inside ncclCommInitRank: do not wait for it at interpreter exit
    return result


if __name__ == "__main__":
    main()
