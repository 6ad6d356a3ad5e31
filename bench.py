#!/usr/bin/env python3
"""bench.py -- flagstat hot path on N MI355X of one node.

One "step" = one pass of the hot path (K1 flagstat_count + K2 flagstat_finalize,
plus the 32-counter all-reduce when N > 1) over this rank's device-resident FLAG
shard.  Workload = the configuration BASELINE.json's metric is quoted on:
8 GiB of uniform-random uint16 (2^32 flags) per GPU, generated on device by the
library's counter-based generator (data: synthetic).  N > 1 is weak scaling:
every rank holds its own 8 GiB shard (seed + rank), 64 GiB at N = 8
(BASELINE config 3), one RCCL all-reduce of int64[32] per step.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: "roofline" (HBM: algorithmic bytes = 2 B/flag over the event-timed
average step on the launch stream) and "cpu_baseline" (the reference's own
dispatcher kernel, oracle/_ref, timed on this host on a bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "Gflags/s + achieved HBM GB/s vs roofline, 8 GiB uint16, 1/2/4/8 MI355X"


def cpu_baseline(seconds: float, sample_flags: int, seed: int):
    """Time the reference's own kernel (what FLAGSTATS_get_function returns on this host,
    libflagstats.h:2976-3022) on a prefix of the rank-0 workload; 1 thread, then all cores."""
    import numpy as np

    import oracle

    a = oracle.generate(oracle.GEN_UNIFORM, seed, 0xFFFF, 0, sample_flags)
    ref = oracle.load_ref()
    lib = oracle.load_c()
    if ref is not None:
        kind = "reference"
        name = ref.ref_dispatch_name(min(sample_flags, 2 ** 30)).decode()
        p16 = ctypes.cast(a.ctypes.data, ctypes.POINTER(ctypes.c_uint16))

        def run(ptr, n):
            out = np.zeros(32, dtype=np.uint64)
            ref.ref_dispatch_x64(ptr, n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
            return out
    else:  # reference build absent on this box: time our C restatement instead
        kind = "port"
        name = "oracle_flagstat_hist_u16"
        lib = oracle.load_c()
        p16 = ctypes.cast(a.ctypes.data, ctypes.POINTER(ctypes.c_uint16))

        def run(ptr, n):
            out = np.zeros(32, dtype=np.uint64)
            lib.oracle_flagstat_hist_u16(ptr, n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
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

    # all cores: a larger sample (beyond the host's last-level caches), one contiguous shard and
    # private counters per thread; the pass loop runs inside C (ref_dispatch_repeat), ctypes drops
    # the GIL, so Python is not what is being timed
    cores = os.cpu_count() or 1
    big_flags = max(sample_flags, min(2 ** 30, 4 * 2 ** 20 * cores))
    per = big_flags // cores
    big = np.empty(per * cores, dtype=np.uint16)

    def fill(k):
        oracle.load_c().oracle_generate_u16(oracle.GEN_UNIFORM, seed, 0xFFFF, per * k, per,
                                            ctypes.cast(big.ctypes.data + 2 * per * k, ctypes.POINTER(ctypes.c_uint16)))

    def timed(reps):
        def worker(k):
            ptr = ctypes.cast(big.ctypes.data + 2 * per * k, ctypes.POINTER(ctypes.c_uint16))
            out = np.zeros(32, dtype=np.uint64)
            p64 = out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
            if ref is not None:
                ref.ref_dispatch_repeat(ptr, per, reps, p64)
            else:
                for _ in range(reps):
                    lib.oracle_flagstat_hist_u16(ptr, per, p64)
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return time.perf_counter() - t0

    ths = [threading.Thread(target=fill, args=(k,)) for k in range(cores)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    probe = timed(2)
    reps = max(2, min(100000, int(2 * 5.0 / max(probe, 1e-6))))   # aim at ~5 s
    dt_all = timed(reps)
    allc = per * cores * reps / dt_all / 1e9

    return {
        "value": round(one, 4), "unit": "Gflags/s", "cores": 1, "kind": kind, "kernel": name,
        "sample": "first %d flags (%.0f MiB) of the rank-0 workload, %d passes in %.1f s, 1 thread"
                  % (sample_flags, sample_flags * 2 / 2 ** 20, passes, dt),
        "scalar_exact_variant": exact,
        "all_cores": {"value": round(allc, 4), "unit": "Gflags/s", "cores": cores,
                      "sample": "%d flags (%.0f MiB) in %d contiguous shards x %d passes in %.1f s"
                                % (per * cores, per * cores * 2 / 2 ** 20, cores, reps, dt_all)},
    }, a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--flags-per-gpu", type=int, default=2 ** 32, help="default 2^32 flags = 8 GiB uint16")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=2 ** 27)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --flags-per-gpu is the TOTAL array, split into contiguous shards over the ranks "
                         "(default is weak scaling: every rank holds its own --flags-per-gpu shard)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; nccl = RCCL over xGMI (default)")
    ap.add_argument("--same-device", action="store_true",
                    help="TEST ONLY: put every rank on GPU 0 (use with --backend gloo) to exercise the multi-rank "
                         "launch contract on a single-GPU box; the number it prints is not a scaling result")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: wait for each step's all-reduce before the next step's K1 (default: the all-reduce "
                         "of step i overlaps K1 of step i+1)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 step (store + RCCL all-reduce) even at world size 1: exercises the multi-GPU "
                         "code path on a single-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    from libflagstats_amd import _lib, device
    from libflagstats_amd.dist import allreduce_counters

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

    if args.same_device:
        local_rank = 0
    ndev = torch.cuda.device_count()
    if ndev and local_rank >= ndev:
        local_rank %= ndev   # a launcher that isolates one visible GPU per rank (HIP_VISIBLE_DEVICES)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(local_rank), "FLAGSTATS_hip_init")
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    n = args.flags_per_gpu
    if args.strong:
        from libflagstats_amd.dist import shard_range
        b, e = shard_range(args.flags_per_gpu, rank, world)
        n = e - b
    flags = torch.empty(n, dtype=torch.int16, device=dev)           # this rank's shard, resident in HBM
    device.generate_torch(flags, device.GEN_UNIFORM, seed=args.seed + rank, mask=0xFFFF)
    counters = torch.zeros(32, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    overlap = multi and not args.no_overlap
    main_stream = torch.cuda.current_stream(dev)
    comm_stream = torch.cuda.Stream(device=dev) if overlap else None
    bufs = [counters, torch.zeros(32, dtype=torch.int64, device=dev)]
    reduced = [None, None]     # event: the all-reduce that last used bufs[k] has finished
    state = {"i": 0}

    def step():
        # N = 1: counters accumulate across steps (the ABI's += contract, as the reference's
        # bench accumulates across blocks, benchmark/flagstats.cpp:304,328-329), so a step is
        # exactly K1 + K2.  N > 1: a step is one whole query: count (K2 stores), all-reduce of the
        # 32 counters.  The all-reduce of query i runs on a side stream while K1 of query i+1 streams
        # its shard (two counter buffers), so the collective's latency is off the critical path;
        # every query's all-reduce still completes inside the timed region (drain() below).
        if not overlap:
            device.count_torch(flags, counters, store=multi)   # K1 + K2 on torch's current stream
            if multi:
                allreduce_counters(counters)      # the path's only exchange: 256 B over xGMI
            return
        k = state["i"] & 1
        state["i"] += 1
        buf = bufs[k]
        if reduced[k] is not None:
            main_stream.wait_event(reduced[k])    # K2 may overwrite buf only after its last all-reduce
        device.count_torch(flags, buf, store=True)
        counted = torch.cuda.Event()
        counted.record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(counted)
            allreduce_counters(buf)
            ev = torch.cuda.Event()
            ev.record(comm_stream)
        reduced[k] = ev

    def drain():
        if overlap:
            main_stream.wait_stream(comm_stream)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    if not multi:
        counters.zero_()
    barrier()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    drain()
    e1.record()
    barrier()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)

    if multi:
        tmax = torch.tensor([wall, ev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tmax[0]), float(tmax[1])

    ms_per_step = wall * 1e3 / args.steps
    total_flags = args.flags_per_gpu if args.strong else n * world
    value = total_flags * args.steps / wall / 1e9          # whole-job Gflags/s
    ev_ms_per_step = ev_ms / args.steps
    achieved = 2.0 * n / (ev_ms_per_step * 1e-3) / 1e9     # GB/s per GPU, algorithmic bytes

    result = None
    if rank == 0:
        last = bufs[(state["i"] - 1) & 1] if overlap else counters
        got = last.cpu().numpy().view(np.uint64)
        passes = 1 if multi else args.steps   # N = 1 accumulated `steps` identical passes
        assert not (got % np.uint64(passes)).any(), "accumulated counters are not a multiple of the step count"
        got = got // np.uint64(passes)
        cpu = None
        parity = "not checked"
        if args.cpu_seconds > 0 and world == 1:
            import oracle
            cpu, sample = cpu_baseline(args.cpu_seconds, min(args.cpu_sample, n), args.seed)
            # checker: the oracle on the same sample bytes vs the HIP path on the same prefix
            want = oracle.flagstat_mt(sample)
            pre = torch.zeros(32, dtype=torch.int64, device=dev)
            device.count_torch(flags[: sample.size], pre)
            torch.cuda.synchronize()
            ok = np.array_equal(pre.cpu().numpy().view(np.uint64), want)
            parity = "bit-exact vs oracle on the CPU-baseline sample" if ok else "MISMATCH vs oracle"
            if not ok:
                print("PARITY MISMATCH", pre.cpu().numpy().view(np.uint64), want, file=sys.stderr)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath))
                if t.get("flags_per_launch") == n:
                    traffic = t.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": METRIC, "value": round(value, 3), "unit": "Gflags/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "%.3g GiB uniform-random uint16 FLAG array (%d flags) per GPU, device-resident, "
                                   "K1 flagstat_count + K2 flagstat_finalize%s"
                                   % (n * 2 / 2 ** 30, n, " + RCCL all-reduce int64[32]" if world > 1 else ""),
                       "flags_per_gpu": n, "global_flags": total_flags, "parallelism": "shard%d" % world, "allreduce": ("overlapped" if overlap else "in-line") if multi else None,
                       "kernel_variant": int(lib.FLAGSTATS_hip_get(b"variant")),
                       "grid_blocks": int(lib.FLAGSTATS_hip_get(b"grid"))},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": 2 * n,
                         "event_ms_per_launch": round(ev_ms_per_step, 5)},
            "cpu_baseline": cpu,
            "parity": parity,
            "counters_fail_qc_reads": int(got[25]),
        }
        if cpu:
            result["gpu_over_cpu_1thread"] = round(value / cpu["value"], 1) if cpu["value"] else None
        print(json.dumps(result), flush=True)

    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
