"""CPU, world_size 2, gloo: the multi-GPU path = shard + one all-reduce of 32 counters.
Per-rank counting is injected (the oracle stands in for the GPU kernel, which cannot run
here); what is under test is libflagstats_amd.dist."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from libflagstats_amd.dist import allreduce_counters, shard_range, sharded_flagstat

        def count_shard(b, e):  # this rank regenerates only its own shard
            return oracle.flagstat_generated(oracle.GEN_UNIFORM, seed, 0xFFFF, b, e - b, threads=1)

        total = sharded_flagstat(count_shard, n, rank, world)
        # the raw collective on counters near 2^63 (uint64 carried as int64)
        big = torch.full((32,), (2 ** 62) + rank, dtype=torch.int64)
        allreduce_counters(big)
        q.put((rank, total.tolist(), shard_range(n, rank, world), int(big[0].item())))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_allreduce():
    import oracle
    n, seed, world = 1_000_003, 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, seed, 0xFFFF, 0, n, threads=2).tolist()
    spans = {}
    for rank, total, span, big in res:
        assert total == want            # every rank holds the whole-array counters
        spans[rank] = span
        assert big == ((2 ** 62) * 2 + 1) - 2 ** 64   # wraps exactly like uint64 addition
    assert spans[0] == (0, n // 2) and spans[1] == (n // 2, n)
    assert np.uint64(2 ** 63 + 1) == np.array([big], dtype=np.int64).view(np.uint64)[0]


def test_eight_rank_shard_and_allreduce():
    """The north_star's world size, rehearsed on the CPU: eight ranks, an array length that is no multiple of eight (the remainder
    goes to the last rank), every rank ends up with the whole array's counters and the shards tile [0, n) exactly."""
    import oracle
    n, seed, world = 8_000_005, 23, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = oracle.flagstat_generated(oracle.GEN_UNIFORM, seed, 0xFFFF, 0, n, threads=2).tolist()
    spans = {}
    for rank, total, span, big in res:
        assert total == want
        spans[rank] = tuple(span)
        assert big == (((2 ** 62) * 8 + sum(range(8))) % 2 ** 64)   # eight times 2^62 wraps to 28 exactly like uint64 addition
    per = n // world
    assert [spans[r] for r in range(world)] == [(per * r, n if r == world - 1 else per * (r + 1)) for r in range(world)]
