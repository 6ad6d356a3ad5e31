"""Multi-GPU flagstat: shard the FLAG array, count locally, one all-reduce of 32 counters.

The reference has no multi-device path at all ("none").  Every
flag is independent and the result is a sum of integers, so the array is split
into contiguous ranges, one per rank (one process per GPU), each rank runs K1+K2
on its shard, and the only exchange step is a single all-reduce of the
``int64[32]`` counters -- RCCL over xGMI on GPUs (``backend="nccl"``), gloo in the
CPU tests.  256 bytes: latency-bound, bit-exact (integer sum, any order).
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of rank's contiguous shard; the remainder goes to the last rank."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    per = n // world
    begin = per * rank
    end = n if rank == world - 1 else begin + per
    return begin, end


def allreduce_counters(counters, group=None):
    """Sum a 32-counter tensor over all ranks in place (one collective) and return it.

    ``counters``: torch int64[32], on the GPU for nccl(RCCL) or on the CPU for gloo.
    uint64 counters travel as int64 bit patterns; two's-complement addition is the
    same operation, so the sum is exact modulo 2^64.
    """
    import torch
    import torch.distributed as dist

    assert counters.dtype == torch.int64 and counters.numel() == 32
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters


def sharded_flagstat(count_shard: Callable[[int, int], "np.ndarray"], n_total: int, rank: int, world: int,
                     device=None, group=None) -> np.ndarray:
    """Counters of the whole array from this rank's point of view.

    ``count_shard(begin, end)`` returns the uint64[32] counters of flags
    [begin, end) (on a GPU rank: ``DeviceFlags.count`` / ``count_torch``).
    """
    import torch

    begin, end = shard_range(n_total, rank, world)
    local = np.asarray(count_shard(begin, end), dtype=np.uint64)
    t = torch.from_numpy(local.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    allreduce_counters(t, group)
    return t.cpu().numpy().view(np.uint64)
