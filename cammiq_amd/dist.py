"""CPU rehearsal of the multi-GPU exchange step (tests and bench.py's one-GPU rehearsal mode only).

The product's multi-GPU path is C++ behind the C ABI (include/cammiq_hip.h, cq_api.cpp): reads
are sharded with cq_shard_range, every GPU classifies its shard against its own replica of the
index, and the query ends with ONE RCCL all-reduce(sum) of the counter block and of rcount --
cq_counts_allreduce (one process per GPU, bench.py) or inside cq_multi_query (one process, one
host thread per GPU, `cammiq --gpus N`).  RCCL needs one GPU per rank, so a box with a single
GPU (or none) cannot run that collective with more than one rank; this module does the same
reduction over torch.distributed's gloo backend so that everything AROUND the collective --
the sharding rule, the counter-block layout, the uint32 wrap of rcount, the flags word -- is
exercised with world_size 2 on the CPU (tests/test_dist.py).

    counter block  int64 [2*(G+1)+8]   cnt_u | cnt_d | nundet nconf nskipped flags nslow ...
                                       (every word is a sum; flags != 0 means lost pair increments)
    rcount         int32 [n_u + n_d]   per-leaf counts, u leaves first (two's-complement add
                                       == the reference's uint32 add, bit for bit)
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_reads: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of reads for `rank` -- the library's own rule (cq_shard_range)."""
    from . import binding
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return binding.shard_range(n_reads, rank, world)


def allreduce_counts(counters, rcount=None, group=None, async_op=False):
    """gloo stand-in for cq_counts_allreduce: sum the counter block (and rcount) over all ranks, in place."""
    import torch.distributed as tdist
    if not tdist.is_initialized() or tdist.get_world_size(group) == 1:
        return []
    works = [tdist.all_reduce(counters, op=tdist.ReduceOp.SUM, group=group, async_op=async_op)]
    if rcount is not None and rcount.numel():
        works.append(tdist.all_reduce(rcount, op=tdist.ReduceOp.SUM, group=group, async_op=async_op))
    return works if async_op else []
