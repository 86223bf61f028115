"""Multi-GPU glue for the classify path: one process per GPU, reads sharded, index replicated.

The reference has no distributed path at all; its only parallel axis
is the OpenMP loop over reads (/root/reference/src/query.cpp:664-665).  Every output of the
path is a commutative integer sum, so N ranks classify disjoint read ranges and ONE
all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests) per FASTQ rebuilds exactly the
counters a single rank would have produced:

    counter block  int64 [2*(G+1)+8]   cnt_u | cnt_d | nundet nconf nskipped flags nslow ...
    rcount         int32 [n_u + n_d]   per-leaf counts, u leaves first (two's-complement add
                                       == the reference's uint32 add, bit for bit)
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_reads: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of reads for `rank`: [n*p/P, n*(p+1)/P)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return n_reads * rank // world, n_reads * (rank + 1) // world


def allreduce_counts(counters, rcount=None, group=None, async_op=False):
    """Sum the per-rank counter block (and per-leaf rcount) over all ranks, in place.

    With ``async_op=True`` returns the list of work handles (``.wait()`` them before touching the
    tensors again): the collective then runs beside the next batch's classify kernel."""
    import torch.distributed as tdist
    if not tdist.is_initialized() or tdist.get_world_size(group) == 1:
        return []
    works = [tdist.all_reduce(counters, op=tdist.ReduceOp.SUM, group=group, async_op=async_op)]
    if rcount is not None and rcount.numel():
        works.append(tdist.all_reduce(rcount, op=tdist.ReduceOp.SUM, group=group, async_op=async_op))
    return works if async_op else []
