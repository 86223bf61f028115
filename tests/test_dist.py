"""N > 1 path on CPU: world_size-2 gloo.  Each rank takes the shard the LIBRARY's rule gives it
(cq_shard_range through the C ABI), classifies it (the oracle stands in for the GPU kernel here --
this is a test), lays the result out as the device counter block, and the blocks are summed like
cq_counts_allreduce sums them (gloo instead of RCCL: RCCL needs one GPU per rank).  The result must
equal the single-rank result exactly, including the uint32 wrap of rcount and a flags word set on
both ranks.  The RCCL path itself runs in tests/test_multi_gpu.py on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as tdist
    from cammiq_amd import dist as cqdist, synth
    import oracle_lib
    from util import golden
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    g = golden("f_deep")
    G = g["G"]
    lo, hi = cqdist.shard_range(len(g["reads"]), rank, world)
    ix = oracle_lib.OracleIndex(g["pu"], g["pd"])
    b, o = synth.concat_reads(g["reads"][lo:hi])
    r = ix.query(b, o, G)
    ctr = torch.zeros(2 * (G + 1) + 8, dtype=torch.int64)
    ctr[:G + 1] = torch.from_numpy(r["cnt_u"].astype(np.int64))
    ctr[G + 1:2 * G + 2] = torch.from_numpy(r["cnt_d"].astype(np.int64))
    ctr[2 * G + 2] = r["nundet"]
    ctr[2 * G + 3] = r["nconf"]
    ctr[2 * G + 5] = 1       # flags word set on BOTH ranks: the sum must still read "set" (!= 0)
    rc = torch.from_numpy(np.concatenate([r["rcount_u"], r["rcount_d"]]).view(np.int32).copy())
    if rank == 1:
        rc[0] += -5          # wrap-around check: int32 add must behave like the uint32 add
    if rank == 0:
        rc[0] += 5
    cqdist.allreduce_counts(ctr, rc)
    np.save(os.path.join(out_dir, f"ctr{rank}.npy"), ctr.numpy())
    np.save(os.path.join(out_dir, f"rc{rank}.npy"), rc.numpy())
    tdist.destroy_process_group()


def test_two_rank_allreduce_equals_single_rank(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from util import golden
    g = golden("f_deep")
    G = g["G"]
    e = g["exp"]["p"]
    for rank in range(world):
        ctr = np.load(tmp_path / f"ctr{rank}.npy")
        rc = np.load(tmp_path / f"rc{rank}.npy").view(np.uint32)
        assert list(ctr[:G + 1]) == e["cnt_u"] and list(ctr[G + 1:2 * G + 2]) == e["cnt_d"]
        assert ctr[2 * G + 2] == e["nundet"] and ctr[2 * G + 3] == e["nconf"]
        assert ctr[2 * G + 5] != 0
        assert list(rc) == e["rcount_u"] + e["rcount_d"]


def test_shard_ranges_partition_the_reads():
    from cammiq_amd.dist import shard_range
    for n in (0, 1, 7, 10_000_001, 2**63 + 12345):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
