#!/usr/bin/env python3
"""tools/multi_leg.py -- the one-process multi-GPU shape (cq_multi_load / cq_multi_query_packed_tight: one host thread
per GPU inside the library, the library's own RCCL reduction) on min(2, visible GPUs) devices, checked against the
single-device counts of the same reads.  bench.py runs this as a CHILD process with a time limit after its own
measurement (N = 1 only): the first contact of RCCL with two ranks must not be able to take the headline line down
with it.  Prints one JSON object.

    python tools/multi_leg.py PATH_U PATH_D|- N_GENOMES GENOME_LEN BOTH(0|1) N_READS READ_LEN N_DEVICES
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    pu, pd, G, glen, both, n, rl, ndev = sys.argv[1:9]
    pd = None if pd == "-" else pd
    G, glen, both, n, rl, ndev = int(G), int(glen), int(both), int(n), int(rl), int(ndev)
    import torch  # noqa: F401  (torch first: its bundled HIP runtime must come up before /opt/rocm's)
    import cammiq_amd as cq
    from cammiq_amd import bigsynth
    w = bigsynth.World(seed=2, n_genomes=G, genome_len=glen, k=26, h=26, lmax=50, pair_share=0.3 if both else 0.0)
    sb = cq.stride_bytes(rl)
    hp = cq.host_array(n * sb, np.uint8).reshape(n, sb)
    hl = cq.host_array(n, np.uint8)
    chunk = 5_000_000
    buf = np.empty(min(chunk, n) * rl, np.uint8)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        w.reads_into(buf, 1000, c0, m, rl)
        _, _, sk = cq.pack_reads_tight(buf[:m * rl], np.arange(m + 1, dtype=np.uint64) * np.uint64(rl), 26, sb,
                                       out=(hp[c0:c0 + m], hl[c0:c0 + m]))
        assert sk == 0
    devs = list(range(ndev))
    ix = cq.Index(pu, pd, device=devs[0])
    single = ix.query_packed_tight(hp, hl, rl, G, out=ix.counts_out(G, pinned=True))
    single = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in single.items()}
    ix.close()
    t0 = time.perf_counter()
    mm = cq.Multi(pu, pd, devs)
    t_load = time.perf_counter() - t0
    out = mm.shards[0].counts_out(G, pinned=True)
    mm.query_packed_tight(hp[:1 << 16], hl[:1 << 16], rl, G, out=out)
    def equal(mq):
        return all(np.array_equal(mq[k], single[k]) for k in ("cnt_u", "cnt_d", "rcount_u", "rcount_d")) \
            and mq["nundet"] == single["nundet"] and mq["nconf"] == single["nconf"]

    # The exchange step both ways (the library reads CAMMIQ_MULTI_ALLREDUCE per query): the default reduce to the device
    # the host reads, then the all-reduce.  With one device both are the one-rank communicator; with two or more this is
    # the A/B nobody could take on a one-GPU lease -- taken automatically by the first run that sees >= 2 GPUs.
    same, by_kind = True, {}
    for kind, flag in (("reduce_to_root", "0"), ("all_reduce", "1")):
        os.environ["CAMMIQ_MULTI_ALLREDUCE"] = flag
        tk = []
        for _ in range(2):
            t0 = time.perf_counter()
            mq = mm.query_packed_tight(hp, hl, rl, G, out=out)
            tk.append(time.perf_counter() - t0)
            same = same and equal(mq)
        by_kind[kind] = {"ms": round(min(tk) * 1e3, 3), "runs_ms": [round(x * 1e3, 3) for x in tk]}
    os.environ.pop("CAMMIQ_MULTI_ALLREDUCE", None)
    ts = [by_kind["reduce_to_root"]["ms"] * 1e-3]
    mm.close()
    print(json.dumps({"devices": devs, "reads": n, "Mreads_s": round(n / min(ts) / 1e6, 2), "ms": round(min(ts) * 1e3, 3),
                      "equals_single_device": bool(same), "load_s": round(t_load, 2), "exchange_ab": by_kind,
                      "what": "cq_multi_query_packed_tight in a child process: reads sharded over the devices by "
                              "cq_shard_range, one host thread per device, RCCL reduce of counter block + rcount to the "
                              "device the host reads, inside libcammiq_hip.so"}), flush=True)
    return 0 if same else 3


if __name__ == "__main__":
    sys.exit(main())
