#!/usr/bin/env python3
"""tools/hostfed_ab.py -- A/B of the host-fed bracket (cq_query_packed_tight: SURVEY 8(d)'s bracket) under the
library's per-query tuning knobs, ONE process, one index, one batch of tight rows in page-locked memory; every
variant's counts must equal the first one's.  Variants are interleaved (A B C A B C ...) so that drift of the box
shows up as spread inside a variant, not as a difference between variants.

    python tools/hostfed_ab.py [--config 2] [--reads N] [--read-len L] [--rounds 4] name:ENV=V,ENV=V ...

e.g.  base:  narrow_off:CAMMIQ_RCOUNT_NARROW=0  lens_copy:CAMMIQ_LENS_FILL=0  tail_off:CAMMIQ_CHUNK_TAIL=0
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=1000)
    ap.add_argument("--genome-len", type=int, default=3_450_000)
    ap.add_argument("--unique", action="store_true")
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--trace", default=None, help="after the A/B: one more query of the FIRST variant with CAMMIQ_PIPE_TRACE=<this file> "
                                                  "(the library's own event timeline) and a summary of it on stdout")
    ap.add_argument("variants", nargs="*", default=["base:"])
    a = ap.parse_args()
    import torch  # noqa: F401
    import cammiq_amd as cq
    from cammiq_amd import bigsynth
    both = not a.unique
    G, n, rl = a.genomes, a.reads, a.read_len
    w = bigsynth.World(seed=2, n_genomes=G, genome_len=a.genome_len, k=26, h=26, lmax=50, pair_share=0.3 if both else 0.0)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    wdir = tempfile.mkdtemp(prefix="cammiq_ab_", dir=shm)
    try:
        pu = os.path.join(wdir, "index_u.bin1")
        pd = os.path.join(wdir, "index_d.bin2") if both else None
        w.write_index(pu, pd)
        ix = cq.Index(pu, pd, device=0)
    finally:
        import shutil
        shutil.rmtree(wdir, ignore_errors=True)
    sb = cq.stride_bytes(rl)
    hp = cq.host_array(n * sb, np.uint8).reshape(n, sb)
    hl = cq.host_array(n, np.uint8)
    chunk = 5_000_000
    buf = np.empty(min(chunk, n) * rl, np.uint8)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        w.reads_into(buf, 1000, c0, m, rl)
        cq.pack_reads_tight(buf[:m * rl], np.arange(m + 1, dtype=np.uint64) * np.uint64(rl), 26, sb, out=(hp[c0:c0 + m], hl[c0:c0 + m]))
    out = ix.counts_out(G, pinned=True)
    variants = []
    for v in a.variants:
        name, _, envs = v.partition(":")
        variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
    knobs = sorted({k for _, e in variants for k in e})
    times = {name: [] for name, _ in variants}
    first = None
    for rnd in range(a.rounds + 1):          # round 0 warms every variant's buffers up and is not counted
        for name, env in variants:
            for k in knobs:
                os.environ.pop(k, None)
            os.environ.update(env)
            t0 = time.perf_counter()
            r = ix.query_packed_tight(hp, hl, rl, G, out=out)
            dt = time.perf_counter() - t0
            sig = (int(r["cnt_u"].sum()), int(r["cnt_d"].sum()), r["nundet"], r["nconf"],
                   int(r["rcount_u"].astype(np.uint64).sum()), int(r["rcount_d"].astype(np.uint64).sum()),
                   int((r["rcount_u"].astype(np.uint64) * (np.arange(len(r["rcount_u"]), dtype=np.uint64) % np.uint64(1009))).sum()))
            if first is None:
                first = sig
            assert sig == first, f"variant {name} changed the counts: {sig} vs {first}"
            if rnd:
                times[name].append(dt * 1e3)
    for name, env in variants:
        ts = times[name]
        print(json.dumps({"variant": name, "env": env, "ms_min": round(min(ts), 3), "ms_median": round(sorted(ts)[len(ts) // 2], 3),
                          "runs_ms": [round(x, 3) for x in ts], "Mreads_s_best": round(n / min(ts) / 1e3, 1)}), flush=True)


    if a.trace:
        name, env = variants[0]
        for k in knobs:
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["CAMMIQ_PIPE_TRACE"] = a.trace
        t0 = time.perf_counter()
        ix.query_packed_tight(hp, hl, rl, G, out=out)
        dt = time.perf_counter() - t0
        os.environ.pop("CAMMIQ_PIPE_TRACE")
        print(json.dumps({"traced_query_ms": round(dt * 1e3, 3), "variant": name}), flush=True)
        sys.stdout.flush()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from pipe_trace import summarise
        summarise(a.trace)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
