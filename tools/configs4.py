#!/usr/bin/env python3
"""tools/configs4.py -- BASELINE.json configs[4]'s per-GPU shard at the size one box can build: ~15 000 genomes,
--both index sized to what host memory allows (target 1.26e9 markers = 15 000 x 3.45 Mbp at the survey's marker
density: ~81 GB of table, ~97 GB on the device), 150-bp reads, 125 M per launch (configs[4]'s 1 B reads over 8 GPUs).

    python tools/configs4.py [--genomes 15000] [--genome-len 3450000] [--reads 125000000] [--out gpurun_out/cfg4.json]

What it checks (run(); tests/test_gpu_configs.py::test_configs4_index_at_size asserts on the record):
  * properties that hold at any size: conservation of reads, idempotence, two unequal halves and the eight
    cq_shard_range shards add up to the whole counter by counter and leaf by leaf, the host-fed door (tight rows in
    page-locked memory, the whole shard in one query; word rows from pageable memory on a part) gives the device
    door's counters and rcount;
  * the oracle on a 100 k-read slice.  The oracle cannot hold 10^9 markers (~100 B per node), so it loads the
    SUB-INDEX the generator writes beside the full one: exactly the full index's markers whose h-mer occurs in
    the slice (either strand).  For those reads every lookup finds in the sub-index what it finds in the full one
    (tests/test_subindex.py proves the construction where the oracle holds both), so all counters must be equal
    and rcount must be the oracle's at the sub-index leaves' positions in the full index and zero elsewhere
    (the slice is classified alone for this check).
What it measures: generator / load stage times, peak host memory, table statistics (overflowed buckets, longest
chain), kernel ms per launch (HIP events inside the library), windows/s against configs[2]'s rate.
"""
from __future__ import annotations

import argparse
import json
import os
import resource
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BYTES_PER_MARKER_HOST_PEAK = 90      # round 4 (table laid out on the device): index files in /dev/shm (~15 B) + decoded keys, codes and
                                     # leaves + leaf refIDs + this tool's own rcount copies; measured 85.6 GB of peak RSS at 1.258e9 markers = 68 B
                                     # (round 3, host layout: 163 GB = 130 B, budgeted at 175)
MARKERS_PER_GENOME_BASE = 2 / 69 * 0.839   # two strands, one marker per 69 positions, minus block-straddling / shared-once losses


def mem_available_bytes() -> int:
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            return int(line.split()[1]) * 1024
    return 0


def disk_usage_free(d: str) -> int:
    try:
        return shutil.disk_usage(d).free
    except OSError:
        return 0


def scratch_dir() -> str:
    """Where the index files go: /dev/shm, unless the temporary directory has more room (a container with a tiny shm)."""
    shm, tmp = "/dev/shm", tempfile.gettempdir()
    if os.path.isdir(shm) and disk_usage_free(shm) >= disk_usage_free(tmp):
        return shm
    return tmp


def peak_rss_gb() -> float:
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def size_for_this_box(genomes: int, want_len: int, host_budget_bytes: float | None = None, shm_bytes: float | None = None):
    """Largest genome length <= want_len whose index this box can build (host memory is the limit, not HBM)."""
    avail = mem_available_bytes() if host_budget_bytes is None else host_budget_bytes
    if shm_bytes is not None:
        shm = shm_bytes
    else:
        shm = disk_usage_free(scratch_dir())
    budget = min(0.72 * avail, 260e9)                 # a one-GPU lease is capped at ~270 GiB of host memory
    markers = budget / BYTES_PER_MARKER_HOST_PEAK
    markers = min(markers, 0.8 * shm / 16)            # ~14.8 bytes of index file per marker
    L = int(markers / (genomes * MARKERS_PER_GENOME_BASE))
    return min(want_len, L), avail, shm


def run(genomes=15000, genome_len=3_450_000, n_reads=125_000_000, rl=150, slice_reads=100_000, log=print, workdir=None):
    import torch
    import cammiq_amd as cq
    from cammiq_amd import bigsynth
    import oracle_lib
    from util import hmers_of_reads

    rec = {"genomes": genomes, "genome_len": genome_len, "reads": n_reads, "read_len": rl, "stages_s": {}}
    t_all = time.time()

    def lap(name, t0):
        rec["stages_s"][name] = round(time.time() - t0, 2)
        log(f"[configs4] {name:34s} {time.time() - t0:8.1f} s   peak RSS {peak_rss_gb():6.1f} GB   t+{time.time() - t_all:6.0f} s")

    G, h = genomes, 26
    wdir = workdir or tempfile.mkdtemp(prefix="cammiq_cfg4_", dir=scratch_dir())
    pu, pd = os.path.join(wdir, "index_u.bin1"), os.path.join(wdir, "index_d.bin2")
    su, sd = os.path.join(wdir, "sub_u.bin1"), os.path.join(wdir, "sub_d.bin2")
    try:
        w = bigsynth.World(seed=2, n_genomes=G, genome_len=genome_len, k=26, h=h, lmax=50, pair_share=0.3)
        # the slice the oracle will see = the first reads of the batch; its h-mers select the sub-index
        t0 = time.time()
        sample, so = w.reads(seed=1000, n=slice_reads, length=rl)
        hm = hmers_of_reads(sample, slice_reads, rl, h)
        lap("slice reads + their h-mers", t0)
        t0 = time.time()
        os.environ.setdefault("CQS_TIMING", "1")
        nu, nd, ids_u, ids_d = w.write_index_with_sub(pu, pd, hm, su, sd)
        del hm
        lap("generate full index + sub-index", t0)
        rec.update(leaves_u=nu, leaves_d=nd, sub_leaves_u=len(ids_u), sub_leaves_d=len(ids_d),
                   index_file_GB=round(sum(os.path.getsize(p) for p in (pu, pu + ".aux", pd, pd + ".aux")) / 1e9, 2))
        log(f"[configs4] index: {nu} + {nd} = {nu + nd} markers, {rec['index_file_GB']} GB on disk; sub-index {len(ids_u)} + {len(ids_d)}")

        t0 = time.time()
        os.environ.setdefault("CAMMIQ_LOAD_TIMING", "1")
        ix = cq.Index(pu, pd, device=0)
        lap("cq_index_load (decode+layout+upload)", t0)
        info = ix.info_dict()
        rec["table"] = {"buckets": info["n_table_buckets"], "table_GB": round(info["n_table_buckets"] * 64 / 1e9, 2),
                        "device_GB": round(info["device_bytes"] / 1e9, 2), "keys": info["n_keys"],
                        "overflowed_buckets": info["n_overflowed"],
                        "overflowed_pct": round(100.0 * info["n_overflowed"] / max(1, info["n_table_buckets"]), 3),
                        "max_chain": info["max_chain"], "trie_nodes": info["n_trie_nodes"],
                        "minimizer_len": info.get("minimizer_len")}
        log(f"[configs4] table {rec['table']}")
        for p in (pu, pu + ".aux", pd, pd + ".aux"):
            os.unlink(p)                       # give the page cache back before the reads are generated

        # ---- the batch: n_reads x rl, packed once: word rows resident in HBM (the device door), tight rows in page-locked
        #      host memory (the host-fed door, what crosses the link), the first nh word rows also in pageable memory
        t0 = time.time()
        sw, sb = cq.stride_words(rl), cq.stride_bytes(rl)
        nh = min(n_reads, 6_000_000)
        packed = np.empty((nh, sw), np.uint32)
        lens = np.empty(nh, np.uint8)
        hp = cq.host_array(n_reads * sb, np.uint8).reshape(n_reads, sb)
        hl = cq.host_array(n_reads, np.uint8)
        dp = torch.empty((n_reads, sw), dtype=torch.int32, device="cuda")
        dl = torch.empty(n_reads, dtype=torch.uint8, device="cuda")
        chunk = 5_000_000
        buf = np.empty(min(chunk, n_reads) * rl, np.uint8)
        for c0 in range(0, n_reads, chunk):
            m = min(chunk, n_reads - c0)
            w.reads_into(buf, 1000, c0, m, rl)
            offs = np.arange(m + 1, dtype=np.uint64) * np.uint64(rl)
            pk, ln, sk = cq.pack_reads(buf[:m * rl], offs, h, sw)
            assert sk == 0
            _, _, tsk = cq.pack_reads_tight(buf[:m * rl], offs, h, sb, out=(hp[c0:c0 + m], hl[c0:c0 + m]))
            assert tsk == 0
            if c0 < nh:
                k = min(m, nh - c0)
                packed[c0:c0 + k] = pk[:k]
                lens[c0:c0 + k] = ln[:k]
            dp[c0:c0 + m].copy_(torch.from_numpy(pk.view(np.int32)))
            dl[c0:c0 + m].copy_(torch.from_numpy(ln))
            if c0 == 0:
                assert np.array_equal(buf[:slice_reads * rl], sample), "the slice is not the head of the batch"
        del buf
        torch.cuda.synchronize()
        lap("generate + pack + upload reads", t0)

        cw = ix.counter_words(G)
        nleaf = nu + nd
        stream = torch.cuda.current_stream().cuda_stream

        def device_query(lo, hi):
            ctr = torch.zeros(cw, dtype=torch.int64, device="cuda")
            rc = torch.zeros(nleaf, dtype=torch.int32, device="cuda")
            ix.query_device(cq.MODE_P, dp[lo:hi].data_ptr(), dl[lo:hi].data_ptr(), hi - lo, sw, rl, G, ctr.data_ptr(), rc.data_ptr(), stream)
            torch.cuda.synchronize()
            kms = ix.last_kernel_times()
            c = ctr.cpu().numpy().astype(np.uint64)
            r = rc.cpu().numpy().view(np.uint32)
            assert int(c[2 * G + 4]) == 0 and int(c[2 * G + 5]) == 0          # nskipped, flags
            return dict(cnt_u=c[:G + 1], cnt_d=c[G + 1:2 * G + 2], nundet=int(c[2 * G + 2]), nconf=int(c[2 * G + 3]),
                        rcount_u=r[:nu].copy(), rcount_d=r[nu:].copy(), nslow=int(c[2 * G + 6]), kernel_ms=kms)

        def add(a, b):
            out = {k: a[k] + b[k] for k in ("cnt_u", "cnt_d", "rcount_u", "rcount_d")}
            out["nundet"], out["nconf"] = a["nundet"] + b["nundet"], a["nconf"] + b["nconf"]
            return out

        def same(a, b):
            return all(np.array_equal(a[k], b[k]) for k in ("cnt_u", "cnt_d", "rcount_u", "rcount_d")) \
                and a["nundet"] == b["nundet"] and a["nconf"] == b["nconf"]

        checks = {}
        t0 = time.time()
        whole = device_query(0, n_reads)                                  # ONE launch of the full shard (also warms up)
        runs = [device_query(0, n_reads) for _ in range(3)]
        checks["idempotence"] = all(same(r_, whole) for r_ in runs)
        kms = [r_["kernel_ms"][0] for r_ in runs]
        rec["kernel_ms_runs"] = [round(x, 3) for x in kms]
        rec["kernel_ms"] = round(float(np.mean(kms)), 3)
        rec["slow_path_kernel_ms"] = round(float(np.mean([r_["kernel_ms"][1] for r_ in runs])), 4)
        rec["kernel_launch"] = ix.last_launch_info()
        rec["kernel_Mreads_s"] = round(n_reads / (rec["kernel_ms"] * 1e-3) / 1e6, 1)
        rec["kernel_Gwindows_s"] = round(n_reads * (rl - h + 1) / (rec["kernel_ms"] * 1e-3) / 1e9, 2)
        cu, cd = int(whole["cnt_u"].sum()), int(whole["cnt_d"].sum())
        base = whole["nundet"] + whole["nconf"] + cu
        checks["conservation"] = bool(base <= n_reads <= base + cd)
        checks["rcount_covers_counted_reads"] = bool(int(whole["rcount_u"].astype(np.uint64).sum()) +
                                                     int(whole["rcount_d"].astype(np.uint64).sum()) >= max(cu, cd // 2))
        checks["genomes_hit"] = int(np.count_nonzero(whole["cnt_u"]))
        rec["outcome"] = {"nundet": whole["nundet"], "nconf": whole["nconf"], "cnt_u_sum": cu, "cnt_d_sum": cd,
                          "slow_path_reads": whole["nslow"]}
        a, b = device_query(0, n_reads // 2 + 3), device_query(n_reads // 2 + 3, n_reads)
        checks["halves_add_up"] = same(add(a, b), whole)
        acc = None
        for p in range(8):
            lo, hi = cq.shard_range(n_reads, p, 8)
            part = device_query(lo, hi)
            acc = part if acc is None else add(acc, part)
        checks["eight_shards_add_up"] = same(acc, whole)
        lap("device queries + additivity checks", t0)
        t0 = time.time()
        # the host-fed door on the WHOLE shard: tight rows in page-locked memory -> H2D in chunks, widened and classified
        # while the next chunk arrives -> rcount back narrow and widened on the host (SURVEY 8(d)'s bracket)
        out = ix.counts_out(G, pinned=True)
        ix.query_packed_tight(hp[:1 << 16], hl[:1 << 16], rl, G, out=out)
        ts = []
        for _ in range(2):
            t1 = time.time()
            hq = ix.query_packed_tight(hp, hl, rl, G, out=out)
            ts.append(time.time() - t1)
        checks["host_fed_door_whole_shard_equals_device_door"] = same(hq, whole)
        rec["host_fed_ms_runs"] = [round(x * 1e3, 1) for x in ts]
        rec["host_fed_Mreads_s"] = round(n_reads / min(ts) / 1e6, 1)
        del out, hq
        hq = ix.query_packed(packed[:nh], lens[:nh], rl, G)               # word rows from pageable memory, a part of the batch
        checks["host_fed_door_equals_device_door"] = same(hq, device_query(0, nh))
        del hq
        lap("host-fed doors", t0)

        # ---- the oracle on the slice, against the sub-index
        t0 = time.time()
        oi = oracle_lib.OracleIndex(su, sd)
        ref = oi.query(sample, so, G, nthreads=min(16, os.cpu_count() or 1), variant="thread_local")
        got = device_query(0, slice_reads)                                # the slice alone, against the FULL index
        ok = all(np.array_equal(got[k], ref[k]) for k in ("cnt_u", "cnt_d")) and got["nundet"] == ref["nundet"] \
            and got["nconf"] == ref["nconf"]
        for k, ids, n_full in (("rcount_u", ids_u, nu), ("rcount_d", ids_d, nd)):
            exp = np.zeros(n_full, np.uint32)
            exp[ids.astype(np.int64)] = ref[k]
            ok = ok and np.array_equal(got[k], exp)
        checks["slice_equals_oracle_on_subindex"] = bool(ok)
        hq = ix.query(sample, so, G)                                      # and through the ASCII door
        checks["slice_ascii_door_equals_device_door"] = same(hq, got)
        rec["oracle_slice"] = {"reads": slice_reads, "branch": ref["branch"], "nundet": ref["nundet"], "nconf": ref["nconf"]}
        lap("oracle on the slice (sub-index)", t0)
        rec["checks"] = checks
        rec["peak_host_RSS_GB"] = round(peak_rss_gb(), 1)
        rec["total_s"] = round(time.time() - t_all, 1)
        ix.close()
        return rec
    finally:
        if workdir is None:
            shutil.rmtree(wdir, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=15000)
    ap.add_argument("--genome-len", type=int, default=3_450_000)
    ap.add_argument("--reads", type=int, default=125_000_000, help="reads of the shard, ONE launch (configs[4]: 1 B reads over 8 GPUs)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cfg4.json"))
    ap.add_argument("--fit", action="store_true", help="shrink --genome-len to what this box's host memory can build")
    a = ap.parse_args()
    L = a.genome_len
    if a.fit:
        L, avail, shm = size_for_this_box(a.genomes, a.genome_len)
        print(f"[configs4] MemAvailable {avail / 1e9:.0f} GB, /dev/shm free {shm / 1e9:.0f} GB -> genome length {L}", flush=True)
    rec = run(a.genomes, L, a.reads, a.read_len, log=lambda s: print(s, flush=True))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec))
    bad = [k for k, v in rec["checks"].items() if v is False]
    if bad:
        raise SystemExit(f"FAILED checks: {bad}")


if __name__ == "__main__":
    main()
