#!/usr/bin/env python3
"""bench.py -- Mreads/s of the CAMMiQ classify hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (what FqReader::query64mt_p does for the reads of one
FASTQ, /root/reference/src/query.cpp:650-889) over one batch of synthetic reads that is already
resident in HBM: the classify kernel(s), accumulating into the device counters like the reference
accumulates into FqReader state.  The K timed steps are the K batches of one query; for N > 1 the
query ends with its one real exchange step, the RCCL all-reduce of the count vectors and of rcount
(BASELINE.json north_star: "count vectors all-reduced ... before the host hands the count matrix to
the ILP"), INSIDE the timed bracket.  --allreduce-every-step instead treats every batch as a query
of its own (reset, classify, all-reduce; the collective of step i is issued asynchronously and
overlaps the kernel of step i+1).  Index load / layout and FASTQ parsing are outside the bracket,
exactly like the reference's own `Time for query` line (query.cpp:459,645-647).

Workload at N = 1 = BASELINE.json configs[1]: 500 synthetic bacterial-size genomes, --unique
index (h = k = 26), 10 M x 100 bp reads.  N > 1 is weak scaling: the index is replicated and
every rank classifies its own 10 M reads (configs[3] shape).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Mreads/sec classified (100 bp, L=26); per-genome hit counts bit-exact"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_read(rl: int, h: int, tables: int) -> int:
    """B = ceil(rl/4) + 2*(rl-h+1)*T*16 (hits omitted -> lower bound)."""
    return (rl + 3) // 4 + 2 * (rl - h + 1) * tables * 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genomes", type=int, default=500)
    ap.add_argument("--genome-len", type=int, default=3_450_000)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--frac-deep", type=float, default=0.07,
                    help="fraction of markers longer than k (trie depth > 0); 0.07 is what the survey measured")
    ap.add_argument("--both", action="store_true", help="unique + doubly-unique index (configs[2] shape)")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="reads timed through the CPU oracle (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU oracle leg and its parity gate (used under rocprofv3 so that every "
                         "classify launch in the trace is a full-size timed step)")
    ap.add_argument("--allreduce-every-step", action="store_true",
                    help="N > 1: reset + all-reduce the counters every step (every batch its own query) instead of "
                         "once per run")
    ap.add_argument("--host-api", action="store_true",
                    help="also time cq_query on host ASCII reads (PCIe-inclusive rate; never `value`)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 through torch.distributed.run (see the docstring)")
        args.gpus = world

    import torch
    import cammiq_amd as cq
    from cammiq_amd import bigsynth, dist as cqdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the classify path has no CPU fallback")
    # Rehearsal knob for a one-GPU box: CAMMIQ_BENCH_REHEARSAL=1 maps every rank to cuda:0 and uses
    # gloo, so that the N > 1 code path (sharding, shared index files, all-reduce, max over ranks) can
    # be exercised without an 8-GPU node.  Never set by the driver; numbers from it mean nothing.
    rehearsal = os.environ.get("CAMMIQ_BENCH_REHEARSAL") == "1"
    dev = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as tdist
        if rehearsal:
            tdist.init_process_group("gloo")
        else:
            tdist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    h = k = 26
    G = args.genomes
    tables = 2 if args.both else 1
    t_setup = time.time()
    # one copy of the index files per node: local rank 0 writes, the others wait
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    wdir = os.path.join(shm, f"cammiq_bench_{os.environ.get('MASTER_PORT', os.getpid())}")
    try:
        w = bigsynth.World(seed=2, n_genomes=G, genome_len=args.genome_len, k=k, h=h, lmax=50,
                           frac_deep=args.frac_deep, pair_share=0.3 if args.both else 0.0)
        pu = os.path.join(wdir, "index_u.bin1")
        pd = os.path.join(wdir, "index_d.bin2") if args.both else None
        if local_rank == 0:
            os.makedirs(wdir, exist_ok=True)
            nu, nd = w.write_index(pu, pd)
            with open(os.path.join(wdir, "leaves.txt"), "w") as f:
                f.write(f"{nu} {nd}\n")
        if world > 1:
            tdist.barrier()
        nu, nd = (int(x) for x in open(os.path.join(wdir, "leaves.txt")).read().split())
        t_gen = time.time() - t_setup
        t0 = time.time()
        ix = cq.Index(pu, pd, device=dev)
        t_load = time.time() - t0
        info = ix.info_dict()

        n = args.reads
        bases, offs = w.reads(seed=1000 + rank, n=n, length=args.read_len)
        t0 = time.time()
        packed, lens, skipped = cq.pack_reads(bases, offs, h)
        t_pack = time.time() - t0
        sw = packed.shape[1]
        d_packed = torch.from_numpy(packed.view(np.int32)).cuda()
        d_lens = torch.from_numpy(lens).cuda()
        # two sets of counters: with --allreduce-every-step the all-reduce of step i overlaps the classify
        # kernel of step i+1 (independent batches, like consecutive FASTQ files of one run)
        ctrs = [torch.zeros(ix.counter_words(G), dtype=torch.int64, device="cuda") for _ in range(2)]
        rcs = [torch.zeros(max(nu + nd, 1), dtype=torch.int32, device="cuda") for _ in range(2)]
        pending = [None, None]
        stream = torch.cuda.current_stream().cuda_stream
        step_no = [0]
        per_step = args.allreduce_every_step and world > 1

        def step():
            b = (step_no[0] & 1) if per_step else 0
            step_no[0] += 1
            if per_step:
                if pending[b] is not None:
                    for wk in pending[b]:
                        wk.wait()             # the collective that last used this buffer pair
                    pending[b] = None
                ctrs[b].zero_()               # resetCounters (query.cpp:1820-1840)
                rcs[b].zero_()
            ix.query_device(cq.MODE_P, d_packed.data_ptr(), d_lens.data_ptr(), n, sw, args.read_len, G,
                            ctrs[b].data_ptr(), rcs[b].data_ptr(), stream)   # accumulates
            if per_step:
                pending[b] = cqdist.allreduce_counts(ctrs[b], rcs[b], async_op=True)
            return b

        def finish_query():
            """End of a query: the one exchange step (N > 1, unless every step already had its own)."""
            if world > 1 and not per_step:
                cqdist.allreduce_counts(ctrs[0], rcs[0])

        def fence():
            for b in (0, 1):
                if pending[b] is not None:
                    for wk in pending[b]:
                        wk.wait()
                    pending[b] = None
            if world > 1:
                tdist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            step()
        finish_query()                        # also warms RCCL up
        fence()
        for b in (0, 1):                      # resetCounters before the timed query
            ctrs[b].zero_()
            rcs[b].zero_()
        fence()
        kms = []
        last = 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step()
            kms.append(ix.last_kernel_ms())   # HIP events on the launch stream, recorded inside the library
        finish_query()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
            dt = float(tmax.item())
        ctr = ctrs[last]
        counted_steps = 1 if per_step else args.steps   # how many batches the final counters hold

        # ---- sanity on the last step's counters: every read lands in exactly one outcome
        c = ctr.cpu().numpy().astype(np.uint64)
        tot_reads = n * world
        held_reads = tot_reads * counted_steps
        nundet, nconf, nskip, nslow = (int(c[2 * (G + 1) + i]) for i in (0, 1, 2, 4))
        cnt_u_sum = int(c[:G + 1].sum())
        if not args.both and not os.environ.get("CAMMIQ_LIB"):
            assert cnt_u_sum + nundet + nconf + nskip == held_reads, "conservation of reads violated"

        result = None
        if rank == 0:
            value = tot_reads * args.steps / dt / 1e6
            B = algorithmic_bytes_per_read(args.read_len, h, tables)
            k_ms = float(np.mean(kms))
            achieved = n * B / (k_ms * 1e-3) / 1e9
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            default_workload = (G == 500 and n == 10_000_000 and args.read_len == 100 and not args.both
                                and args.genome_len == 3_450_000)
            if default_workload and os.path.exists(tp):   # the PMC passes were made on exactly this workload
                try:
                    traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            result = {
                "metric": METRIC, "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
                "data": "synthetic",
                "config": {"workload": ("configs[2]-shape: " if args.both else "configs[1]: ") +
                           f"{G} synthetic genomes x {args.genome_len} bp, "
                           f"{'--both' if args.both else '--unique'} index h=k=26 ({nu}+{nd} leaves), "
                           f"{n} x {args.read_len} bp reads per GPU per step (1% subst. errors, 10% off-database)",
                           "reads_per_gpu": n, "read_len": args.read_len, "hash_len": h, "n_genomes": G,
                           "leaves_u": nu, "leaves_d": nd, "table_GB": round(info["n_table_buckets"] * 64 / 1e9, 3),
                           "index_device_GB": round(info["device_bytes"] / 1e9, 3),
                           "table_buckets_overflowed": info["n_overflowed"], "table_max_chain": info["max_chain"],
                           "parallelism": f"reads sharded x{world}, index replicated" +
                                          ((", RCCL all-reduce of counts + rcount per step, overlapped with the next step's kernel"
                                            if per_step else
                                            ", one RCCL all-reduce of counts + rcount at the end of the query, inside the timed bracket")
                                           if world > 1 else "")},
                "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                             "kernel": "classify_kernel<8,16,false>", "kernel_ms": round(k_ms, 4),
                             "algorithmic_bytes_per_read": B},
                "kernel_Mreads_s": round(n / (k_ms * 1e-3) / 1e6, 3),
                "outcome": {"reads": held_reads, "nundet": nundet, "nconf": nconf, "nskipped": nskip,
                            "slow_path_reads": nslow, "cnt_u_sum": cnt_u_sum},
                "setup_s": {"generate": round(t_gen, 2), "index_load_layout_upload": round(t_load, 2),
                            "pack_reads": round(t_pack, 2)},
            }

        # ---- PCIe-inclusive rate of the host-buffer API (never `value`): ASCII reads in host
        #      memory -> pack -> H2D -> kernel -> D2H, chunks pipelined inside cq_query
        if rank == 0 and world == 1 and args.host_api:
            ix.query(bases[:args.read_len * 1000], offs[:1001], G)          # warm the staging buffers
            t0 = time.perf_counter()
            hq = ix.query(bases, offs, G)
            th = time.perf_counter() - t0
            result["host_api"] = {"Mreads_s": round(n / th / 1e6, 2), "seconds": round(th, 4),
                                  "what": "cq_query on ASCII reads in pageable host memory, counters back on the host "
                                          "(pack + H2D + kernel + D2H, 2 M-read chunks double-buffered)"}
            assert int(hq["cnt_u"].sum()) * counted_steps == cnt_u_sum and hq["nundet"] * counted_steps == nundet, \
                "host API disagrees with device API"

        # ---- CPU baseline (rank 0, N = 1 only): the oracle, a restatement of query64mt_p
        if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_sample > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            ns = min(args.cpu_sample, n)
            cores = min(os.cpu_count() or 1, 16)   # a one-GPU box's CPU share is 16 cores
            oi = oracle_lib.OracleIndex(pu, pd)
            sb, so = bases[:ns * args.read_len], offs[:ns + 1]
            t0 = time.perf_counter()
            ref = oi.query(sb, so, G, mode=0, nthreads=cores)
            tc = time.perf_counter() - t0
            t0 = time.perf_counter()
            fair = oi.query(sb, so, G, mode=0, nthreads=-cores)      # same work, atomic counters instead of the lock
            tf = time.perf_counter() - t0
            assert all(np.array_equal(fair[kk], ref[kk]) for kk in ("cnt_u", "cnt_d", "rcount_u", "rcount_d"))
            t0 = time.perf_counter()
            ns1 = max(ns // 8, 1)
            oi.query(bases[:ns1 * args.read_len], offs[:ns1 + 1], G, mode=0, nthreads=1)
            tc1 = time.perf_counter() - t0
            # parity gate on the very same sample, through the product's host API
            got = ix.query(sb, so, G)
            parity = all(np.array_equal(got[kk], ref[kk]) for kk in ("cnt_u", "cnt_d", "rcount_u", "rcount_d")) \
                and got["nundet"] == ref["nundet"] and got["nconf"] == ref["nconf"]
            if not parity:
                raise SystemExit("PARITY FAILURE: GPU counters differ from the CPU oracle on the bench sample")
            result["cpu_baseline"] = {
                "value": round(ns / tc / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
                "sample": f"first {ns} reads of the same batch, same index; OpenMP over reads with one global "
                          f"critical section per update as query64mt_p (oracle/cammiq_oracle.c); "
                          f"same sample with atomic counter updates instead of the lock ('fair' variant): "
                          f"{ns / tf / 1e6:.4f} Mreads/s; single-thread rate on {ns1} reads: {ns1 / tc1 / 1e6:.4f} Mreads/s",
                "cpu_model": _cpu_model(), "seconds": round(tc + tf + tc1, 2)}
            result["parity_checked_reads"] = ns
        if rank == 0:
            print(json.dumps(result), flush=True)
    finally:
        if world > 1:
            tdist.barrier()
        if local_rank == 0:
            shutil.rmtree(wdir, ignore_errors=True)
        if world > 1:
            tdist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
