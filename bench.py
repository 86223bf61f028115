#!/usr/bin/env python3
"""bench.py -- Mreads/s of the CAMMiQ classify hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: called WITHOUT a launcher (no WORLD_SIZE in the environment) `--gpus N` starts its N
ranks itself, as fresh child processes, before anything in this process has touched torch or the GPU, forwards
rank 0's JSON line and exits with the ranks' status.  The control plane (the 128-byte communicator id, barriers,
the max over ranks) is torch.distributed over **gloo**; the ONLY RCCL user in a rank is the product's own
collective (cq_counts_allreduce in libcammiq_hip.so, /opt/rocm's librccl).  There is no fallback collective: a
communicator that cannot be set up ends the run with a non-zero status.

Workload at N = 1 = BASELINE.json configs[2] (the configuration north_star quotes the metric on):
1000 synthetic bacterial-size genomes, --both index (unique + doubly-unique markers, h = k = 26),
50 M x 100 bp reads per step.  `--config 1` runs configs[1] (500 genomes, --unique, 10 M reads);
`--config 4` configs[4]'s per-GPU shard: ~15 000 genomes x 3.45 Mbp --both (1.26e9 markers, 92 GB on the device --
or what this box's host can build: the line says which), 125 M x 150 bp reads per step, one resident batch, the
CPU leg and the parity gate on a slice against the generator's sub-index.
N > 1 is weak scaling (configs[3] shape): the index is replicated and every rank classifies its own
50 M reads per step.

One "step" = one pass of the hot path (what FqReader::query64mt_p does for the reads of one FASTQ,
/root/reference/src/query.cpp:650-889) over one batch of packed reads already resident in HBM: the
classify kernels, accumulating into the device counters like the reference accumulates into FqReader
state, plus the D2H copy of the counter block.  The K timed steps are the K batches of one query
(three distinct resident batches, rotated); the query ends inside the timed bracket with its
exchange step -- for N > 1 the RCCL all-reduce of the counter block and of rcount, issued by the
product itself (cq_counts_allreduce, C ABI) -- and the D2H copy of rcount into pinned host memory,
i.e. with everything the host hands to the ILP (query.cpp:251-258).  Index load / layout and FASTQ
parsing are outside, exactly like the reference's own `Time for query` (query.cpp:459,645-647).

`value` counts reads with the inputs resident in HBM when the clock starts.  The PCIe-inclusive
rate of the same workload -- tight 2-bit rows in pinned HOST memory, pipelined H2D + kernels,
counters and rcount back in the caller's arrays through cq_query_packed_tight: SURVEY.md 8(d)'s
bracket -- is reported next to it as `host_fed` / `value_survey_8d_bracket`.  `roofline.board`
says what THIS board's memory system gives (cq_calibrate, ~0.15 s before the timed region).  `reads_door` is the rate
through cq_query_reads on the bench sample: the reference's own reads[] / rlengths[] arrays (ASCII, pageable), host packing
included -- what a binding that keeps readFastq gets.
Prints ONE JSON line on rank 0.
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
GEN_CHUNK = 5_000_000  # reads generated + packed per piece (one reusable ASCII buffer)

PRESETS = {
    1: dict(genomes=500, both=False, reads=10_000_000, read_len=100, batches=3, label="configs[1]"),
    2: dict(genomes=1000, both=True, reads=50_000_000, read_len=100, batches=3, label="configs[2]"),
    # configs[4]'s per-GPU shard: ~15 000 genomes, --both index "sized to 288 GB HBM", 1 B x 150 bp reads over 8 GPUs = 125 M
    # per GPU.  The index is as large as THIS box's host can build (tools/configs4.py size_for_this_box: 15 000 x 3.45 Mbp =
    # 1.26e9 markers where the host has the memory, shorter genomes where it has not -- the line says which); one
    # resident batch; the CPU leg and the parity gate run on a slice against the generator's sub-index (the oracle cannot
    # hold 10^9 markers: tests/test_subindex.py proves the construction).
    4: dict(genomes=15000, both=True, reads=125_000_000, read_len=150, batches=1, label="configs[4] per-GPU shard"),
}
BIG_INDEX_MARKERS = 3e8   # beyond: the oracle works on the sub-index of its slice


def algorithmic_bytes_per_read(rl: int, h: int, tables: int) -> int:
    """B = ceil(rl/4) + 2*(rl-h+1)*T*16 (hits omitted -> lower bound)."""
    return (rl + 3) // 4 + 2 * (rl - h + 1) * tables * 16


def workload_key(G, genome_len, both, n, rl):
    return f"G{G}_L{genome_len}_{'both' if both else 'unique'}_n{n}_rl{rl}"


STAGES = ("started", "index_loaded", "reads_resident", "comm_ready", "preflight_ok", "timed_done", "done")


def stage(name: str) -> None:
    """A rank says how far it got: one line on stderr and -- under bench.py's own launcher -- one file per rank in
    CAMMIQ_BENCH_STAGE_DIR, which the launcher's per-stage deadline watches (self_launch).  CAMMIQ_BENCH_TEST_STALL=
    "rank:stage:seconds" makes that rank ("*": every rank) sleep when it reaches that stage (test hook for the deadline)."""
    r = os.environ.get("RANK", "0")
    print(f"[bench rank {r}] stage={name}", file=sys.stderr, flush=True)
    d = os.environ.get("CAMMIQ_BENCH_STAGE_DIR")
    if d:
        tmp = os.path.join(d, f".rank_{r}.tmp")
        with open(tmp, "w") as f:
            f.write(name)
        os.replace(tmp, os.path.join(d, f"rank_{r}"))
    stall = os.environ.get("CAMMIQ_BENCH_TEST_STALL", "")
    if stall:
        sr, ss, sec = stall.split(":")
        if sr in (r, "*") and ss == name:
            time.sleep(float(sec))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(PRESETS),
                    help="BASELINE.json configs[i]: 2 = 1000 genomes, --both, 50 M reads (default); 1 = 500, --unique, 10 M")
    ap.add_argument("--genomes", type=int, default=None)
    ap.add_argument("--genome-len", type=int, default=None,
                    help="default 3 450 000; --config 4: the largest up to that whose index this box's host memory can build")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=None, help="default: the preset's (100; 150 for --config 4)")
    ap.add_argument("--batches", type=int, default=None, help="distinct resident batches the steps rotate over (default 3; 1 for --config 4)")
    ap.add_argument("--frac-deep", type=float, default=0.07,
                    help="fraction of markers longer than k (trie depth > 0); 0.07 is what the survey measured")
    ap.add_argument("--both", action="store_true", default=None, help="unique + doubly-unique index")
    ap.add_argument("--unique", dest="both", action="store_false", help="unique-only index")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="reads timed through the CPU oracle (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU oracle leg and its parity gate (used under rocprofv3 so that every "
                         "classify launch in the trace is a full-size timed step)")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the PCIe-inclusive leg (cq_query_packed)")
    ap.add_argument("--pipe-trace", default=None,
                    help="host-fed leg: one more query with CAMMIQ_PIPE_TRACE=<this file> (the library's own event timeline; "
                         "summarise with tools/pipe_trace.py), after the measured ones")
    ap.add_argument("--no-calibrate", action="store_true", help="skip the board calibrators (cq_calibrate, ~1 s before the timed region)")
    ap.add_argument("--multi-leg", choices=["auto", "on", "off"], default="auto",
                    help="N = 1 only: also run the same host-fed query through cq_multi_load / cq_multi_query_packed_tight (one "
                         "process, one host thread per GPU, RCCL inside the library) on min(2, visible GPUs) devices and "
                         "demand the single-device counts; auto = when at least two GPUs are visible")
    ap.add_argument("--ascii-api", action="store_true",
                    help="also time cq_query on ASCII reads in pageable memory (host packing included)")
    args = ap.parse_args()
    preset = PRESETS[args.config]
    G = args.genomes if args.genomes is not None else preset["genomes"]
    both = preset["both"] if args.both is None else args.both
    n = args.reads if args.reads is not None else preset["reads"]
    if args.read_len is None:
        args.read_len = preset["read_len"]
    if args.batches is None:
        args.batches = preset["batches"]
    fitted = None
    if args.genome_len is None:
        args.genome_len = 3_450_000
        if args.config == 4:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import configs4
            args.genome_len, avail, shm_free = configs4.size_for_this_box(G, 3_450_000)
            fitted = f"host: {avail / 1e9:.0f} GB available, {shm_free / 1e9:.0f} GB of scratch -> genome length {args.genome_len} of 3450000"
            if "WORLD_SIZE" not in os.environ or os.environ.get("RANK", "0") == "0":
                print(f"[bench] --config 4: {fitted}", file=sys.stderr, flush=True)
    is_preset = (G == preset["genomes"] and both == preset["both"] and n == preset["reads"] and args.read_len == preset["read_len"]
                 and args.genome_len == 3_450_000)
    label = preset["label"] if is_preset else f"configs[{args.config}]-shape (modified)"
    if args.config == 4 and fitted and args.genome_len != 3_450_000:
        label = f"configs[4] per-GPU shard, index shortened to what this host can build ({args.genome_len}-bp genomes)"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))      # nothing below has run yet: no torch import, no GPU touched
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    stage("started")

    import torch
    import cammiq_amd as cq
    from cammiq_amd import bigsynth, dist as cqdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the classify path has no CPU fallback")
    # Rehearsal knob for a one-GPU box: CAMMIQ_BENCH_REHEARSAL=1 maps every rank to cuda:0 and reduces over
    # gloo (RCCL needs one GPU per rank), so that the N > 1 code path (sharding, shared index files, max over
    # ranks) can be exercised without an 8-GPU node.  Never set by the driver; numbers from it mean nothing.
    rehearsal = os.environ.get("CAMMIQ_BENCH_REHEARSAL") == "1"
    dev = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        # control plane only (id, barriers, max over ranks): gloo on the CPU.  torch's bundled RCCL is never brought
        # up, so the product's collective is the one RCCL instance in the process.
        import torch.distributed as tdist
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: bind the loopback, the box's hostname may not resolve
        tdist.init_process_group("gloo")

    h = k = 26
    rl = args.read_len
    tables = 2 if both else 1
    t_setup = time.time()
    # one copy of the index files per node: local rank 0 writes, the others wait
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    wdir = os.path.join(shm, f"cammiq_bench_{os.environ.get('MASTER_PORT', os.getpid())}")
    try:
        w = bigsynth.World(seed=2, n_genomes=G, genome_len=args.genome_len, k=k, h=h, lmax=50,
                           frac_deep=args.frac_deep, pair_share=0.3 if both else 0.0)
        pu = os.path.join(wdir, "index_u.bin1")
        pd = os.path.join(wdir, "index_d.bin2") if both else None
        # an index the oracle cannot hold (configs[4]: ~100 bytes of heap per trie node): the generator also writes the
        # SUB-INDEX of the slice the CPU leg will see -- exactly the full index's markers whose h-mer occurs in those
        # reads -- and each of its leaves' position in the full index's decode order (tools/configs4.py, tests/test_subindex.py)
        est_markers = G * args.genome_len * (2 / 69 * 0.839) * (1.0 if both else 0.82)
        big = est_markers > BIG_INDEX_MARKERS
        want_cpu = rank == 0 and world == 1 and args.cpu_sample > 0 and not args.no_cpu_baseline
        su = os.path.join(wdir, "sub_u.bin1")
        sd = os.path.join(wdir, "sub_d.bin2") if both else None
        sub_ids = None
        if local_rank == 0:
            os.makedirs(wdir, exist_ok=True)
            if big and want_cpu:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from util import hmers_of_reads
                args.cpu_sample = min(args.cpu_sample, 100_000, n)
                sl_b, _ = w.reads(seed=1000 + 16 * rank, n=args.cpu_sample, length=args.read_len)
                nu, nd, ids_u, ids_d = w.write_index_with_sub(pu, pd, hmers_of_reads(sl_b, args.cpu_sample, args.read_len, h), su, sd)
                sub_ids = (ids_u, ids_d)
                del sl_b
            else:
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
        stage("index_loaded")
        info = ix.info_dict()
        n_vis = torch.cuda.device_count()
        # under a profiler the child would inherit the profiler's environment and put its kernels and copies into the
        # trace (and touch a second GPU): auto skips the leg there and the line says so
        profiled = any(k.startswith(("ROCP", "ROCPROF", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        multi_leg = world == 1 and not args.no_host_fed and (args.multi_leg == "on" or (args.multi_leg == "auto" and n_vis >= 2 and not profiled))
        multi_leg_skipped = ("running under a profiler" if world == 1 and not args.no_host_fed and args.multi_leg == "auto"
                             and n_vis >= 2 and profiled else None)
        if local_rank == 0 and world == 1 and not multi_leg and not (args.cpu_sample > 0 and not args.no_cpu_baseline):
            shutil.rmtree(wdir, ignore_errors=True)    # the CPU leg and the multi leg are the only later readers of the files
        elif local_rank == 0 and world == 1 and big and not multi_leg:
            for f in (pu, pu + ".aux", pd, (pd or "") + ".aux"):   # ~19 GB of /dev/shm at configs[4]: only the sub-index is read again
                if f and os.path.exists(f):
                    os.unlink(f)

        # ---- the batches: generated piecewise, packed once, resident in HBM (and, for the host-fed leg, in
        #      page-locked host memory).  Batch j of rank r is read stream 1000 + 16 r + j of the generator.
        host_fed = (world == 1 and not args.no_host_fed)
        sw = cq.stride_words(rl)
        sb = cq.stride_bytes(rl)
        nb = max(1, args.batches)
        t0 = time.time()
        ascii_buf = np.empty(min(GEN_CHUNK, n) * rl, np.uint8)
        sample_bases = None
        ns = min(args.cpu_sample, n)
        h_packed, h_lens, d_packed, d_lens = [], [], [], []
        # batch 0 also stays as what the reference holds after readFastq -- ASCII, pageable -- for the reads-door leg
        ascii_all = np.empty(n * rl, np.uint8) if (host_fed and rank == 0) else None
        for j in range(nb):
            keep_host = host_fed and j == 0           # one batch in pinned host memory feeds the host-fed leg
            hp = cq.host_array(n * sb, np.uint8).reshape(n, sb) if keep_host else None   # tight rows: what crosses the link
            hl = cq.host_array(n, np.uint8) if keep_host else None
            dp = torch.empty((n, sw), dtype=torch.int32, device="cuda")
            dl = torch.empty(n, dtype=torch.uint8, device="cuda")
            for c0 in range(0, n, GEN_CHUNK):
                m = min(GEN_CHUNK, n - c0)
                w.reads_into(ascii_buf, 1000 + 16 * rank + j, c0, m, rl)
                offs = np.arange(m + 1, dtype=np.uint64) * np.uint64(rl)
                pk, ln, skipped = cq.pack_reads(ascii_buf[:m * rl], offs, h, sw)
                assert skipped == 0
                if j == 0 and c0 == 0 and ns:
                    sample_bases = ascii_buf[:min(ns, m) * rl].copy()
                    ns = min(ns, m)
                if keep_host:
                    _, _, tsk = cq.pack_reads_tight(ascii_buf[:m * rl], offs, h, sb, out=(hp[c0:c0 + m], hl[c0:c0 + m]))
                    assert tsk == 0
                    if ascii_all is not None:
                        ascii_all[c0 * rl:(c0 + m) * rl] = ascii_buf[:m * rl]
                dp[c0:c0 + m].copy_(torch.from_numpy(pk.view(np.int32)))
                dl[c0:c0 + m].copy_(torch.from_numpy(ln))
            h_packed.append(hp); h_lens.append(hl); d_packed.append(dp); d_lens.append(dl)
        del ascii_buf
        torch.cuda.synchronize()
        t_reads = time.time() - t0
        stage("reads_resident")

        cw = ix.counter_words(G)
        nleaf = nu + nd
        ctr = torch.zeros(cw, dtype=torch.int64, device="cuda")
        rcd = torch.zeros(max(nleaf, 1), dtype=torch.int32, device="cuda")
        h_ctr = torch.zeros(cw, dtype=torch.int64).pin_memory()
        h_rc = torch.zeros(max(nleaf, 1), dtype=torch.int32).pin_memory()
        stream = torch.cuda.current_stream().cuda_stream

        # ---- N > 1: the product's own communicator (RCCL through the C ABI); torch.distributed only carries
        #      the 128-byte id, the barriers and the max over ranks
        comm, reduce_how = None, None
        if world > 1:
            if rehearsal:
                reduce_how = "gloo (one-GPU rehearsal; not RCCL)"
            else:
                # No fallback: if the product's communicator or its collective is broken the scaling run must say so
                # (non-zero exit on every rank), not measure somebody else's all-reduce.
                uid = [cq.comm_unique_id() if rank == 0 else None]
                tdist.broadcast_object_list(uid, src=0)
                comm = cq.Comm(ix, uid[0], rank, world)
                stage("comm_ready")
                # pre-flight: a 16-word block of ones must come back as 16 x world from the library's collective
                probe_t = torch.ones(16, dtype=torch.int64, device="cuda")
                probe_r = torch.ones(8, dtype=torch.int32, device="cuda")
                comm.allreduce_counts(probe_t.data_ptr(), 16, probe_r.data_ptr(), 8, stream)
                torch.cuda.synchronize()
                if int(probe_t.sum().item()) != 16 * world or int(probe_r.sum().item()) != 8 * world:
                    raise SystemExit(f"[bench rank {rank}] cq_counts_allreduce pre-flight returned a wrong sum")
                reduce_how = "cq_counts_allreduce (RCCL in libcammiq_hip.so; control plane: gloo)"
        if world > 1:
            stage("preflight_ok")

        # ---- what THIS board gives the kernel to work with, before the timed region (cq_calibrate, ~1 s): the chip's rate
        #      of random 16-byte loads from the handle's own table, the same with atomics + LDS traffic beside the loads,
        #      and the shader clock held under both -- so that the line itself says whether a slow run was the board
        calib = None
        if rank == 0 and not args.no_calibrate:
            try:
                calib = ix.calibrate()
            except Exception as e:       # diagnostic only: never takes the measurement down
                calib = {"error": str(e)[:200]}

        step_no = [0]

        def step():
            j = step_no[0] % nb
            step_no[0] += 1
            ix.query_device(cq.MODE_P, d_packed[j].data_ptr(), d_lens[j].data_ptr(), n, sw, rl, G,
                            ctr.data_ptr(), rcd.data_ptr(), stream)   # accumulates
            h_ctr.copy_(ctr, non_blocking=True)                        # D2H of the counter block, per batch

        def finish_query():
            """End of a query: the exchange step (N > 1), then everything the host hands on comes back."""
            if world > 1:
                if comm is not None:
                    comm.allreduce_counts(ctr.data_ptr(), cw, rcd.data_ptr() if nleaf else None, nleaf, stream)
                else:                                                  # one-GPU rehearsal only: sum over gloo on the host
                    torch.cuda.synchronize()
                    hc, hr = ctr.cpu(), rcd.cpu()
                    cqdist.allreduce_counts(hc, hr if nleaf else None)
                    ctr.copy_(hc); rcd.copy_(hr)
            h_ctr.copy_(ctr, non_blocking=True)
            if nleaf:                                                  # rcount once per query: the library's narrow way back
                rc_host = h_rc.numpy().view(np.uint32)                 # (cq_rcount_fetch: a byte per leaf + escapes over the link,
                ix.rcount_fetch(rcd.data_ptr(), stream, rc_host[:nu], rc_host[nu:nu + nd])   # widened into these pinned arrays)

        def fence():
            if world > 1:
                tdist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            step()
        finish_query()                        # also warms RCCL up
        fence()
        ctr.zero_()                           # resetCounters before the timed query
        rcd.zero_()
        step_no[0] = 0
        fence()
        kms, kslow = [], []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            a, b = ix.last_kernel_times()     # HIP events on the launch stream, recorded inside the library
            kms.append(a); kslow.append(b)
        finish_query()
        fence()
        dt = time.perf_counter() - t0
        stage("timed_done")
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64)
            tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
            dt = float(tmax.item())

        # ---- sanity on the query's counters: every read lands in exactly one outcome
        c = h_ctr.numpy().astype(np.uint64)
        tot_reads = n * world
        held_reads = tot_reads * args.steps
        nundet, nconf, nskip, nflag, nslow = (int(c[2 * (G + 1) + i]) for i in (0, 1, 2, 3, 4))
        cnt_u_sum, cnt_d_sum = int(c[:G + 1].sum()), int(c[G + 1:2 * G + 2].sum())
        rc_sum = int(h_rc.numpy().view(np.uint32).astype(np.uint64).sum()) if nleaf else 0
        assert nskip == 0 and nflag == 0
        if not both and not os.environ.get("CAMMIQ_LIB"):
            assert cnt_u_sum + nundet + nconf + nskip == held_reads, "conservation of reads violated"
        assert rc_sum >= max(cnt_u_sum, cnt_d_sum // 2) or os.environ.get("CAMMIQ_LIB"), "rcount did not come back"

        result = None
        if rank == 0:
            value = tot_reads * args.steps / dt / 1e6
            k_ms = float(np.mean(kms))
            k_slow_ms = float(np.mean(kslow))
            # Algorithmic bytes per read, SURVEY.md 8(d): ceil(rl/4) + 2 W T 16 -- one 16-byte slot (8-byte key,
            # 8-byte payload) per window, strand and table.  The merged table keeps val_u and val_d in ONE 16-byte
            # slot, so a --both probe touches the same 16 bytes as a --unique probe: the figure this design has to
            # move per read is the T = 1 one for either index; the T = 2 contract figure is given beside it.
            B = algorithmic_bytes_per_read(rl, h, 1)
            B_contract = algorithmic_bytes_per_read(rl, h, tables)
            achieved = n * B / (k_ms * 1e-3) / 1e9
            li = ix.last_launch_info()
            roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "kernel": li["kernel"], "kernel_launch": {kk: li[kk] for kk in ("reads_per_subtile", "hit_slots", "lds_hist",
                                                                                   "fixed_shape", "blocks_per_cu")},
                    "kernel_ms": round(k_ms, 4), "kernel_ms_runs": [round(x, 4) for x in kms],
                    "algorithmic_T": 1,
                    "slow_path_kernel_ms": round(k_slow_ms, 4),
                    "algorithmic_bytes_per_read": B,
                    "algorithmic_note": "ceil(rl/4) + 2*(rl-h+1)*16: one 16-B slot per window and strand; the merged "
                                        "u+d table answers both tables from that one slot",
                    "contract_T_tables_bytes_per_read": B_contract,
                    "contract_T_tables_GBs": round(n * B_contract / (k_ms * 1e-3) / 1e9, 2)}
            # measured HBM traffic + gather ceiling of this exact workload, from the committed PMC passes
            key = workload_key(G, args.genome_len, both, n, rl)
            ent = None
            for tp in ("traffic_r04.json", "traffic_r03.json", "traffic_r02.json"):   # the newest round that profiled this workload
                tp = os.path.join(ROOT, "profiles", tp)
                if ent is None and os.path.exists(tp):
                    try:
                        ent = json.load(open(tp)).get(key)
                    except Exception:
                        ent = None
            if ent:
                # the committed rocprofv3 --stats mean of this kernel on this workload (profiles/): the fraction a
                # reader recomputing from profiles/ gets, next to the one measured live in this run
                if ent.get("kernel_ms_rocprof_stats"):
                    pm = float(ent["kernel_ms_rocprof_stats"])
                    roof["profiled_kernel_ms"] = pm
                    roof["frac_at_profiled_ms"] = round(n * B / (pm * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                    roof["profiled_source"] = ent.get("kernel_stats_file")
                    if ent.get("in_kernel_clock_MHz"):
                        roof["profiled_in_kernel_clock_MHz"] = ent["in_kernel_clock_MHz"]
                hb = float(ent["hbm_bytes_per_launch"])
                roof["traffic"] = hb
                roof["physical_GBs"] = round(hb / (k_ms * 1e-3) / 1e9, 2)
                roof["physical_frac"] = round(hb / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                if ent.get("gather_ceiling_lines_per_s"):
                    # 64-byte lines READ per second (FETCH_SIZE; writes left out) against the chip's measured rate of
                    # random 16-byte loads from a table of this size
                    lines = float(ent.get("fetch_bytes_per_launch", hb)) / 64.0 / (k_ms * 1e-3)
                    roof["gather_ceiling_frac"] = round(lines / float(ent["gather_ceiling_lines_per_s"]), 5)
                    roof["gather_ceiling_note"] = ent.get("gather_ceiling_note")
            if calib and "error" not in calib:
                roof["board"] = {
                    "gather_ceiling_here_Glines_s": round(calib["gather16_Glines_s"], 3),
                    "gather_mix_here_Glines_s": round(calib["gather16_mix_Glines_s"], 3),
                    "chase_here_Glines_s": round(calib["chase16_Glines_s"], 3),
                    "chase_latency_ns": round(calib["chase_latency_ns"], 1),
                    "shader_clock_MHz_chase": round(calib["clock_MHz_chase"], 1),
                    "shader_clock_MHz_gather": round(calib["clock_MHz_gather"], 1),
                    "shader_clock_MHz_mix": round(calib["clock_MHz_mix"], 1),
                    "blocks_per_cu_best": [calib["gather_blocks_per_cu"], calib["mix_blocks_per_cu"]],
                    "table_GB": round(calib["table_bytes"] / 1e9, 3), "seconds": round(calib["seconds"], 2),
                    "what": "cq_calibrate on this run's board, before the timed region: random 16-byte loads from the "
                            "handle's own table (4 in flight per lane, best of 4/6/8 workgroups per CU); mix = the same "
                            "with a returnless atomic per 16 loads into an rcount-sized array and LDS stores/reads beside "
                            "them; chase = DEPENDENT random 16-byte loads (one in flight per lane) from one wave per CU, latency = "
                            "lanes in flight / rate: what the memory system answers a lone request in; clocks = shader "
                            "cycles per 100 MHz tick inside those kernels (median over workgroups)"}
                if ent and ent.get("fetch_bytes_per_launch"):
                    lines = float(ent["fetch_bytes_per_launch"]) / 64.0 / (k_ms * 1e-3)
                    roof["gather_ceiling_frac_here"] = round(lines / (calib["gather16_Glines_s"] * 1e9), 5)
                    roof["gather_mix_frac_here"] = round(lines / (calib["gather16_mix_Glines_s"] * 1e9), 5)
            elif calib:
                roof["board"] = calib
            result = {
                "metric": METRIC, "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
                "data": "synthetic",
                "config": {"workload": f"{label}: {G} synthetic genomes x {args.genome_len} bp, "
                           f"{'--both' if both else '--unique'} index h=k=26 ({nu}+{nd} leaves), "
                           f"{n} x {rl} bp reads per GPU per step (1% subst. errors, 10% off-database), "
                           f"{nb} distinct HBM-resident batches rotated",
                           "workload_key": key,
                           "reads_per_gpu": n, "read_len": rl, "hash_len": h, "n_genomes": G,
                           "leaves_u": nu, "leaves_d": nd, "table_GB": round(info["n_table_buckets"] * 64 / 1e9, 3),
                           "index_device_GB": round(info["device_bytes"] / 1e9, 3),
                           "table_buckets_overflowed": info["n_overflowed"], "table_max_chain": info["max_chain"],
                           "bracket": "K x (classify kernels + D2H of the counter block)"
                                      + (", RCCL all-reduce of counts + rcount" if world > 1 else "")
                                      + ", rcount back in pinned host arrays once (cq_rcount_fetch); inputs resident in HBM",
                           "parallelism": f"reads sharded x{world}, index replicated"
                                          + (f"; exchange step: {reduce_how}" if world > 1 else "")},
                "roofline": roof,
                "kernel_Mreads_s": round(n / ((k_ms + k_slow_ms) * 1e-3) / 1e6, 3),
                "outcome": {"reads": held_reads, "nundet": nundet, "nconf": nconf, "nskipped": nskip,
                            "slow_path_reads": nslow, "cnt_u_sum": cnt_u_sum, "cnt_d_sum": cnt_d_sum, "rcount_sum": rc_sum},
                "setup_s": {"generate_index": round(t_gen, 2), "index_load_layout_upload": round(t_load, 2),
                            "generate_pack_upload_reads": round(t_reads, 2)},
            }

        # ---- PCIe-inclusive rate (never `value`): SURVEY.md 8(d)'s bracket on the same batch -- packed reads in
        #      pinned host memory -> pipelined H2D -> kernels -> D2H of counters and rcount (cq_query_packed)
        if rank == 0 and host_fed:
            out = ix.counts_out(G, pinned=True)
            ix.query_packed_tight(h_packed[0][:1 << 16], h_lens[0][:1 << 16], rl, G, out=out)     # warm the staging buffers
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                hq = ix.query_packed_tight(h_packed[0], h_lens[0], rl, G, out=out)
                ts.append(time.perf_counter() - t0)
            th = min(ts)
            row_bytes = sb        # every read of the batch has one length: the lengths are filled on the device, not sent (else sb + 1)
            # SURVEY.md 8(d) defines the metric's bracket as H2D of packed reads + kernels + D2H: that number is this one
            # (`value` is the task contract's HBM-resident rate); quote both whenever one is quoted
            result["value_survey_8d_bracket"] = round(n / th / 1e6, 2)
            result["host_fed"] = {
                "Mreads_s": round(n / th / 1e6, 2), "ms": round(th * 1e3, 3), "runs_ms": [round(x * 1e3, 3) for x in ts],
                "rows_over_bracket_GBs": round(n * row_bytes / th / 1e9, 2), "bytes_per_read_on_the_wire": row_bytes,
                "what": "cq_query_packed_tight: tight 2-bit rows (ceil(len/4) bytes) in pinned host memory -> H2D in 2 M-read "
                        "chunks on a copy stream, widened to word rows on the device and classified while the next chunk "
                        "arrives (lengths of a uniform-length chunk are filled on the device, not sent) -> D2H of the counter "
                        "block; rcount comes back as one byte per leaf + an escape list and is widened into the caller's pinned "
                        "uint32 arrays by host threads while later pieces arrive (SURVEY 8(d) bracket, = the reference's "
                        "Time-for-query bracket); best of 3"}
            if args.pipe_trace:
                os.environ["CAMMIQ_PIPE_TRACE"] = args.pipe_trace
                ix.query_packed_tight(h_packed[0], h_lens[0], rl, G, out=out)
                os.environ.pop("CAMMIQ_PIPE_TRACE")
            # one query of batch 0 alone must agree with itself through both doors
            ctr.zero_(); rcd.zero_()
            ix.query_device(cq.MODE_P, d_packed[0].data_ptr(), d_lens[0].data_ptr(), n, sw, rl, G, ctr.data_ptr(),
                            rcd.data_ptr(), stream)
            torch.cuda.synchronize()
            c0 = ctr.cpu().numpy().astype(np.uint64)
            assert np.array_equal(c0[:G + 1], hq["cnt_u"]) and np.array_equal(c0[G + 1:2 * G + 2], hq["cnt_d"]), \
                "host-fed path disagrees with the device path"
            assert int(c0[2 * G + 2]) == hq["nundet"] and int(c0[2 * G + 3]) == hq["nconf"]
            if nleaf:
                rc_dev = rcd.cpu().numpy().view(np.uint32)
                assert np.array_equal(rc_dev[:nu], hq["rcount_u"]) and np.array_equal(rc_dev[nu:nu + nd], hq["rcount_d"]), \
                    "rcount through the host-fed door (narrow over the link, widened on the host) differs from the device's"

        # ---- second N > 1 shape: ONE process, one host thread per GPU inside the library, the library's
        #      own RCCL reduction -- the same kind of host-fed query over min(2, visible GPUs) devices must give the counts
        #      of a single device.  It runs in a CHILD process with a time limit (tools/multi_leg.py), after everything this
        #      line reports has been measured: RCCL's first contact with two ranks cannot take the headline down with it.
        #      A failure is recorded in the line (and fails the run only when the counts DIFFER), never hidden.
        if rank == 0 and multi_leg_skipped:
            result["multi_in_process"] = {"skipped": multi_leg_skipped}
        if rank == 0 and multi_leg:
            import subprocess
            ndev = min(2, n_vis)
            nm = min(n, 20_000_000)
            cmd = [sys.executable, os.path.join(ROOT, "tools", "multi_leg.py"), pu, pd or "-", str(G), str(args.genome_len),
                   "1" if both else "0", str(nm), str(rl), str(ndev)]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get("CAMMIQ_MULTI_LEG_TIMEOUT", "300")))
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if lines:
                    result["multi_in_process"] = json.loads(lines[-1])
                else:
                    result["multi_in_process"] = {"devices": list(range(ndev)), "error": f"exit status {r.returncode}: {r.stderr[-600:]}"}
                if r.returncode == 3:
                    raise SystemExit("multi-GPU (one process) counts differ from the single-device counts")
            except subprocess.TimeoutExpired:
                result["multi_in_process"] = {"devices": list(range(ndev)), "error": "no result within the time limit (child killed)"}

        if rank == 0 and world == 1 and host_fed and ascii_all is not None:
            # the door a reference-side binding uses without touching readFastq: FqReader's own arrays, one
            # pointer and one length byte per read, ASCII in pageable memory -- host packing + H2D + kernels + D2H, the whole
            # batch the host-fed leg classified
            ptrs = (np.uint64(ascii_all.ctypes.data) + np.arange(n, dtype=np.uint64) * np.uint64(rl)).astype(np.uint64)
            rl8 = np.full(n, rl, np.uint8)
            ix.query_reads(ptrs[:1000], rl8[:1000], G)
            out_pg = ix.counts_out(G, pinned=False)    # a binding's std::vectors
            out_pin = ix.counts_out(G, pinned=True)
            tr, tp = [], []
            for _ in range(2):
                t0 = time.perf_counter()
                rd = ix.query_reads(ptrs, rl8, G, out=out_pg)
                tr.append(time.perf_counter() - t0)
            for _ in range(2):
                t0 = time.perf_counter()
                ix.query_reads(ptrs, rl8, G, out=out_pin)
                tp.append(time.perf_counter() - t0)
            same = (int(rd["nundet"]) == int(hq["nundet"]) and int(rd["nconf"]) == int(hq["nconf"])
                    and np.array_equal(rd["cnt_u"], hq["cnt_u"]) and np.array_equal(rd["cnt_d"], hq["cnt_d"])
                    and np.array_equal(rd["rcount_u"], hq["rcount_u"]) and np.array_equal(rd["rcount_d"], hq["rcount_d"]))
            if not same:
                raise SystemExit("PARITY FAILURE: cq_query_reads disagrees with the host-fed door on the same batch")
            result["reads_door"] = {"Mreads_s": round(n / min(tr) / 1e6, 2), "Mreads_s_pinned_outputs": round(n / min(tp) / 1e6, 2),
                                    "reads": n, "runs_ms": [round(x * 1e3, 3) for x in tr], "runs_ms_pinned_outputs": [round(x * 1e3, 3) for x in tp],
                                    "what": "cq_query_reads on the reference's own arrays (one pointer + one length byte per read, "
                                            "ASCII in pageable host memory): host packing + H2D + kernels + D2H on the batch the "
                                            "host-fed leg classified, into pageable arrays (a binding's std::vectors) and into "
                                            "page-locked ones; counters equal to the host-fed door's"}
            del rd, ptrs, rl8, out_pg, out_pin
        ascii_all = None

        if rank == 0 and world == 1 and args.ascii_api and sample_bases is not None:
            so = np.arange(ns + 1, dtype=np.uint64) * np.uint64(rl)
            ix.query(sample_bases[:rl * 1000], so[:1001], G)
            t0 = time.perf_counter()
            ix.query(sample_bases, so, G)
            th = time.perf_counter() - t0
            result["ascii_api"] = {"Mreads_s": round(ns / th / 1e6, 2), "reads": ns,
                                   "what": "cq_query on ASCII reads in pageable host memory (pack + H2D + kernel + D2H)"}

        # ---- CPU baseline (rank 0, N = 1 only): the oracle, a restatement of query64mt_p
        if rank == 0 and world == 1 and not args.no_cpu_baseline and ns > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            # all host cores this process may run on (the lease's share of the box; os.cpu_count() is the box)
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cores = max(1, min(cores, oracle_lib.lib().cqo_omp_max_threads() if not os.environ.get("OMP_NUM_THREADS") else cores))
            oi = oracle_lib.OracleIndex(su, sd) if big else oracle_lib.OracleIndex(pu, pd)
            sb, so = sample_bases, np.arange(ns + 1, dtype=np.uint64) * np.uint64(rl)
            keys4 = ("cnt_u", "cnt_d", "rcount_u", "rcount_d")
            # ~25 s of CPU work in all: the thread-local and the atomic variant on the whole sample, the locked one (which
            # gets SLOWER with many threads: one critical section per read) on a quarter of it when more than 32 threads
            # run, one core on an eighth
            # rates are reads / the oracle's loop over the reads (`loop_s`) -- the reference's own "Time for query" bracket
            # (query.cpp:459,645-647); zeroing and reading back rcount of ~10^8 heap nodes around it is not classify time
            t_cpu0 = time.perf_counter()
            ref = oi.query(sb, so, G, mode=0, nthreads=cores, variant="thread_local")   # SURVEY 8(d)'s optimised variant
            tt = ref["loop_s"]
            fair = oi.query(sb, so, G, mode=0, nthreads=cores, variant="atomic")        # the locked loop with atomics instead
            tf = fair["loop_s"]
            # query64mt_p as written (one global critical section per read, query.cpp:742-878) gets SLOWER with many
            # threads; a reference user picks -t, so the faithful figure is its best over a few thread counts -- on a
            # quarter of the sample, every run's counters compared with a thread-local run of the same quarter
            nsl = max(ns // 4, 1)
            refq = ref if nsl == ns else oi.query(sb[:nsl * rl], so[:nsl + 1], G, mode=0, nthreads=cores, variant="thread_local")
            assert all(np.array_equal(fair[kk], ref[kk]) for kk in keys4) and fair["nundet"] == ref["nundet"] \
                and fair["nconf"] == ref["nconf"], "CPU variant atomic differs from the thread-local one"
            crit = {}
            for t in sorted({x for x in (8, 16, 32, cores) if x <= cores}):
                lk = oi.query(sb[:nsl * rl], so[:nsl + 1], G, mode=0, nthreads=t, variant="critical")
                assert all(np.array_equal(lk[kk], refq[kk]) for kk in keys4) and lk["nundet"] == refq["nundet"] \
                    and lk["nconf"] == refq["nconf"], f"CPU variant critical ({t} threads) differs from the thread-local one"
                crit[t] = nsl / lk["loop_s"] / 1e6
            ns1 = max(ns // 8, 1)
            tc1 = oi.query(sb[:ns1 * rl], so[:ns1 + 1], G, mode=0, nthreads=1)["loop_s"]
            crit[1] = ns1 / tc1 / 1e6
            best_t = max(crit, key=crit.get)
            t_cpu = time.perf_counter() - t_cpu0
            # parity gate on the very same sample, through the product's host API
            got = ix.query(sb, so, G)
            if big:
                # the oracle saw the sub-index: its rcount belongs at the sub-index leaves' positions in the full index's
                # decode order, every other leaf of the full index must read zero
                parity = all(np.array_equal(got[kk], ref[kk]) for kk in ("cnt_u", "cnt_d"))
                for kk, ids, n_full in (("rcount_u", sub_ids[0], nu), ("rcount_d", sub_ids[1], nd)):
                    exp = np.zeros(n_full, np.uint32)
                    exp[ids.astype(np.int64)] = ref[kk]
                    parity = parity and np.array_equal(got[kk], exp)
                    del exp
            else:
                parity = all(np.array_equal(got[kk], ref[kk]) for kk in keys4)
            parity = parity and got["nundet"] == ref["nundet"] and got["nconf"] == ref["nconf"]
            if not parity:
                raise SystemExit("PARITY FAILURE: GPU counters differ from the CPU oracle on the bench sample")
            result["cpu_baseline"] = {
                "value": round(crit[best_t], 4), "unit": "Mreads/s", "cores": best_t, "kind": "port",
                "value_as_written_by_threads": {str(t): round(v, 4) for t, v in sorted(crit.items())},
                "value_thread_local": round(ns / tt / 1e6, 4), "value_atomic": round(ns / tf / 1e6, 4),
                "value_one_core": round(crit[1], 4), "best_value": round(max(ns / tt / 1e6, ns / tf / 1e6, crit[best_t]), 4),
                "cores_available": cores, "host_cores_online": os.cpu_count(),
                "sample": ("ORACLE ON THE SUB-INDEX of its slice (the full index's markers whose h-mer occurs in these reads: same "
                           "answers, a far smaller map -- the CPU rates are an UPPER bound of what the full index would give). "
                           if big else "") +
                          f"first {ns} reads of batch 0, same index; {cores} cores available to this process "
                          f"({os.cpu_count()} online on the box). value: OpenMP over reads with one global critical "
                          f"section per read as query64mt_p (oracle/cammiq_oracle.c), best over "
                          f"{sorted(crit)} threads (`cores` = the best count) on the first {nsl} reads ({ns1} for one "
                          f"thread), every run equal to the thread-local counters; value_thread_local: per-thread "
                          f"counters merged after the loop, rcount by atomics, {cores} threads, all {ns} reads (SURVEY 8(d)'s "
                          f"optimised variant -- the number to hold the GPU against, = best_value when it is the fastest); "
                          f"value_atomic: the locked loop with atomics instead, {cores} threads",
                "bracket": "the loop over the reads alone, as the reference's Time for query (query.cpp:459,645-647)",
                "cpu_model": _cpu_model(), "seconds": round(t_cpu, 2)}
            result["parity_checked_reads"] = ns
        if rank == 0:
            print(json.dumps(result), flush=True)
        if comm is not None:
            comm.close()
        stage("done")
    finally:
        if world > 1:
            tdist.barrier()
        if local_rank == 0:
            shutil.rmtree(wdir, ignore_errors=True)
        if world > 1:
            tdist.destroy_process_group()


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has
    not imported torch or touched the GPU, and never will), rank 0 inherits stdout so its ONE JSON line is this
    program's output; the other ranks' stdout goes to stderr.  Returns the exit status: 0 only if every rank
    returned 0.  A rank that fails takes the others down (exact PIDs, no patterns).

    The launcher has a deadline and a voice: every rank reports the stage it reached (stage()); a rank that stays in
    one stage longer than CAMMIQ_BENCH_STAGE_TIMEOUT seconds (default 600) -- stuck in ncclCommInitRank, in the gloo
    rendezvous, in the first collective -- or a run longer than CAMMIQ_BENCH_TIMEOUT in all (default 3000) ends the run:
    every live rank is terminated by PID, killed after a grace period, the last stage of every rank is printed and
    the status is non-zero.  SIGTERM / SIGINT to the launcher do the same, so no rank is left holding a GPU.  Ranks
    are never restarted in place."""
    import signal
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    stage_limit = float(os.environ.get("CAMMIQ_BENCH_STAGE_TIMEOUT", "600"))
    total_limit = float(os.environ.get("CAMMIQ_BENCH_TIMEOUT", "3000"))
    grace = float(os.environ.get("CAMMIQ_BENCH_KILL_GRACE", "5"))
    sdir = tempfile.mkdtemp(prefix="cammiq_bench_stages_")
    procs = []
    got_signal = []

    def on_signal(signo, _frame):
        got_signal.append(signo)

    old_handlers = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}

    def last_stage(r):
        try:
            return open(os.path.join(sdir, f"rank_{r}")).read().strip() or "not_started"
        except OSError:
            return "not_started"

    def stop_all(why):
        live = [r for r in range(len(procs)) if procs[r].poll() is None]
        print(f"[bench launcher] {why}; last stage per rank: "
              + ", ".join(f"rank {r}: {last_stage(r)}" + ("" if procs[r].poll() is None else f" (exited {procs[r].returncode})")
                          for r in range(len(procs))), file=sys.stderr, flush=True)
        for r in live:
            procs[r].terminate()
        t_end = time.monotonic() + grace
        for r in live:
            try:
                procs[r].wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                print(f"[bench launcher] rank {r} (pid {procs[r].pid}) ignored SIGTERM: SIGKILL", file=sys.stderr, flush=True)
                procs[r].kill()
                procs[r].wait()

    status = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CAMMIQ_BENCH_STAGE_DIR=sdir)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC is the only kind this host driver supports (RCCL)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else sys.stderr))
            print(f"[bench launcher] rank {r} pid {procs[r].pid}", file=sys.stderr, flush=True)
        t_start = time.monotonic()
        seen = {r: ("not_started", t_start) for r in range(n)}
        alive = set(range(n))
        while alive:
            now = time.monotonic()
            if got_signal:
                status = 128 + got_signal[0]
                stop_all(f"signal {got_signal[0]} received: stopping all ranks")
                break
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    st = last_stage(r)
                    if st != seen[r][0]:
                        seen[r] = (st, now)
                    continue
                alive.discard(r)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 1
                    stop_all(f"rank {r} exited with status {rc}: stopping the other ranks")
                    alive.clear()
                    break
            if not alive or status:
                break
            stuck = [r for r in sorted(alive) if now - seen[r][1] > stage_limit]
            if stuck:
                status = 124
                stop_all("deadline: " + ", ".join(f"rank {r} spent more than {stage_limit:g} s in stage={seen[r][0]}" for r in stuck))
                break
            if now - t_start > total_limit:
                status = 124
                stop_all(f"deadline: the run took more than {total_limit:g} s")
                break
            time.sleep(0.05)
    finally:
        if any(p.poll() is None for p in procs):     # an exception in the loop above: leave nothing behind
            stop_all("launcher is exiting")
            status = status or 1
        for sg, hd in old_handlers.items():
            signal.signal(sg, hd)
        shutil.rmtree(sdir, ignore_errors=True)
    return status


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
