"""BASELINE.json configs at their own per-GPU shard shapes, on one MI355X.

configs[2] / configs[3]: 1000 genomes, --both index, 50 M x 100 bp reads (configs[3] = this shard on each
of 8 GPUs).  configs[4]'s shard shape: ~15 000 genomes (per-genome counters beyond the LDS histogram ->
global atomics), --both, 150-bp reads, 10 M of them.  Full size is far beyond the oracle, so what is
checked is what must hold at any size -- shard additivity (the multi-GPU contract: halves and thirds add
up to the whole, counter by counter and leaf by leaf), idempotence, device API == host API == packed
host-fed API, rcount consistent with the per-genome counters -- plus the oracle itself on a 100 k-read
slice of the same batch."""
import numpy as np
import pytest

import cammiq_amd as cq
import oracle_lib
from util import assert_same

pytestmark = pytest.mark.gpu

KEYS = ("cnt_u", "cnt_d", "rcount_u", "rcount_d")


def _add(a, b):
    out = {k: a[k] + b[k] for k in KEYS}
    out["nundet"] = a["nundet"] + b["nundet"]
    out["nconf"] = a["nconf"] + b["nconf"]
    return out


def _device_query(ix, packed, lens, lo, hi, rl, G):
    import torch
    dp = torch.from_numpy(packed[lo:hi].view(np.int32)).cuda()
    dl = torch.from_numpy(lens[lo:hi]).cuda()
    ctr = torch.zeros(ix.counter_words(G), dtype=torch.int64, device="cuda")
    rc = torch.zeros(sum(ix.n_leaves), dtype=torch.int32, device="cuda")
    ix.query_device(cq.MODE_P, dp.data_ptr(), dl.data_ptr(), hi - lo, packed.shape[1], rl, G, ctr.data_ptr(),
                    rc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = ctr.cpu().numpy().astype(np.uint64)
    r = rc.cpu().numpy().view(np.uint32)
    assert int(c[2 * G + 4]) == 0 and int(c[2 * G + 5]) == 0          # nskipped, flags
    return dict(cnt_u=c[:G + 1], cnt_d=c[G + 1:2 * G + 2], nundet=int(c[2 * G + 2]), nconf=int(c[2 * G + 3]),
                rcount_u=r[:ix.n_leaves[0]].copy(), rcount_d=r[ix.n_leaves[0]:].copy(), nslow=int(c[2 * G + 6]))


def _shape_checks(tmp_path, G, genome_len, n, rl, min_leaves):
    from cammiq_amd import bigsynth
    w = bigsynth.World(seed=2, n_genomes=G, genome_len=genome_len, pair_share=0.3)
    pu, pd = str(tmp_path / "index_u.bin1"), str(tmp_path / "index_d.bin2")
    nu, nd = w.write_index(pu, pd)
    assert nu > min_leaves[0] and nd > min_leaves[1]
    ix = cq.Index(pu, pd, device=0)
    sw = cq.stride_words(rl)
    packed = np.empty((n, sw), np.uint32)
    lens = np.empty(n, np.uint8)
    chunk = 5_000_000
    buf = np.empty(chunk * rl, np.uint8)
    sample = None
    ns = 100_000
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        w.reads_into(buf, 1000, c0, m, rl)
        pk, ln, sk = cq.pack_reads(buf[:m * rl], np.arange(m + 1, dtype=np.uint64) * np.uint64(rl), 26, sw)
        assert sk == 0
        packed[c0:c0 + m] = pk
        lens[c0:c0 + m] = ln
        if c0 == 0:
            sample = buf[:ns * rl].copy()
    del buf
    so = np.arange(ns + 1, dtype=np.uint64) * np.uint64(rl)

    whole = _device_query(ix, packed, lens, 0, n, rl, G)               # ONE launch of the full shard
    assert int(whole["cnt_u"].sum()) > 0.2 * n and int(whole["cnt_d"].sum()) > 0
    # every counted read bumps rcount of >= 1 leaf; a read counted in d only via one pair bumps cnt_d twice
    counted_lo = max(int(whole["cnt_u"].sum()), int(whole["cnt_d"].sum()) // 2)
    assert int(whole["rcount_u"].astype(np.uint64).sum()) + int(whole["rcount_d"].astype(np.uint64).sum()) >= counted_lo
    # every read is undetermined, conflicting or counted; a counted read adds 1 to cnt_u and/or 1-2 to cnt_d
    base = whole["nundet"] + whole["nconf"] + int(whole["cnt_u"].sum())
    assert base <= n <= base + int(whole["cnt_d"].sum())
    again = _device_query(ix, packed, lens, 0, n, rl, G)
    assert_same(again, whole, "idempotence")
    # shard additivity: two unequal halves, then the eight shards cq_shard_range gives an 8-GPU node
    a, b = _device_query(ix, packed, lens, 0, n // 2 + 3, rl, G), _device_query(ix, packed, lens, n // 2 + 3, n, rl, G)
    assert_same(_add(a, b), whole, "halves")
    acc = None
    for p in range(8):
        lo, hi = cq.shard_range(n, p, 8)
        part = _device_query(ix, packed, lens, lo, hi, rl, G)
        acc = part if acc is None else _add(acc, part)
    assert_same(acc, whole, "eight shards")
    # the host-fed door (pipelined H2D of 2 M-read chunks) gives the same counters
    assert_same(ix.query_packed(packed, lens, rl, G), whole, "cq_query_packed")
    # the oracle on a slice, through ASCII (cq_query) and through the packed rows
    ref = oracle_lib.OracleIndex(pu, pd).query(sample, so, G, nthreads=16)
    assert_same(ix.query(sample, so, G), ref, "slice vs oracle")
    assert_same(_device_query(ix, packed, lens, 0, ns, rl, G), ref, "packed slice vs oracle")
    assert sum(ref["branch"][k] for k in ("U1_P0", "U1_Pall", "U0_P1")) > 0.5 * ns
    return ix, whole


def test_configs2_full_size_shard(tmp_path):
    """configs[2] = configs[3]'s per-GPU shard: 1000 genomes x 3.45 Mbp, --both, 50 M x 100 bp."""
    ix, whole = _shape_checks(tmp_path, 1000, 3_450_000, 50_000_000, 100, (60_000_000, 10_000_000))
    assert whole["nslow"] == 0


def test_configs4_shard_shape(tmp_path):
    """configs[4]'s shape at test scale: 15 000 genomes (global-atomic per-genome counters), --both,
    150-bp reads, 10 M per launch."""
    ix, whole = _shape_checks(tmp_path, 15_000, 100_000, 10_000_000, 150, (15_000_000, 3_000_000))
    assert int(np.count_nonzero(whole["cnt_u"])) > 14_000


def test_inline_pair_limit_falls_back_silently(tmp_path):
    """A depth-0 doubly-unique leaf keeps its two refIDs inline in the slot only when both are below 2^15
    (cq_device.h CQ_INLINE_PAIR_BIT); above that the kernel reads them from leaf_rids.  Same results."""
    from cammiq_amd import synth
    from util import build_index
    gen = synth.clade_genomes(41, 3, 3, 2500, 0.03)
    u, d = synth.select_markers(gen, 22, 22, keep_every=2, seed=2)      # all keys of length h: depth-0 leaves
    assert len(d) > 50
    remap = {i + 1: 32_700 + i * 40 for i in range(len(gen))}           # straddles 2^15 = 32768
    u2 = {k: (remap[r], c) for k, (r, c) in u.items()}
    d2 = {k: (remap[a], remap[b], c1, c2) for k, (a, b, c1, c2) in d.items()}
    pu, pd = build_index(tmp_path, u2, d2, 22)
    reads = synth.simulate_reads(gen, 6000, (30, 200), 0.01, 3, frac_random=0.1)
    G = 33_100
    b, o = synth.concat_reads(reads)
    for mode in (0, 1):
        got = cq.Index(pu, pd, device=0).query(b, o, G, mode=mode)
        ref = oracle_lib.OracleIndex(pu, pd).query(b, o, G, mode=mode)
        assert_same(got, ref, f"refIDs around 2^15, mode {mode}", rcount=(mode == 0))
        assert got["pairs"] == ref["pairs"]
    assert int(ref["cnt_d"][remap[9]]) > 0 and int(ref["cnt_d"][remap[1]]) > 0


def test_configs4_index_at_size():
    """configs[4]'s per-GPU shard with the index at the size this box can build -- 15 000 genomes x 3.45 Mbp at the
    survey's marker density = 1.26e9 markers (81 GB of table, ~97 GB on the device) where the host has the ~170 GB the
    build peaks at, proportionally shorter genomes where it has not (CAMMIQ_CFG4_GENOME_LEN pins a length) -- and
    125 M x 150 bp reads (configs[4]'s per-GPU read count) in ONE launch, through the device door and -- the whole shard --
    through the host-fed door.  tools/configs4.py does the work: property checks that hold at any size
    (conservation, idempotence, halves and the eight cq_shard_range shards add up leaf by leaf, host-fed door ==
    device door) and the oracle on a 100 k-read slice against the generator's sub-index (the full index's markers
    whose h-mer occurs in the slice: tests/test_subindex.py)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import configs4
    G = 15_000
    L, avail, shm = configs4.size_for_this_box(G, 3_450_000)
    if os.environ.get("CAMMIQ_CFG4_GENOME_LEN"):
        L = int(os.environ["CAMMIQ_CFG4_GENOME_LEN"])
    if L < 400_000:
        pytest.skip(f"host memory ({avail / 1e9:.0f} GB available, {shm / 1e9:.0f} GB of /dev/shm) is too small for an index beyond the shape test's")
    n_reads = int(os.environ.get("CAMMIQ_CFG4_READS", "125000000"))      # configs[4]: 1 B x 150 bp over 8 GPUs = 125 M per GPU
    rec = configs4.run(G, L, n_reads, 150, log=lambda s: print(s, flush=True))
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(rec, open(os.path.join(out, "cfg4_test_record.json"), "w"), indent=1)
    # the size this ran at, where a -q run keeps it: pytest's warnings summary (a pass prints nothing else)
    import warnings
    warnings.warn(f"configs[4] ran at {G} genomes x {L} bp = {rec['leaves_u'] + rec['leaves_d']} markers "
                  f"({rec['table']['table_GB']} GB table, {rec['table']['device_GB']} GB on the device, m = {rec['table']['minimizer_len']}), "
                  f"{n_reads} x 150 bp in one launch: kernel {rec['kernel_ms']} ms ({rec['kernel_Gwindows_s']} G windows/s), "
                  f"host-fed door on the whole shard {rec['host_fed_Mreads_s']} Mreads/s, cq_index_load "
                  f"{rec['stages_s'].get('cq_index_load (decode+layout+upload)')} s, peak host RSS {rec['peak_host_RSS_GB']} GB; "
                  f"host had {avail / 1e9:.0f} GB available")
    bad = [k for k, v in rec["checks"].items() if v is False]
    assert not bad, f"{bad} failed at {rec['leaves_u']} + {rec['leaves_d']} markers"
    assert rec["checks"]["genomes_hit"] > 14_900
    assert rec["leaves_u"] + rec["leaves_d"] > 0.9 * G * L * configs4.MARKERS_PER_GENOME_BASE
    assert rec["outcome"]["slow_path_reads"] < 0.001 * rec["reads"]
    assert rec["kernel_launch"]["fixed_shape"] == 1 and rec["kernel_launch"]["lds_hist"] == 0    # 15 000 genomes: global counters
