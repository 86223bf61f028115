"""the multi-GPU path behind the C ABI (cq_multi_*, cq_comm_*, RCCL linked into
libcammiq_hip.so).  The GPU box of this pool has ONE MI355X, so what runs here is
  * n_dev = min(2, device count) distinct devices through ncclCommInitAll + ncclAllReduce (on a
    one-GPU box: the one-rank communicator, same code path),
  * two and three shards on the SAME device (the library sums shards that share a device with a
    kernel and sends the per-device sums through RCCL) -- the sharding, the threads, the counter
    layout and the SC pair merge with more than one shard,
  * the one-process-per-GPU entry points (cq_comm_*) with world size 1.
Every result must equal the single-handle result exactly."""
import os

import numpy as np
import pytest

import cammiq_amd as cq
from cammiq_amd import synth
import oracle_lib
from util import assert_same, build_index, golden

pytestmark = pytest.mark.gpu


def _n_devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("devices", ["distinct", [0, 0], [0, 0, 0]])
def test_multi_query_equals_single_device(tmp_path, devices):
    if devices == "distinct":
        devices = list(range(min(2, _n_devices())))
    gen = synth.clade_genomes(77, 4, 3, 4000, 0.03)
    u, d = synth.select_markers(gen, 26, 40, keep_every=3, seed=7)
    pu, pd = build_index(tmp_path, u, d, 26)
    G = len(gen)
    reads = synth.simulate_reads(gen, 30011, (26, 200), 0.01, 11, frac_random=0.1)
    b, o = synth.concat_reads(reads)
    one = cq.Index(pu, pd, device=0)
    m = cq.Multi(pu, pd, devices)
    assert m.size == len(devices) and m.n_leaves == one.n_leaves
    ref = oracle_lib.OracleIndex(pu, pd).query(b, o, G, nthreads=8)
    for mode in (cq.MODE_P, cq.MODE_SC):
        want = one.query(b, o, G, mode=mode)
        got = m.query(b, o, G, mode=mode)
        assert_same(got, want, f"multi {devices} mode {mode}")
        assert got["pairs"] == want["pairs"] and got["nskipped"] == want["nskipped"]
    assert_same(m.query(b, o, G), ref, "multi vs oracle")
    # pre-packed rows through the same shards
    packed, lens, _ = cq.pack_reads(b, o, one.hash_len)
    assert_same(m.query_packed(packed, lens, 200, G), ref, "multi packed")
    tight, tlens, _ = cq.pack_reads_tight(b, o, one.hash_len)           # 50 bytes per read instead of 52
    assert tight.shape[1] == 50
    for mode in (cq.MODE_P, cq.MODE_SC):
        got = m.query_packed_tight(tight, tlens, 200, G, mode=mode)
        want = one.query(b, o, G, mode=mode)
        assert_same(got, want, f"multi tight mode {mode}", rcount=(mode == cq.MODE_P))
        assert got["pairs"] == want["pairs"]
    # fewer reads than shards, and none at all
    for k in (0, 1, len(devices)):
        bb, oo = synth.concat_reads(reads[:k])
        assert_same(m.query(bb, oo, G), one.query(bb, oo, G), f"{k} reads")
    m.close()


def test_query_packed_equals_query_and_kernel_times(tmp_path):
    g = golden("f_deep")
    b, o = synth.concat_reads(g["reads"])
    ix = cq.Index(g["pu"], g["pd"], device=0)
    packed, lens, sk = cq.pack_reads(b, o, ix.hash_len)
    assert sk == 0
    got = ix.query_packed(packed, lens, 255, g["G"])
    assert_same(got, g["exp"]["p"], "packed, pageable")
    # page-locked inputs and outputs (what bench.py's host-fed leg uses)
    pp = cq.host_array(packed.size, np.uint32).reshape(packed.shape)
    pl = cq.host_array(lens.size, np.uint8)
    pp[:] = packed
    pl[:] = lens
    out = ix.counts_out(g["G"], pinned=True)
    for _ in range(2):      # reused output arrays are overwritten
        got = ix.query_packed(pp, pl, 255, g["G"], out=out)
        assert_same(got, g["exp"]["p"], "packed, pinned")
    fast, slow = ix.last_kernel_times()
    assert fast > 0 and slow >= 0 and abs(ix.last_kernel_ms() - (fast + slow)) < 1e-3
    # max_len is only a hint: unknown (0) or too small, the lengths themselves size the work
    for hint in (0, 30):
        assert_same(ix.query_packed(packed, lens, hint, g["G"]), g["exp"]["p"], f"hint {hint}")
    bad = lens.copy()
    bad[5] = 255
    narrow = np.ascontiguousarray(packed[:, :4])            # rows of 4 words hold 64 bases
    with pytest.raises(cq.CammiqError) as e:
        ix.query_packed(narrow, bad, 64, g["G"])
    assert e.value.code == -1


@pytest.mark.parametrize("rl", [(26, 255), 100, 27, 150])
def test_tight_rows_equal_ascii_reads(tmp_path, rl):
    """cq_query_packed_tight: rows at a byte stride cross the link and are widened on the device.  Ragged and fixed
    lengths (strides of 64, 25, 7 and 38 bytes: every alignment of a row start), reads outside the parity domain in
    between, more than one 2 M-read chunk, pinned and pageable inputs, a stride too small for a length."""
    gen = synth.clade_genomes(91, 3, 3, 3000, 0.03)
    u, d = synth.select_markers(gen, 26, 40, keep_every=2, seed=4)
    pu, pd = build_index(tmp_path, u, d, 26)
    G = len(gen)
    base = synth.simulate_reads(gen, 7001, rl, 0.01, 5, frac_random=0.1)
    hi = rl if isinstance(rl, int) else rl[1]
    bad = [b"ACGT", gen[0][:hi - 1] + b"N", b""]
    reads = base[:3000] + bad + base[3000:]
    reads = reads * (301 if rl == 100 else 1)              # 100 bp: 2.1 M reads = two chunks, the second one short
    b, o = synth.concat_reads(reads)
    ix = cq.Index(pu, pd, device=0)
    tight, lens, sk = cq.pack_reads_tight(b, o, ix.hash_len)
    assert tight.shape[1] == cq.stride_bytes(hi) and sk == len(bad) * (len(reads) // len(base + bad))
    for mode in (cq.MODE_P, cq.MODE_SC):
        want = ix.query(b, o, G, mode=mode)
        got = ix.query_packed_tight(tight, lens, hi, G, mode=mode)
        assert_same(got, want, f"tight rl={rl} mode={mode}", rcount=(mode == cq.MODE_P))
        assert got["pairs"] == want["pairs"] and got["nskipped"] == want["nskipped"] == sk
    if rl != 100:
        ref = oracle_lib.OracleIndex(pu, pd).query(*synth.concat_reads(base), G, nthreads=8)
        assert_same(ix.query_packed_tight(tight, lens, 0, G), ref, f"tight vs oracle rl={rl}")
    pt = cq.host_array(tight.size, np.uint8).reshape(tight.shape)
    pl = cq.host_array(lens.size, np.uint8)
    pt[:] = tight
    pl[:] = lens
    assert_same(ix.query_packed_tight(pt, pl, hi, G), ix.query(b, o, G), "tight, pinned")
    if tight.shape[1] > 7:
        bad_l = lens.copy()
        bad_l[1] = 4 * 7 + 1
        with pytest.raises(cq.CammiqError) as e:
            ix.query_packed_tight(np.ascontiguousarray(tight[:, :7]), bad_l, 0, G)
        assert e.value.code == -1


@pytest.mark.parametrize("fused", ["1", "0"])
def test_tight_rows_through_the_exact_slow_path_and_every_kernel_shape(tmp_path, monkeypatch, fused):
    """The classify kernel's own staging widens tight rows (round 4; CAMMIQ_FUSED_WIDEN=0: the separate widening kernel of
    rounds 2-3).  Every kernel shape that reads them: the exact slow path (dense markers: far more than 16 hits per read,
    rows read byte by byte), h = 26 at 100 / 150 bp (shape folded in), 101 / 125 / 151 bp (stride folded in), the generic
    kernel (another h; ragged lengths up to 255: strides that are not a multiple of four bytes, R = 4 for long reads)."""
    monkeypatch.setenv("CAMMIQ_FUSED_WIDEN", fused)
    gen = synth.clade_genomes(31, 1, 3, 1200, 0.05)
    u, d = synth.select_markers(gen, 10, 16, keep_every=1, seed=0)
    pu, pd = build_index(tmp_path, u, d, 8, name="dense")
    reads = synth.simulate_reads(gen, 700, (60, 255), 0.005, 3)
    b, o = synth.concat_reads(reads)
    ix = cq.Index(pu, pd, device=0)
    tight, lens, _ = cq.pack_reads_tight(b, o, 8)
    ref = oracle_lib.OracleIndex(pu, pd).query(b, o, len(gen))
    got = ix.query_packed_tight(tight, lens, 0, len(gen))
    assert_same(got, ref, "dense markers through the tight door")
    assert int(np.max(ref["rcount_u"])) > 0
    gen = synth.clade_genomes(77, 3, 4, 4000, 0.03)
    u, d = synth.select_markers(gen, 26, 48, keep_every=2, seed=6)
    pu, pd = build_index(tmp_path, u, d, 26, name="h26")
    ix = cq.Index(pu, pd, device=0)
    oi = oracle_lib.OracleIndex(pu, pd)
    for rl in (100, 150, 101, 125, 151, (26, 255), 250):
        reads = synth.simulate_reads(gen, 9001, rl, 0.01, 4, frac_random=0.1)
        b, o = synth.concat_reads(reads)
        tight, lens, sk = cq.pack_reads_tight(b, o, 26)
        assert sk == 0
        for mode in (cq.MODE_P, cq.MODE_SC):
            ref = oi.query(b, o, len(gen), mode=mode, nthreads=8)
            got = ix.query_packed_tight(tight, lens, 0, len(gen), mode=mode)
            assert_same(got, ref, f"tight door, rl={rl}, mode={mode}, fused={fused}", rcount=(mode == cq.MODE_P))
            assert got["pairs"] == ref["pairs"]


def test_large_rcount_comes_back_through_the_bounce_buffers(tmp_path):
    """rcount arrays in pageable memory take the pinned double-buffered D2H path: rcount_u here is larger
    than one 16 MiB bounce buffer (several pieces), rcount_d larger than the 1 MiB direct-copy limit."""
    from cammiq_amd import bigsynth
    G = 200
    w = bigsynth.World(seed=5, n_genomes=G, genome_len=1_200_000, pair_share=0.3)
    pu, pd = str(tmp_path / "index_u.bin1"), str(tmp_path / "index_d.bin2")
    nu, nd = w.write_index(pu, pd)
    assert nu * 4 > (16 << 20) and nd * 4 > (1 << 20)
    b, o = w.reads(seed=3, n=60000, length=100)
    ix = cq.Index(pu, pd, device=0)
    got = ix.query(b, o, G)
    ref = oracle_lib.OracleIndex(pu, pd).query(b, o, G, nthreads=8)
    assert_same(got, ref, "bounce")
    packed, lens, _ = cq.pack_reads(b, o, 26)
    assert_same(ix.query_packed(packed, lens, 100, G, out=ix.counts_out(G, pinned=True)), ref, "pinned out")


@pytest.mark.parametrize("how", ["ring", "escapes_overrun", "off"])
def test_rcount_comes_back_narrow_and_exact(tmp_path, monkeypatch, how):
    """rcount crosses the link as one byte per leaf + an escape list for counts of 255 and more and is widened into the
    caller's uint32 arrays by the library (cq_api.cpp narrow_start): bit-exact with the oracle with leaves forced far
    past 255 (every read of a block repeated 700 times), through many small segments (each written to page-locked host memory and flagged by its own
    workgroup, one straddling the u / d boundary), into pinned and into pageable arrays, for every host-fed door; when
    the escape list overruns (forced: 3 entries) the plain uint32 copy takes over, silently and exactly."""
    from cammiq_amd import bigsynth
    G = 40
    w = bigsynth.World(seed=11, n_genomes=G, genome_len=400_000, pair_share=0.3)
    pu, pd = str(tmp_path / "index_u.bin1"), str(tmp_path / "index_d.bin2")
    nu, nd = w.write_index(pu, pd)
    b, o = w.reads(seed=4, n=3000, length=100)
    rb = b.reshape(3000, 100)
    hot = np.repeat(rb[:40], 700, axis=0)                       # 40 reads x 700: their leaves count 700 and more
    allb = np.ascontiguousarray(np.concatenate([rb, hot]).ravel())
    n = 3000 + 40 * 700
    allo = np.arange(n + 1, dtype=np.uint64) * np.uint64(100)
    ref = oracle_lib.OracleIndex(pu, pd).query(allb, allo, G, nthreads=8)
    assert max(int(ref["rcount_u"].max()), int(ref["rcount_d"].max())) >= 700
    assert int((ref["rcount_u"] >= 255).sum() + (ref["rcount_d"] >= 255).sum()) > 3
    monkeypatch.setenv("CAMMIQ_NARROW_FROM", "1")
    if how == "ring":
        monkeypatch.setenv("CAMMIQ_NARROW_SEG", "4096")       # (nu + nd) / 4096 segments, each flagged by its workgroup
        assert (nu + nd) // 4096 > 24 and nu % 4096 != 0         # ... and one of them straddles the u / d boundary
    elif how == "escapes_overrun":
        monkeypatch.setenv("CAMMIQ_ESC_CAP", "3")
    else:
        monkeypatch.setenv("CAMMIQ_RCOUNT_NARROW", "0")
    ix = cq.Index(pu, pd, device=0)
    assert_same(ix.query(allb, allo, G), ref, f"{how}: ascii door, pageable out")
    packed, lens, _ = cq.pack_reads(allb, allo, 26)
    assert_same(ix.query_packed(packed, lens, 100, G, out=ix.counts_out(G, pinned=True)), ref, f"{how}: packed door, pinned out")
    tight, tl, _ = cq.pack_reads_tight(allb, allo, 26)
    out = ix.counts_out(G, pinned=True)
    out.ru[:] = 0xABABABAB                                      # stale contents must be overwritten, every entry
    out.rd[:] = 0xABABABAB
    assert_same(ix.query_packed_tight(tight, tl, 100, G, out=out), ref, f"{how}: tight door, pinned out")
    assert_same(ix.query_packed_tight(tight, tl, 100, G), ref, f"{how}: tight door again (buffers reused)")
    m = cq.Multi(pu, pd, [0, 0])
    assert_same(m.query_packed_tight(tight, tl, 100, G), ref, f"{how}: two shards")


@pytest.mark.parametrize("narrow_from", ["1", None])
def test_rcount_fetch_brings_a_device_array_back(tmp_path, monkeypatch, narrow_from):
    """cq_rcount_fetch: the device door's way back for rcount (what cq_query_device accumulated into) -- narrow over the
    link when the index has enough leaves (forced here by CAMMIQ_NARROW_FROM=1), the plain copy otherwise; into
    page-locked and into plain arrays; ordered behind the caller's stream."""
    import torch
    from cammiq_amd import bigsynth
    if narrow_from:
        monkeypatch.setenv("CAMMIQ_NARROW_FROM", narrow_from)
        monkeypatch.setenv("CAMMIQ_NARROW_SEG", "8192")
    G = 60
    w = bigsynth.World(seed=21, n_genomes=G, genome_len=300_000, pair_share=0.3)
    pu, pd = str(tmp_path / "index_u.bin1"), str(tmp_path / "index_d.bin2")
    nu, nd = w.write_index(pu, pd)
    b, o = w.reads(seed=6, n=40_000, length=100)
    hot = np.repeat(b.reshape(-1, 100)[:20], 400, axis=0)       # leaves past 255: the escape list
    b = np.ascontiguousarray(np.concatenate([b.reshape(-1, 100), hot]).ravel())
    n = 40_000 + 8_000
    o = np.arange(n + 1, dtype=np.uint64) * np.uint64(100)
    ref = oracle_lib.OracleIndex(pu, pd).query(b, o, G, nthreads=8)
    ix = cq.Index(pu, pd, device=0)
    packed, lens, _ = cq.pack_reads(b, o, 26)
    dp = torch.from_numpy(packed.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    ctr = torch.zeros(ix.counter_words(G), dtype=torch.int64, device="cuda")
    rc = torch.zeros(nu + nd, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ix.query_device(cq.MODE_P, dp.data_ptr(), dl.data_ptr(), n, packed.shape[1], 100, G, ctr.data_ptr(), rc.data_ptr(), st)
    for pinned in (True, False):
        ru = cq.host_array(nu, np.uint32) if pinned else np.full(nu, 0xABABABAB, np.uint32)
        rd = cq.host_array(nd, np.uint32) if pinned else np.full(nd, 0xABABABAB, np.uint32)
        ix.rcount_fetch(rc.data_ptr(), st, ru, rd)              # queued behind the classify kernels on `st`
        assert np.array_equal(ru, ref["rcount_u"]) and np.array_equal(rd, ref["rcount_d"]), f"pinned={pinned}"
    assert int(ref["rcount_u"].max()) >= 400


def test_pair_map_grows_instead_of_failing(tmp_path, monkeypatch):
    """query64_sc's read_cnts_b with more distinct pairs than the device map has slots: the library
    grows the map and classifies again (it used to return CQ_ERR_LIMIT); output arrays that are too
    small are reported with the number needed and nothing is lost."""
    rng = np.random.default_rng(3)
    G = 60
    keys_d, reads = {}, []
    seen = set()
    while len(keys_d) < 400:
        k = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 12))
        a, b_ = sorted(rng.choice(np.arange(1, G + 1), 2, replace=False))
        if k in seen or synth.revcomp(k) in seen or (a, b_) in {(v[0], v[1]) for v in keys_d.values()}:
            continue
        seen.add(k)
        keys_d[k] = (int(a), int(b_), 1, 1)
        reads.append(b"GT" + k + b"GT")
    pu, pd = build_index(tmp_path, {b"ACGTACGTACGTAA": (1, 1)}, keys_d, 10)
    b, o = synth.concat_reads(reads * 3)
    ref = oracle_lib.OracleIndex(pu, pd).query(b, o, G, mode=1)
    assert len(ref["pairs"]) > 300
    monkeypatch.setenv("CAMMIQ_PAIR_SLOTS", "16")      # start with a 16-slot map
    ix = cq.Index(pu, pd, device=0)
    got = ix.query(b, o, G, mode=cq.MODE_SC, pair_cap=1 << 12)
    assert got["pairs"] == ref["pairs"]
    assert_same(got, ref, "grown pair map", rcount=False)
    with pytest.raises(cq.CammiqError) as e:            # arrays too small: told how many, nothing cleared
        ix.query(b, o, G, mode=cq.MODE_SC, pair_cap=8)
    assert e.value.code == -9
    assert ix.count_pairs() == len(ref["pairs"])
    assert ix.fetch_pairs(1 << 12) == ref["pairs"]
    assert ix.count_pairs() == 0
    # a failed query must not leak into the next one ("counters are OVERWRITTEN"): the pairs kept after
    # CQ_ERR_LIMIT live only until the next query on the handle, which starts from an empty map
    with pytest.raises(cq.CammiqError) as e:
        ix.query(b, o, G, mode=cq.MODE_SC, pair_cap=8)
    assert e.value.code == -9
    got = ix.query(b, o, G, mode=cq.MODE_SC, pair_cap=1 << 12)
    assert got["pairs"] == ref["pairs"], "pairs of a failed query were added onto the retry"
    assert_same(got, ref, "retry after CQ_ERR_LIMIT", rcount=False)
    m = cq.Multi(pu, pd, [0, 0])                        # and with shards: every shard's map grows, maps are merged
    got = m.query(b, o, G, mode=cq.MODE_SC, pair_cap=1 << 12)
    assert got["pairs"] == ref["pairs"]
    assert_same(got, ref, "multi grown pair map", rcount=False)
    with pytest.raises(cq.CammiqError) as e:            # same contract through the shards
        m.query(b, o, G, mode=cq.MODE_SC, pair_cap=8)
    assert e.value.code == -9
    got = m.query(b, o, G, mode=cq.MODE_SC, pair_cap=1 << 12)
    assert got["pairs"] == ref["pairs"]


def test_comm_entry_points_world_size_one(tmp_path):
    """cq_comm_unique_id / cq_comm_init_rank / cq_counts_allreduce with one rank: RCCL is initialised
    and the collective runs on the caller's stream (the N > 1 case is the same calls on N processes)."""
    import torch
    g = golden("f_deep")
    b, o = synth.concat_reads(g["reads"])
    ix = cq.Index(g["pu"], g["pd"], device=0)
    uid = cq.comm_unique_id()
    assert len(uid) == 128
    comm = cq.Comm(ix, uid, 0, 1)
    packed, lens, _ = cq.pack_reads(b, o, ix.hash_len)
    dp = torch.from_numpy(packed.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    G = g["G"]
    ctr = torch.zeros(ix.counter_words(G), dtype=torch.int64, device="cuda")
    rc = torch.zeros(sum(ix.n_leaves), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lo, hi = cq.shard_range(len(lens), 0, 1)
    assert (lo, hi) == (0, len(lens))
    ix.query_device(cq.MODE_P, dp.data_ptr(), dl.data_ptr(), hi - lo, packed.shape[1], 255, G, ctr.data_ptr(),
                    rc.data_ptr(), st)
    comm.allreduce_counts(ctr.data_ptr(), ctr.numel(), rc.data_ptr(), rc.numel(), st)
    torch.cuda.synchronize()
    c = ctr.cpu().numpy().astype(np.uint64)
    e = g["exp"]["p"]
    assert list(c[:G + 1]) == e["cnt_u"] and list(c[G + 1:2 * G + 2]) == e["cnt_d"]
    assert int(c[2 * G + 2]) == e["nundet"] and int(c[2 * G + 3]) == e["nconf"]
    assert list(rc.cpu().numpy().view(np.uint32)) == e["rcount_u"] + e["rcount_d"]
    comm.close()


def test_cli_gpus_flag(tmp_path):
    """`cammiq --query --gpus N` / `--devices a,b` (extensions): same TSV as one GPU.  --devices always
    takes the cq_multi_* path, also with a repeated ordinal (two shards on the one GPU of this pool)."""
    import subprocess
    g = golden("survey_F1")
    fq = tmp_path / "q1.fastq"
    synth.write_fastq(str(fq), g["reads"])
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cammiq_amd", "cammiq")
    n = min(2, _n_devices())
    out = tmp_path / "out.txt"
    r = subprocess.run([cli, "--query", "--read_cnts", "-f", os.path.join(g["dir"], "genome_map.out"), "-i", g["pu"], g["pd"],
                        "-q", str(fq), "-o", str(out), "--gpus", str(n)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_text() == "QUERY/TAXID\t1001\t1002\t1003\t1004\nq1.fastq\t412\t317\t404\t417\n"
    out2 = tmp_path / "out2.txt"
    r = subprocess.run([cli, "--query", "--read_cnts", "-f", os.path.join(g["dir"], "genome_map.out"), "-i", g["pu"], g["pd"],
                        "-q", str(fq), "-o", str(out2), "--devices", "0,0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out2.read_text() == out.read_text() and "Number of unlabeled reads: 263." in r.stderr
    r = subprocess.run([cli, "--query", "-f", os.path.join(g["dir"], "genome_map.out"), "-i", g["pu"], g["pd"],
                        "-q", str(fq), "--gpus", "99"], capture_output=True, text=True)
    assert r.returncode != 0 and "range [1, 64]" in r.stderr
