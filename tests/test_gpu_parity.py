"""Parity tests proper: the HIP path (through the C ABI) against the oracle, bit for bit.
Run on the GPU box:  python -m pytest tests -m gpu -x -q"""
import os

import numpy as np
import pytest

import cammiq_amd as cq
from cammiq_amd import synth
import oracle_lib
from util import assert_same, build_index, golden, read_pointers

pytestmark = pytest.mark.gpu


def _both(pu, pd, reads, G, mode=0):
    b, o = synth.concat_reads(reads)
    got = cq.Index(pu, pd, device=0).query(b, o, G, mode=mode)
    ref = oracle_lib.OracleIndex(pu, pd).query(b, o, G, mode=mode)
    return got, ref


@pytest.mark.parametrize("name", ["f_deep", "f_flat"])
def test_golden_fixtures(name):
    g = golden(name)
    b, o = synth.concat_reads(g["reads"])
    ix = cq.Index(g["pu"], g["pd"], device=0)
    got = ix.query(b, o, g["G"])
    assert_same(got, g["exp"]["p"], name)
    assert got["nskipped"] == 0
    sc = ix.query(b, o, g["G"], mode=cq.MODE_SC)
    assert_same(sc, g["exp"]["sc"], name + " sc", rcount=False)
    assert sorted([a, b_, c] for (a, b_), c in sc["pairs"].items()) == g["exp"]["sc"]["pairs"]
    assert int(sc["rcount_u"].sum()) == 0
    # a second call on the same handle overwrites, it does not accumulate
    assert_same(ix.query(b, o, g["G"]), g["exp"]["p"], name + " again")


@pytest.mark.parametrize("name", ["f_deep", "f_flat"])
def test_reads_door_takes_the_references_own_arrays(name):
    """cq_query_reads: one pointer and one length byte per read -- FqReader::reads[f] / rlengths[f] as readFastq leaves
    them (query.hpp:35-36, query.cpp:371-393), every read a heap block of its own.  Same counts as the flattened door and
    as the golden vectors, in both modes, on one device and through cq_multi_query_reads; blocks shorter than h, blocks
    with a byte outside ACGTacgt, empty blocks and a NULL block of length 0 are skipped, not read past."""
    import ctypes
    g = golden(name)
    reads = list(g["reads"])
    # every read in a buffer of its own, as the reference allocates them (no terminator, nothing valid behind the block)
    blocks = [ctypes.create_string_buffer(r, len(r)) for r in reads]
    ptrs = np.array([ctypes.addressof(b) for b in blocks], np.uint64)
    rl8 = np.array([len(r) for r in reads], np.uint8)
    ix = cq.Index(g["pu"], g["pd"], device=0)
    got = ix.query_reads(ptrs, rl8, g["G"])
    assert_same(got, g["exp"]["p"], name + " reads door")
    assert got["nskipped"] == 0
    b, o = synth.concat_reads(reads)
    assert_same(ix.query(b, o, g["G"]), got, name + " flattened door")
    sc = ix.query_reads(ptrs, rl8, g["G"], mode=cq.MODE_SC)
    assert_same(sc, g["exp"]["sc"], name + " reads door sc", rcount=False)
    assert sorted([a, b_, c] for (a, b_), c in sc["pairs"].items()) == g["exp"]["sc"]["pairs"]
    # blocks outside the parity domain between the good ones: counts unchanged, every one of them in nskipped
    h = ix.hash_len
    bad = [b"", b"ACGT"[: max(0, min(4, h - 1))], b"N" * (h + 3), reads[0][:h - 1], reads[1][:h] + b"n" + reads[1][h + 1:]]
    bad_blocks = [ctypes.create_string_buffer(x, max(1, len(x))) for x in bad]
    p2 = list(ptrs[:100]) + [ctypes.addressof(x) for x in bad_blocks] + [0] + list(ptrs[100:])
    l2 = list(rl8[:100]) + [len(x) for x in bad] + [0] + list(rl8[100:])
    got2 = ix.query_reads(np.array(p2, np.uint64), np.array(l2, np.uint8), g["G"])
    assert got2["nskipped"] == len(bad) + 1
    assert_same(got2, g["exp"]["p"], name + " reads door with blocks outside the domain")
    # no reads at all: every counter zero, nothing dereferenced
    none = ix.query_reads(np.zeros(0, np.uint64), np.zeros(0, np.uint8), g["G"])
    assert none["nundet"] == 0 and none["nconf"] == 0 and none["nskipped"] == 0 and not none["cnt_u"].any() and not none["rcount_u"].any()
    # several shards (the same device twice: a one-GPU box): cq_multi_query_reads cuts the pointer array, not a buffer
    m = cq.Multi(g["pu"], g["pd"], [0, 0])
    assert_same(m.query_reads(ptrs, rl8, g["G"]), g["exp"]["p"], name + " reads door, two shards")
    m.close()
    del blocks, bad_blocks


def test_image_cache_gives_identical_results(tmp_path, monkeypatch):
    """An index loaded from the CAMMIQ_IMAGE_CACHE file classifies exactly like a decoded one."""
    import shutil
    g = golden("f_deep")
    pu, pd = str(tmp_path / "index_u.bin1"), str(tmp_path / "index_d.bin2")
    for src, dst in ((g["pu"], pu), (g["pd"], pd)):
        shutil.copy(src, dst)
        shutil.copy(src + ".aux", dst + ".aux")
    b, o = synth.concat_reads(g["reads"])
    monkeypatch.setenv("CAMMIQ_IMAGE_CACHE", "1")
    for expect_cached in (0, 1):
        ix = cq.Index(pu, pd, device=0)
        assert ix.info.reserved_ == expect_cached
        assert_same(ix.query(b, o, g["G"]), g["exp"]["p"], f"cached={expect_cached}")
        sc = ix.query(b, o, g["G"], mode=cq.MODE_SC)
        assert sorted([a, b_, c] for (a, b_), c in sc["pairs"].items()) == g["exp"]["sc"]["pairs"]


def test_survey_fixtures():
    """Indices written by the reference's build; numbers as recorded in SURVEY.md 8(c)."""
    g = golden("survey_F1")
    b, o = synth.concat_reads(g["reads"])
    got = cq.Index(g["pu"], g["pd"], device=0).query(b, o, g["G"])
    assert list(map(int, got["cnt_u"])) == g["exp"]["survey"]["cnt_u"]
    assert got["nundet"] == g["exp"]["survey"]["nundet"]
    g = golden("survey_F2")
    b, o = synth.concat_reads(g["reads"])
    got = cq.Index(g["pu"], g["pd"], device=0).query(b, o, g["G"])
    s = g["exp"]["survey"]
    assert got["nundet"] == s["nundet"] and got["nconf"] == s["nconf"]
    assert int(got["rcount_u"].sum()) == s["sum_rcount_u"]
    assert 2 * int(got["rcount_d"].sum()) == s["sum_rcount_d_over_map_sp"]
    ref = oracle_lib.OracleIndex(g["pu"], g["pd"]).query(b, o, g["G"])
    assert_same(got, ref, "survey_F2")


@pytest.mark.parametrize("h,k,lmax", [(5, 12, 20), (12, 12, 12), (16, 20, 40), (17, 20, 30), (26, 26, 50),
                                     (29, 30, 44), (31, 31, 45)])
def test_random_indices_all_hash_lengths(tmp_path, h, k, lmax):
    gen = synth.clade_genomes(100 + h, 2, 3, 2000, 0.04)
    u, d = synth.select_markers(gen, k, lmax, keep_every=3, seed=h)
    pu, pd = build_index(tmp_path, u, d, h)
    reads = synth.simulate_reads(gen, 3000, (max(h, 20), 255), 0.02, h, frac_random=0.15, lower_frac=0.1)
    for mode in (0, 1):
        got, ref = _both(pu, pd, reads, len(gen), mode)
        assert_same(got, ref, f"h={h} mode={mode}", rcount=(mode == 0))
        assert got["pairs"] == ref["pairs"]


def test_unique_only_index(tmp_path):
    gen = synth.clade_genomes(5, 5, 1, 3000, 0.0)
    u, _ = synth.select_markers(gen, 26, 26, keep_every=2, seed=3)
    pu, _pd = build_index(tmp_path, u, {}, 26)
    reads = synth.simulate_reads(gen, 5000, 100, 0.01, 2, frac_random=0.1)
    got, ref = _both(pu, None, reads, len(gen))
    assert_same(got, ref, "unique-only")
    assert int(got["cnt_d"].sum()) == 0 and got["rcount_d"].size == 0


def test_edge_reads(tmp_path):
    """rl == h, rl == 255, mixed lengths, lower case, reads with no window at all."""
    gen = synth.clade_genomes(9, 2, 2, 1500, 0.03)
    u, d = synth.select_markers(gen, 20, 30, keep_every=1, seed=1)
    pu, pd = build_index(tmp_path, u, d, 20)
    g0 = gen[0]
    reads = [g0[10:30], g0[10:30].lower(), synth.revcomp(g0[100:120]), g0[0:255], synth.revcomp(g0[300:555]),
             g0[50:71], g0[400:431].lower(), b"ACGT" * 5, b"A" * 255, b"T" * 20]
    reads += synth.simulate_reads(gen, 500, (20, 255), 0.0, 4)
    got, ref = _both(pu, pd, reads, len(gen))
    assert_same(got, ref, "edge")
    assert int(got["cnt_u"].sum()) + int(got["cnt_d"].sum()) > 0


def test_reads_outside_the_parity_domain_are_skipped(tmp_path):
    gen = synth.clade_genomes(9, 2, 2, 1500, 0.03)
    u, d = synth.select_markers(gen, 20, 30, keep_every=2, seed=1)
    pu, pd = build_index(tmp_path, u, d, 20)
    good = synth.simulate_reads(gen, 300, (20, 200), 0.01, 4)
    bad = [b"ACGT", b"", gen[0][:50] + b"N" + gen[0][51:100], b"A" * 300, gen[1][:19]]
    mixed = good[:100] + bad[:2] + good[100:200] + bad[2:] + good[200:]
    b, o = synth.concat_reads(mixed)
    got = cq.Index(pu, pd, device=0).query(b, o, len(gen))
    bg, og = synth.concat_reads(good)
    ref = oracle_lib.OracleIndex(pu, pd).query(bg, og, len(gen))
    assert_same(got, ref, "skipped")
    assert got["nskipped"] == len(bad)


def test_empty_input(tmp_path):
    pu, pd = build_index(tmp_path, {b"ACGTACG": (1, 1)}, {}, 6)
    b, o = synth.concat_reads([])
    got = cq.Index(pu, pd, device=0).query(b, o, 1)
    assert got["nundet"] == 0 and got["nconf"] == 0 and int(got["cnt_u"].sum()) == 0


def test_dense_markers_take_the_exact_slow_path(tmp_path):
    """Every position a marker (keep_every=1, small k): far more than 16 hits per read, so the
    fast kernel hands every read to the slow kernel.  Results must not change."""
    gen = synth.clade_genomes(31, 1, 3, 1200, 0.05)
    u, d = synth.select_markers(gen, 10, 16, keep_every=1, seed=0)
    pu, pd = build_index(tmp_path, u, d, 8)
    reads = synth.simulate_reads(gen, 700, (60, 255), 0.005, 3)
    got, ref = _both(pu, pd, reads, len(gen))
    assert_same(got, ref, "dense")
    assert int(np.max(ref["rcount_u"])) > 0


def test_every_decision_branch_on_gpu(tmp_path):
    K = [b"AAAAAC", b"AAAACC", b"AAACCC", b"AACCCC", b"ACCCCC", b"CCCCCA", b"CCCCAA"]
    u = {K[0]: (1, 1), K[1]: (2, 1), b"GGGGGGT": (3, 1)}
    d = {K[2]: (1, 2, 1, 1), K[3]: (1, 3, 1, 1), K[4]: (2, 3, 1, 1), K[5]: (4, 5, 1, 1), K[6]: (6, 6, 1, 1)}
    pu, pd = build_index(tmp_path, u, d, 6)
    reads = [b"TTTTTTGTGTGT", b"GTGT" + K[0] + b"GT", K[0] + b"GT" + K[1], b"GT" + K[2] + b"GT",
             K[0] + b"G" + K[2] + b"G" + K[3], K[0] + b"G" + K[4], K[2] + b"G" + K[3], K[2] + b"G" + K[5],
             b"ACGGGGGGT", b"ACGGGGGG", b"GT" + K[6] + b"GT", synth.revcomp(b"GTGT" + K[0] + b"GT")]
    for mode in (0, 1):
        for r in reads:   # one read at a time: any wrong branch shows up by name
            got, ref = _both(pu, pd, [r], 6, mode)
            assert_same(got, ref, f"read {r!r} mode {mode}", rcount=(mode == 0))
            assert got["pairs"] == ref["pairs"]
        got, ref = _both(pu, pd, reads * 50, 6, mode)
        assert_same(got, ref, f"all mode {mode}", rcount=(mode == 0))
        assert got["pairs"] == ref["pairs"]
        assert all(v > 0 for v in ref["branch"].values())


def test_refid_above_n_genomes_is_an_error(tmp_path):
    pu, pd = build_index(tmp_path, {b"ACGTACG": (5, 1)}, {}, 6)
    b, o = synth.concat_reads([b"ACGTACGT"])
    with pytest.raises(cq.CammiqError) as e:
        cq.Index(pu, pd, device=0).query(b, o, 4)
    assert e.value.code == -5


def test_size_independent_properties_at_scale(tmp_path):
    """BASELINE-sized batches are beyond the oracle's reach in seconds; check what must hold
    at any size: shard additivity (the multi-GPU contract), permutation invariance, every
    read lands in exactly one outcome, device API == host API."""
    import torch
    gen = synth.clade_genomes(77, 4, 3, 4000, 0.03)
    u, d = synth.select_markers(gen, 26, 40, keep_every=3, seed=7)
    pu, pd = build_index(tmp_path, u, d, 26)
    G = len(gen)
    base = synth.simulate_reads(gen, 20000, 100, 0.01, 11, frac_random=0.1)
    reads = base * 25                       # 500k reads
    b, o = synth.concat_reads(reads)
    ix = cq.Index(pu, pd, device=0)
    whole = ix.query(b, o, G)
    # oracle on the 20k base; the 25-fold repeat must be exactly 25x
    ref = oracle_lib.OracleIndex(pu, pd).query(*synth.concat_reads(base), G)
    for k in ("cnt_u", "cnt_d", "rcount_u", "rcount_d"):
        assert np.array_equal(whole[k], ref[k] * 25), k
    assert whole["nundet"] == 25 * ref["nundet"] and whole["nconf"] == 25 * ref["nconf"]
    # shards add up
    n = len(reads)
    acc = None
    for lo, hi in ((0, n // 3), (n // 3, n // 3 + 7), (n // 3 + 7, n)):
        part = ix.query(*synth.concat_reads(reads[lo:hi]), G)
        if acc is None:
            acc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in part.items()}
        else:
            for k in ("cnt_u", "cnt_d", "rcount_u", "rcount_d"):
                acc[k] += part[k]
            acc["nundet"] += part["nundet"]; acc["nconf"] += part["nconf"]
    assert_same(acc, whole, "shards")
    # permutation invariance
    rng = np.random.default_rng(0)
    perm = [reads[i] for i in rng.permutation(n)]
    assert_same(ix.query(*synth.concat_reads(perm), G), whole, "permuted")
    # conservation: each read is counted, undetermined or conflicting -- exactly one of them
    counted = sum(ref["branch"][k] for k in ("U1_P0", "U0_P1", "U1_Pall", "U0_PI1"))
    assert counted + ref["nundet"] + ref["nconf"] == len(base)
    # device-resident API (what bench.py and a multi-GPU host use) == host API
    packed, lens, sk = cq.pack_reads(b, o, ix.hash_len)
    assert sk == 0
    dp = torch.from_numpy(packed.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    ctr = torch.zeros(ix.counter_words(G), dtype=torch.int64, device="cuda")
    rc = torch.zeros(sum(ix.n_leaves), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    half = n // 2
    ix.query_device(0, dp.data_ptr(), dl.data_ptr(), half, packed.shape[1], 100, G, ctr.data_ptr(), rc.data_ptr(), st)
    ix.query_device(0, dp[half:].data_ptr(), dl[half:].data_ptr(), n - half, packed.shape[1], 100, G,
                    ctr.data_ptr(), rc.data_ptr(), st)
    torch.cuda.synchronize()
    c = ctr.cpu().numpy().astype(np.uint64)
    got = dict(cnt_u=c[:G + 1], cnt_d=c[G + 1:2 * G + 2], nundet=int(c[2 * G + 2]), nconf=int(c[2 * G + 3]),
               rcount_u=rc.cpu().numpy().view(np.uint32)[:ix.n_leaves[0]],
               rcount_d=rc.cpu().numpy().view(np.uint32)[ix.n_leaves[0]:])
    assert_same(got, whole, "device api")
    assert ix.last_kernel_ms() > 0


@pytest.mark.parametrize("share", [0.0, 0.3])
def test_benchmark_generator_world_matches_oracle(tmp_path, share):
    """The generator bench.py uses (csrc/cq_synth.cpp), at a size the oracle handles in seconds:
    unique-only and unique + doubly-unique, 150 bp reads (configs[4] read length)."""
    from cammiq_amd import bigsynth
    w = bigsynth.World(seed=9, n_genomes=24, genome_len=60000, pair_share=share)
    pu = str(tmp_path / "index_u.bin1")
    pd = str(tmp_path / "index_d.bin2") if share else None
    nu, nd = w.write_index(pu, pd)
    assert nu > 0 and (nd > 0) == bool(share)
    for rl in (100, 150):
        b, o = w.reads(seed=3, n=40000, length=rl)
        got = cq.Index(pu, pd, device=0).query(b, o, 24)
        ref = oracle_lib.OracleIndex(pu, pd).query(b, o, 24, nthreads=8)
        assert_same(got, ref, f"bigsynth share={share} rl={rl}")
        assert got["nskipped"] == 0
        if share:   # query64_sc on the same reads: counters and the (a, b) -> count map of read_cnts_b
            got = cq.Index(pu, pd, device=0).query(b, o, 24, mode=cq.MODE_SC)
            ref = oracle_lib.OracleIndex(pu, pd).query(b, o, 24, mode=1, nthreads=8)
            assert_same(got, ref, f"bigsynth SC share={share} rl={rl}", rcount=False)
            assert got["pairs"] == ref["pairs"] and len(ref["pairs"]) > 10


def test_many_genomes_use_global_counters(tmp_path):
    """n_genomes above the LDS-histogram limit (8191) switches the per-genome counters to
    global atomics; results must not change."""
    gen = synth.clade_genomes(41, 3, 3, 2500, 0.03)
    u, d = synth.select_markers(gen, 22, 34, keep_every=2, seed=2)
    # spread the nine genomes' refIDs over a large id space
    remap = {i + 1: 1 + i * 1700 for i in range(len(gen))}
    u2 = {k: (remap[r], c) for k, (r, c) in u.items()}
    d2 = {k: (remap[a], remap[b], c1, c2) for k, (a, b, c1, c2) in d.items()}
    pu, pd = build_index(tmp_path, u2, d2, 22)
    reads = synth.simulate_reads(gen, 6000, (30, 200), 0.01, 3, frac_random=0.1)
    G = 15000
    for mode in (0, 1):
        got, ref = _both(pu, pd, reads, G, mode)
        assert_same(got, ref, f"G={G} mode={mode}", rcount=(mode == 0))
        assert got["pairs"] == ref["pairs"]
    assert int(ref["cnt_u"][remap[9]]) > 0


def test_full_size_configs1_properties(tmp_path):
    """BASELINE.json configs[1] at full size (500 genomes, ~49 M markers, 10 M x 100 bp reads):
    far beyond the oracle, so check what must hold at any size -- every read lands in exactly
    one outcome, halves add up to the whole (the multi-GPU contract), a repeated call gives the
    same answer, the host API equals the device API -- plus the oracle on a 100 k-read slice."""
    import torch
    from cammiq_amd import bigsynth
    G, n, rl = 500, 10_000_000, 100
    w = bigsynth.World(seed=2, n_genomes=G, genome_len=3_450_000)
    pu = str(tmp_path / "index_u.bin1")
    nu, nd = w.write_index(pu, None)
    assert nu > 45_000_000 and nd == 0
    ix = cq.Index(pu, None, device=0)
    b, o = w.reads(seed=1000, n=n, length=rl)
    whole = ix.query(b, o, G)
    assert whole["nskipped"] == 0
    assert int(whole["cnt_u"].sum()) + whole["nundet"] + whole["nconf"] == n          # conservation (unique-only)
    assert int(whole["rcount_u"].sum()) >= int(whole["cnt_u"].sum())                    # >= 1 leaf per counted read
    again = ix.query(b, o, G)
    assert_same(again, whole, "idempotence")
    half = n // 2
    a1 = ix.query(b[:half * rl], o[:half + 1], G)
    a2 = ix.query(b[half * rl:], o[half:] - o[half], G)
    for k in ("cnt_u", "cnt_d", "rcount_u"):
        assert np.array_equal(a1[k] + a2[k], whole[k]), k
    assert a1["nundet"] + a2["nundet"] == whole["nundet"] and a1["nconf"] + a2["nconf"] == whole["nconf"]
    # device-resident API in one launch
    packed, lens, sk = cq.pack_reads(b, o, ix.hash_len)
    dp = torch.from_numpy(packed.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    ctr = torch.zeros(ix.counter_words(G), dtype=torch.int64, device="cuda")
    rc = torch.zeros(sum(ix.n_leaves), dtype=torch.int32, device="cuda")
    ix.query_device(0, dp.data_ptr(), dl.data_ptr(), n, packed.shape[1], rl, G, ctr.data_ptr(), rc.data_ptr(),
                    torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = ctr.cpu().numpy().astype(np.uint64)
    assert np.array_equal(c[:G + 1], whole["cnt_u"]) and int(c[2 * G + 2]) == whole["nundet"]
    assert np.array_equal(rc.cpu().numpy().view(np.uint32), whole["rcount_u"])
    # and the oracle on a slice
    ns = 100_000
    ref = oracle_lib.OracleIndex(pu, None).query(b[:ns * rl], o[:ns + 1], G, nthreads=16)
    assert_same(ix.query(b[:ns * rl], o[:ns + 1], G), ref, "slice vs oracle")


def test_long_inputs_are_cut_into_several_launches(tmp_path, monkeypatch):
    """The LDS histogram packs cnt_u | cnt_d of a genome into one word, so a workgroup takes at most 32767 reads
    per launch and the launcher cuts longer inputs into several launches.  Force small launches (test knobs) and
    check against the oracle: reads with few hits, reads that overflow into the exact slow path (their indices
    must stay relative to the whole call), a pair leaf (g, g) that bumps cnt_d twice per read."""
    gen = synth.clade_genomes(31, 1, 3, 1200, 0.05)
    u, d = synth.select_markers(gen, 10, 16, keep_every=1, seed=0)        # dense: > 16 hits per read
    d[b"ACGGTTCAAGGT"] = (2, 2, 1, 1)
    pu, pd = build_index(tmp_path, u, d, 8)
    dense = synth.simulate_reads(gen, 300, (60, 255), 0.005, 3)
    sparse = [b"TTTTTTGTGTGTACGGTTCAAGGTAC", b"GTGTGTGTGTGTGTGTGTGTGTGT"] * 9000 + [b"A" * 30] * 2000
    reads = sparse[:7000] + dense[:150] + sparse[7000:] + dense[150:]
    b, o = synth.concat_reads(reads)
    ref = oracle_lib.OracleIndex(pu, pd).query(b, o, len(gen), nthreads=8)
    assert int(ref["cnt_d"][2]) >= 2 * 9000
    monkeypatch.setenv("CAMMIQ_MAX_SUB_PER_WAVE", "1")
    monkeypatch.setenv("CAMMIQ_BLOCKS_PER_CU", "1")                         # 256 x 4 waves x 1 sub-tile x 8 = 8192 reads per launch
    got = cq.Index(pu, pd, device=0).query(b, o, len(gen))
    assert_same(got, ref, "several launches")
    monkeypatch.setenv("CAMMIQ_FAST_R", "4")                                # four reads per sub-tile: 4096 reads per launch
    assert_same(cq.Index(pu, pd, device=0).query(b, o, len(gen)), ref, "several launches, R = 4")


@pytest.mark.parametrize("rl", [100, 255])
def test_reads_per_sub_tile_do_not_change_the_result(tmp_path, monkeypatch, rl):
    """The launcher gives a wave eight reads per sub-tile, or four when eight reads' rows and hash words would
    leave three workgroups or fewer resident (long reads).  Both, and the automatic choice, against the oracle,
    with the per-genome counters in LDS and as global atomics."""
    gen = synth.clade_genomes(52, 3, 3, 3000, 0.03)
    u, d = synth.select_markers(gen, 26, 40, keep_every=2, seed=5)
    pu, pd = build_index(tmp_path, u, d, 26)
    reads = synth.simulate_reads(gen, 20011, rl, 0.01, 8, frac_random=0.1)   # ragged last sub-tile for 4 and for 8
    b, o = synth.concat_reads(reads)
    oi = oracle_lib.OracleIndex(pu, pd)
    for mode in (cq.MODE_P, cq.MODE_SC):
        ref = oi.query(b, o, len(gen), mode=mode, nthreads=8)
        for R in ("8", "4", None):
            for hist in ("1000000", "0"):
                monkeypatch.setenv("CAMMIQ_LDS_HIST_MAX", hist)
                if R:
                    monkeypatch.setenv("CAMMIQ_FAST_R", R)
                else:
                    monkeypatch.delenv("CAMMIQ_FAST_R", raising=False)
                got = cq.Index(pu, pd, device=0).query(b, o, len(gen), mode=mode)
                assert_same(got, ref, f"rl={rl} R={R} hist={hist} mode={mode}", rcount=(mode == cq.MODE_P))
                assert got["pairs"] == ref["pairs"]


@pytest.mark.parametrize("rl", [100, 150, 101, 125, 151])
def test_fixed_shape_instantiation_equals_the_generic_kernel(tmp_path, monkeypatch, rl):
    """h = 26 (CAMMiQ's default) with a batch whose longest read is 100 / 150 bp runs the instantiation that has hash
    length and batch shape folded in as constants; a batch of any other length whose rows are 7, 8 or 10 words (97-112,
    113-128, 145-160 bases: the 101-, 125-, 151-bp reads real FASTQs hold) runs the one with h, m and the row stride
    folded in and the window counts as launch arguments (fixed_read_len = minus the stride); CAMMIQ_NO_FIXED_SHAPE=1
    forces the generic one.  All against the oracle: deep keys, mixed read lengths below the batch maximum (ragged
    windows), both modes, counters in LDS and as global atomics."""
    want_rl = rl if rl in (100, 150) else -((rl + 15) // 16)
    gen = synth.clade_genomes(77, 3, 4, 4000, 0.03)
    u, d = synth.select_markers(gen, 26, 48, keep_every=2, seed=6)
    pu, pd = build_index(tmp_path, u, d, 26)
    reads = synth.simulate_reads(gen, 15003, (26, rl), 0.01, 4, frac_random=0.1)
    reads += synth.simulate_reads(gen, 5000, rl, 0.01, 5)                     # the batch maximum is exactly rl
    b, o = synth.concat_reads(reads)
    oi = oracle_lib.OracleIndex(pu, pd)
    for mode in (cq.MODE_P, cq.MODE_SC):
        ref = oi.query(b, o, len(gen), mode=mode, nthreads=8)
        for nofix in ("0", "1"):
            for hist in ("1000000", "0"):
                monkeypatch.setenv("CAMMIQ_NO_FIXED_SHAPE", nofix)
                monkeypatch.setenv("CAMMIQ_LDS_HIST_MAX", hist)
                ix = cq.Index(pu, pd, device=0)
                got = ix.query(b, o, len(gen), mode=mode)
                li = ix.last_launch_info()
                assert li["fixed_shape"] == (0 if nofix == "1" else 1), li
                assert li["fixed_read_len"] == (0 if nofix == "1" else want_rl) and li["reads_per_subtile"] == 8
                assert li["lds_hist"] == (1 if hist != "0" else 0)
                assert_same(got, ref, f"rl={rl} nofix={nofix} hist={hist} mode={mode}", rcount=(mode == cq.MODE_P))
                assert got["pairs"] == ref["pairs"]
    # another hash length or another batch shape never takes it
    monkeypatch.delenv("CAMMIQ_NO_FIXED_SHAPE")
    ix = cq.Index(pu, pd, device=0)
    ix.query(*synth.concat_reads(reads[:2000] + [b"ACGT" * 50]), len(gen))       # 200 bases: 13-word rows
    assert ix.last_launch_info()["fixed_shape"] == 0


@pytest.mark.parametrize("mlen", ["14", "17", "18", "21"])
def test_minimizer_length_does_not_change_the_result(tmp_path, monkeypatch, mlen):
    """The layout addresses large tables by 18-mer minimizers (36-bit m-mers folded to a 32-bit hash) instead of
    16-mers (cq_device.h); CAMMIQ_MINIMIZER_LEN forces a length on any table.  Generic and fixed-shape
    instantiations (h = 26: 100-bp and 150-bp batches have their own m = 18 builds), deep keys, both modes, against
    the oracle; then another hash length through the generic kernel."""
    monkeypatch.setenv("CAMMIQ_MINIMIZER_LEN", mlen)
    gen = synth.clade_genomes(91, 3, 4, 4000, 0.03)
    u, d = synth.select_markers(gen, 26, 48, keep_every=2, seed=8)
    pu, pd = build_index(tmp_path, u, d, 26)
    oi = oracle_lib.OracleIndex(pu, pd)
    for rl in (100, 150, (30, 220)):
        reads = synth.simulate_reads(gen, 12007, rl, 0.01, 6, frac_random=0.1)
        b, o = synth.concat_reads(reads)
        for mode in (cq.MODE_P, cq.MODE_SC):
            ref = oi.query(b, o, len(gen), mode=mode, nthreads=8)
            ix = cq.Index(pu, pd, device=0)
            assert ix.info_dict()["minimizer_len"] == int(mlen)
            got = ix.query(b, o, len(gen), mode=mode)
            li = ix.last_launch_info()
            assert li["minimizer_len"] == int(mlen)
            assert li["fixed_shape"] == (1 if (mlen == "18" and isinstance(rl, int)) else 0), (li, rl)
            assert_same(got, ref, f"m={mlen} rl={rl} mode={mode}", rcount=(mode == cq.MODE_P))
            assert got["pairs"] == ref["pairs"]
    gen = synth.clade_genomes(5, 2, 3, 2500, 0.04)
    u, d = synth.select_markers(gen, 20, 36, keep_every=2, seed=3)
    pu, pd = build_index(tmp_path, u, d, 19, "h19")
    got, ref = _both(pu, pd, synth.simulate_reads(gen, 6000, (19, 255), 0.01, 2, frac_random=0.1), len(gen))
    assert_same(got, ref, f"h=19 m={mlen}")
