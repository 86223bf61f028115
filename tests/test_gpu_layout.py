"""The table of the flat image laid out ON THE DEVICE (cammiq_amd/csrc/cq_layout_gpu.hip: home buckets, counting sort
into home groups, merge of duplicates, the placement sweep as a prefix maximum) against the host builder
(cq_layout.cpp finish_image_host), which stays the reference: with CAMMIQ_GPU_LAYOUT=verify cq_index_load builds
both and fails unless the device's table equals the host's word for word and the statistics agree.  Replaces
Hash::loadIdx64_p's map64 inserts (/root/reference/src/hashtrie.cpp:486-507)."""
import os

import numpy as np
import pytest

import cammiq_amd as cq
from cammiq_amd import synth
import oracle_lib
from util import assert_same, build_index, golden

pytestmark = pytest.mark.gpu


def _verify_load(monkeypatch, pu, pd, **env):
    monkeypatch.setenv("CAMMIQ_GPU_LAYOUT", "verify")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ix = cq.Index(pu, pd, device=0)          # raises CammiqError(CQ_ERR_FORMAT) on the first differing word
    for k in env:
        monkeypatch.delenv(k)
    return ix


@pytest.mark.parametrize("mlen", [None, "11", "17", "18", "21"])
@pytest.mark.parametrize("name", ["f_deep", "f_flat", "survey_F1", "survey_F2"])
def test_device_layout_equals_host_layout_on_the_fixtures(name, mlen, monkeypatch):
    g = golden(name)
    env = {"CAMMIQ_MINIMIZER_LEN": mlen} if mlen else {}
    ix = _verify_load(monkeypatch, g["pu"], g["pd"], **env)
    monkeypatch.setenv("CAMMIQ_GPU_LAYOUT", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    host = cq.Index(g["pu"], g["pd"], device=0)
    a, b = ix.info_dict(), host.info_dict()
    for k in ("n_keys", "n_table_buckets", "n_overflowed", "max_chain", "minimizer_len", "device_bytes"):
        assert a[k] == b[k], k


@pytest.mark.parametrize("kpb", ["0.3", "1.0", "2.0", "3.9"])
def test_device_layout_dense_and_sparse_tables(tmp_path, monkeypatch, kpb):
    """Table densities from 0.3 to 3.9 keys per 4-slot bucket (long chains, overflow flags everywhere); world 20260038 at
    3.9 is the one whose carry outruns the 64 spill buckets, so the tail has to grow on the device exactly as on the host."""
    gen = synth.clade_genomes(20260038, 3, 2, 400, 0.08)
    u, d = synth.select_markers(gen, 27, 27, keep_every=1, seed=20260038)
    pu, pd = build_index(tmp_path, u, d, 26, name="dense", seed=20260038)
    ix = _verify_load(monkeypatch, pu, pd, CAMMIQ_KEYS_PER_BUCKET=kpb)
    i = ix.info_dict()
    if kpb == "3.9":
        assert i["n_table_buckets"] > int(i["n_keys"] / 3.9) + 1 + 64, "the tail did not have to grow: pick another world"
    reads = synth.simulate_reads(gen, 3000, (30, 200), 0.01, 5, frac_random=0.1)
    b, o = synth.concat_reads(reads)
    assert_same(ix.query(b, o, len(gen)), oracle_lib.OracleIndex(pu, pd).query(b, o, len(gen)), f"kpb {kpb}")


def test_device_layout_merges_duplicates_like_map64(tmp_path, monkeypatch):
    """The same h-mer as a bucket of BOTH tables (one merged slot with val_u and val_d), keys that differ only past the
    hash prefix (one bucket, a trie below it), and depth-0 leaves whose refIDs ride inline in the free value word."""
    rng = np.random.default_rng(5)
    h = 12
    ku, kd = {}, {}
    seen = set()
    while len(ku) < 600:
        k = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), h))
        if k in seen:
            continue
        seen.add(k)
        r = int(rng.integers(1, 30))
        sel = len(ku) % 4
        if sel == 0:                                   # depth-0 leaf in ht_u only
            ku[k] = (r, 1)
        elif sel == 1:                                 # the same h-mer in both tables, depth 0 in both
            ku[k] = (r, 1)
            kd[k] = (r % 29 + 1, (r + 7) % 29 + 1, 1, 1)
        elif sel == 2:                                 # two longer keys below one bucket of ht_u, a depth-0 leaf in ht_d
            ku[k + b"AC"] = (r, 1)
            ku[k + b"GT"] = (r % 29 + 1, 1)
            kd[k] = (3, 9, 1, 1)
        else:                                          # deep in both
            ku[k + b"ACGTA"] = (r, 1)
            kd[k + b"TTG"] = (2, 11, 1, 1)
    pu, pd = build_index(tmp_path, ku, kd, h, name="dup")
    ix = _verify_load(monkeypatch, pu, pd)
    reads = [b"GG" + k + b"CCA" for k in list(ku)[:400]] + [b"TT" + k + b"AAC" for k in list(kd)[:300]]
    b, o = synth.concat_reads(reads)
    for mode in (0, 1):
        got = ix.query(b, o, 30, mode=mode)
        ref = oracle_lib.OracleIndex(pu, pd).query(b, o, 30, mode=mode)
        assert_same(got, ref, f"duplicates, mode {mode}", rcount=(mode == 0))


def test_device_layout_at_a_few_million_keys_and_queries(tmp_path, monkeypatch):
    """A generator world of ~2.4 M markers (three scan levels on the bucket array, deep tries, shared blocks): verify,
    then the device-built handle answers like the oracle; cq_multi builds the table once per shard handle."""
    from cammiq_amd import bigsynth
    G = 200
    w = bigsynth.World(seed=9, n_genomes=G, genome_len=1_000_000, pair_share=0.3)
    pu, pd = str(tmp_path / "index_u.bin1"), str(tmp_path / "index_d.bin2")
    nu, nd = w.write_index(pu, pd)
    assert nu + nd > 2_000_000
    ix = _verify_load(monkeypatch, pu, pd)
    b, o = w.reads(seed=3, n=50_000, length=100)
    ref = oracle_lib.OracleIndex(pu, pd).query(b, o, G, nthreads=8)
    assert_same(ix.query(b, o, G), ref, "device-built table")
    monkeypatch.setenv("CAMMIQ_GPU_LAYOUT", "1")
    m = cq.Multi(pu, pd, [0, 0])
    assert_same(m.query(b, o, G), ref, "two shard handles, each built on the device")
    assert m.shards[0].info_dict()["n_keys"] == ix.info_dict()["n_keys"]


def test_a_home_group_too_large_for_the_device_falls_back_to_the_host_builder(tmp_path, monkeypatch):
    """70 000 keys of 9 bases squeezed into 16 buckets (CAMMIQ_KEYS_PER_BUCKET far above 4 is clamped by the table's
    minimum of 16 buckets): ~4 400 keys per home bucket, more than the device path sorts per group.  The load falls back
    to the host builder, silently and exactly."""
    rng = np.random.default_rng(11)
    h = 9
    codes = rng.choice(4 ** h, size=70_000, replace=False)
    ku = {bytes(b"ACGT"[(int(c) >> (2 * (h - 1 - j))) & 3] for j in range(h)): (int(c) % 50 + 1, 1) for c in codes}
    pu = os.path.join(str(tmp_path), "big_groups_u.bin1")
    synth.write_index(pu, ku, h, False, order_seed=1)
    monkeypatch.setenv("CAMMIQ_GPU_LAYOUT", "1")
    monkeypatch.setenv("CAMMIQ_KEYS_PER_BUCKET", "100000")
    ix = cq.Index(pu, None, device=0)
    i = ix.info_dict()
    assert i["n_keys"] == 70_000 and i["n_table_buckets"] >= 70_000 // 4
    reads = [b"AC" + k + b"GT" for k in list(ku)[:3000]]
    b, o = synth.concat_reads(reads)
    assert_same(ix.query(b, o, 50), oracle_lib.OracleIndex(pu, None).query(b, o, 50), "host fall-back")
