"""Randomised parity campaign: many small worlds with every parameter drawn at random (hash length, key
lengths, relatedness of the genomes, marker density, read lengths and error rate, reads outside the parity
domain, table density, launch chunking, counter placement, entry point, mode) -- HIP path against the oracle,
bit for bit.  Every world is a pure function of its seed; a failure names the seed.

    python -m pytest tests/test_gpu_fuzz.py -m gpu -q                       # 25 s of worlds (the suite's share)
    CAMMIQ_FUZZ_SECONDS=600 CAMMIQ_FUZZ_LOG=gpurun_out/fuzz.log python -m pytest tests/test_gpu_fuzz.py -m gpu -q
    CAMMIQ_FUZZ_SEED=1234 CAMMIQ_FUZZ_WORLDS=1 ...                          # replay one world
"""
import os
import random
import time

import numpy as np
import pytest

import cammiq_amd as cq
from cammiq_amd import synth
import oracle_lib
from util import assert_same, build_index, read_pointers

pytestmark = pytest.mark.gpu

_KNOBS = ("CAMMIQ_MAX_SUB_PER_WAVE", "CAMMIQ_LDS_HIST_MAX", "CAMMIQ_KEYS_PER_BUCKET", "CAMMIQ_BLOCKS_PER_CU",
          "CAMMIQ_PAIR_SLOTS", "CAMMIQ_FAST_R", "CAMMIQ_MINIMIZER_LEN", "CAMMIQ_NO_FIXED_SHAPE", "CAMMIQ_GPU_LAYOUT",
          "CAMMIQ_NARROW_FROM", "CAMMIQ_NARROW_SEG")


def _draw(seed):
    """All parameters of world `seed`."""
    r = random.Random(seed)
    h = r.choice([5, 8, 11, 12, 13, 15, 16, 17, 20, 23, 26, 26, 26, 29, 31])
    k = max(h, r.choice([h, h + 1, h + 4, 12, 20, 26, 31]))
    lmax = min(k + r.choice([0, 1, 5, 20, 60]), 200)
    w = dict(seed=seed, h=h, k=k, lmax=lmax,
             n_clades=r.randint(1, 4), per_clade=r.randint(1, 4), glen=r.choice([400, 900, 1500, 2500]),
             div=r.choice([0.0, 0.01, 0.03, 0.08]), keep_every=r.choice([1, 1, 2, 3, 5]),
             n_reads=r.choice([1, 7, 64, 513, 2000, 5000]), err=r.choice([0.0, 0.005, 0.02, 0.06]),
             frac_random=r.choice([0.0, 0.1, 0.5]), lower_frac=r.choice([0.0, 0.2]),
             n_bad=r.choice([0, 0, 3, 40]), unique_only=r.random() < 0.2,
             extra_genomes=r.choice([0, 0, 1, 37, 8191, 9000]),          # ids the index never names; 9000 > LDS histogram
             route=r.choice(["ascii", "reads", "packed", "tight", "multi", "multi_reads", "multi_packed", "multi_tight"]),
             env={})
    lo = max(h, 1)
    shape = r.choice(["fixed", "ragged", "short", "long"])
    w["rl"] = {"fixed": r.choice([lo, max(lo, 50), 100, 150, 250, 255]),
               "ragged": (lo, 255), "short": (lo, min(255, lo + 6)), "long": (200, 255)}[shape]
    if r.random() < 0.4:
        w["env"]["CAMMIQ_MAX_SUB_PER_WAVE"] = str(r.choice([1, 2, 5, 33]))
    if r.random() < 0.3:
        w["env"]["CAMMIQ_LDS_HIST_MAX"] = r.choice(["0", "64", "1000000"])
    if r.random() < 0.4:
        w["env"]["CAMMIQ_KEYS_PER_BUCKET"] = r.choice(["0.3", "2.0", "3.2", "3.9"])
    if r.random() < 0.2:
        w["env"]["CAMMIQ_BLOCKS_PER_CU"] = r.choice(["1", "2", "4"])
    if r.random() < 0.3:
        w["env"]["CAMMIQ_PAIR_SLOTS"] = r.choice(["16", "64"])               # SC mode has to grow the pair table
    if r.random() < 0.4:
        w["env"]["CAMMIQ_FAST_R"] = r.choice(["4", "8"])                     # reads per wave sub-tile, either way at any length
    # (drawn last: worlds of earlier campaigns keep every other parameter)
    if r.random() < 0.45:
        w["env"]["CAMMIQ_MINIMIZER_LEN"] = r.choice(["9", "13", "15", "17", "18", "18", "19", "21"])   # the table's address length (16 / 18 by size otherwise)
    if r.random() < 0.25:
        w["env"]["CAMMIQ_NO_FIXED_SHAPE"] = "1"                              # h = 26 at 100 / 150 bp: the generic instantiation instead
    # round 4 (drawn last again): where the table is laid out -- on the device with the host builder's image compared
    # byte for byte (verify), on the device, or on the host -- and rcount's narrow way back, through a ring of small pieces
    w["env"]["CAMMIQ_GPU_LAYOUT"] = r.choice(["verify", "verify", "1", "0"])
    if r.random() < 0.5:
        w["env"]["CAMMIQ_NARROW_FROM"] = "1"
        w["env"]["CAMMIQ_NARROW_SEG"] = r.choice(["64", "1024", "65536"])
    return w


def _bad_reads(r, gen, h, n):
    out = []
    for _ in range(n):
        g = gen[r.randrange(len(gen))]
        kind = r.randrange(5)
        if kind == 0:
            out.append(g[:max(h - 1, 0)][:r.randint(0, max(h - 1, 0))])             # shorter than h (possibly empty)
        elif kind == 1:
            L = min(len(g), r.randint(max(h, 2), 255))
            p = r.randrange(L)
            out.append(g[:p] + b"N" + g[p + 1:L])                                   # an N anywhere
        elif kind == 2:
            out.append((g * 2)[:r.randint(256, 400)])                               # longer than 255
        elif kind == 3:
            L = min(len(g), r.randint(max(h, 2), 255))
            out.append(g[:L - 1] + r.choice([b"n", b"R", b"-", b".", b"*"]))      # another non-ACGT byte, at the end
        else:
            out.append(b"")
    return out


def _run_world(w, tmpdir):
    r = random.Random(w["seed"] ^ 0x5EED)
    gen = synth.clade_genomes(w["seed"], w["n_clades"], w["per_clade"], w["glen"], w["div"])
    u, d = synth.select_markers(gen, w["k"], w["lmax"], keep_every=w["keep_every"], seed=w["seed"])
    if w["unique_only"]:
        d = {}
    pu, pd = build_index(tmpdir, u, d, w["h"], name=f"w{w['seed']}", seed=w["seed"])
    if w["unique_only"]:
        os.remove(pd), os.remove(pd + ".aux")
        pd = None
    G = len(gen) + w["extra_genomes"]
    good = synth.simulate_reads(gen, w["n_reads"], w["rl"], w["err"], w["seed"], frac_random=w["frac_random"],
                                lower_frac=w["lower_frac"])
    good = [x for x in good if len(x) >= w["h"]]         # a genome shorter than the drawn length cuts the read
    mixed = list(good)
    for b in _bad_reads(r, gen, w["h"], w["n_bad"]):
        mixed.insert(r.randint(0, len(mixed)), b)
    n_bad = len(mixed) - len(good)
    bm, om = synth.concat_reads(mixed)
    bg, og = synth.concat_reads(good)
    oi = oracle_lib.OracleIndex(pu, pd)
    for kname in _KNOBS:
        os.environ.pop(kname, None)
    os.environ.update(w["env"])
    try:
        multi = w["route"].startswith("multi")
        ix = cq.Multi(pu, pd, [0]) if multi else cq.Index(pu, pd, device=0)
        for mode in (cq.MODE_P, cq.MODE_SC):
            ref = oi.query(bg, og, G, mode=mode)
            if w["route"].endswith("packed"):
                packed, lens, sk = cq.pack_reads(bm, om, w["h"])
                assert sk == n_bad, f"packer skipped {sk}, want {n_bad}"
                ml = int(lens.max()) if len(lens) else 0
                got = ix.query_packed(packed, lens, ml, G, mode=mode)
            elif w["route"].endswith("tight"):
                tight, lens, sk = cq.pack_reads_tight(bm, om, w["h"])
                assert sk == n_bad, f"tight packer skipped {sk}, want {n_bad}"
                got = ix.query_packed_tight(tight, lens, 0, G, mode=mode)
            elif w["route"].endswith("reads"):   # the reference's own arrays: a pointer and a length byte per read
                short = [x for x in mixed if len(x) <= 255]
                bs, os_ = synth.concat_reads(short)
                ptrs, rl8 = read_pointers(bs, os_)
                got = ix.query_reads(ptrs, rl8, G, mode=mode)
                assert got["nskipped"] == n_bad - (len(mixed) - len(short)), f"nskipped {got['nskipped']}"
            else:
                got = ix.query(bm, om, G, mode=mode)
                assert got["nskipped"] == n_bad, f"nskipped {got['nskipped']}, want {n_bad}"
            assert_same(got, ref, f"mode={mode}", rcount=(mode == cq.MODE_P))
            assert got["pairs"] == ref["pairs"], f"mode={mode}: pair map differs"
        ix.close()
    finally:
        for kname in _KNOBS:
            os.environ.pop(kname, None)
    for p in (pu, pd):
        if p:
            os.remove(p), os.remove(p + ".aux")
    return len(good), len(u), len(d)


def _draw_generator(seed):
    """Worlds of the benchmark generator (csrc/cq_synth.cpp): realistic marker density, deep tries, shared blocks."""
    r = random.Random(seed)
    h = r.choice([12, 16, 17, 20, 24, 26, 26, 26, 31])
    k = r.randint(h, 31)
    me = r.choice([8, 20, 69, 69, 100])
    G = r.choice([2, 5, 24, 80, 200])
    markers = r.choice([3_000, 30_000, 150_000])
    w = dict(seed=seed, kind="generator", h=h, k=k, lmax=k + r.choice([0, 3, 24, 40]), marker_every=me, n_genomes=G,
             genome_len=max(markers * me // G, 600), frac_deep=r.choice([0.0, 0.07, 0.5]),
             pair_share=r.choice([0.0, 0.1, 0.5]), block=r.choice([256, 2048, 8192]),
             n_reads=r.choice([20_000, 60_000]), err=r.choice([0.0, 0.01, 0.05]), frac_random=r.choice([0.0, 0.1, 0.6]),
             route=r.choice(["ascii", "reads", "packed", "tight", "multi", "multi_reads", "multi_packed", "multi_tight"]), env={})
    w["rl"] = r.choice([h, 50, 75, 100, 100, 150, 250, 255])
    w["rl"] = min(max(w["rl"], h), w["genome_len"])
    if r.random() < 0.4:
        w["env"]["CAMMIQ_MAX_SUB_PER_WAVE"] = str(r.choice([1, 7, 100]))
    if r.random() < 0.3:
        w["env"]["CAMMIQ_LDS_HIST_MAX"] = r.choice(["0", "1000000"])
    if r.random() < 0.4:
        w["env"]["CAMMIQ_KEYS_PER_BUCKET"] = r.choice(["0.5", "2.0", "3.2", "3.9"])
    if r.random() < 0.3:
        w["env"]["CAMMIQ_PAIR_SLOTS"] = "64"
    if r.random() < 0.4:
        w["env"]["CAMMIQ_FAST_R"] = r.choice(["4", "8"])
    if r.random() < 0.5:
        w["env"]["CAMMIQ_MINIMIZER_LEN"] = r.choice(["14", "17", "18", "18", "20"])   # large tables use 18 by themselves
    if r.random() < 0.25:
        w["env"]["CAMMIQ_NO_FIXED_SHAPE"] = "1"
    w["env"]["CAMMIQ_GPU_LAYOUT"] = r.choice(["verify", "verify", "1", "0"])    # round 4, drawn last: see _draw
    if r.random() < 0.5:
        w["env"]["CAMMIQ_NARROW_FROM"] = "1"
        w["env"]["CAMMIQ_NARROW_SEG"] = r.choice(["64", "4096", "65536"])
    return w


def _run_generator_world(w, tmpdir):
    from cammiq_amd import bigsynth
    world = bigsynth.World(seed=w["seed"], n_genomes=w["n_genomes"], genome_len=w["genome_len"], k=w["k"], h=w["h"],
                           lmax=w["lmax"], marker_every=w["marker_every"], frac_deep=w["frac_deep"],
                           pair_share=w["pair_share"], block=w["block"])
    pu = os.path.join(str(tmpdir), f"g{w['seed']}_u.bin1")
    pd = os.path.join(str(tmpdir), f"g{w['seed']}_d.bin2") if w["pair_share"] else None
    nu, nd = world.write_index(pu, pd)
    G = w["n_genomes"]
    b, o = world.reads(seed=w["seed"] + 1, n=w["n_reads"], length=w["rl"], err=w["err"], frac_random=w["frac_random"])
    oi = oracle_lib.OracleIndex(pu, pd)
    for kname in _KNOBS:
        os.environ.pop(kname, None)
    os.environ.update(w["env"])
    try:
        ix = cq.Multi(pu, pd, [0]) if w["route"].startswith("multi") else cq.Index(pu, pd, device=0)
        for mode in (cq.MODE_P, cq.MODE_SC):
            ref = oi.query(b, o, G, mode=mode, nthreads=8)
            if w["route"].endswith("packed"):
                packed, lens, sk = cq.pack_reads(b, o, w["h"])
                assert sk == 0
                got = ix.query_packed(packed, lens, w["rl"], G, mode=mode, pair_cap=1 << 18)
            elif w["route"].endswith("tight"):
                tight, lens, sk = cq.pack_reads_tight(b, o, w["h"])
                assert sk == 0
                got = ix.query_packed_tight(tight, lens, w["rl"], G, mode=mode, pair_cap=1 << 18)
            elif w["route"].endswith("reads"):
                ptrs, rl8 = read_pointers(b, o)
                got = ix.query_reads(ptrs, rl8, G, mode=mode, pair_cap=1 << 18)
                assert got["nskipped"] == 0
            else:
                got = ix.query(b, o, G, mode=mode, pair_cap=1 << 18)
                assert got["nskipped"] == 0
            assert_same(got, ref, f"mode={mode}", rcount=(mode == cq.MODE_P))
            assert got["pairs"] == ref["pairs"], f"mode={mode}: pair map differs"
        ix.close()
    finally:
        for kname in _KNOBS:
            os.environ.pop(kname, None)
    world.close()
    for p in (pu, pd):
        if p:
            os.remove(p), os.remove(p + ".aux")
    return w["n_reads"], nu, nd


def test_fuzz_campaign(tmp_path):
    _campaign(tmp_path, _draw, _run_world, 20260000, 25)


def test_fuzz_campaign_generator_worlds(tmp_path):
    _campaign(tmp_path, _draw_generator, _run_generator_world, 7260000, 20)


def _campaign(tmp_path, _draw, _run_world, default_seed, default_seconds):
    budget = float(os.environ.get("CAMMIQ_FUZZ_SECONDS", str(default_seconds)))
    seed0 = int(os.environ.get("CAMMIQ_FUZZ_SEED", str(default_seed)))
    max_worlds = int(os.environ.get("CAMMIQ_FUZZ_WORLDS", "1000000"))
    log = open(os.environ["CAMMIQ_FUZZ_LOG"], "a") if os.environ.get("CAMMIQ_FUZZ_LOG") else None
    t0 = time.time()
    done = reads = hits_u = hits_d = 0
    while done < max_worlds and (done == 0 or time.time() - t0 < budget):
        w = _draw(seed0 + done)
        try:
            n, nu, nd = _run_world(w, tmp_path)
        except Exception as e:
            msg = f"world seed={w['seed']} failed: {e}\n  parameters: {w}"
            if log:
                log.write(msg + "\n"), log.flush()
            raise AssertionError(msg) from e
        done += 1
        reads += n
        hits_u += nu
        hits_d += nd
        if log and done % 10 == 0:
            log.write(f"{done} worlds, {reads} reads, {time.time() - t0:.0f} s\n"), log.flush()
    if log:
        log.write(f"done: {done} worlds from seed {seed0}, {reads} reads, {hits_u}+{hits_d} markers, all equal to the oracle\n")
        log.close()
    assert done >= 1
