"""CPU-side tests of the product: the C ABI loads and exports what include/cammiq_hip.h
declares, the decoder + flat layout + packer are right, errors come back as status codes.
No compute call is made here (no GPU in this container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cammiq_amd as cq
from cammiq_amd import binding, synth
import oracle_lib
import pyref
from util import build_index, golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cammiq_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cq_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = C.CDLL(binding.lib_path())
    for name in declared:
        assert hasattr(L, name), f"{name} declared in cammiq_hip.h but not exported"
    assert declared == set(binding.SIGNATURES), (declared ^ set(binding.SIGNATURES))
    assert binding.lib().cq_abi_version() == 6


def test_oracle_is_not_linked_into_the_product():
    out = os.popen(f"nm -D {binding.lib_path()}").read()
    assert "cqo_" not in out
    src = os.path.join(ROOT, "cammiq_amd")
    for dp, _, fs in os.walk(src):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "cqo_" not in txt, f


@pytest.mark.parametrize("mlen", [None, "11", "17", "18", "21"])
@pytest.mark.parametrize("name", ["f_deep", "f_flat", "survey_F1", "survey_F2"])
def test_decoder_matches_oracle_and_pyref(name, mlen, monkeypatch):
    """mlen: the minimizer length the table is addressed by is a property of the index (cq_device.h: 16, or 18 for
    large tables; CAMMIQ_MINIMIZER_LEN overrides).  Whatever it is, the table must find exactly the file's buckets."""
    g = golden(name)
    if mlen:
        monkeypatch.setenv("CAMMIQ_MINIMIZER_LEN", mlen)
    ix = cq.Index(g["pu"], g["pd"], device=-1)
    assert ix.info_dict()["minimizer_len"] == min(ix.hash_len, int(mlen) if mlen else 16)
    oi = oracle_lib.OracleIndex(g["pu"], g["pd"])
    assert ix.hash_len == oi.hash_len and ix.n_leaves == oi.n_leaves
    assert list(ix.info.n_file_buckets) == oi.n_buckets
    for t in (0, 1):
        a, b = ix.leaves(t), oi.leaves(t)
        for f in ("refID1", "refID2", "depth", "ucount1", "ucount2"):
            assert np.array_equal(a[f], b[f]), (t, f)
    # the table finds every bucket, with the right root code for depth-0 keys, and nothing else
    h = ix.hash_len
    nu = ix.n_leaves[0]
    hv = lambda key: int("".join("{:02b}".format(synth.SYM[c]) for c in key[:h]), 2)
    tabs = [pyref.decode_index(g["pu"])[2], pyref.decode_index(g["pd"])[2] if g["pd"] else []]
    present = [set(), set()]
    for t in (0, 1):
        for i, (key, *_r) in enumerate(tabs[t]):
            present[t].add(hv(key))
            if len(key) == h:
                code = ix.probe(hv(key))[t]
                assert code == (0x80000000 | (i + (nu if t else 0)))
    for k in present[0] | present[1]:
        cu, cd, chain = ix.probe(k)
        assert (cu != 0) == (k in present[0]) and (cd != 0) == (k in present[1])
        assert 1 <= chain <= ix.info.max_chain
    rng = np.random.default_rng(1)
    for k in rng.integers(0, 1 << (2 * h), size=5000):
        cu, cd, _ = ix.probe(int(k))
        assert (cu != 0) == (int(k) in present[0]) and (cd != 0) == (int(k) in present[1])
    i = ix.info_dict()
    assert i["n_keys"] == len(present[0] | present[1])
    assert i["n_overflowed"] < 0.5 * i["n_table_buckets"] and i["max_chain"] < 64


def test_duplicate_bucket_later_wins(tmp_path):
    """map64[bucket] = root overwrites (hashtrie.cpp:500): craft a file with the same bucket twice."""
    s = synth._Sink()
    s.bit(0); s.bits(7, 64); s.bits(8, 6)
    hv = int("000110110001", 2)  # ACGTAC
    for rid in (7, 9):
        s.u64(hv)
        s.bit(1)
        for _ in range(4):
            s.bit(0)
        s.u32(rid); s.u16(1)
    s.flush64()
    p = str(tmp_path / "dup.bin1")
    open(p, "wb").write(bytes(s.ints)); open(p + ".aux", "wb").write(bytes(s.aux))
    ix = cq.Index(p, None, device=-1)
    assert ix.n_leaves == [2, 0] and list(ix.leaves(0)["refID1"]) == [7, 9]
    assert ix.probe(hv)[0] == (0x80000000 | 1)
    oi = oracle_lib.OracleIndex(p, None)
    b, o = synth.concat_reads([b"ACGTACGG"])
    assert list(map(int, oi.query(b, o, 9)["rcount_u"])) == [0, 1]


def test_load_errors(tmp_path):
    with pytest.raises(cq.CammiqError) as e:
        cq.Index(str(tmp_path / "nope.bin1"), None, device=-1)
    assert e.value.code == -2 and "Cannot open file" in str(e.value)
    pu, pd = build_index(tmp_path, {b"ACGTACG": (1, 1)}, {b"ACGTACGA": (1, 2, 1, 1)}, 6, "m")
    # truncated byte stream
    raw = open(pu, "rb").read()
    open(pu, "wb").write(raw[:11])
    with pytest.raises(cq.CammiqError) as e:
        cq.Index(pu, None, device=-1)
    assert e.value.code == -3
    open(pu, "wb").write(raw)
    # truncated bit stream
    aux = open(pu + ".aux", "rb").read()
    open(pu + ".aux", "wb").write(aux[:2])
    with pytest.raises(cq.CammiqError) as e:
        cq.Index(pu, None, device=-1)
    assert e.value.code == -3
    open(pu + ".aux", "wb").write(aux)
    # hash length mismatch between the two files (assert at query.cpp:460)
    pu2, pd2 = build_index(tmp_path, {b"ACGTACG": (1, 1)}, {b"ACGTACGA": (1, 2, 1, 1)}, 7, "m7")
    with pytest.raises(cq.CammiqError) as e:
        cq.Index(pu, pd2, device=-1)
    assert e.value.code == -4
    # header option != 64
    open(pu + ".aux", "wb").write(bytes([0x3F]) + aux[1:])
    with pytest.raises(cq.CammiqError) as e:
        cq.Index(pu, None, device=-1)
    assert e.value.code == -3


def test_bucket_index_limit_is_an_error_not_a_wrap(tmp_path, monkeypatch):
    """Bucket numbers are 32-bit on the device: a table that would need 2^32 buckets or more is refused with
    CQ_ERR_LIMIT and a message, before anything is allocated (the density knob makes a tiny index ask for one)."""
    pu, pd = build_index(tmp_path, {b"ACGTACG": (1, 1), b"TTGTACG": (2, 1)}, {b"ACGTACGA": (1, 2, 1, 1)}, 6, "lim")
    monkeypatch.setenv("CAMMIQ_KEYS_PER_BUCKET", "1e-10")
    with pytest.raises(cq.CammiqError) as e:
        cq.Index(pu, pd, device=-1)
    assert e.value.code == -9 and "2^32 buckets" in str(e.value)
    monkeypatch.setenv("CAMMIQ_KEYS_PER_BUCKET", "0.01")          # a sparse but legal table still loads and probes
    ix = cq.Index(pu, pd, device=-1)
    assert ix.info.n_table_buckets >= 300


def test_no_cpu_fallback(tmp_path):
    """A host-only handle must refuse to classify; there is no CPU path to fall back to."""
    pu, pd = build_index(tmp_path, {b"ACGTACG": (1, 1)}, {}, 6, "n")
    ix = cq.Index(pu, None, device=-1)
    b, o = synth.concat_reads([b"ACGTACGT"])
    with pytest.raises(cq.CammiqError) as e:
        ix.query(b, o, 1)
    assert e.value.code == -6


def test_packer():
    reads = [b"ACGT" * 8, b"acgtACGTTTGA", b"ACGTN", b"AC", b"T" * 255, b"G" * 256, b"ACGT\xe6ACGT", b""]
    b, o = synth.concat_reads(reads)
    h = 4
    packed, lens, sk = cq.pack_reads(b, o, h)
    assert packed.shape == (len(reads), 16)
    assert list(lens) == [32, 12, 0, 0, 255, 0, 0, 0] and sk == 5
    for r, read in enumerate(reads):
        if lens[r] == 0:
            assert not packed[r].any()
            continue
        for j, c in enumerate(read):
            got = (int(packed[r, j >> 4]) >> (30 - 2 * (j & 15))) & 3
            assert got == synth.SYM[c]
        # bits past the end are zero
        full = int.from_bytes(packed[r].astype(">u4").tobytes(), "big")
        assert full & ((1 << (16 * 32 - 2 * len(read))) - 1) == 0
    assert cq.stride_words(100) == 7 and cq.stride_words(150) == 10 and cq.stride_words(255) == 16
    assert cq.stride_words(1) == 1 and cq.stride_words(64) == 4 and cq.stride_words(65) == 5 and cq.stride_words(0) == 1


def test_packer_large_multithreaded():
    rng = np.random.default_rng(0)
    n, L = 70000, 100
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=n * L)]
    offs = (np.arange(n + 1, dtype=np.uint64) * L)
    packed, lens, sk = cq.pack_reads(bases, offs, 26)
    assert sk == 0 and (lens == L).all()
    sym = np.searchsorted(np.frombuffer(b"ACGT", np.uint8), bases).reshape(n, L).astype(np.uint32)
    want = np.zeros((n, cq.stride_words(L)), np.uint32)
    for j in range(L):
        want[:, j >> 4] |= sym[:, j] << np.uint32(30 - 2 * (j & 15))
    assert np.array_equal(packed, want)


def test_packer_fuzz_all_byte_values():
    """Every byte value at every alignment: the SWAR/PEXT fast path must agree with the
    byte-by-byte definition (valid iff all bytes in ACGTacgt; 2-bit codes A0 C1 G2 T3)."""
    rng = np.random.default_rng(5)
    reads = []
    for i in range(4000):
        L = int(rng.integers(5, 256))
        r = np.frombuffer(b"ACGTacgt", np.uint8)[rng.integers(0, 8, size=L)].copy()
        if i % 3 == 0:
            r[int(rng.integers(0, L))] = int(rng.integers(0, 256))
        reads.append(r.tobytes())
    for v in range(256):            # each byte value once, mid-word
        reads.append(b"ACGTACGTACG" + bytes([v]) + b"TTGCA" * 4)
    b, o = synth.concat_reads(reads)
    packed, lens, sk = cq.pack_reads(b, o, 5)
    ok = set(b"ACGTacgt")
    nbad = 0
    for r, read in enumerate(reads):
        valid = all(c in ok for c in read)
        if not valid:
            nbad += 1
            assert lens[r] == 0 and not packed[r].any(), (r, read)
            continue
        assert lens[r] == len(read)
        want = np.zeros(packed.shape[1], np.uint32)
        for j, c in enumerate(read):
            want[j >> 4] |= np.uint32(synth.SYM[c] << (30 - 2 * (j & 15)))
        assert np.array_equal(packed[r], want), r
    assert sk == nbad


@pytest.mark.parametrize("decoder", ["serial", "parallel", "scan"])
def test_loader_survives_corrupted_files(tmp_path, monkeypatch, decoder):
    """Bit flips, truncations and garbage must come back as status codes (the reference asserts
    or walks off its buffers); a file that still loads must decode like the oracle does.  Both
    decoders: the chunked one (scan + parallel chunks) hands every stream its scan does not accept to the
    serial one, so the two must agree on what loads and on the status of what does not."""
    import random
    monkeypatch.delenv("CAMMIQ_DECODE_STEP", raising=False)
    monkeypatch.delenv("CAMMIQ_DECODE_SEG", raising=False)
    if decoder == "parallel":     # serial shape scan, chunks of 5 buckets
        monkeypatch.setenv("CAMMIQ_DECODE_THREADS", "4")
        monkeypatch.setenv("CAMMIQ_DECODE_STEP", "5")
    elif decoder == "scan":       # the shape scan on all cores (prefix sum + prefix minimum), segments of 3 bytes
        monkeypatch.setenv("CAMMIQ_DECODE_THREADS", "4")
        monkeypatch.setenv("CAMMIQ_DECODE_SEG", "3")
    else:
        monkeypatch.setenv("CAMMIQ_DECODE_THREADS", "1")
    gen = synth.clade_genomes(3, 1, 3, 800, 0.05)
    u, d = synth.select_markers(gen, 12, 20, keep_every=2, seed=1)
    pu, pd = build_index(tmp_path, u, d, 10, "fz")
    good = {p: open(p, "rb").read() for p in (pu, pu + ".aux", pd, pd + ".aux")}
    rng = random.Random(7)
    loaded = failed = 0
    outcome = []
    for trial in range(120):
        victim = rng.choice(list(good))
        data = bytearray(good[victim])
        kind = trial % 3
        if kind == 0:
            for _ in range(rng.randrange(1, 4)):
                i = rng.randrange(len(data)); data[i] ^= 1 << rng.randrange(8)
        elif kind == 1:
            data = data[:rng.randrange(0, len(data))]
        else:
            i = rng.randrange(len(data)); data[i:i + 8] = bytes(rng.randrange(256) for _ in range(8))
        open(victim, "wb").write(bytes(data))
        try:
            ix = cq.Index(pu, pd, device=-1)
            loaded += 1
            assert ix.n_leaves[0] >= 0 and ix.info.max_chain < 1000
            outcome.append(("ok", tuple(ix.n_leaves), ix.info_dict()["n_trie_nodes"]))
            ix.close()
        except cq.CammiqError as e:
            failed += 1
            assert e.code in (-2, -3, -4, -9), e
            outcome.append(("err", e.code))
        open(victim, "wb").write(good[victim])
    assert failed > 20 and loaded + failed == 120
    _CORRUPTION_OUTCOMES[decoder] = outcome
    for other in ("parallel", "scan"):
        if "serial" in _CORRUPTION_OUTCOMES and other in _CORRUPTION_OUTCOMES:
            assert _CORRUPTION_OUTCOMES["serial"] == _CORRUPTION_OUTCOMES[other], other


_CORRUPTION_OUTCOMES = {}


@pytest.mark.parametrize("name", ["f_deep", "survey_F2", "f_flat"])
def test_chunked_decoder_equals_serial_decoder(name, tmp_path, monkeypatch):
    """cq_decode.cpp: a scan of the bit stream finds where every chunk of buckets starts (bit position, bucket / leaf /
    node counts), the chunks are decoded by all cores straight into the final arrays.  The finished image --
    leaves in decode order, table, trie nodes -- must be byte-identical to the serial decoder's for any chunking."""
    import shutil
    g = golden(name)
    monkeypatch.setenv("CAMMIQ_IMAGE_CACHE", "1")
    images = []
    # step: buckets per chunk after the serial shape scan; "seg N": the shape scan on all cores, N bytes per segment
    for threads, step in (("1", None), ("2", "1"), ("5", "7"), ("8", "1000"), ("3", "100000000"),
                          ("4", "seg 1"), ("3", "seg 2"), ("8", "seg 7"), ("2", "seg 4096")):
        d = tmp_path / f"t{threads}_s{step}".replace(" ", "_")
        d.mkdir()
        pu = str(d / "index_u.bin1")
        pd = str(d / "index_d.bin2") if g["pd"] else None
        for src, dst in ((g["pu"], pu), (g["pd"], pd)):
            if src:
                shutil.copy2(src, dst)
                shutil.copy2(src + ".aux", dst + ".aux")
        monkeypatch.setenv("CAMMIQ_DECODE_THREADS", threads)
        monkeypatch.delenv("CAMMIQ_DECODE_STEP", raising=False)
        monkeypatch.delenv("CAMMIQ_DECODE_SEG", raising=False)
        if step and step.startswith("seg "):
            monkeypatch.setenv("CAMMIQ_DECODE_SEG", step[4:])
        elif step:
            monkeypatch.setenv("CAMMIQ_DECODE_STEP", step)
        ix = cq.Index(pu, pd, device=-1)
        images.append((ix.info_dict(), open(pu + ".cqimg", "rb").read()))
    for other in images[1:]:
        assert other[0] == images[0][0]
        assert other[1] == images[0][1], "chunked decode differs from the serial decode"


def _snapshot(ix, probes):
    d = ix.info_dict()
    d.pop("device_bytes", None)
    lv = [ix.leaves(t).tobytes() for t in (0, 1)]
    return d, lv, [ix.probe(int(k)) for k in probes]


@pytest.mark.parametrize("name", ["f_deep", "survey_F1"])
def test_image_cache_roundtrip_and_invalidation(name, tmp_path, monkeypatch):
    """CAMMIQ_IMAGE_CACHE=1: a cached load is indistinguishable from a decode,
    and the cache is ignored + rewritten when a source file or the cache itself changes."""
    import shutil
    g = golden(name)
    pu = str(tmp_path / "index_u.bin1")
    pd = str(tmp_path / "index_d.bin2") if g["pd"] else None
    for src, dst in ((g["pu"], pu), (g["pd"], pd)):
        if src:
            shutil.copy(src, dst)
            shutil.copy(src + ".aux", dst + ".aux")
    cache = pu + ".cqimg"
    rng = np.random.default_rng(5)
    base = cq.Index(pu, pd, device=-1)
    h = base.hash_len
    keys = [int(k) for k in rng.integers(0, 1 << (2 * h), 2000)]
    # plus keys known to be present
    tab = pyref.decode_index(pu)[2]
    keys += [int("".join("{:02b}".format(synth.SYM[c]) for c in key[:h]), 2) for key, *_ in tab[:500]]
    want = _snapshot(base, keys)
    assert base.info.reserved_ == 0 and not os.path.exists(cache)      # opt-in: nothing written by default

    monkeypatch.setenv("CAMMIQ_IMAGE_CACHE", "1")
    first = cq.Index(pu, pd, device=-1)
    assert first.info.reserved_ == 0 and os.path.exists(cache)
    assert _snapshot(first, keys) == want
    second = cq.Index(pu, pd, device=-1)
    assert second.info.reserved_ == 1
    assert _snapshot(second, keys) == want

    # damaged cache -> silently rebuilt
    raw = bytearray(open(cache, "rb").read())
    open(cache, "wb").write(raw[:len(raw) - 7])
    third = cq.Index(pu, pd, device=-1)
    assert third.info.reserved_ == 0 and _snapshot(third, keys) == want
    assert os.path.getsize(cache) == len(raw)
    raw2 = bytearray(open(cache, "rb").read())
    raw2[-3] ^= 0x40
    open(cache, "wb").write(raw2)
    assert cq.Index(pu, pd, device=-1).info.reserved_ == 0
    assert cq.Index(pu, pd, device=-1).info.reserved_ == 1

    # source touched -> stale cache ignored
    st = os.stat(pu)
    os.utime(pu, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000_000))
    fourth = cq.Index(pu, pd, device=-1)
    assert fourth.info.reserved_ == 0 and _snapshot(fourth, keys) == want
    assert cq.Index(pu, pd, device=-1).info.reserved_ == 1
    # opting out again never reads it
    monkeypatch.delenv("CAMMIQ_IMAGE_CACHE")
    assert cq.Index(pu, pd, device=-1).info.reserved_ == 0


@pytest.mark.parametrize("name", ["f_deep", "survey_F2"])
def test_parallel_layout_equals_serial_layout(name, tmp_path, monkeypatch):
    """The table is laid out by 256 independent part sweeps plus a boundary fix-up
    (cq_layout.cpp); the image must be byte-identical to the one a single serial sweep gives,
    for any number of workers and any table density (dense tables spill across part borders)."""
    import shutil
    g = golden(name)
    monkeypatch.setenv("CAMMIQ_IMAGE_CACHE", "1")
    images = {}
    for kpb in ("1.0", "3.0", "3.9"):
        for threads in ("1", "3", "8"):
            d = tmp_path / f"k{kpb}_t{threads}"
            d.mkdir()
            pu = str(d / "index_u.bin1")
            pd = str(d / "index_d.bin2") if g["pd"] else None
            for src, dst in ((g["pu"], pu), (g["pd"], pd)):
                if src:
                    shutil.copy2(src, dst)
                    shutil.copy2(src + ".aux", dst + ".aux")
            monkeypatch.setenv("CAMMIQ_LAYOUT_THREADS", threads)
            monkeypatch.setenv("CAMMIQ_KEYS_PER_BUCKET", kpb)
            ix = cq.Index(pu, pd, device=-1)
            raw = open(pu + ".cqimg", "rb").read()
            # skip the header (source mtimes differ between the copies only if copy2 failed to keep them)
            images.setdefault(kpb, []).append((ix.info_dict(), raw))
        ref = images[kpb][0]
        for other in images[kpb][1:]:
            assert other[0] == ref[0]
            assert other[1] == ref[1], f"kpb {kpb}: parallel layout differs from the serial one"
    # the dense layouts really did spill (otherwise the fix-up path was not exercised)
    assert images["3.9"][0][0]["n_overflowed"] > images["1.0"][0][0]["n_overflowed"]


def test_spill_tail_grows_when_a_dense_table_overruns_it(tmp_path, monkeypatch):
    """World 20260038 of the randomised campaign: 3.9 keys per 4-slot bucket on a small table whose keys crowd
    their minimizers' buckets.  The carry outruns the 64 buckets past the hash range; the tail grows by what is
    left (it used to be CQ_ERR_LIMIT) and every key is still found, nothing else is."""
    gen = synth.clade_genomes(20260038, 3, 2, 400, 0.08)
    u, d = synth.select_markers(gen, 27, 27, keep_every=1, seed=20260038)
    pu, pd = build_index(tmp_path, u, d, 26, name="dense", seed=20260038)
    monkeypatch.setenv("CAMMIQ_KEYS_PER_BUCKET", "3.9")
    ix = cq.Index(pu, pd, device=-1)
    monkeypatch.setenv("CAMMIQ_KEYS_PER_BUCKET", "1.0")
    sparse = cq.Index(pu, pd, device=-1)
    i = ix.info_dict()
    assert i["n_keys"] == sparse.info.n_keys
    assert i["n_table_buckets"] > int(i["n_keys"] / 3.9) + 1 + 64, "the tail did not have to grow: pick another world"
    hv = lambda key: int("".join("{:02b}".format(synth.SYM[c]) for c in key[:26]), 2)
    present = {hv(k) for k in u} | {hv(k) for k in d}
    assert len(present) == i["n_keys"]
    for k in present:
        cu, cd, chain = ix.probe(k)
        assert (cu, cd) == sparse.probe(k)[:2] and (cu or cd) and 1 <= chain <= i["max_chain"]
    rng = np.random.default_rng(2)
    for k in rng.integers(0, 1 << 52, size=3000):
        assert ix.probe(int(k))[:2] == (0, 0) or int(k) in present


def test_header_is_plain_c99_and_the_abi_works_from_c(tmp_path):
    """include/cammiq_hip.h compiles as strict C99 and the library is usable from plain C
    (the boundary a cgo / JNI / ctypes binding would sit on)."""
    import subprocess
    exe = str(tmp_path / "abi_c99")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_c99.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "cammiq_amd"), "-lcammiq_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "cammiq_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    g = golden("f_deep")
    r = subprocess.run([exe, g["pu"], g["pd"]], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    ix = cq.Index(g["pu"], g["pd"], device=-1)
    assert r.stdout.split() == ["ok", "hash_len", str(ix.hash_len), "leaves", f"{ix.n_leaves[0]}+{ix.n_leaves[1]}",
                                "keys", str(ix.info.n_keys)]


def test_pack_read_single_row_equals_batch_packer():
    """cq_pack_read (one sequence line at a time, what the CLI's FASTQ loader calls from its threads) gives the
    rows and lengths of cq_pack_reads; hash_len = 1 leaves short reads to the kernel."""
    import ctypes as C
    reads = [b"ACGT" * 8, b"acgtACGTTTGA" * 3, b"ACGTN" * 7, b"AC", b"T" * 255, b"G" * 100, b"ACGT\xe6ACGT" * 4]
    b, o = synth.concat_reads(reads)
    h = 26
    packed, lens, _ = cq.pack_reads(b, o, h)
    sw = packed.shape[1]
    L = binding.lib()
    for i, r in enumerate(reads):
        row = np.full(sw, 0xDEADBEEF, np.uint32)
        ln = C.c_uint8(77)
        buf = np.frombuffer(r, np.uint8)
        assert L.cq_pack_read(buf.ctypes.data_as(C.c_void_p), len(r), h, sw, row.ctypes.data_as(C.c_void_p), C.byref(ln)) == 0
        assert ln.value == lens[i] and np.array_equal(row, packed[i]), i
        assert L.cq_pack_read(buf.ctypes.data_as(C.c_void_p), len(r), 1, sw, row.ctypes.data_as(C.c_void_p), C.byref(ln)) == 0
        ok = all(c in b"ACGTacgt" for c in r) and 1 <= len(r) <= 255
        assert ln.value == (len(r) if ok else 0)
    assert L.cq_pack_read(None, 5, 1, sw, None, None) == -1


def test_tight_rows_are_the_word_rows_cut_to_whole_bytes():
    """cq_pack_reads_tight / cq_pack_read_tight: base j in byte j/4, first base on top -- the word row written most
    significant byte first and cut after ceil(len_max / 4) bytes (25 bytes per 100-bp read over the host link).
    Every byte value, every length 0..300, strides that do not hold the longest read, guard bytes around each row."""
    import ctypes as C
    assert [cq.stride_bytes(x) for x in (0, 1, 4, 5, 100, 150, 255, 9999)] == [1, 1, 1, 2, 25, 38, 64, 64]
    rng = np.random.default_rng(11)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), size=n)) for n in list(range(0, 301, 1))]
    reads += [b"ACGTN" * 7, b"ACGT\xe6ACGT" * 4, bytes(range(256))[:200]]
    b, o = synth.concat_reads(reads)
    for h in (1, 26):
        words, wl, wsk = cq.pack_reads(b, o, h)
        for sb in (64, 25, 7):
            tight, tl, tsk = cq.pack_reads_tight(b, o, h, sb=sb)
            fits = np.array([len(r) <= 4 * sb for r in reads])
            assert np.array_equal(tl, np.where(fits, wl, 0))
            assert tsk == wsk + int((~fits & (wl > 0)).sum())
            be = words.astype(">u4").view(np.uint8).reshape(len(reads), -1)[:, :sb]
            assert np.array_equal(tight[tl > 0], be[tl > 0])
            assert not tight[tl == 0].any()                                  # skipped reads leave zero rows
    L = binding.lib()
    tight, tl, _ = cq.pack_reads_tight(b, o, 26)
    for i, r in enumerate(reads):
        sb = cq.stride_bytes(min(len(r), 255))
        row = np.full(sb + 2, 0xA5, np.uint8)                                # one guard byte on either side
        ln = C.c_uint8(77)
        buf = np.frombuffer(r, np.uint8)
        assert L.cq_pack_read_tight(buf.ctypes.data_as(C.c_void_p), len(r), 26, sb, row[1:].ctypes.data_as(C.c_void_p), C.byref(ln)) == 0
        assert row[0] == 0xA5 and row[-1] == 0xA5, i
        assert ln.value == tl[i] and np.array_equal(row[1:-1], tight[i, :sb]), i
    assert L.cq_pack_read_tight(None, 5, 1, 25, None, None) == -1
    assert L.cq_pack_reads_tight(None, None, 0, 26, 0, None, None, None) == -1


@pytest.mark.parametrize("isa", ["scalar", "bmi2"])
def test_packers_agree_on_every_instruction_set(isa):
    """The packer picks AVX2, BMI2 (PEXT) or the scalar table at run time; the tests above ran on the best one.
    Run them again capped to the others (CAMMIQ_PACK_ISA is read once per process): all must give the rows the
    independent Python packing expects."""
    import subprocess
    import sys
    if os.environ.get("CAMMIQ_PACK_ISA_CHILD"):
        pytest.skip("already inside the capped run")            # never recurse, whatever -k selects
    env = dict(os.environ, CAMMIQ_PACK_ISA=isa, CAMMIQ_PACK_ISA_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "(test_packer or pack_read or tight_rows) and not instruction_set"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
