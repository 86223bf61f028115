"""Shared helpers for the tests: golden fixtures and result comparison."""
import gzip
import json
import os

import numpy as np

from cammiq_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    d = os.path.join(GOLDEN, name)
    reads = gzip.open(os.path.join(d, "reads.txt.gz"), "rb").read().split()
    exp = json.load(open(os.path.join(d, "expected.json")))
    pu = os.path.join(d, "index_u.bin1")
    pd = os.path.join(d, "index_d.bin2")
    if not os.path.exists(pd):
        pd = None
    return dict(dir=d, pu=pu, pd=pd, reads=reads, exp=exp, G=exp["n_genomes"])


def assert_same(got, ref, what="", rcount=True):
    for k in ("cnt_u", "cnt_d") + (("rcount_u", "rcount_d") if rcount else ()):
        a = np.asarray(got[k]).astype(np.uint64)
        b = np.asarray(ref[k]).astype(np.uint64)
        assert a.shape == b.shape, f"{what}: {k} shape {a.shape} vs {b.shape}"
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            raise AssertionError(f"{what}: {k} differs at {bad[:8]} (got {a[bad[:8]]}, want {b[bad[:8]]}), "
                                 f"{len(bad)} of {len(a)}")
    assert got["nundet"] == ref["nundet"], f"{what}: nundet {got['nundet']} vs {ref['nundet']}"
    assert got["nconf"] == ref["nconf"], f"{what}: nconf {got['nconf']} vs {ref['nconf']}"


def build_index(tmpdir, keys_u, keys_d, h, name="ix", seed=0):
    pu = os.path.join(str(tmpdir), f"{name}_u.bin1")
    pd = os.path.join(str(tmpdir), f"{name}_d.bin2")
    synth.write_index(pu, keys_u, h, False, order_seed=seed)
    synth.write_index(pd, keys_d, h, True, order_seed=seed + 1)
    return pu, pd


def hmers_of_reads(bases: np.ndarray, n: int, rl: int, h: int) -> np.ndarray:
    """Every h-mer (forward and reverse complement, 2h-bit packed as the reference's rolling hash packs it,
    query.cpp:482-485) of n fixed-length ACGT reads: the set of keys map64.find can be asked for by these reads."""
    lut = np.full(256, 255, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    codes = lut[np.ascontiguousarray(bases[:n * rl], np.uint8)].reshape(n, rl)
    assert codes.max() < 4
    out = []
    for c in (codes, (3 - codes)[:, ::-1]):          # the read, then its reverse complement (getRC, query.cpp:447-450)
        W = rl - h + 1
        hv = np.zeros((n, W), np.uint64)
        for j in range(h):
            hv = (hv << np.uint64(2)) | c[:, j:j + W].astype(np.uint64)
        out.append(hv.ravel())
    return np.unique(np.concatenate(out))


def read_pointers(bases, offs):
    """(bases, offsets) -> the arrays cq_query_reads takes: one address per read into `bases` (which the caller keeps alive)
    and one length byte per read -- the shape of FqReader::reads[f] / rlengths[f] (query.hpp:35-36).  Reads longer than 255
    cannot be expressed in a length byte (the reference's uint8_t wraps, query.cpp:387): the caller leaves them out."""
    import numpy as np
    lens = np.diff(offs.astype(np.int64))
    assert lens.size == 0 or int(lens.max()) <= 255, "a length byte holds at most 255"
    ptrs = (np.uint64(bases.ctypes.data) + offs[:-1].astype(np.uint64)).astype(np.uint64)
    return np.ascontiguousarray(ptrs), np.ascontiguousarray(lens.astype(np.uint8))
