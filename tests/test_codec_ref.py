"""Pin the on-disk codec against the REAL reference: oracle/_ref/libref_binaryio.so is the
reference's own binaryio.cpp (BitWriter/BitReader), compiled unmodified by oracle/Makefile.

* the Python writer used for every fixture must produce byte-identical files when its call
  trace is replayed through the reference's BitWriter;
* the reference's BitReader must read back from our files exactly what was written;
* the oracle's and the product's readers must decode reference-written files.
"""
import ctypes as C
import os
import random

import numpy as np
import pytest

import cammiq_amd as cq
from cammiq_amd import synth
import oracle_lib
import pyref

REF_SO = os.path.join(oracle_lib.ORACLE_DIR, "_ref", "libref_binaryio.so")
REF_SRC = "/root/reference/src/binaryio.cpp"


def _ensure_ref_built():
    """oracle/_ref is built HERE, when this module is imported (pytest collects before any session fixture runs: a
    skipif evaluated against a file the fixture builds later would silently drop the only reference-pinned layer on
    a fresh checkout).  The tests skip only where neither the reference's sources nor a prebuilt .so exist."""
    if not os.path.exists(REF_SO) and os.path.exists(REF_SRC):
        oracle_lib.build()
    return os.path.exists(REF_SO)


pytestmark = pytest.mark.skipif(not _ensure_ref_built(),
                                reason="no /root/reference and no prebuilt oracle/_ref/libref_binaryio.so")


def _ref():
    L = C.CDLL(REF_SO)
    for f in (L.ref_write_ops, L.ref_read_ops):
        f.restype = C.c_int
        f.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    return L


def _arrays(trace):
    op = np.array([t[0] for t in trace], np.uint8)
    arg = np.array([t[1] for t in trace], np.uint8)
    val = np.array([t[2] for t in trace], np.uint64)
    return op, arg, val


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _keys(seed, doubly, n=300, h=12):
    rng = random.Random(seed)
    keys = {}
    while len(keys) < n:
        L = h + rng.choice([0, 0, 0, 1, 2, 5, 9])
        k = bytes(rng.choice(b"ACGT") for _ in range(L))
        if any(k.startswith(o) or o.startswith(k) for o in keys):
            continue
        keys[k] = (rng.randrange(1, 9), rng.randrange(1, 9), rng.randrange(1, 400), rng.randrange(1, 400)) \
            if doubly else (rng.randrange(1, 9), rng.randrange(1, 60000))
    return keys


@pytest.mark.parametrize("doubly", [False, True])
def test_writer_is_byte_identical_to_reference_bitwriter(tmp_path, doubly):
    keys = _keys(1 + doubly, doubly)
    ours = str(tmp_path / ("ours.bin2" if doubly else "ours.bin1"))
    trace = []
    synth.write_index(ours, keys, 12, doubly, order_seed=7, trace=trace)
    ref = str(tmp_path / ("ref.bin2" if doubly else "ref.bin1"))
    op, arg, val = _arrays(trace)
    assert _ref().ref_write_ops(ref.encode(), _ptr(op), _ptr(arg), _ptr(val), len(op)) == 0
    assert open(ours, "rb").read() == open(ref, "rb").read()
    assert open(ours + ".aux", "rb").read() == open(ref + ".aux", "rb").read()


@pytest.mark.parametrize("doubly", [False, True])
def test_reference_bitreader_reads_back_our_files(tmp_path, doubly):
    keys = _keys(5 + doubly, doubly)
    path = str(tmp_path / "x.bin")
    trace = []
    synth.write_index(path, keys, 12, doubly, order_seed=9, trace=trace)
    # replay as reads: every write op except the trailing flush64 has a read twin
    rtrace = [t for t in trace if t[0] != 5]
    op, arg, want = _arrays(rtrace)
    got = np.zeros(len(op), np.uint64)
    assert _ref().ref_read_ops(path.encode(), _ptr(op), _ptr(arg), _ptr(got), len(op)) == 0
    assert np.array_equal(got, want)
    # and then the terminator: END64 in the byte stream, one-bits in the aux stream
    op2 = np.concatenate([op, np.array([4, 2, 1], np.uint8)])
    arg2 = np.concatenate([arg, np.array([0, 0, 8], np.uint8)])
    got2 = np.zeros(len(op2), np.uint64)
    assert _ref().ref_read_ops(path.encode(), _ptr(op2), _ptr(arg2), _ptr(got2), len(op2)) == 0
    assert int(got2[-3]) == 0xFFFFFFFFFFFFFFFF and int(got2[-2]) == 0xFFFF and int(got2[-1]) == 0xFF


@pytest.mark.parametrize("doubly", [False, True])
def test_all_readers_decode_reference_written_files(tmp_path, doubly):
    """Files produced by the reference's BitWriter itself -> oracle, product, pyref agree
    with what went in."""
    keys = _keys(11 + doubly, doubly, n=500, h=10)
    tmp = str(tmp_path / "tmp.bin")
    trace = []
    synth.write_index(tmp, keys, 10, doubly, order_seed=2, trace=trace)
    ref = str(tmp_path / ("r.bin2" if doubly else "r.bin1"))
    op, arg, val = _arrays(trace)
    assert _ref().ref_write_ops(ref.encode(), _ptr(op), _ptr(arg), _ptr(val), len(op)) == 0
    d, h, leaves = pyref.decode_index(ref)
    assert d == int(doubly) and h == 10 and len(leaves) == len(keys)
    want = {k: v for k, v in keys.items()}
    for key, r1, r2, c1, c2 in leaves:
        assert (want[key] == (r1, r2, c1, c2)) if doubly else (want[key] == (r1, c1))
    # a d-flagged file in the "u" slot is legal for both loaders (the header bit decides)
    oi = oracle_lib.OracleIndex(ref, None)
    pi = cq.Index(ref, None, device=-1)
    a, b = oi.leaves(0), pi.leaves(0)
    assert len(b) == len(keys)
    for f in ("refID1", "refID2", "depth", "ucount1", "ucount2"):
        assert np.array_equal(a[f], b[f]), f
    assert [x[1] for x in leaves] == list(b["refID1"])
    assert [len(x[0]) for x in leaves] == list(b["depth"])


def test_bit_level_fuzz_against_reference(tmp_path):
    """Random op scripts: reference writer -> reference reader is the identity, and our
    Python sink produces the same bytes (covers partial-byte drop and MSB-first order)."""
    rng = random.Random(3)
    for trial in range(20):
        trace = []
        sink = synth._Sink(trace)
        for _ in range(rng.randrange(1, 400)):
            k = rng.randrange(5)
            if k == 0:
                sink.bit(rng.randrange(2))
            elif k == 1:
                c = rng.randrange(1, 33)
                sink.bits(c, rng.randrange(1 << c))
            elif k == 2:
                sink.u16(rng.randrange(1 << 16))
            elif k == 3:
                sink.u32(rng.randrange(1 << 32))
            else:
                sink.u64(rng.randrange(1 << 64))
        path = str(tmp_path / f"f{trial}.bin")
        op, arg, val = _arrays(trace)
        assert _ref().ref_write_ops(path.encode(), _ptr(op), _ptr(arg), _ptr(val), len(op)) == 0
        assert open(path, "rb").read() == bytes(sink.ints)
        assert open(path + ".aux", "rb").read() == bytes(sink.aux)
