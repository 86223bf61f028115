"""ctypes binding to libcammiq_hip.so -- the product's C ABI (include/cammiq_hip.h).

Mirrors the reference seam (FqReader::loadIdx_p + query64_p/query64mt_p/query64_sc,
/root/reference/src/query.cpp:109-123, 458-1080) one call per function.  There is no
Python or CPU fallback: if the library is missing or no GPU is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

MODE_P, MODE_SC = 0, 1
DEVICE_NONE = -1
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class CammiqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cammiq_hip error {code}: {msg}")
        self.code = code


class _Info(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("hash_len", C.c_uint32), ("max_refid", C.c_uint32),
                ("device", C.c_int32), ("doubly_flag", C.c_uint32 * 2), ("n_leaves", C.c_uint64 * 2),
                ("n_file_buckets", C.c_uint64 * 2), ("n_trie_nodes", C.c_uint64), ("n_keys", C.c_uint64),
                ("n_table_buckets", C.c_uint64), ("n_overflowed", C.c_uint64), ("max_chain", C.c_uint32),
                ("reserved_", C.c_uint32), ("device_bytes", C.c_uint64)]


class _Counts(C.Structure):
    _fields_ = [("cnt_u", C.c_void_p), ("cnt_d", C.c_void_p), ("rcount_u", C.c_void_p),
                ("rcount_d", C.c_void_p), ("nundet", C.c_uint64), ("nconf", C.c_uint64),
                ("nskipped", C.c_uint64), ("pair_a", C.c_void_p), ("pair_b", C.c_void_p),
                ("pair_cnt", C.c_void_p), ("pair_cap", C.c_uint64), ("n_pairs", C.c_uint64)]


LEAF_DTYPE = np.dtype([("refID1", "<u4"), ("refID2", "<u4"), ("ucount1", "<u2"), ("ucount2", "<u2"),
                       ("depth", "u1"), ("pad", "u1", (3,))])

# name -> (restype, argtypes); tests check every symbol declared in include/cammiq_hip.h is here
SIGNATURES = {
    "cq_abi_version": (C.c_int, []),
    "cq_last_error": (C.c_char_p, []),
    "cq_index_load": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "cq_index_get_info": (C.c_int, [C.c_void_p, C.POINTER(_Info)]),
    "cq_index_leaves": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cq_index_probe": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint32)]),
    "cq_index_free": (None, [C.c_void_p]),
    "cq_query": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                           C.POINTER(_Counts)]),
    "cq_pack_stride_words": (C.c_uint32, [C.c_uint32]),
    "cq_pack_reads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                C.c_void_p, C.POINTER(C.c_uint64)]),
    "cq_counter_words": (C.c_uint64, [C.c_uint32]),
    "cq_query_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                  C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cq_pairs_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                 C.POINTER(C.c_uint64)]),
    "cq_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
}


def lib_path() -> str:
    # CAMMIQ_LIB: kernel-tuning experiments load an alternative build of the same library
    return os.environ.get("CAMMIQ_LIB") or os.path.join(_HERE, "libcammiq_hip.so")


def lib():
    """Load libcammiq_hip.so (built in-tree by __graft_entry__.build()).  Fails loudly."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise CammiqError(-100, f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise CammiqError(rc, (lib().cq_last_error() or b"").decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def stride_words(max_len: int) -> int:
    return int(lib().cq_pack_stride_words(int(max_len)))


def pack_reads(bases: np.ndarray, offsets: np.ndarray, hash_len: int, sw: int | None = None):
    """ASCII -> (packed uint32 [n, sw], lens uint8 [n], n_skipped)."""
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n = len(offsets) - 1
    if sw is None:
        ml = int(np.diff(offsets.astype(np.int64)).clip(max=255).max()) if n else 0
        sw = stride_words(ml)
    packed = np.zeros((n, sw), np.uint32)
    lens = np.zeros(n, np.uint8)
    sk = C.c_uint64(0)
    _check(lib().cq_pack_reads(_p(bases), _p(offsets), n, hash_len, sw, _p(packed), _p(lens), C.byref(sk)))
    return packed, lens, int(sk.value)


class Index:
    """Opaque index handle: replaces FqReader::ht_u / ht_d (query.hpp:57-58)."""

    def __init__(self, path_u: str, path_d: str | None = None, device: int = 0):
        h = C.c_void_p()
        _check(lib().cq_index_load(path_u.encode(), path_d.encode() if path_d else None, device, C.byref(h)))
        self._h = h
        info = _Info()
        _check(lib().cq_index_get_info(self._h, C.byref(info)))
        self.info = info
        self.hash_len = info.hash_len
        self.n_leaves = [int(info.n_leaves[0]), int(info.n_leaves[1])]
        self.max_refid = info.max_refid
        self.device = info.device

    def info_dict(self):
        i = self.info
        return dict(hash_len=i.hash_len, max_refid=i.max_refid, device=i.device,
                    n_leaves=list(i.n_leaves), n_file_buckets=list(i.n_file_buckets),
                    n_trie_nodes=i.n_trie_nodes, n_keys=i.n_keys, n_table_buckets=i.n_table_buckets,
                    n_overflowed=i.n_overflowed, max_chain=i.max_chain, device_bytes=i.device_bytes,
                    doubly_flag=list(i.doubly_flag))

    def leaves(self, table: int) -> np.ndarray:
        out = np.zeros(self.n_leaves[table], LEAF_DTYPE)
        assert LEAF_DTYPE.itemsize == 16
        _check(lib().cq_index_leaves(self._h, table, _p(out)))
        return out

    def probe(self, hv: int):
        cu, cd, ch = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().cq_index_probe(self._h, hv, C.byref(cu), C.byref(cd), C.byref(ch)))
        return cu.value, cd.value, ch.value

    def query(self, bases: np.ndarray, offsets: np.ndarray, n_genomes: int, mode: int = MODE_P,
              pair_cap: int = 1 << 16):
        """One call of query64_p/_mt_p (MODE_P) or query64_sc (MODE_SC) on host ASCII reads."""
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        cu = np.zeros(n_genomes + 1, np.uint64)
        cd = np.zeros(n_genomes + 1, np.uint64)
        ru = np.zeros(self.n_leaves[0], np.uint32)
        rd = np.zeros(self.n_leaves[1], np.uint32)
        pa = np.zeros(pair_cap, np.uint32)
        pb = np.zeros(pair_cap, np.uint32)
        pc = np.zeros(pair_cap, np.uint64)
        c = _Counts()
        c.cnt_u, c.cnt_d = _p(cu).value, _p(cd).value
        c.rcount_u = _p(ru).value if ru.size else None
        c.rcount_d = _p(rd).value if rd.size else None
        c.pair_a, c.pair_b, c.pair_cnt, c.pair_cap = _p(pa).value, _p(pb).value, _p(pc).value, pair_cap
        _check(lib().cq_query(self._h, mode, _p(bases), _p(offsets), n, n_genomes, C.byref(c)))
        k = int(c.n_pairs)
        return dict(cnt_u=cu, cnt_d=cd, rcount_u=ru, rcount_d=rd, nundet=int(c.nundet), nconf=int(c.nconf),
                    nskipped=int(c.nskipped),
                    pairs={(int(pa[i]), int(pb[i])): int(pc[i]) for i in range(k)})

    def counter_words(self, n_genomes: int) -> int:
        return int(lib().cq_counter_words(n_genomes))

    def query_device(self, mode: int, d_packed_ptr: int, d_lens_ptr: int, n_reads: int, sw: int,
                     max_len: int, n_genomes: int, d_counters_ptr: int, d_rcount_ptr: int | None,
                     stream_ptr: int | None = None):
        """Asynchronous classify of HBM-resident packed reads; accumulates into the device
        counter block / rcount array (raw device pointers, e.g. torch tensor .data_ptr())."""
        _check(lib().cq_query_device(self._h, mode, C.c_void_p(d_packed_ptr), C.c_void_p(d_lens_ptr), n_reads,
                                     sw, max_len, n_genomes, C.c_void_p(d_counters_ptr),
                                     C.c_void_p(d_rcount_ptr) if d_rcount_ptr else None,
                                     C.c_void_p(stream_ptr) if stream_ptr else None))

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        _check(lib().cq_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def fetch_pairs(self, pair_cap: int = 1 << 16):
        pa = np.zeros(pair_cap, np.uint32)
        pb = np.zeros(pair_cap, np.uint32)
        pc = np.zeros(pair_cap, np.uint64)
        n = C.c_uint64(0)
        _check(lib().cq_pairs_fetch(self._h, _p(pa), _p(pb), _p(pc), pair_cap, C.byref(n)))
        return {(int(pa[i]), int(pb[i])): int(pc[i]) for i in range(int(n.value))}

    def close(self):
        if getattr(self, "_h", None):
            lib().cq_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
