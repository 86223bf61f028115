"""ctypes wrapper of libcq_synth.so: benchmark-scale synthetic genomes / indices / reads
(see csrc/cq_synth.cpp).  Not on the hot path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Params(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_genomes", C.c_uint32), ("genome_len", C.c_uint32),
                ("k", C.c_uint32), ("h", C.c_uint32), ("lmax", C.c_uint32), ("marker_every", C.c_uint32),
                ("block", C.c_uint32), ("frac_deep", C.c_double), ("pair_share", C.c_double)]


def _lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(os.path.join(_HERE, "libcq_synth.so"))
        L.cqs_create.restype = C.c_void_p
        L.cqs_create.argtypes = [C.POINTER(_Params)]
        L.cqs_free.argtypes = [C.c_void_p]
        L.cqs_write_index.restype = C.c_int
        L.cqs_write_index.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.cqs_write_index_with_sub.restype = C.c_int
        L.cqs_write_index_with_sub.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                               C.c_void_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64,
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.cqs_genome_bases.restype = C.c_int
        L.cqs_genome_bases.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p]
        L.cqs_make_reads.restype = C.c_int
        L.cqs_make_reads.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_void_p]
        L.cqs_make_reads_at.restype = C.c_int
        L.cqs_make_reads_at.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double,
                                        C.c_void_p]
        _LIB = L
    return _LIB


class World:
    """G random genomes (+ optional pairwise shared blocks), their marker index, their reads."""

    def __init__(self, seed: int, n_genomes: int, genome_len: int, k: int = 26, h: int = 26, lmax: int = 50,
                 marker_every: int = 69, frac_deep: float = 0.07, pair_share: float = 0.0, block: int = 2048):
        self.p = _Params(seed, n_genomes, genome_len, k, h, lmax, marker_every, block, frac_deep, pair_share)
        self.n_genomes = n_genomes
        self._h = _lib().cqs_create(C.byref(self.p))
        if not self._h:
            raise ValueError("cq_synth: n_genomes < 2^23, h <= k, h <= 31 and lmax <= 255 are required")

    def write_index(self, path_u: str, path_d: str | None = None):
        nu, nd = C.c_uint64(0), C.c_uint64(0)
        rc = _lib().cqs_write_index(self._h, path_u.encode(), path_d.encode() if path_d else None,
                                    C.byref(nu), C.byref(nd))
        if rc != 0:
            raise IOError(f"cannot write {path_u}")
        return int(nu.value), int(nd.value)

    def write_index_with_sub(self, path_u: str, path_d: str | None, hmers: np.ndarray, sub_u: str, sub_d: str | None):
        """The full index AND a sub-index holding exactly its markers whose h-mer is in `hmers` (uint64, any order).
        -> (n_u, n_d, ids_u, ids_d): ids_* = position of each sub-index leaf in the FULL index's decode order.
        Reads all of whose windows (both strands) are in `hmers` classify identically against either index."""
        hv = np.unique(np.ascontiguousarray(hmers, np.uint64))
        ids = np.zeros(2 * len(hv) + 1, np.uint64)
        nu, nd, su, sd = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = _lib().cqs_write_index_with_sub(self._h, path_u.encode(), path_d.encode() if path_d else None, C.byref(nu),
                                             C.byref(nd), hv.ctypes.data_as(C.c_void_p), len(hv), sub_u.encode(),
                                             sub_d.encode() if sub_d else None, ids.ctypes.data_as(C.c_void_p), len(ids),
                                             C.byref(su), C.byref(sd))
        if rc != 0:
            raise IOError(f"cannot write {path_u} / {sub_u} (rc {rc})")
        su, sd = int(su.value), int(sd.value)
        return int(nu.value), int(nd.value), ids[:su].copy(), ids[su:su + sd].copy()

    def genome_bases(self, g: int, start: int, n: int) -> bytes:
        out = np.empty(n, np.uint8)
        if _lib().cqs_genome_bases(self._h, g, start, n, out.ctypes.data_as(C.c_void_p)) != 0:
            raise ValueError("range outside the genome")
        return out.tobytes()

    def reads(self, seed: int, n: int, length: int = 100, err: float = 0.01, frac_random: float = 0.1):
        """-> (bases uint8[n*length], offsets uint64[n+1])"""
        bases = np.empty(n * length, np.uint8)
        rc = _lib().cqs_make_reads(self._h, seed, n, length, err, frac_random, bases.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise ValueError("read length exceeds genome length")
        return bases, np.arange(n + 1, dtype=np.uint64) * np.uint64(length)

    def reads_into(self, out: np.ndarray, seed: int, first: int, n: int, length: int = 100, err: float = 0.01,
                   frac_random: float = 0.1):
        """Reads first .. first+n-1 of stream `seed` into out[:n*length] (a reusable ASCII buffer)."""
        assert out.dtype == np.uint8 and out.size >= n * length and out.flags.c_contiguous
        rc = _lib().cqs_make_reads_at(self._h, seed, first, n, length, err, frac_random, out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise ValueError("read length exceeds genome length")

    def close(self):
        if self._h:
            _lib().cqs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
