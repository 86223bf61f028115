"""ctypes binding to oracle/liboracle.so -- the CHECKER (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

BRANCHES = ["undet", "U1_P0", "Umulti", "U0_P1", "U1_Pall", "U1_Pconf", "U0_PI1", "U0_Pconf"]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(
                os.path.join(ORACLE_DIR, "cammiq_oracle.c")):
            build()
        L = C.CDLL(so)
        L.cqo_load.restype = C.c_void_p
        L.cqo_load.argtypes = [C.c_char_p, C.c_char_p]
        L.cqo_free.argtypes = [C.c_void_p]
        L.cqo_hash_len.restype = C.c_uint32
        L.cqo_hash_len.argtypes = [C.c_void_p]
        L.cqo_num_leaves.restype = C.c_uint64
        L.cqo_num_leaves.argtypes = [C.c_void_p, C.c_int]
        L.cqo_num_buckets.restype = C.c_uint64
        L.cqo_num_buckets.argtypes = [C.c_void_p, C.c_int]
        L.cqo_max_refid.restype = C.c_uint32
        L.cqo_max_refid.argtypes = [C.c_void_p]
        L.cqo_leaves.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.cqo_query.restype = C.c_int64
        L.cqo_query.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                C.c_uint32] + [C.c_void_p] * 9 + [C.c_uint64, C.c_void_p]
        L.cqo_query_variant.restype = C.c_int64
        L.cqo_query_variant.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.c_uint32] + [C.c_void_p] * 9 + [C.c_uint64, C.c_void_p]
        L.cqo_omp_max_threads.restype = C.c_int
        L.cqo_last_loop_seconds.restype = C.c_double
        _LIB = L
    return _LIB


VARIANTS = ["serial", "critical", "atomic", "thread_local"]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleIndex:
    def __init__(self, path_u: str, path_d: str | None):
        self.h = lib().cqo_load(path_u.encode(), path_d.encode() if path_d else None)
        if not self.h:
            raise RuntimeError(f"oracle: cannot load {path_u} / {path_d}")
        self.hash_len = lib().cqo_hash_len(self.h)
        self.n_leaves = [lib().cqo_num_leaves(self.h, t) for t in (0, 1)]
        self.n_buckets = [lib().cqo_num_buckets(self.h, t) for t in (0, 1)]
        self.max_refid = lib().cqo_max_refid(self.h)

    def leaves(self, table: int):
        n = self.n_leaves[table]
        r1 = np.zeros(n, np.uint32); r2 = np.zeros(n, np.uint32)
        dp = np.zeros(n, np.uint8); c1 = np.zeros(n, np.uint16); c2 = np.zeros(n, np.uint16)
        lib().cqo_leaves(self.h, table, _p(r1), _p(r2), _p(dp), _p(c1), _p(c2))
        return dict(refID1=r1, refID2=r2, depth=dp, ucount1=c1, ucount2=c2)

    def query(self, bases: np.ndarray, offsets: np.ndarray, n_genomes: int, mode: int = 0,
              nthreads: int = 1, variant: str | None = None):
        """nthreads: 1 serial (query64_p); > 1 OpenMP with one global critical section (query64mt_p);
        < 0 atomics.  variant ("serial" | "critical" | "atomic" | "thread_local") overrides that choice and
        takes |nthreads| threads: "thread_local" = per-thread counters merged after the loop."""
        n = len(offsets) - 1
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        cu = np.zeros(n_genomes + 1, np.uint64); cd = np.zeros(n_genomes + 1, np.uint64)
        ru = np.zeros(self.n_leaves[0], np.uint32); rd = np.zeros(self.n_leaves[1], np.uint32)
        scal = np.zeros(2, np.uint64); br = np.zeros(8, np.uint64)
        cap = 1 << 16
        pa = np.zeros(cap, np.uint32); pb = np.zeros(cap, np.uint32); pc = np.zeros(cap, np.uint64)
        npairs = np.zeros(1, np.uint64)
        if variant is None:
            rc = lib().cqo_query(self.h, mode, nthreads, _p(bases), _p(offsets), n, n_genomes,
                                 _p(cu), _p(cd), _p(ru), _p(rd), _p(scal), _p(br),
                                 _p(pa), _p(pb), _p(pc), cap, _p(npairs))
        else:
            v = VARIANTS.index(variant)
            rc = lib().cqo_query_variant(self.h, mode, abs(nthreads), v, _p(bases), _p(offsets), n, n_genomes,
                                         _p(cu), _p(cd), _p(ru), _p(rd), _p(scal), _p(br),
                                         _p(pa), _p(pb), _p(pc), cap, _p(npairs))
        if rc != 0:
            raise ValueError(f"oracle: read {-rc - 1} outside the parity domain" if rc > -10**9
                             else "oracle: refID above n_genomes")
        k = int(npairs[0])
        pairs = {(int(pa[i]), int(pb[i])): int(pc[i]) for i in range(min(k, cap))}
        return dict(cnt_u=cu, cnt_d=cd, rcount_u=ru, rcount_d=rd, nundet=int(scal[0]),
                    nconf=int(scal[1]), branch=dict(zip(BRANCHES, (int(x) for x in br))), pairs=pairs,
                    loop_s=float(lib().cqo_last_loop_seconds()))   # the loop over the reads alone ("Time for query" bracket)

    def close(self):
        if self.h:
            lib().cqo_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
