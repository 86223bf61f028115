"""ctypes binding to libcammiq_hip.so -- the product's C ABI (include/cammiq_hip.h).

Mirrors the reference seam (FqReader::loadIdx_p + query64_p/query64mt_p/query64_sc,
/root/reference/src/query.cpp:109-123, 458-1080) one call per function.  There is no
Python or CPU fallback: if the library is missing or no GPU is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

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
                ("reserved_", C.c_uint32), ("device_bytes", C.c_uint64), ("minimizer_len", C.c_uint32),
                ("reserved2_", C.c_uint32)]


class _LaunchInfo(C.Structure):
    _fields_ = [("reads_per_subtile", C.c_int32), ("hit_slots", C.c_int32), ("lds_hist", C.c_int32),
                ("fixed_shape", C.c_int32), ("fixed_hash_len", C.c_int32), ("fixed_read_len", C.c_int32),
                ("blocks_per_cu", C.c_int32), ("minimizer_len", C.c_int32)]


class _Calibration(C.Structure):
    _fields_ = [("gather16_Glines_s", C.c_double), ("gather16_mix_Glines_s", C.c_double), ("clock_MHz_gather", C.c_double),
                ("clock_MHz_mix", C.c_double), ("table_bytes", C.c_double), ("seconds", C.c_double),
                ("gather_blocks_per_cu", C.c_int32), ("mix_blocks_per_cu", C.c_int32),
                ("chase16_Glines_s", C.c_double), ("chase_latency_ns", C.c_double), ("clock_MHz_chase", C.c_double)]


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
    "cq_query_reads": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                 C.POINTER(_Counts)]),
    "cq_pack_stride_words": (C.c_uint32, [C.c_uint32]),
    "cq_pack_reads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                C.c_void_p, C.POINTER(C.c_uint64)]),
    "cq_pack_read": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "cq_counter_words": (C.c_uint64, [C.c_uint32]),
    "cq_query_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                  C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cq_pairs_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                 C.POINTER(C.c_uint64)]),
    "cq_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "cq_last_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "cq_last_launch_info": (C.c_int, [C.c_void_p, C.POINTER(_LaunchInfo)]),
    "cq_query_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                  C.c_uint32, C.c_uint32, C.POINTER(_Counts)]),
    "cq_pack_stride_bytes": (C.c_uint32, [C.c_uint32]),
    "cq_pack_reads_tight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.POINTER(C.c_uint64)]),
    "cq_pack_read_tight": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "cq_query_packed_tight": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                        C.c_uint32, C.c_uint32, C.POINTER(_Counts)]),
    "cq_multi_query_packed_tight": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.POINTER(_Counts)]),
    "cq_calibrate": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cq_rcount_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cq_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "cq_host_free": (None, [C.c_void_p]),
    "cq_pairs_reserve": (C.c_int, [C.c_void_p, C.c_uint64]),
    "cq_shard_range": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cq_multi_load": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "cq_multi_size": (C.c_int, [C.c_void_p]),
    "cq_multi_index": (C.c_void_p, [C.c_void_p, C.c_int]),
    "cq_multi_query": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                 C.POINTER(_Counts)]),
    "cq_multi_query_reads": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                       C.POINTER(_Counts)]),
    "cq_multi_query_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                        C.c_uint32, C.c_uint32, C.POINTER(_Counts)]),
    "cq_multi_free": (None, [C.c_void_p]),
    "cq_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cq_comm_init_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cq_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cq_counts_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]),
    "cq_comm_free": (None, [C.c_void_p]),
}
COMM_ID_BYTES = 128


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
            if not hasattr(L, name) and os.environ.get("CAMMIQ_LIB"):
                continue   # an older experiment build named by CAMMIQ_LIB (tools/kexp.py A/B): calling the entry raises
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


def stride_bytes(max_len: int) -> int:
    return int(lib().cq_pack_stride_bytes(int(max_len)))


def pack_reads_tight(bases: np.ndarray, offsets: np.ndarray, hash_len: int, sb: int | None = None, out=None):
    """ASCII -> (tight rows uint8 [n, sb], lens uint8 [n], n_skipped): 25 bytes per 100-bp read for the host link."""
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n = len(offsets) - 1
    if sb is None:
        ml = int(np.diff(offsets.astype(np.int64)).clip(max=255).max()) if n else 0
        sb = stride_bytes(ml)
    packed, lens = out if out is not None else (np.zeros((n, sb), np.uint8), np.zeros(n, np.uint8))
    assert packed.shape == (n, sb) and packed.flags.c_contiguous and lens.shape == (n,)
    sk = C.c_uint64(0)
    _check(lib().cq_pack_reads_tight(_p(bases), _p(offsets), n, hash_len, sb, _p(packed), _p(lens), C.byref(sk)))
    return packed, lens, int(sk.value)


def shard_range(n_reads: int, rank: int, world: int):
    """[n*p/P, n*(p+1)/P) -- the library's own sharding rule (cq_shard_range)."""
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    _check(lib().cq_shard_range(n_reads, rank, world, C.byref(lo), C.byref(hi)))
    return int(lo.value), int(hi.value)


def host_array(n: int, dtype) -> np.ndarray:
    """numpy array in page-locked host memory (cq_host_alloc); released when the last view dies."""
    dt = np.dtype(dtype)
    p = C.c_void_p()
    nbytes = max(int(n) * dt.itemsize, 1)
    _check(lib().cq_host_alloc(C.byref(p), nbytes))
    buf = (C.c_uint8 * nbytes).from_address(p.value)
    weakref.finalize(buf, lib().cq_host_free, C.c_void_p(p.value))   # every view keeps `buf` alive through .base
    return np.frombuffer(buf, dtype=dt, count=int(n))


class _CountsOut:
    """Caller-owned output arrays of one classify call (cq_counts)."""

    def __init__(self, n_genomes, n_leaves, pair_cap, pinned=False):
        mk = host_array if pinned else (lambda n, dt: np.zeros(n, dt))
        self.cu = np.zeros(n_genomes + 1, np.uint64)
        self.cd = np.zeros(n_genomes + 1, np.uint64)
        self.ru = mk(n_leaves[0], np.uint32)
        self.rd = mk(n_leaves[1], np.uint32)
        self.pa = np.zeros(pair_cap, np.uint32)
        self.pb = np.zeros(pair_cap, np.uint32)
        self.pc = np.zeros(pair_cap, np.uint64)
        c = _Counts()
        c.cnt_u, c.cnt_d = _p(self.cu).value, _p(self.cd).value
        c.rcount_u = _p(self.ru).value if self.ru.size else None
        c.rcount_d = _p(self.rd).value if self.rd.size else None
        c.pair_a, c.pair_b, c.pair_cnt, c.pair_cap = _p(self.pa).value, _p(self.pb).value, _p(self.pc).value, pair_cap
        self.c = c

    def result(self):
        c = self.c
        k = int(c.n_pairs)
        return dict(cnt_u=self.cu, cnt_d=self.cd, rcount_u=np.asarray(self.ru), rcount_d=np.asarray(self.rd),
                    nundet=int(c.nundet), nconf=int(c.nconf), nskipped=int(c.nskipped),
                    pairs={(int(self.pa[i]), int(self.pb[i])): int(self.pc[i]) for i in range(k)})


class Index:
    """Opaque index handle: replaces FqReader::ht_u / ht_d (query.hpp:57-58)."""

    def __init__(self, path_u: str, path_d: str | None = None, device: int = 0, _borrowed=None):
        if _borrowed is not None:
            h = C.c_void_p(_borrowed)
            self._owned = False
        else:
            h = C.c_void_p()
            _check(lib().cq_index_load(path_u.encode(), path_d.encode() if path_d else None, device, C.byref(h)))
            self._owned = True
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
                    doubly_flag=list(i.doubly_flag), minimizer_len=i.minimizer_len)

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
              pair_cap: int = 1 << 16, out: "_CountsOut | None" = None):
        """One call of query64_p/_mt_p (MODE_P) or query64_sc (MODE_SC) on host ASCII reads."""
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        o = out or _CountsOut(n_genomes, self.n_leaves, pair_cap)
        _check(lib().cq_query(self._h, mode, _p(bases), _p(offsets), n, n_genomes, C.byref(o.c)))
        return o.result()

    def query_reads(self, read_ptrs: np.ndarray, rlengths: np.ndarray, n_genomes: int, mode: int = MODE_P,
                    pair_cap: int = 1 << 16, out: "_CountsOut | None" = None):
        """cq_query_reads: the reference's own arrays -- one address per read (uint64 array of pointers to ASCII blocks the
        caller keeps alive) and one length byte per read (FqReader::reads[f] / rlengths[f])."""
        assert read_ptrs.dtype == np.uint64 and read_ptrs.flags.c_contiguous and rlengths.dtype == np.uint8
        o = out or _CountsOut(n_genomes, self.n_leaves, pair_cap)
        _check(lib().cq_query_reads(self._h, mode, _p(read_ptrs), _p(rlengths), len(rlengths), n_genomes, C.byref(o.c)))
        return o.result()

    def query_packed(self, packed: np.ndarray, lens: np.ndarray, max_len: int, n_genomes: int,
                     mode: int = MODE_P, pair_cap: int = 1 << 16, out: "_CountsOut | None" = None):
        """cq_query_packed: pre-packed host rows -> pipelined H2D + classify + D2H of the counters."""
        assert packed.dtype == np.uint32 and packed.flags.c_contiguous and lens.dtype == np.uint8
        o = out or _CountsOut(n_genomes, self.n_leaves, pair_cap)
        _check(lib().cq_query_packed(self._h, mode, _p(packed), _p(lens), len(lens), packed.shape[1], max_len,
                                     n_genomes, C.byref(o.c)))
        return o.result()

    def query_packed_tight(self, packed: np.ndarray, lens: np.ndarray, max_len: int, n_genomes: int,
                           mode: int = MODE_P, pair_cap: int = 1 << 16, out: "_CountsOut | None" = None):
        """cq_query_packed_tight: rows at a byte stride (pack_reads_tight) -> H2D, widened on the device, classified."""
        assert packed.dtype == np.uint8 and packed.flags.c_contiguous and lens.dtype == np.uint8
        o = out or _CountsOut(n_genomes, self.n_leaves, pair_cap)
        _check(lib().cq_query_packed_tight(self._h, mode, _p(packed), _p(lens), len(lens), packed.shape[1], max_len,
                                           n_genomes, C.byref(o.c)))
        return o.result()

    def counts_out(self, n_genomes: int, pair_cap: int = 1 << 16, pinned: bool = True) -> "_CountsOut":
        """Reusable output arrays (rcount in page-locked memory when pinned)."""
        return _CountsOut(n_genomes, self.n_leaves, pair_cap, pinned=pinned)

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

    def rcount_fetch(self, d_rcount_ptr: int, stream_ptr: int | None, rcount_u: np.ndarray, rcount_d: np.ndarray):
        """cq_rcount_fetch: a device rcount array (raw pointer) -> the two host uint32 arrays, narrow over the link."""
        assert rcount_u.dtype == np.uint32 and rcount_d.dtype == np.uint32
        assert len(rcount_u) == self.n_leaves[0] and len(rcount_d) == self.n_leaves[1]
        _check(lib().cq_rcount_fetch(self._h, C.c_void_p(d_rcount_ptr), C.c_void_p(stream_ptr) if stream_ptr else None,
                                     _p(rcount_u) if rcount_u.size else None, _p(rcount_d) if rcount_d.size else None))

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        _check(lib().cq_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def last_kernel_times(self):
        """(main kernel ms, exact slow-path kernel ms) of the most recent launch."""
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().cq_last_kernel_times(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def last_launch_info(self) -> dict:
        """Which instantiation of the classify kernel the most recent launch ran (cq_last_launch_info)."""
        li = _LaunchInfo()
        _check(lib().cq_last_launch_info(self._h, C.byref(li)))
        d = {k: int(getattr(li, k)) for k, _ in _LaunchInfo._fields_}
        fx = f",{d['fixed_hash_len']},{d['fixed_read_len']},{d['minimizer_len']}" if d["fixed_shape"] else ",0,0,0"
        d["kernel"] = f"classify_kernel<{d['reads_per_subtile']},{d['hit_slots']},false{fx}>"
        return d

    def calibrate(self) -> dict:
        """cq_calibrate: this board's random-load ceiling on the handle's own table and the clock it holds (diagnostic)."""
        c = _Calibration()
        _check(lib().cq_calibrate(self._h, C.byref(c)))
        return {k: (int(getattr(c, k)) if k.endswith("per_cu") else float(getattr(c, k))) for k, _ in _Calibration._fields_}

    def pairs_reserve(self, n_slots: int):
        _check(lib().cq_pairs_reserve(self._h, n_slots))

    def count_pairs(self) -> int:
        n = C.c_uint64(0)
        _check(lib().cq_pairs_fetch(self._h, None, None, None, 0, C.byref(n)))
        return int(n.value)

    def fetch_pairs(self, pair_cap: int = 1 << 16):
        pa = np.zeros(pair_cap, np.uint32)
        pb = np.zeros(pair_cap, np.uint32)
        pc = np.zeros(pair_cap, np.uint64)
        n = C.c_uint64(0)
        _check(lib().cq_pairs_fetch(self._h, _p(pa), _p(pb), _p(pc), pair_cap, C.byref(n)))
        return {(int(pa[i]), int(pb[i])): int(pc[i]) for i in range(int(n.value))}

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                lib().cq_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Multi:
    """cq_multi: one process, one host thread per GPU inside the library, RCCL all-reduce of the counts."""

    def __init__(self, path_u: str, path_d: str | None, devices):
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(lib().cq_multi_load(path_u.encode(), path_d.encode() if path_d else None, devs, len(devices), C.byref(h)))
        self._h = h
        self.size = int(lib().cq_multi_size(h))
        self.shards = [Index(None, _borrowed=lib().cq_multi_index(h, i)) for i in range(self.size)]
        self.n_leaves = self.shards[0].n_leaves
        self.hash_len = self.shards[0].hash_len

    def query(self, bases, offsets, n_genomes, mode=MODE_P, pair_cap=1 << 16):
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        o = _CountsOut(n_genomes, self.n_leaves, pair_cap)
        _check(lib().cq_multi_query(self._h, mode, _p(bases), _p(offsets), len(offsets) - 1, n_genomes, C.byref(o.c)))
        return o.result()

    def query_reads(self, read_ptrs, rlengths, n_genomes, mode=MODE_P, pair_cap=1 << 16):
        assert read_ptrs.dtype == np.uint64 and read_ptrs.flags.c_contiguous and rlengths.dtype == np.uint8
        o = _CountsOut(n_genomes, self.n_leaves, pair_cap)
        _check(lib().cq_multi_query_reads(self._h, mode, _p(read_ptrs), _p(rlengths), len(rlengths), n_genomes, C.byref(o.c)))
        return o.result()

    def query_packed(self, packed, lens, max_len, n_genomes, mode=MODE_P, pair_cap=1 << 16, out=None):
        assert packed.dtype == np.uint32 and packed.flags.c_contiguous and lens.dtype == np.uint8
        o = out or _CountsOut(n_genomes, self.n_leaves, pair_cap)
        _check(lib().cq_multi_query_packed(self._h, mode, _p(packed), _p(lens), len(lens), packed.shape[1], max_len,
                                           n_genomes, C.byref(o.c)))
        return o.result()

    def query_packed_tight(self, packed, lens, max_len, n_genomes, mode=MODE_P, pair_cap=1 << 16, out=None):
        assert packed.dtype == np.uint8 and packed.flags.c_contiguous and lens.dtype == np.uint8
        o = out or _CountsOut(n_genomes, self.n_leaves, pair_cap)
        _check(lib().cq_multi_query_packed_tight(self._h, mode, _p(packed), _p(lens), len(lens), packed.shape[1], max_len,
                                                 n_genomes, C.byref(o.c)))
        return o.result()

    def close(self):
        if getattr(self, "_h", None):
            for s_ in self.shards:
                s_._h = None
            lib().cq_multi_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().cq_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """cq_comm: this process's rank in the RCCL communicator of a one-process-per-GPU job."""

    def __init__(self, index: Index, uid: bytes, rank: int, world: int):
        assert len(uid) == COMM_ID_BYTES
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        _check(lib().cq_comm_init_rank(index._h, buf, rank, world, C.byref(h)))
        self._h = h
        self.rank, self.world = rank, world

    def allreduce_counts(self, d_counters_ptr: int, n_counter_words: int, d_rcount_ptr: int | None, n_rcount: int,
                         stream_ptr: int | None = None):
        """In-place sum over all ranks of the device counter block and rcount array; asynchronous on the stream."""
        _check(lib().cq_counts_allreduce(self._h, C.c_void_p(d_counters_ptr), n_counter_words,
                                         C.c_void_p(d_rcount_ptr) if d_rcount_ptr else None, n_rcount,
                                         C.c_void_p(stream_ptr) if stream_ptr else None))

    def close(self):
        if getattr(self, "_h", None):
            lib().cq_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
