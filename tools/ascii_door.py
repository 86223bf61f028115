#!/usr/bin/env python3
"""tools/ascii_door.py [n_reads]: where the time of cq_query (ASCII reads in pageable memory -- the door the
reference-side binding of INTEGRATION.md uses) goes, on configs[1]'s index: packing alone, the packed door with
pageable / pinned inputs and outputs, the ASCII door."""
import os
import sys
import tempfile
import time

import numpy as np
import torch  # noqa: F401  (first: library order, see tests/conftest.py)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cammiq_amd as cq  # noqa: E402
from cammiq_amd import bigsynth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G, rl, h = 500, 100, 26
w = bigsynth.World(seed=2, n_genomes=G, genome_len=3_450_000, k=26, h=h, lmax=50)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
pu = os.path.join(d, "index_u.bin1")
w.write_index(pu, None)
ix = cq.Index(pu, None, device=0)
b, o = w.reads(seed=1000, n=n, length=rl)


def best(f, k=3):
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return min(ts), r


t_pack, (pk, ln, _) = best(lambda: cq.pack_reads(b, o, h))
t_tight, (tp, tl, _) = best(lambda: cq.pack_reads_tight(b, o, h))
print(f"reads {n}, host cores {os.cpu_count()}")
print(f"cq_pack_reads alone            {t_pack * 1e3:8.1f} ms  {n / t_pack / 1e6:8.1f} Mreads/s")
print(f"cq_pack_reads_tight alone      {t_tight * 1e3:8.1f} ms  {n / t_tight / 1e6:8.1f} Mreads/s")
out_pin = ix.counts_out(G, pinned=True)
t, _ = best(lambda: ix.query_packed(pk, ln, rl, G))
print(f"packed door, pageable in/out   {t * 1e3:8.1f} ms  {n / t / 1e6:8.1f} Mreads/s")
t, _ = best(lambda: ix.query_packed(pk, ln, rl, G, out=out_pin))
print(f"packed door, pageable in, pinned out {t * 1e3:8.1f} ms  {n / t / 1e6:8.1f} Mreads/s")
pp = cq.host_array(pk.size, np.uint32).reshape(pk.shape); pp[:] = pk
pl = cq.host_array(ln.size, np.uint8); pl[:] = ln
t, _ = best(lambda: ix.query_packed(pp, pl, rl, G, out=out_pin))
print(f"packed door, pinned in/out     {t * 1e3:8.1f} ms  {n / t / 1e6:8.1f} Mreads/s")
t, r_ascii = best(lambda: ix.query(b, o, G))
print(f"ASCII door (cq_query), pageable out {t * 1e3:8.1f} ms  {n / t / 1e6:8.1f} Mreads/s")
t, _ = best(lambda: ix.query(b, o, G, out=out_pin))
print(f"ASCII door (cq_query), pinned out   {t * 1e3:8.1f} ms  {n / t / 1e6:8.1f} Mreads/s")
if hasattr(cq.binding.lib(), "cq_query_reads"):
    # the reference's own arrays: every read a heap block of its own.  glibc hands consecutive `new uint8_t[100]` out 112 bytes
    # apart (16-byte header, 16-byte granule): the arena below has that layout, reads[r] = arena + 112 r
    stride = (rl + 8 + 15) // 16 * 16
    arena = np.zeros(n * stride, np.uint8)
    arena.reshape(n, stride)[:, :rl] = b.reshape(n, rl)
    ptrs = (np.uint64(arena.ctypes.data) + np.arange(n, dtype=np.uint64) * np.uint64(stride)).astype(np.uint64)
    rl8 = np.full(n, rl, np.uint8)
    t, r_reads = best(lambda: ix.query_reads(ptrs, rl8, G))
    print(f"reads door (cq_query_reads: one block per read, {stride} B apart), pageable out {t * 1e3:8.1f} ms  {n / t / 1e6:8.1f} Mreads/s")
    t, _ = best(lambda: ix.query_reads(ptrs, rl8, G, out=out_pin))
    print(f"reads door (cq_query_reads), pinned out {t * 1e3:8.1f} ms  {n / t / 1e6:8.1f} Mreads/s")
    for k in ("cnt_u", "cnt_d", "rcount_u", "rcount_d"):
        assert np.array_equal(r_ascii[k], r_reads[k]), k
    assert (r_ascii["nundet"], r_ascii["nconf"]) == (r_reads["nundet"], r_reads["nconf"])
    # what the stub of INTEGRATION.md did before this door existed: flatten reads[] on one thread, then cq_query
    t0 = time.perf_counter()
    flat = np.ascontiguousarray(arena.reshape(n, stride)[:, :rl]).reshape(-1)
    t_flat = time.perf_counter() - t0
    print(f"(flattening the blocks into one buffer first, numpy, one thread: {t_flat * 1e3:8.1f} ms)")
    del flat
