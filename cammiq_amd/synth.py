"""Synthetic inputs for the CAMMiQ query path: genomes, marker indices, reads.

Nothing here is on the hot path.  RefSeq cannot be fetched in this environment and
the reference's build side (build.cpp / gsa.cpp) is out of scope, so tests, smoke()
and bench.py need their own source of *format-conformant* inputs:

* ``write_index``   emits ``index_u.bin1`` / ``index_d.bin2`` and their ``.aux`` bit
  streams exactly as ``Hash::encodeIdx64`` / ``encodeIdx64_d`` lay them out
  (/root/reference/src/hashtrie.cpp:625-699, ``encodeTrie`` :599-623) through the
  byte/bit conventions of ``BitWriter`` (/root/reference/src/binaryio.cpp:11-123).
* ``select_markers`` picks shortest unique / doubly-unique substrings (length k..Lmax,
  both strands) by brute force -- the *definition* CAMMiQ's build implements with
  suffix arrays (/root/reference/src/build.cpp:336-629), not its sparsification.
* ``simulate_reads`` mimics ``CAMMiQ-simulate`` (/root/reference/CAMMiQ-simulate:242-273):
  uniform start, uniform strand, per-base substitution errors, no N.

For benchmark-scale indices (1e8 leaves) see ``csrc/cq_synth.cpp``; this module is
the small-scale, readable twin used by the parity tests.
"""
from __future__ import annotations

import os
import random
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

_COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
SYM = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
END64 = 0xFFFFFFFFFFFFFFFF


def revcomp(s: bytes) -> bytes:
    return s.translate(_COMP)[::-1]


# --------------------------------------------------------------------------- genomes

def random_genome(rng: random.Random, n: int) -> bytes:
    return bytes(rng.choice(b"ACGT") for _ in range(n))


def mutate(rng: random.Random, g: bytes, p: float) -> bytes:
    out = bytearray(g)
    for i in range(len(out)):
        if rng.random() < p:
            out[i] = rng.choice([c for c in b"ACGT" if c != out[i]])
    return bytes(out)


def clade_genomes(seed: int, n_clades: int, per_clade: int, length: int, div: float) -> List[bytes]:
    """``n_clades`` random ancestors, ``per_clade`` children each at per-base
    substitution probability ``div`` -- related genomes give doubly-unique markers
    and key lengths spread over k..Lmax."""
    rng = random.Random(seed)
    out = []
    for _ in range(n_clades):
        anc = random_genome(rng, length)
        for _ in range(per_clade):
            out.append(mutate(rng, anc, div))
    return out


# --------------------------------------------------------------------------- markers

def select_markers(genomes: Sequence[bytes], k: int, lmax: int, keep_every: int = 1,
                   seed: int = 0):
    """Shortest unique (|G|=1) and doubly-unique (|G|=2) substrings of length k..lmax.

    Genome ids are 1-based.  Both strands of every genome count as text.  Returns
    ``(u, d)``: ``u[key] = (rid, ucount)``, ``d[key] = (rid1, rid2, uc1, uc2)`` with
    rid1 < rid2.  Within each dict the keys are prefix-free (argument in DESIGN.md).
    ``keep_every`` sparsifies: a candidate is kept when a seeded hash of its
    (genome, strand, position) is 0 mod keep_every.
    """
    texts = []  # (gid, strand, bytes)
    for gi, g in enumerate(genomes):
        texts.append((gi + 1, 0, g))
        texts.append((gi + 1, 1, revcomp(g)))
    # positions still unresolved for the unique / doubly-unique search
    live = [(ti, p) for ti, (_, _, t) in enumerate(texts) for p in range(len(t) - k + 1)]
    d_open = set(live)  # positions that have not yet seen |G| <= 2
    u: Dict[bytes, Tuple[int, int]] = {}
    d: Dict[bytes, Tuple[int, int, int, int]] = {}
    rng = random.Random(seed)
    salt = rng.getrandbits(32)

    def keep(ti, p):
        if keep_every <= 1:
            return True
        return ((ti * 1000003 + p) * 2654435761 + salt) % 4294967296 % keep_every == 0

    for L in range(k, lmax + 1):
        if not live:
            break
        gsets: Dict[bytes, set] = {}
        occ: Dict[bytes, Dict[int, int]] = {}
        alive = []
        for ti, p in live:
            t = texts[ti][2]
            if p + L > len(t):
                continue
            s = t[p:p + L]
            gsets.setdefault(s, set()).add(texts[ti][0])
            o = occ.setdefault(s, {})
            o[texts[ti][0]] = o.get(texts[ti][0], 0) + 1
            alive.append((ti, p, s))
        nxt = []
        for ti, p, s in alive:
            G = gsets[s]
            if len(G) == 1:
                if keep(ti, p) or s in u:
                    (g1,) = tuple(G)
                    u[s] = (g1, min(occ[s][g1], 0xFFFF))
                d_open.discard((ti, p))
                continue
            if len(G) == 2 and (ti, p) in d_open:
                if keep(ti, p) or s in d:
                    g1, g2 = sorted(G)
                    d[s] = (g1, g2, min(occ[s][g1], 0xFFFF), min(occ[s][g2], 0xFFFF))
                d_open.discard((ti, p))
            nxt.append((ti, p))
        live = nxt
    return u, d


# --------------------------------------------------------------------------- writer

class _Sink:
    """The two output streams of BitWriter (binaryio.cpp:11-123).

    .aux: writeBit is MSB first, a byte is emitted when its 8th bit arrives and a trailing
    partial byte is never written.  .binN: big-endian 16/32/64-bit integers.
    ``trace`` (optional list) records every call as (op, arg, value) with the op codes of
    oracle/ref_binaryio_driver.cpp, so a test can replay it through the reference's own
    BitWriter and demand byte-identical files.
    """

    def __init__(self, trace=None):
        self.aux = bytearray()
        self.ints = bytearray()
        self.cur = 0
        self.n = 0
        self.trace = trace

    def _bit(self, b: int):
        self.cur = (self.cur << 1) | (b & 1)
        self.n += 1
        if self.n == 8:
            self.aux.append(self.cur)
            self.cur = 0
            self.n = 0

    def bit(self, b: int):
        if self.trace is not None:
            self.trace.append((0, 0, b & 1))
        self._bit(b)

    def bits(self, count: int, value: int):
        if self.trace is not None:
            self.trace.append((1, count, value))
        for i in range(count):
            self._bit((value >> (count - 1 - i)) & 1)

    def u16(self, v: int):
        if self.trace is not None:
            self.trace.append((2, 0, v))
        self.ints += v.to_bytes(2, "big")

    def u32(self, v: int):
        if self.trace is not None:
            self.trace.append((3, 0, v))
        self.ints += v.to_bytes(4, "big")

    def u64(self, v: int):
        if self.trace is not None:
            self.trace.append((4, 0, v))
        self.ints += v.to_bytes(8, "big")

    def flush64(self):
        """flush64 = flush64i (72 one-bits, binaryio.cpp:115-118) + flush64a (END64 then a
        16-bit 0xFFFF, :120-123)."""
        if self.trace is not None:
            self.trace.append((5, 0, 0))
        for _ in range(72):
            self._bit(1)
        self.ints += END64.to_bytes(8, "big") + (0xFFFF).to_bytes(2, "big")


def _emit_trie(node, out: _Sink, doubly: bool):
    """encodeTrie / encodeTrie_d (hashtrie.cpp:599-623): '1', four children in A,C,G,T
    order ('0' when absent), then the leaf record if the node is a leaf."""
    out.bit(1)
    if isinstance(node, tuple):  # leaf record
        for _ in range(4):
            out.bit(0)
        if doubly:
            r1, r2, c1, c2 = node
            out.u32(r1); out.u32(r2); out.u16(c1); out.u16(c2)
        else:
            r1, c1 = node
            out.u32(r1); out.u16(c1)
        return
    for c in range(4):
        ch = node.get(c)
        if ch is None:
            out.bit(0)
        else:
            _emit_trie(ch, out, doubly)


def build_buckets(keys: Dict[bytes, tuple], h: int):
    """Group keys by their first h symbols; build one dict-trie per bucket.
    Raises on a prefix conflict, like Hash::insert64 aborts (hashtrie.cpp:146-149)."""
    buckets: Dict[int, object] = {}
    for key, rec in keys.items():
        if len(key) < h:
            raise ValueError("key shorter than hash length")
        hv = 0
        for c in key[:h]:
            hv = (hv << 2) | SYM[c]
        rest = key[h:]
        if not rest:
            if hv in buckets:
                raise ValueError("prefix conflict")
            buckets[hv] = rec
            continue
        node = buckets.setdefault(hv, {})
        if isinstance(node, tuple):
            raise ValueError("prefix conflict")
        for c in rest[:-1]:
            node = node.setdefault(SYM[c], {})
            if isinstance(node, tuple):
                raise ValueError("prefix conflict")
        last = SYM[rest[-1]]
        if last in node:
            raise ValueError("prefix conflict")
        node[last] = rec
    return buckets


def write_index(path: str, keys: Dict[bytes, tuple], h: int, doubly: bool,
                order_seed: int | None = 0, trace=None) -> int:
    """Write ``path`` and ``path + '.aux'``.  Returns the number of leaves.

    Bucket order in the reference is robin_hood's iteration order, i.e. arbitrary
    (SURVEY.md 8a row a12); ``order_seed`` shuffles, ``None`` keeps insertion order.
    """
    buckets = build_buckets(keys, h)
    order = list(buckets.keys())
    if order_seed is not None:
        random.Random(order_seed).shuffle(order)
    out = _Sink(trace)
    out.bit(1 if doubly else 0)      # encodeIdx64[_d]: flag (hashtrie.cpp:644,681)
    out.bits(7, 64)                  # option
    out.bits(8, h)                   # hash length
    for hv in order:
        out.u64(hv)
        _emit_trie(buckets[hv], out, doubly)
    out.flush64()
    with open(path, "wb") as f:
        f.write(bytes(out.ints))
    with open(path + ".aux", "wb") as f:
        f.write(bytes(out.aux))
    return len(keys)


def write_meta(dirpath: str, genomes: Sequence[bytes], u: Dict, d: Dict):
    """genome_map.out + the three meta files the query driver opens
    (query.cpp:125-205; written by build.cpp:673-736)."""
    G = len(genomes)
    with open(os.path.join(dirpath, "genome_map.out"), "w") as f:
        for i in range(1, G + 1):
            f.write(f"g{i}.fna\t{i}\t{2000 + i}\tsynthetic genome {i}\n")
    with open(os.path.join(dirpath, "genome_lengths.out"), "w") as f:
        for i in range(1, G + 1):
            f.write(f"{i}\t{len(genomes[i - 1])}\n")
    nu = [0] * (G + 1)
    nd = [0] * (G + 1)
    for rid, uc in u.values():
        nu[rid] += uc
    for r1, r2, c1, c2 in d.values():
        nd[r1] += c1
        nd[r2] += c2
    with open(os.path.join(dirpath, "unique_lmer_count_u.out"), "w") as f:
        for i in range(1, G + 1):
            f.write(f"{i}\t{nu[i]}\n")
    with open(os.path.join(dirpath, "unique_lmer_count_d.out"), "w") as f:
        for i in range(1, G + 1):
            f.write(f"{i}\t{nd[i]}\n")


# --------------------------------------------------------------------------- reads

def simulate_reads(genomes: Sequence[bytes], n: int, rl: int | Tuple[int, int], err: float,
                   seed: int, frac_random: float = 0.0, lower_frac: float = 0.0) -> List[bytes]:
    """Uniform genome / start / strand, per-base substitution probability ``err``,
    no N (CAMMiQ-simulate:242-273).  ``rl`` may be a (lo, hi) range for ragged reads.
    ``frac_random`` adds off-database reads; ``lower_frac`` lower-cases some reads."""
    rng = np.random.default_rng(seed)
    out = []
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for _ in range(n):
        L = rl if isinstance(rl, int) else int(rng.integers(rl[0], rl[1] + 1))
        if rng.random() < frac_random:
            r = acgt[rng.integers(0, 4, size=L)]
        else:
            g = genomes[int(rng.integers(0, len(genomes)))]
            L = min(L, len(g))
            st = int(rng.integers(0, len(g) - L + 1))
            s = g[st:st + L]
            if rng.random() < 0.5:
                s = revcomp(s)
            r = np.frombuffer(s, dtype=np.uint8).copy()
            if err > 0:
                m = rng.random(L) < err
                k = int(m.sum())
                if k:
                    # substitute with one of the three other bases
                    idx = np.searchsorted(acgt, r[m])
                    r[m] = acgt[(idx + rng.integers(1, 4, size=k)) % 4]
        b = r.tobytes()
        if lower_frac and rng.random() < lower_frac:
            b = b.lower()
        out.append(b)
    return out


def write_fastq(path: str, reads: Iterable[bytes]):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")


def concat_reads(reads: Sequence[bytes]):
    """ASCII reads -> (uint8 bases, uint64 offsets[n+1]) as the C ABI takes them."""
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return bases, offs
