"""Second, independent statement of the hot path -- pure Python, small cases only.

Written from SURVEY.md section 8(a) rows a6/a9/a10/a12 (format and closed-form decision
rule), NOT from oracle/cammiq_oracle.c: no trie, no rolling hash, no pointer sets.
It decodes an index into a flat ``{key string: leaf id}`` dict and answers
"which key is a prefix of the read suffix starting here" by brute force, so an
error shared with the C restatement would have to be an error in the reading of
the reference itself.  tests/test_oracle.py requires both to agree exactly.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

_COMP = bytes.maketrans(b"ACGTacgt", b"TGCATGCA")  # rcIdx folds to upper case
_ALPH = b"ACGT"


class Bits:
    def __init__(self, data: bytes):
        self.d = data
        self.i = 0

    def bit(self) -> int:
        byte = self.i >> 3
        v = 1 if byte >= len(self.d) else (self.d[byte] >> (7 - (self.i & 7))) & 1
        self.i += 1
        return v

    def bits(self, n: int) -> int:
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


def decode_index(path: str):
    """-> (doubly, h, leaves) with leaves = [(key_bytes, rid1, rid2, uc1, uc2)] in
    file (pre-order, decode) order.  Row a12 of SURVEY.md."""
    ints = open(path, "rb").read()
    aux = Bits(open(path + ".aux", "rb").read())
    doubly = aux.bit()
    assert aux.bits(7) == 64
    h = aux.bits(8)
    pos = 0
    leaves = []

    def u(n):
        nonlocal pos
        v = int.from_bytes(ints[pos:pos + n], "big")
        pos += n
        return v

    def walk(prefix: bytes):
        if aux.bit() == 0:
            return False
        kids = [walk(prefix + _ALPH[c:c + 1]) for c in range(4)]
        if not any(kids):
            if doubly:
                r1, r2, c1, c2 = u(4), u(4), u(2), u(2)
            else:
                r1, r2, c1, c2 = u(4), 0, u(2), 0
            leaves.append((prefix, r1, r2, c1, c2))
        return True

    while True:
        hv = u(8)
        if hv == 0xFFFFFFFFFFFFFFFF:
            break
        root = bytes(_ALPH[(hv >> (2 * (h - 1 - j))) & 3] for j in range(h))
        assert walk(root)
    return doubly, h, leaves


def classify(path_u: str, path_d: str | None, reads: List[bytes], n_genomes: int, mode: str = "p"):
    """Brute-force classify.  Returns dict with cnt_u, cnt_d (len n_genomes+1),
    rcount_u, rcount_d (decode order), nundet, nconf, pairs ({(a,b): n}, sc mode)."""
    _, h, lu = decode_index(path_u)
    ld = []
    if path_d:
        _, h2, ld = decode_index(path_d)
        assert h2 == h
    tables = []
    for leaves in (lu, ld):
        by_len: Dict[int, Dict[bytes, int]] = {}
        for i, (key, *_r) in enumerate(leaves):
            by_len.setdefault(len(key), {})[key] = i  # later duplicate bucket wins
        tables.append(by_len)
    cnt_u = [0] * (n_genomes + 1)
    cnt_d = [0] * (n_genomes + 1)
    rc = [[0] * len(lu), [0] * len(ld)]
    nundet = nconf = 0
    pairs: Dict[Tuple[int, int], int] = {}
    metas = (lu, ld)
    for read in reads:
        rl = len(read)
        assert h <= rl <= 255
        hits = set()
        for strand in (read.upper(), read.translate(_COMP)[::-1]):
            for i in range(rl - h + 1):
                for t in (0, 1):
                    for L, tab in tables[t].items():
                        if i + L <= rl:
                            j = tab.get(strand[i:i + L])
                            if j is not None:
                                hits.add((t, j))
        U = set()
        P = set()
        for t, j in hits:
            _, r1, r2, _, _ = metas[t][j]
            if r2 == 0:
                U.add(r1)
            else:
                P.add((min(r1, r2), max(r1, r2)))
        counted = False
        if len(U) >= 2:
            nconf += 1
        elif len(U) == 1:
            (r,) = U
            if all(r in p for p in P):
                cnt_u[r] += 1
                if P:
                    cnt_d[r] += 1
                counted = True
            else:
                nconf += 1
        else:
            if not P:
                nundet += 1
            elif len(P) == 1:
                (p,) = P
                cnt_d[p[0]] += 1
                cnt_d[p[1]] += 1
                pairs[p] = pairs.get(p, 0) + 1
                counted = True
            else:
                inter = set.intersection(*[set(p) for p in P])
                if len(inter) == 1:
                    (x,) = inter
                    cnt_d[x] += 1
                    if mode == "sc":
                        cnt_u[x] += 1
                    counted = True
                else:
                    nconf += 1
        if counted and mode == "p":
            for t, j in hits:
                rc[t][j] += 1
    return dict(cnt_u=cnt_u, cnt_d=cnt_d, rcount_u=rc[0], rcount_d=rc[1],
                nundet=nundet, nconf=nconf, pairs=pairs, h=h)
