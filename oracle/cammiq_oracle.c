/*
 * cammiq_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT part of the product.
 *
 * A plain-C, CPU restatement of the read-classification hot path of CAMMiQ
 * (reference tree mounted at /root/reference, citations below are relative to
 * /root/reference/src).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the
 * checker / reported baseline -- never as the thing shipped or measured.
 * The product (cammiq_amd/csrc) never links, loads or calls anything here.
 *
 * PARITY STATUS: "parity unpinned" for the classify layer.
 *   The reference holds no tests, golden vectors or fixtures for this path
 *   (SURVEY.md section 4), and its query path does not compile in this image:
 *   hashtrie.hpp:5 includes <robin_hood.h>, which is absent and may not be
 *   replaced by a stand-in.  The one layer that DOES build from the reference's
 *   own sources is the on-disk codec (binaryio.cpp, no external dependency);
 *   oracle/Makefile compiles it into oracle/_ref/ and tests/test_codec_ref.py
 *   pins this file's bit/byte reader against it.  Everything above the codec
 *   (trie decode, find, decision rule) is a line-by-line restatement checked
 *   against a second, independent brute-force implementation (tests/pyref.py).
 *
 * What is restated (same control flow, same data structures in spirit):
 *   BitReader::readBit/readBits/readBits16/32/64      binaryio.cpp:141-182
 *   BitReader::openFile (slurp .binN and .binN.aux)    binaryio.cpp:184-214
 *   Hash::loadIdx64_p                                  hashtrie.cpp:486-507
 *   Hash::decodeTrie_p                                 hashtrie.cpp:425-484
 *   Hash::find64_p                                     hashtrie.cpp:350-369
 *   FqReader::query64_p / query64mt_p                  query.cpp:458-648 / 650-889
 *   FqReader::query64_sc                               query.cpp:891-1080
 *   FqReader::getRC + rcIdx                            query.cpp:447-450,1875-1881
 *   symbolIdx                                          query.cpp:1860-1873, hashtrie.cpp:701-714
 *
 * Parity domain: read bytes in ACGTacgt, h <= len <= 255,
 * refIDs in 1..n_genomes.  Outside it the reference has undefined behaviour;
 * this oracle returns an error instead.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ tables */

/* symbolIdx: query.cpp:1860-1873 (A/a 0, C/c 1, G/g 2, T/t 3; the +165 codes
 * 230/232/236/249 also map; everything else -1). */
static int cqo_sym(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': case 230: return 0;
    case 'C': case 'c': case 232: return 1;
    case 'G': case 'g': case 236: return 2;
    case 'T': case 't': case 249: return 3;
    default: return -1;
    }
}

/* rcIdx: query.cpp:1875-1881 -- complement, folding to upper case. */
static int cqo_rc(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 'T';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    default: return -1;
    }
}

/* ------------------------------------------------------------- trie nodes */

/* trieNode / pleafNode: hashtrie.hpp:8-13,37-47. */
typedef struct cqo_node {
    struct cqo_node *children[4];
    int isEnd;
    /* leaf payload (valid when isEnd) */
    uint32_t refID1, refID2;
    uint8_t depth;
    uint16_t ucount1, ucount2;
    uint32_t rcount;
    uint64_t order; /* decode order within its table (0-based) */
    int table;      /* 0 = ht_u, 1 = ht_d */
} cqo_node;

/* exact map u64 -> node*  (hashtrie.hpp:49; any exact container is equivalent) */
typedef struct {
    uint64_t *keys;
    cqo_node **vals;
    uint64_t cap, n;
} cqo_map;

static uint64_t cqo_mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static void cqo_map_init(cqo_map *m, uint64_t cap)
{
    m->cap = cap; m->n = 0;
    m->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    m->vals = (cqo_node **)calloc(cap, sizeof(cqo_node *));
    memset(m->keys, 0xff, cap * sizeof(uint64_t));
}

/* Slot holding `key`, claiming a free one (and stamping the key) if absent. */
static cqo_node **cqo_map_slot(cqo_map *m, uint64_t key)
{
    uint64_t i = cqo_mix(key) & (m->cap - 1);
    while (m->vals[i] != NULL && m->keys[i] != key)
        i = (i + 1) & (m->cap - 1);
    if (m->vals[i] == NULL) m->keys[i] = key;
    return &m->vals[i];
}

static void cqo_map_grow(cqo_map *m)
{
    cqo_map o = *m;
    cqo_map_init(m, o.cap * 2);
    for (uint64_t i = 0; i < o.cap; i++)
        if (o.vals[i]) { *cqo_map_slot(m, o.keys[i]) = o.vals[i]; m->n++; }
    free(o.keys); free(o.vals);
}

/* map64[bucket] = root   (hashtrie.cpp:500): insert or overwrite. */
static void cqo_map_put(cqo_map *m, uint64_t key, cqo_node *v)
{
    if ((m->n + 1) * 2 > m->cap) cqo_map_grow(m);
    cqo_node **s = cqo_map_slot(m, key);
    if (*s == NULL) m->n++;
    *s = v;
}

static cqo_node *cqo_map_get(const cqo_map *m, uint64_t key)
{
    uint64_t i = cqo_mix(key) & (m->cap - 1);
    while (m->vals[i] != NULL) {
        if (m->keys[i] == key) return m->vals[i];
        i = (i + 1) & (m->cap - 1);
    }
    return NULL;
}

/* --------------------------------------------------------------- BitReader */

typedef struct {
    uint8_t *buffer_INT, *buffer_AUX;
    size_t fsize_INT, fsize_AUX, cur_INT, cur_AUX;
    int curBits, curByte;
    int past_eof; /* set once a bit was synthesised beyond the .aux file */
} cqo_reader;

static uint8_t *cqo_slurp(const char *fn, size_t *sz)
{
    FILE *f = fopen(fn, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *b = (uint8_t *)malloc((size_t)n + 100);
    memset(b, 0, (size_t)n + 100);
    if (n > 0 && fread(b, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(b); return NULL; }
    fclose(f);
    *sz = (size_t)n;
    return b;
}

/* binaryio.cpp:141-156 -- MSB first; past EOF every bit reads as 1 (curByte=-1). */
static uint32_t cqo_readBit(cqo_reader *r)
{
    if (r->curBits == 0) {
        r->curBits = 8;
        if (r->cur_AUX < r->fsize_AUX) r->curByte = (int)(int8_t)r->buffer_AUX[r->cur_AUX++];
        else { r->curByte = -1; r->past_eof = 1; }
    }
    uint32_t v = ((uint32_t)(r->curByte >> (r->curBits - 1))) & 1u;
    r->curBits -= 1;
    return v;
}

static uint32_t cqo_readBits(cqo_reader *r, int count)
{
    uint32_t v = 0;
    for (int i = 0; i < count; i++) v = (v << 1) + cqo_readBit(r);
    return v;
}

/* binaryio.cpp:165-182 -- big-endian integers from the .binN byte stream. */
static uint16_t cqo_readBits16(cqo_reader *r)
{
    uint16_t v = r->buffer_INT[r->cur_INT++];
    v = (uint16_t)((v << 8) | r->buffer_INT[r->cur_INT++]);
    return v;
}

static uint32_t cqo_readBits32(cqo_reader *r)
{
    uint32_t v = r->buffer_INT[r->cur_INT++];
    for (int i = 0; i < 3; i++) v = (v << 8) | r->buffer_INT[r->cur_INT++];
    return v;
}

static uint64_t cqo_readBits64(cqo_reader *r)
{
    uint64_t v = r->buffer_INT[r->cur_INT++];
    for (int i = 0; i < 7; i++) v = (v << 8) | r->buffer_INT[r->cur_INT++];
    return v;
}

/* -------------------------------------------------------------------- Hash */

typedef struct {
    uint32_t hash_len_;
    int doubly_unique;
    cqo_map map64;
    cqo_node **leaves;  /* decode order; map_sp is derived from this */
    uint64_t leaf_cnt, leaf_cap;
    uint64_t n_buckets_in_file;
    int table;
    int error;
} cqo_hash;

static cqo_node *cqo_new_node(void)
{
    return (cqo_node *)calloc(1, sizeof(cqo_node));
}

static void cqo_free_node(cqo_node *n)
{
    if (!n) return;
    for (int i = 0; i < 4; i++) cqo_free_node(n->children[i]);
    free(n);
}

/* hashtrie.cpp:425-484.  A node whose four children are all absent is a leaf;
 * its record is read from the byte stream right after its fourth child bit. */
static cqo_node *cqo_decodeTrie_p(cqo_hash *h, cqo_reader *r, int d_flag, uint8_t depth)
{
    int b = (int)cqo_readBit(r);
    if (b == 0) return NULL;
    if (r->past_eof) { h->error = 1; return NULL; } /* truncated .aux: would recurse forever */
    cqo_node *root = cqo_new_node();
    int isleaf = 1;
    for (int i = 0; i < 4; i++) {
        cqo_node *child = cqo_decodeTrie_p(h, r, d_flag, (uint8_t)(depth + 1));
        root->children[i] = child;
        if (child != NULL) isleaf = 0;
        if (h->error) return root;
    }
    if (isleaf) {
        if (r->cur_INT + (d_flag ? 12 : 6) > r->fsize_INT) { h->error = 1; return root; }
        root->isEnd = 1;
        root->depth = (uint8_t)(depth + h->hash_len_);
        if (d_flag) {
            root->refID1 = cqo_readBits32(r);
            root->refID2 = cqo_readBits32(r);
            if (root->refID1 == 0 || root->refID2 == 0) h->error = 2; /* assert :446 */
            root->ucount1 = cqo_readBits16(r);
            root->ucount2 = cqo_readBits16(r);
        } else {
            root->refID1 = cqo_readBits32(r);
            root->refID2 = 0;
            root->ucount1 = cqo_readBits16(r);
        }
        root->rcount = 0;
        root->table = h->table;
        root->order = h->leaf_cnt;
        if (h->leaf_cnt == h->leaf_cap) {
            h->leaf_cap = h->leaf_cap ? h->leaf_cap * 2 : 1024;
            h->leaves = (cqo_node **)realloc(h->leaves, h->leaf_cap * sizeof(cqo_node *));
        }
        h->leaves[h->leaf_cnt++] = root;
    }
    return root;
}

/* hashtrie.cpp:486-507. */
static int cqo_loadIdx64_p(cqo_hash *h, const char *fn)
{
    cqo_reader r;
    memset(&r, 0, sizeof r);
    char aux[4096];
    snprintf(aux, sizeof aux, "%s.aux", fn);
    r.buffer_INT = cqo_slurp(fn, &r.fsize_INT);
    r.buffer_AUX = cqo_slurp(aux, &r.fsize_AUX);
    if (!r.buffer_INT || !r.buffer_AUX) { free(r.buffer_INT); free(r.buffer_AUX); return -1; }
    h->doubly_unique = (int)cqo_readBit(&r);
    int option = (int)cqo_readBits(&r, 7);
    if (option != 64) { free(r.buffer_INT); free(r.buffer_AUX); return -2; } /* assert :492 */
    h->hash_len_ = cqo_readBits(&r, 8);
    cqo_map_init(&h->map64, 1024);
    int rc = 0;
    for (;;) {
        if (r.cur_INT + 8 > r.fsize_INT) { rc = -3; break; }
        uint64_t bucket = cqo_readBits64(&r);
        if (bucket == 0xFFFFFFFFFFFFFFFFULL) break; /* END64 */
        cqo_node *root = cqo_decodeTrie_p(h, &r, h->doubly_unique, 0);
        if (h->error || root == NULL) { rc = -4; break; }
        cqo_map_put(&h->map64, bucket, root);
        h->n_buckets_in_file++;
    }
    free(r.buffer_INT); free(r.buffer_AUX);
    return rc;
}

/* hashtrie.cpp:350-369. */
static cqo_node *cqo_find64_p(const cqo_hash *h, uint64_t bucket_, const uint8_t *cand, size_t len_)
{
    cqo_node *cur = cqo_map_get(&h->map64, bucket_);
    if (cur == NULL) return NULL;
    for (size_t i = 0; i < len_; i++) {
        int index = cqo_sym(cand[i]);
        if (cur->isEnd) return cur;
        if (cur->children[index] == NULL) return NULL;
        cur = cur->children[index];
    }
    if (cur != NULL && cur->isEnd) return cur;
    return NULL;
}

/* ------------------------------------------------------------ public index */

typedef struct cqo_index {
    cqo_hash ht[2]; /* [0] = ht_u, [1] = ht_d */
} cqo_index;

void cqo_free(cqo_index *ix)
{
    if (!ix) return;
    for (int t = 0; t < 2; t++) {
        cqo_map *m = &ix->ht[t].map64;
        if (m->vals) {
            for (uint64_t i = 0; i < m->cap; i++) cqo_free_node(m->vals[i]);
            free(m->keys); free(m->vals);
        }
        free(ix->ht[t].leaves);
    }
    free(ix);
}

/* FqReader::loadIdx_p (query.cpp:109-123).  path_d == NULL or "" stands for the
 * empty-but-valid .bin2 a --unique build needs (SURVEY.md 8b "Quirk"). */
cqo_index *cqo_load(const char *path_u, const char *path_d)
{
    cqo_index *ix = (cqo_index *)calloc(1, sizeof *ix);
    ix->ht[0].table = 0; ix->ht[1].table = 1;
    if (cqo_loadIdx64_p(&ix->ht[0], path_u) != 0) { cqo_free(ix); return NULL; }
    if (path_d && path_d[0]) {
        if (cqo_loadIdx64_p(&ix->ht[1], path_d) != 0) { cqo_free(ix); return NULL; }
        /* assert(hash_len_u == hash_len_d)  query.cpp:460 */
        if (ix->ht[0].hash_len_ != ix->ht[1].hash_len_) { cqo_free(ix); return NULL; }
    } else {
        ix->ht[1].hash_len_ = ix->ht[0].hash_len_;
        ix->ht[1].doubly_unique = 1;
        cqo_map_init(&ix->ht[1].map64, 16);
    }
    return ix;
}

uint32_t cqo_hash_len(const cqo_index *ix) { return ix->ht[0].hash_len_; }
uint64_t cqo_num_leaves(const cqo_index *ix, int table) { return ix->ht[table].leaf_cnt; }
uint64_t cqo_num_buckets(const cqo_index *ix, int table) { return ix->ht[table].map64.n; }
int cqo_is_doubly_unique(const cqo_index *ix, int table) { return ix->ht[table].doubly_unique; }

/* Leaf records in decode order -- the order map_sp is filled in
 * (hashtrie.cpp:452-453,476). */
void cqo_leaves(const cqo_index *ix, int table, uint32_t *refID1, uint32_t *refID2,
                uint8_t *depth, uint16_t *ucount1, uint16_t *ucount2)
{
    const cqo_hash *h = &ix->ht[table];
    for (uint64_t i = 0; i < h->leaf_cnt; i++) {
        refID1[i] = h->leaves[i]->refID1;
        refID2[i] = h->leaves[i]->refID2;
        depth[i] = h->leaves[i]->depth;
        ucount1[i] = h->leaves[i]->ucount1;
        ucount2[i] = h->leaves[i]->ucount2;
    }
}

uint32_t cqo_max_refid(const cqo_index *ix)
{
    uint32_t m = 0;
    for (int t = 0; t < 2; t++)
        for (uint64_t i = 0; i < ix->ht[t].leaf_cnt; i++) {
            if (ix->ht[t].leaves[i]->refID1 > m) m = ix->ht[t].leaves[i]->refID1;
            if (ix->ht[t].leaves[i]->refID2 > m) m = ix->ht[t].leaves[i]->refID2;
        }
    return m;
}

/* ------------------------------------------------ small ordered sets (std::set) */

typedef struct { uint32_t a, b; } cqo_pair;

static int cqo_set_u32_insert(uint32_t *s, int n, uint32_t v)
{
    int i = 0;
    while (i < n && s[i] < v) i++;
    if (i < n && s[i] == v) return n;
    memmove(s + i + 1, s + i, (size_t)(n - i) * sizeof *s);
    s[i] = v;
    return n + 1;
}

static int cqo_set_ptr_insert(cqo_node **s, int n, cqo_node *v)
{
    for (int i = 0; i < n; i++) if (s[i] == v) return n;
    s[n] = v;
    return n + 1;
}

static int cqo_pair_lt(cqo_pair x, cqo_pair y) { return x.a < y.a || (x.a == y.a && x.b < y.b); }

static int cqo_set_pair_insert(cqo_pair *s, int n, cqo_pair v)
{
    int i = 0;
    while (i < n && cqo_pair_lt(s[i], v)) i++;
    if (i < n && s[i].a == v.a && s[i].b == v.b) return n;
    memmove(s + i + 1, s + i, (size_t)(n - i) * sizeof *s);
    s[i] = v;
    return n + 1;
}

/* ------------------------------------------------------------------- query */

#define CQO_MAX_RL 256           /* query.hpp:34 */
#define CQO_MAX_HITS (4 * 256)   /* 2 strands x 2 tables x <=251 windows */

enum { CQO_MODE_P = 0, CQO_MODE_SC = 1 };

/* Which outcome of the decision switch a read took (coverage bookkeeping only;
 * SURVEY.md 8c asks to keep the eight outcomes visible). */
enum {
    CQO_BR_UNDET = 0,     /* P=0, U=0                         query.cpp:544-545 */
    CQO_BR_U1_P0,         /* P=0, |U|=1                       :547-551 */
    CQO_BR_UMULTI,        /* |U|>=2 (any P)                   :553,568,589 */
    CQO_BR_U0_P1,         /* |P|=1, U=0                       :557-563 */
    CQO_BR_U1_PALL,       /* |U|=1, every pair contains r     :572-576,597-600 */
    CQO_BR_U1_PCONF,      /* |U|=1, some pair without r       :571,596 */
    CQO_BR_U0_PI1,        /* |P|>=2, U=0, |I|=1               :624-628 */
    CQO_BR_U0_PCONF,      /* |P|>=2, U=0, |I|!=1              :621,630 */
    CQO_BR_N
};

typedef struct {
    uint64_t *cnt_u, *cnt_d;     /* [n_genomes + 1], index 0 unused */
    uint64_t nundet, nconf;
    uint64_t branch[CQO_BR_N];
    /* SC mode: read_cnts_b (query.hpp:49) as an append-only list merged later */
    cqo_pair *pb; uint64_t *pbc; uint64_t npb, pbcap;
} cqo_acc;

static void cqo_pb_add(cqo_acc *a, cqo_pair p)
{
    for (uint64_t i = 0; i < a->npb; i++)
        if (a->pb[i].a == p.a && a->pb[i].b == p.b) { a->pbc[i]++; return; }
    if (a->npb == a->pbcap) {
        a->pbcap = a->pbcap ? a->pbcap * 2 : 64;
        a->pb = (cqo_pair *)realloc(a->pb, a->pbcap * sizeof *a->pb);
        a->pbc = (uint64_t *)realloc(a->pbc, a->pbcap * sizeof *a->pbc);
    }
    a->pb[a->npb] = p; a->pbc[a->npb] = 1; a->npb++;
}

/* One strand: query.cpp:480-501 (forward) / :506-527 (reverse complement). */
static int cqo_scan_strand(const cqo_index *ix, const uint8_t *read, size_t rl,
                           cqo_node **pnodes, int np)
{
    const uint32_t hash_len_u = ix->ht[0].hash_len_;
    uint32_t hs = 2 * hash_len_u - 2;
    uint64_t hv = 0;
    cqo_node *pln;
    for (size_t i = 0; i < hash_len_u; i++)
        hv = ((hv << 2) | (uint64_t)cqo_sym(read[i]));
    for (size_t i = 0; i < rl - hash_len_u; i++) {
        pln = cqo_find64_p(&ix->ht[0], hv, read + i + hash_len_u, rl - hash_len_u - i);
        if (pln != NULL) np = cqo_set_ptr_insert(pnodes, np, pln);
        pln = cqo_find64_p(&ix->ht[1], hv, read + i + hash_len_u, rl - hash_len_u - i);
        if (pln != NULL) np = cqo_set_ptr_insert(pnodes, np, pln);
        hv = hv - ((uint64_t)cqo_sym(read[i]) << hs); /* Next hash. */
        hv = ((hv << 2) | (uint64_t)cqo_sym(read[i + hash_len_u]));
    }
    pln = cqo_find64_p(&ix->ht[0], hv, read, 0);
    if (pln != NULL) np = cqo_set_ptr_insert(pnodes, np, pln);
    pln = cqo_find64_p(&ix->ht[1], hv, read, 0);
    if (pln != NULL) np = cqo_set_ptr_insert(pnodes, np, pln);
    return np;
}

/* Per-read body of query64_p (query.cpp:471-636) / query64_sc (:904-1071).
 * `locked` mirrors query64mt_p's single unnamed omp critical (:742-878). */
static void cqo_classify_read(const cqo_index *ix, int mode, const uint8_t *read, size_t rl,
                              cqo_acc *acc, int locked)
{
    cqo_node *pnodes[CQO_MAX_HITS];
    uint32_t rids[CQO_MAX_HITS];
    cqo_pair rid_pairs[CQO_MAX_HITS];
    uint32_t intersection[2];
    uint8_t rc_read[CQO_MAX_RL];
    int np = 0, nr = 0, npair = 0, ni = 0;

    /* Forward strand, then reverse complement (getRC: query.cpp:447-450). */
    np = cqo_scan_strand(ix, read, rl, pnodes, np);
    for (size_t i = 0; i < rl; i++) rc_read[i] = (uint8_t)cqo_rc(read[rl - i - 1]);
    np = cqo_scan_strand(ix, rc_read, rl, pnodes, np);

    /* Record results: query.cpp:529-540. */
    for (int k = 0; k < np; k++) {
        cqo_node *pn = pnodes[k];
        if (pn->refID2 == 0)
            nr = cqo_set_u32_insert(rids, nr, pn->refID1);
        else {
            cqo_pair p;
            if (pn->refID1 < pn->refID2) { p.a = pn->refID1; p.b = pn->refID2; }
            else { p.a = pn->refID2; p.b = pn->refID1; }
            npair = cqo_set_pair_insert(rid_pairs, npair, p);
        }
    }

    /* Decision switch: query.cpp:542-636.  Effects are collected first and
     * applied in one place so the serial and the "critical" variant share it. */
    int d_nundet = 0, d_nconf = 0, bump = 0, br = -1;
    uint32_t inc_u[2]; int n_inc_u = 0;
    uint32_t inc_d[4]; int n_inc_d = 0;
    int add_pb = 0;

    switch (npair) {
    case 0:
        if (nr == 0) { d_nundet++; br = CQO_BR_UNDET; }
        else if (nr == 1) { inc_u[n_inc_u++] = rids[0]; bump = 1; br = CQO_BR_U1_P0; }
        else { d_nconf++; br = CQO_BR_UMULTI; }
        break;
    case 1:
        if (nr == 0) {
            inc_d[n_inc_d++] = rid_pairs[0].a;
            inc_d[n_inc_d++] = rid_pairs[0].b;
            bump = 1; add_pb = 1; br = CQO_BR_U0_P1;
        } else if (nr > 1) { d_nconf++; br = CQO_BR_UMULTI; }
        else {
            uint32_t rid = rids[0];
            if (rid_pairs[0].a != rid && rid_pairs[0].b != rid) { d_nconf++; br = CQO_BR_U1_PCONF; }
            else { inc_u[n_inc_u++] = rid; inc_d[n_inc_d++] = rid; bump = 1; br = CQO_BR_U1_PALL; }
        }
        break;
    default:
        if (nr != 0) {
            if (nr > 1) { d_nconf++; br = CQO_BR_UMULTI; break; }
            uint32_t rid = rids[0];
            int conf = 0;
            for (int k = 0; k < npair; k++)
                if (rid_pairs[k].a != rid && rid_pairs[k].b != rid) { conf = 1; break; }
            if (conf) { d_nconf++; br = CQO_BR_U1_PCONF; }
            else { inc_u[n_inc_u++] = rid; inc_d[n_inc_d++] = rid; bump = 1; br = CQO_BR_U1_PALL; }
        } else {
            /* intersection over all pairs, seeded with the first (smallest) pair:
             * query.cpp:604-619 (a std::set, so first==second collapses). */
            for (int k = 0; k < npair; k++) {
                if (k == 0) {
                    intersection[0] = rid_pairs[0].a; ni = 1;
                    if (rid_pairs[0].b != rid_pairs[0].a) { intersection[1] = rid_pairs[0].b; ni = 2; }
                } else {
                    int w = 0;
                    for (int q = 0; q < ni; q++) {
                        uint32_t rid = intersection[q];
                        if (!(rid_pairs[k].a != rid && rid_pairs[k].b != rid)) intersection[w++] = rid;
                    }
                    ni = w;
                }
            }
            if (ni == 1) {
                inc_d[n_inc_d++] = intersection[0];
                /* query64_sc bumps BOTH counters here (query.cpp:1055-1059) */
                if (mode == CQO_MODE_SC) inc_u[n_inc_u++] = intersection[0];
                bump = 1; br = CQO_BR_U0_PI1;
            } else { d_nconf++; br = CQO_BR_U0_PCONF; }
        }
        break;
    }

    /* Apply.  query64_sc never touches rcount (query.cpp:891-1080). */
    if (locked == 2) {
        /* "fair" CPU variant for the reported baseline: same per-read work, but the
         * counters are bumped with atomics instead of one global critical section, so the threads
         * do not serialise on every read the way query64mt_p does. */
        if (d_nundet) {
#pragma omp atomic
            acc->nundet += 1;
        }
        if (d_nconf) {
#pragma omp atomic
            acc->nconf += 1;
        }
        for (int k = 0; k < n_inc_u; k++) {
#pragma omp atomic
            acc->cnt_u[inc_u[k]] += 1;
        }
        for (int k = 0; k < n_inc_d; k++) {
#pragma omp atomic
            acc->cnt_d[inc_d[k]] += 1;
        }
        if (bump && mode == CQO_MODE_P)
            for (int k = 0; k < np; k++) {
#pragma omp atomic
                pnodes[k]->rcount += 1;
            }
        if (add_pb && mode == CQO_MODE_SC) {
#pragma omp critical
            cqo_pb_add(acc, rid_pairs[0]);
        }
#pragma omp atomic
        acc->branch[br] += 1;
    } else if (locked == 3) {
        /* SURVEY.md 8(d)'s optimised CPU variant: `acc` is this THREAD's own counter block (merged after the
         * loop by cqo_query_variant), so nothing here is shared except the per-leaf rcount, which stays one
         * atomic add per distinct hit of a counted read.  Removes the lock of query.cpp:742-878 altogether. */
        acc->nundet += (uint64_t)d_nundet; acc->nconf += (uint64_t)d_nconf;
        for (int k = 0; k < n_inc_u; k++) acc->cnt_u[inc_u[k]]++;
        for (int k = 0; k < n_inc_d; k++) acc->cnt_d[inc_d[k]]++;
        if (bump && mode == CQO_MODE_P)
            for (int k = 0; k < np; k++) {
#pragma omp atomic
                pnodes[k]->rcount += 1;
            }
        if (add_pb && mode == CQO_MODE_SC) cqo_pb_add(acc, rid_pairs[0]);
        acc->branch[br]++;
    } else if (locked) {
#pragma omp critical
        {
            acc->nundet += (uint64_t)d_nundet; acc->nconf += (uint64_t)d_nconf;
            for (int k = 0; k < n_inc_u; k++) acc->cnt_u[inc_u[k]]++;
            for (int k = 0; k < n_inc_d; k++) acc->cnt_d[inc_d[k]]++;
            if (bump && mode == CQO_MODE_P) for (int k = 0; k < np; k++) pnodes[k]->rcount += 1;
            if (add_pb && mode == CQO_MODE_SC) cqo_pb_add(acc, rid_pairs[0]);
            acc->branch[br]++;
        }
    } else {
        acc->nundet += (uint64_t)d_nundet; acc->nconf += (uint64_t)d_nconf;
        for (int k = 0; k < n_inc_u; k++) acc->cnt_u[inc_u[k]]++;
        for (int k = 0; k < n_inc_d; k++) acc->cnt_d[inc_d[k]]++;
        if (bump && mode == CQO_MODE_P) for (int k = 0; k < np; k++) pnodes[k]->rcount += 1;
        if (add_pb && mode == CQO_MODE_SC) cqo_pb_add(acc, rid_pairs[0]);
        acc->branch[br]++;
    }
}

/* Validates the parity domain; returns the index of the first bad read or -1. */
static int64_t cqo_check_reads(const cqo_index *ix, const uint8_t *bases, const uint64_t *offsets,
                               uint64_t n_reads)
{
    uint32_t h = ix->ht[0].hash_len_;
    for (uint64_t r = 0; r < n_reads; r++) {
        uint64_t rl = offsets[r + 1] - offsets[r];
        if (rl < h || rl > 255) return (int64_t)r;
        for (uint64_t i = offsets[r]; i < offsets[r + 1]; i++)
            if (cqo_sym(bases[i]) < 0 || bases[i] >= 128) return (int64_t)r;
    }
    return -1;
}

/*
 * Classify all reads of one FASTQ.  Counters are zeroed first, i.e. one call ==
 * one file after resetCounters (query.cpp:1820-1840).
 *   mode        CQO_MODE_P (query64_p / query64mt_p) or CQO_MODE_SC (query64_sc)
 *   nthreads    1 -> serial body (query64_p); >1 -> OpenMP over reads with one
 *               global critical section per update, as query64mt_p does; <0 -> |nthreads|
 *               threads with atomic counter updates (fair CPU variant for bench.py).
 *   cnt_u/cnt_d [n_genomes+1]; rcount_u/rcount_d per leaf in decode order;
 *   scal[0]=nundet scal[1]=nconf; branch[8]; pair_* (SC): up to pair_cap triples,
 *   *n_pairs receives the number of distinct pairs.
 * Returns 0, or -(1+index) of the first read outside the parity domain, or -1e9
 * for a refID above n_genomes.
 */
enum { CQO_VAR_SERIAL = 0, CQO_VAR_CRITICAL = 1, CQO_VAR_ATOMIC = 2, CQO_VAR_THREAD_LOCAL = 3 };

#include <time.h>
static double cqo_last_loop_s = 0.0;
static double cqo_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* Seconds the most recent cqo_query* spent in its loop over the reads (the reference's "Time for query" bracket). */
double cqo_last_loop_seconds(void) { return cqo_last_loop_s; }

/* variant: CQO_VAR_SERIAL (query64_p), CQO_VAR_CRITICAL (query64mt_p as written: one global critical section per
 * read), CQO_VAR_ATOMIC (same loop, every shared counter an atomic), CQO_VAR_THREAD_LOCAL (per-thread cnt_u /
 * cnt_d / nundet / nconf / branch / read_cnts_b merged after the loop, rcount via atomics: SURVEY.md 8(d)'s
 * "optimised variant (thread-local counters)").  All four give identical outputs (tests/test_oracle.py). */
int64_t cqo_query_variant(cqo_index *ix, int mode, int nthreads, int variant,
                          const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                          uint32_t n_genomes,
                          uint64_t *cnt_u, uint64_t *cnt_d, uint32_t *rcount_u, uint32_t *rcount_d,
                          uint64_t *scal, uint64_t *branch,
                          uint32_t *pair_a, uint32_t *pair_b, uint64_t *pair_cnt, uint64_t pair_cap,
                          uint64_t *n_pairs);

int64_t cqo_query(cqo_index *ix, int mode, int nthreads,
                  const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                  uint32_t n_genomes,
                  uint64_t *cnt_u, uint64_t *cnt_d, uint32_t *rcount_u, uint32_t *rcount_d,
                  uint64_t *scal, uint64_t *branch,
                  uint32_t *pair_a, uint32_t *pair_b, uint64_t *pair_cnt, uint64_t pair_cap,
                  uint64_t *n_pairs)
{
    const int variant = (nthreads >= 0 && nthreads <= 1) ? CQO_VAR_SERIAL : nthreads < 0 ? CQO_VAR_ATOMIC : CQO_VAR_CRITICAL;
    return cqo_query_variant(ix, mode, nthreads < 0 ? -nthreads : nthreads, variant, bases, offsets, n_reads, n_genomes,
                             cnt_u, cnt_d, rcount_u, rcount_d, scal, branch, pair_a, pair_b, pair_cnt, pair_cap, n_pairs);
}

int64_t cqo_query_variant(cqo_index *ix, int mode, int nthreads, int variant,
                          const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                          uint32_t n_genomes,
                          uint64_t *cnt_u, uint64_t *cnt_d, uint32_t *rcount_u, uint32_t *rcount_d,
                          uint64_t *scal, uint64_t *branch,
                          uint32_t *pair_a, uint32_t *pair_b, uint64_t *pair_cnt, uint64_t pair_cap,
                          uint64_t *n_pairs)
{
    if (cqo_max_refid(ix) > n_genomes) return -1000000000LL;
    int64_t bad = cqo_check_reads(ix, bases, offsets, n_reads);
    if (bad >= 0) return -(1 + bad);

    cqo_acc acc;
    memset(&acc, 0, sizeof acc);
    acc.cnt_u = cnt_u; acc.cnt_d = cnt_d;
    memset(cnt_u, 0, (size_t)(n_genomes + 1) * sizeof *cnt_u);
    memset(cnt_d, 0, (size_t)(n_genomes + 1) * sizeof *cnt_d);
    /* resetCounters (query.cpp:1820-1840).  10^8 heap nodes: on all cores, or this alone outlasts a bench sample */
    for (int t = 0; t < 2; t++) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)ix->ht[t].leaf_cnt; i++) ix->ht[t].leaves[i]->rcount = 0;
    }

    if (nthreads < 1) nthreads = 1;
    /* The reference's own "Time for query" (query.cpp:459,645-647) brackets the loop over the reads and nothing
     * else -- resetCounters runs between files, outside it -- so the loop is timed on its own here: zeroing and
     * reading back rcount of 10^8 heap nodes above / below takes longer than classifying a bench sample. */
    const double t_loop0 = cqo_now();
    if (variant == CQO_VAR_SERIAL) {
        for (uint64_t r = 0; r < n_reads; r++)
            cqo_classify_read(ix, mode, bases + offsets[r], (size_t)(offsets[r + 1] - offsets[r]), &acc, 0);
    } else if (variant == CQO_VAR_THREAD_LOCAL) {
#ifdef _OPENMP
        omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
        {
            cqo_acc loc;
            memset(&loc, 0, sizeof loc);
            loc.cnt_u = (uint64_t *)calloc((size_t)n_genomes + 1, sizeof *loc.cnt_u);
            loc.cnt_d = (uint64_t *)calloc((size_t)n_genomes + 1, sizeof *loc.cnt_d);
#pragma omp for schedule(dynamic, 512) nowait
            for (int64_t r = 0; r < (int64_t)n_reads; r++)
                cqo_classify_read(ix, mode, bases + offsets[r], (size_t)(offsets[r + 1] - offsets[r]), &loc, 3);
#pragma omp critical
            {
                for (uint32_t g = 0; g <= n_genomes; g++) { acc.cnt_u[g] += loc.cnt_u[g]; acc.cnt_d[g] += loc.cnt_d[g]; }
                acc.nundet += loc.nundet; acc.nconf += loc.nconf;
                for (int b = 0; b < CQO_BR_N; b++) acc.branch[b] += loc.branch[b];
                for (uint64_t i = 0; i < loc.npb; i++) {
                    uint64_t j = 0;
                    while (j < acc.npb && !(acc.pb[j].a == loc.pb[i].a && acc.pb[j].b == loc.pb[i].b)) j++;
                    if (j < acc.npb) acc.pbc[j] += loc.pbc[i];
                    else { cqo_pb_add(&acc, loc.pb[i]); acc.pbc[acc.npb - 1] = loc.pbc[i]; }
                }
            }
            free(loc.cnt_u); free(loc.cnt_d); free(loc.pb); free(loc.pbc);
        }
    } else {
        /* CQO_VAR_CRITICAL: one global critical section per update, as query64mt_p does;
         * CQO_VAR_ATOMIC: the same loop with atomic counter updates. */
        const int lock_mode = variant == CQO_VAR_ATOMIC ? 2 : 1;
#ifdef _OPENMP
        omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for
        for (int64_t r = 0; r < (int64_t)n_reads; r++)
            cqo_classify_read(ix, mode, bases + offsets[r], (size_t)(offsets[r + 1] - offsets[r]), &acc, lock_mode);
    }

    cqo_last_loop_s = cqo_now() - t_loop0;
    if (rcount_u) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)ix->ht[0].leaf_cnt; i++) rcount_u[i] = ix->ht[0].leaves[i]->rcount;
    }
    if (rcount_d) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)ix->ht[1].leaf_cnt; i++) rcount_d[i] = ix->ht[1].leaves[i]->rcount;
    }
    scal[0] = acc.nundet; scal[1] = acc.nconf;
    if (branch) memcpy(branch, acc.branch, sizeof acc.branch);
    if (n_pairs) {
        *n_pairs = acc.npb;
        for (uint64_t i = 0; i < acc.npb && i < pair_cap; i++) {
            pair_a[i] = acc.pb[i].a; pair_b[i] = acc.pb[i].b; pair_cnt[i] = acc.pbc[i];
        }
    }
    free(acc.pb); free(acc.pbc);
    return 0;
}

int cqo_omp_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
