// cq_synth.cpp -- benchmark-scale synthetic inputs (libcq_synth.so).  NOT on the hot path and
// not part of libcammiq_hip.so: this is the generator SURVEY.md section 8(d) calls for,
// because RefSeq cannot be downloaded here and the reference's build side is out of scope.
//
// It produces, deterministically from a seed:
//   * G random genomes (optionally in pairs that share blocks, so that doubly-unique
//     markers exist);
//   * index_u.bin1 / index_d.bin2 (+ .aux) in CAMMiQ's exact on-disk format
//     (/root/reference/src/hashtrie.cpp:599-699 through the conventions of
//     /root/reference/src/binaryio.cpp:11-123), with the marker statistics the survey
//     measured on real builds: one marker per ~69 positions per strand, ~93 % of keys of
//     length k, a geometric tail up to Lmax, ucount = 1;
//   * reads the way CAMMiQ-simulate draws them (/root/reference/CAMMiQ-simulate:242-273):
//     uniform genome, start and strand, per-base substitution errors, no N, plus a
//     fraction of random off-database reads.
// The small-scale, readable twin is cammiq_amd/synth.py; tests check that files written
// here decode identically through the oracle, the product and the Python decoder.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    inline uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    inline double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    inline uint64_t below(uint64_t n) { return (uint64_t)(((__uint128_t)next() * n) >> 64); }
};

inline uint64_t mix(uint64_t a, uint64_t b) { Rng r(a * 0x9E3779B97F4A7C15ull + b); r.next(); return r.next(); }

const char kAlpha[4] = {'A', 'C', 'G', 'T'};
inline uint32_t sym(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }
inline uint8_t comp(uint8_t c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }

struct Params {
    uint64_t seed;
    uint32_t n_genomes, genome_len, k, h, lmax, marker_every, block;
    double frac_deep, pair_share;
};

struct Marker {      // one key
    uint64_t hv;     // first h symbols
    uint32_t rid1, rid2;
    uint32_t genome; // 0-based source genome
    uint32_t pos;    // start on the forward strand of the source genome
    uint8_t len;     // key length
    uint8_t strand;  // 1: key is the reverse complement of genome[pos, pos+len)
    uint8_t table;   // 0 unique, 1 doubly unique
};

struct World {
    Params p;
    std::vector<std::vector<uint8_t>> genomes;  // ASCII
    std::vector<std::vector<uint8_t>> shared;   // per genome, per block: 1 = shared with its partner
};

void thread_pool(unsigned n_items, const std::function<void(unsigned)> &fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = std::max(1u, std::min(hw ? hw : 1u, 32u));
    std::atomic<unsigned> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&] { for (;;) { unsigned i = next.fetch_add(1); if (i >= n_items) break; fn(i); } });
    for (auto &x : th) x.join();
}

void fill_random(uint8_t *dst, size_t n, Rng &r)
{
    size_t i = 0;
    while (i < n) {
        uint64_t v = r.next();
        for (int j = 0; j < 32 && i < n; j++, v >>= 2) dst[i++] = (uint8_t)kAlpha[v & 3];
    }
}

// Key bytes of a marker (forward orientation of the key itself).
void key_bytes(const World &w, const Marker &m, uint8_t *out)
{
    const uint8_t *g = w.genomes[m.genome].data() + m.pos;
    if (!m.strand) memcpy(out, g, m.len);
    else for (uint32_t i = 0; i < m.len; i++) out[i] = comp(g[m.len - 1 - i]);
}

struct BitOut {
    std::vector<uint8_t> buf;
    uint32_t cur = 0, n = 0;
    inline void bit(uint32_t b)
    {
        cur = (cur << 1) | (b & 1); n++;
        if (n == 8) { buf.push_back((uint8_t)cur); cur = 0; n = 0; }
    }
    inline void bits(int c, uint32_t v) { for (int i = c - 1; i >= 0; i--) bit((v >> i) & 1); }
};

inline void be(std::vector<uint8_t> &o, uint64_t v, int nbytes)
{
    for (int i = nbytes - 1; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i)));
}

}  // namespace

extern "C" {

struct cqs_params {
    uint64_t seed;
    uint32_t n_genomes;
    uint32_t genome_len;
    uint32_t k;            /* minimum key length */
    uint32_t h;            /* hash length (<= k) */
    uint32_t lmax;         /* maximum key length */
    uint32_t marker_every; /* mean gap between marker starts, per strand */
    uint32_t block;        /* block size for pair sharing */
    double frac_deep;      /* fraction of keys longer than k */
    double pair_share;     /* 0: unrelated genomes, unique markers only; else P(block shared) */
};

void *cqs_create(const cqs_params *pp)
{
    World *w = new World();
    w->p = Params{pp->seed, pp->n_genomes, pp->genome_len, pp->k, pp->h, pp->lmax, pp->marker_every,
                  pp->block ? pp->block : 2048, pp->frac_deep, pp->pair_share};
    const Params &p = w->p;
    w->genomes.resize(p.n_genomes);
    w->shared.resize(p.n_genomes);
    const uint32_t nblk = (p.genome_len + p.block - 1) / p.block;
    thread_pool(p.n_genomes, [&](unsigned g) {
        w->genomes[g].resize(p.genome_len);
        Rng r(mix(p.seed, g));
        fill_random(w->genomes[g].data(), p.genome_len, r);
        w->shared[g].assign(nblk, 0);
    });
    if (p.pair_share > 0) {
        // genome 2i+1 copies the shared blocks of genome 2i
        thread_pool(p.n_genomes / 2, [&](unsigned i) {
            uint32_t a = 2 * i, b = 2 * i + 1;
            Rng r(mix(p.seed ^ 0xABCDEFull, i));
            for (uint32_t k = 0; k < nblk; k++)
                if (r.unit() < p.pair_share) {
                    w->shared[a][k] = w->shared[b][k] = 1;
                    size_t lo = (size_t)k * p.block, hi = std::min<size_t>(lo + p.block, p.genome_len);
                    memcpy(w->genomes[b].data() + lo, w->genomes[a].data() + lo, hi - lo);
                }
        });
    }
    return w;
}

void cqs_free(void *h) { delete (World *)h; }

/* Writes index_u (and index_d when path_d != NULL and pair_share > 0).  Returns 0. */
int cqs_write_index(void *hh, const char *path_u, const char *path_d, uint64_t *n_leaves_u, uint64_t *n_leaves_d)
{
    World &w = *(World *)hh;
    const Params &p = w.p;
    const bool both = path_d && path_d[0] && p.pair_share > 0;
    std::vector<std::vector<Marker>> per(p.n_genomes);
    thread_pool(p.n_genomes, [&](unsigned g) {
        Rng r(mix(p.seed ^ 0x5151ull, g));
        std::vector<Marker> &out = per[g];
        const uint8_t *G = w.genomes[g].data();
        for (int strand = 0; strand < 2; strand++) {
            uint64_t pos = r.below(p.marker_every);
            for (;;) {
                uint32_t len = p.k;
                if (p.lmax > p.k && r.unit() < p.frac_deep) {
                    len = p.k + 1;
                    while (len < p.lmax && r.unit() < 0.75) len++;
                }
                if (pos + len > p.genome_len) break;
                const uint32_t b0 = (uint32_t)(pos / p.block), b1 = (uint32_t)((pos + len - 1) / p.block);
                uint64_t gap = 1 + r.below(2ull * p.marker_every - 1);
                if (b0 == b1) {   // keys never straddle a block boundary
                    const bool sh = w.shared[g][b0] != 0;
                    // a shared block is indexed once, from the even genome of the pair
                    if (!sh || (both && (g & 1u) == 0)) {
                        Marker m;
                        m.genome = g; m.pos = (uint32_t)pos; m.len = (uint8_t)len; m.strand = (uint8_t)strand;
                        m.table = sh ? 1 : 0;
                        m.rid1 = g + 1; m.rid2 = sh ? g + 2 : 0;
                        uint64_t hv = 0;
                        if (!strand) for (uint32_t i = 0; i < p.h; i++) hv = (hv << 2) | sym(G[pos + i]);
                        else for (uint32_t i = 0; i < p.h; i++) hv = (hv << 2) | (3u - sym(G[pos + len - 1 - i]));
                        m.hv = hv;
                        out.push_back(m);
                    }
                }
                pos += gap;
            }
        }
    });
    for (int table = 0; table < (both ? 2 : 1); table++) {
        std::vector<Marker> all;
        size_t tot = 0;
        for (auto &v : per) tot += v.size();
        all.reserve(tot);
        for (auto &v : per) for (auto &m : v) if (m.table == table) all.push_back(m);
        // one key per bucket: sort by hv, drop every bucket that occurs more than once
        std::sort(all.begin(), all.end(), [](const Marker &a, const Marker &b) { return a.hv < b.hv; });
        size_t wr = 0;
        for (size_t i = 0; i < all.size();) {
            size_t j = i + 1;
            while (j < all.size() && all[j].hv == all[i].hv) j++;
            if (j == i + 1) all[wr++] = all[i];
            i = j;
        }
        all.resize(wr);
        // deterministic shuffle: file order is arbitrary in the reference (robin_hood iteration)
        {
            Rng r(mix(p.seed ^ 0x77ull, table));
            for (size_t i = all.size(); i > 1; i--) std::swap(all[i - 1], all[r.below(i)]);
        }
        BitOut aux;
        std::vector<uint8_t> ints;
        ints.reserve(all.size() * (table ? 20 : 14) + 16);
        aux.buf.reserve(all.size() + 64);
        aux.bit(table ? 1 : 0);
        aux.bits(7, 64);
        aux.bits(8, p.h);
        uint8_t key[256];
        for (const Marker &m : all) {
            be(ints, m.hv, 8);
            key_bytes(w, m, key);
            // single-path trie below the bucket root: for every inner level '1' then the four
            // child slots ('0' except the path symbol, which recurses); the leaf is '1 0000'.
            // Pre-order means the closing '0's of a level come after the whole subtree.
            uint32_t depth = m.len - p.h;
            std::vector<uint8_t> closing;  // zeros still owed per level
            for (uint32_t d = 0; d < depth; d++) {
                aux.bit(1);
                uint32_t s = sym(key[p.h + d]);
                for (uint32_t c = 0; c < s; c++) aux.bit(0);
                closing.push_back((uint8_t)(3 - s));
            }
            aux.bit(1); aux.bit(0); aux.bit(0); aux.bit(0); aux.bit(0);
            for (size_t d = closing.size(); d-- > 0;)
                for (uint8_t c = 0; c < closing[d]; c++) aux.bit(0);
            if (table) { be(ints, m.rid1, 4); be(ints, m.rid2, 4); be(ints, 1, 2); be(ints, 1, 2); }
            else { be(ints, m.rid1, 4); be(ints, 1, 2); }
        }
        for (int i = 0; i < 72; i++) aux.bit(1);
        be(ints, 0xFFFFFFFFFFFFFFFFull, 8);
        be(ints, 0xFFFF, 2);
        std::string path = table ? path_d : path_u;
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) return -1;
        fwrite(ints.data(), 1, ints.size(), f);
        fclose(f);
        f = fopen((path + ".aux").c_str(), "wb");
        if (!f) return -1;
        fwrite(aux.buf.data(), 1, aux.buf.size(), f);
        fclose(f);
        if (table == 0 && n_leaves_u) *n_leaves_u = all.size();
        if (table == 1 && n_leaves_d) *n_leaves_d = all.size();
    }
    if (!both && n_leaves_d) *n_leaves_d = 0;
    return 0;
}

/* Reads first .. first+n-1 of the stream `seed`, fixed length len, into bases_out[n*len] (ASCII).
 * Deterministic per (seed, read index): any chunking of a stream gives the same reads.  Substitution
 * errors are placed by geometric gaps (one draw per error, not one per base). */
int cqs_make_reads_at(void *hh, uint64_t seed, uint64_t first, uint64_t n, uint32_t len, double err, double frac_random,
                      uint8_t *bases_out)
{
    World &w = *(World *)hh;
    const Params &p = w.p;
    if (len > p.genome_len) return -1;
    const unsigned chunks = 256;
    const double inv_log_q = (err > 0 && err < 1) ? 1.0 / std::log1p(-err) : 0.0;
    thread_pool(chunks, [&](unsigned c) {
        uint64_t lo = n * c / chunks, hi = n * (c + 1) / chunks;
        for (uint64_t i = lo; i < hi; i++) {
            Rng r(mix(seed, first + i));
            uint8_t *out = bases_out + i * len;
            if (r.unit() < frac_random) { fill_random(out, len, r); continue; }
            uint32_t g = (uint32_t)r.below(p.n_genomes);
            uint64_t st = r.below((uint64_t)p.genome_len - len + 1);
            const uint8_t *G = w.genomes[g].data() + st;
            if (r.next() & 1) for (uint32_t j = 0; j < len; j++) out[j] = comp(G[len - 1 - j]);
            else memcpy(out, G, len);
            if (err >= 1) { for (uint32_t j = 0; j < len; j++) out[j] = (uint8_t)kAlpha[(sym(out[j]) + 1 + r.below(3)) & 3]; }
            else if (err > 0)
                for (double j = std::floor(std::log(1.0 - r.unit()) * inv_log_q); j < (double)len;
                     j += 1.0 + std::floor(std::log(1.0 - r.unit()) * inv_log_q)) {
                    const uint32_t q = (uint32_t)j;
                    out[q] = (uint8_t)kAlpha[(sym(out[q]) + 1 + r.below(3)) & 3];
                }
        }
    });
    return 0;
}

int cqs_make_reads(void *hh, uint64_t seed, uint64_t n, uint32_t len, double err, double frac_random, uint8_t *bases_out)
{
    return cqs_make_reads_at(hh, seed, 0, n, len, err, frac_random, bases_out);
}

}  // extern "C"
