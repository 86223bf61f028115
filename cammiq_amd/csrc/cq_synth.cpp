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
//
// Built for configs[4]'s size (15 000 genomes, > 10^9 markers): genomes are never stored -- every base is a pure
// function of (seed, genome, position) through a counter-based generator, so a worker re-creates the genome it is
// enumerating (3 MB) and a read fetches its 150 bases directly --, markers are 16-byte records partitioned by a
// bijective mix of their h-mer, sorted and de-duplicated per partition on all cores, and both files of a table
// are written in place into a mapping of the output file, partition by partition, at bit offsets known from a
// prefix sum.  File order = ascending mix(h-mer): arbitrary but deterministic, like the reference's robin_hood
// iteration order is arbitrary.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr uint64_t kGamma = 0x9E3779B97F4A7C15ull;

inline uint64_t finish(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Rng {   // splitmix64: the j-th output is a pure function of (seed, j) -- see rng_at
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    inline uint64_t next() { return finish(s += kGamma); }
    inline double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    inline uint64_t below(uint64_t n) { return (uint64_t)(((__uint128_t)next() * n) >> 64); }
};

// Output of call number j (0-based) of Rng(seed).next(): random access into the stream.
inline uint64_t rng_at(uint64_t seed, uint64_t j) { return finish(seed + (j + 1) * kGamma); }

inline uint64_t mix(uint64_t a, uint64_t b) { Rng r(a * kGamma + b); r.next(); return r.next(); }

const char kAlpha[4] = {'A', 'C', 'G', 'T'};
inline uint32_t sym(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }
inline uint8_t comp(uint8_t c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }

struct Params {
    uint64_t seed;
    uint32_t n_genomes, genome_len, k, h, lmax, marker_every, block;
    double frac_deep, pair_share;
};

struct World { Params p; };

unsigned g_worker_cap = 32;   // (64 workers measured on a 1.26e9-marker world: in-place emission into the mapping got slower)
unsigned n_workers()
{
    unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hw ? hw : 1u, g_worker_cap));
}

// fn(item, worker): items handed out dynamically, worker = index of the thread that runs it.
void thread_pool(size_t n_items, const std::function<void(size_t, unsigned)> &fn)
{
    const unsigned nt = (unsigned)std::min<size_t>(n_workers(), std::max<size_t>(n_items, 1));
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] { for (;;) { size_t i = next.fetch_add(1); if (i >= n_items) break; fn(i, t); } });
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

// ---- the virtual genomes.  Genome g is the stream Rng(mix(seed, g)), 32 bases per draw, low bits first.  With
// pair_share > 0 the genomes come in pairs (2i, 2i+1): block k of the pair is shared when draw k of
// Rng(mix(seed ^ 0xABCDEF, i)) falls below pair_share, and a shared block of the odd genome IS the even genome's.
inline bool block_shared(const Params &p, uint32_t g, uint32_t blk)
{
    if (!(p.pair_share > 0) || g >= 2u * (p.n_genomes / 2u)) return false;
    const uint64_t v = rng_at(mix(p.seed ^ 0xABCDEFull, g / 2u), blk);
    return (double)(v >> 11) * (1.0 / 9007199254740992.0) < p.pair_share;
}

// Bases [st, st + n) of genome g (ASCII).
void genome_fetch(const Params &p, uint32_t g, uint64_t st, uint32_t n, uint8_t *out)
{
    uint64_t pos = st;
    const uint64_t end = st + n;
    while (pos < end) {
        const uint32_t blk = (uint32_t)(pos / p.block);
        const uint64_t bend = std::min<uint64_t>(end, (uint64_t)(blk + 1) * p.block);
        const uint32_t src = ((g & 1u) && block_shared(p, g, blk)) ? g - 1u : g;
        const uint64_t sd = mix(p.seed, src);
        while (pos < bend) {
            uint64_t v = rng_at(sd, pos / 32) >> (2 * (pos % 32));
            const uint64_t stop = std::min<uint64_t>(bend, (pos / 32 + 1) * 32);
            for (; pos < stop; pos++, v >>= 2) *out++ = (uint8_t)kAlpha[v & 3];
        }
    }
}

// The whole genome g into buf (genome_len bytes) + its blocks' shared flags.
void genome_materialise(const Params &p, uint32_t g, uint8_t *buf, std::vector<uint8_t> &shared)
{
    Rng r(mix(p.seed, g));
    fill_random(buf, p.genome_len, r);
    const uint32_t nblk = (p.genome_len + p.block - 1) / p.block;
    shared.assign(nblk, 0);
    for (uint32_t k = 0; k < nblk; k++)
        if (block_shared(p, g, k)) {
            shared[k] = 1;
            if (g & 1u) {
                const size_t lo = (size_t)k * p.block, hi = std::min<size_t>(lo + p.block, p.genome_len);
                genome_fetch(p, g, lo, (uint32_t)(hi - lo), buf + lo);
            }
        }
}

// One key: 16 bytes.  refIDs follow from genome and table (unique: g+1; doubly unique: g+1, g+2).
struct Marker {
    uint64_t hv;     // first h symbols of the key
    uint32_t pos;    // start on the forward strand of the source genome
    uint32_t gsl;    // genome << 9 | strand << 8 | key length
    inline uint32_t genome() const { return gsl >> 9; }
    inline uint32_t strand() const { return (gsl >> 8) & 1u; }
    inline uint32_t len() const { return gsl & 255u; }
};

// File order: ascending in a bijective mix of the h-mer (equal h-mers stay adjacent for the duplicate test).
inline uint64_t order_key(uint64_t hv) { return finish(hv * kGamma + 0x1234567ull); }
constexpr unsigned kPartBits = 10, kParts = 1u << kPartBits;

// Key symbols h .. len-1 of a marker (the trie part), 0..3, key orientation.
void deep_symbols(const Params &p, const Marker &m, uint8_t *out)
{
    uint8_t tmp[256];
    const uint32_t len = m.len(), d = len - p.h;
    if (!m.strand()) {
        genome_fetch(p, m.genome(), (uint64_t)m.pos + p.h, d, tmp);
        for (uint32_t i = 0; i < d; i++) out[i] = (uint8_t)sym(tmp[i]);
    } else {   // key = reverse complement of genome[pos, pos + len): key[h + i] = comp(genome[pos + len - 1 - h - i])
        genome_fetch(p, m.genome(), m.pos, d, tmp);
        for (uint32_t i = 0; i < d; i++) out[i] = (uint8_t)(3u - sym(tmp[d - 1 - i]));
    }
}

struct OutMap {   // output file, written in place
    uint8_t *p = nullptr;
    size_t n = 0;
    bool open(const std::string &path, size_t bytes)
    {
        int fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return false;
        n = bytes;
        if (bytes == 0) { ::close(fd); return true; }
        if (ftruncate(fd, (off_t)bytes) != 0) { ::close(fd); return false; }
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return false;
        p = (uint8_t *)m;
        return true;
    }
    ~OutMap() { if (p) munmap(p, n); }
};

inline void be(uint8_t *&o, uint64_t v, int nbytes)
{
    for (int i = nbytes - 1; i >= 0; i--) *o++ = (uint8_t)(v >> (8 * i));
}

}  // namespace

// Steps 3 + 4 of cqs_write_index for one table: `part` = kParts lists of markers in file order.
static bool emit_table(const Params &p, int table, const std::vector<std::vector<Marker>> &part, const std::string &path,
                       uint64_t &n_leaves_out)
{
    const size_t rec = table ? 12 : 6;
    std::vector<uint64_t> nbits(kParts, 0), nbytes(kParts, 0);
    thread_pool(kParts, [&](size_t q, unsigned) {
        uint64_t bits = 0;
        for (const Marker &m : part[q]) bits += 5u + 4u * (m.len() - p.h);   // inner level: '1' + 4 child slots; leaf: '1 0000'
        nbits[q] = bits;
        nbytes[q] = (uint64_t)part[q].size() * (8 + rec);
    });
    // ---- 3. offsets.  aux: 16 header bits, the partitions' bits, 72 one-bits, the final partial byte dropped
    //         (binaryio.cpp:115-118 as the reference's writer leaves it); ints: records, END64, 0xFFFF
    std::vector<uint64_t> bit0(kParts + 1), byte0(kParts + 1);
    bit0[0] = 16; byte0[0] = 0;
    uint64_t n_leaves = 0;
    for (unsigned q = 0; q < kParts; q++) {
        bit0[q + 1] = bit0[q] + nbits[q];
        byte0[q + 1] = byte0[q] + nbytes[q];
        n_leaves += part[q].size();
    }
    const uint64_t aux_bytes = (bit0[kParts] + 72) / 8, int_bytes = byte0[kParts] + 10;
    OutMap fi, fa;
    if (!fi.open(path, int_bytes) || !fa.open(path + ".aux", aux_bytes)) return false;
    // ---- 4. emit in place.  A partition's bit range shares its first and last byte with its neighbours: those
    //         two bytes are OR-ed in afterwards, everything between is written by the partition's worker alone.
    std::vector<uint64_t> edge_off(2 * kParts, ~0ull);
    std::vector<uint8_t> edge_val(2 * kParts, 0);
    thread_pool(kParts, [&](size_t q, unsigned) {
        const std::vector<Marker> &all = part[q];
        uint8_t *o = fi.p + byte0[q];
        const uint64_t b_lo = bit0[q], b_hi = bit0[q + 1];
        const uint64_t first = b_lo / 8, last = b_hi ? (b_hi - 1) / 8 : 0;
        std::vector<uint8_t> loc(b_hi > b_lo ? (size_t)(last - first + 1) : 0, 0);
        uint64_t bp = b_lo - first * 8;   // bit cursor within loc
        auto put = [&](uint32_t bit) { if (bit) loc[bp >> 3] |= (uint8_t)(0x80u >> (bp & 7)); bp++; };
        uint8_t syms[256];
        for (const Marker &m : all) {
            be(o, m.hv, 8);
            const uint32_t depth = m.len() - p.h;
            if (depth) deep_symbols(p, m, syms);
            // single-path trie below the bucket root, pre-order: per inner level '1', then '0' for the child slots
            // before the path symbol; the leaf is '1 0000'; the slots after the path symbol close level by level
            for (uint32_t d = 0; d < depth; d++) { put(1); for (uint32_t c = 0; c < syms[d]; c++) put(0); }
            put(1); put(0); put(0); put(0); put(0);
            for (uint32_t d = depth; d-- > 0;) for (uint32_t c = syms[d]; c < 3; c++) put(0);
            const uint32_t g = m.genome();
            if (table) { be(o, g + 1, 4); be(o, g + 2, 4); be(o, 1, 2); be(o, 1, 2); }
            else { be(o, g + 1, 4); be(o, 1, 2); }
        }
        if (loc.empty()) return;
        if (loc.size() > 2) memcpy(fa.p + first + 1, loc.data() + 1, loc.size() - 2);
        edge_off[2 * q] = first; edge_val[2 * q] = loc[0];
        if (loc.size() > 1) { edge_off[2 * q + 1] = last; edge_val[2 * q + 1] = loc.back(); }
    });
    fa.p[0] = (uint8_t)((table ? 0x80u : 0u) | 64u);
    fa.p[1] = (uint8_t)p.h;
    for (size_t e = 0; e < edge_off.size(); e++)
        if (edge_off[e] != ~0ull && edge_off[e] < aux_bytes) fa.p[edge_off[e]] |= edge_val[e];
    for (uint64_t b = bit0[kParts]; b < aux_bytes * 8; b++) fa.p[b >> 3] |= (uint8_t)(0x80u >> (b & 7));
    uint8_t *o = fi.p + byte0[kParts];
    be(o, 0xFFFFFFFFFFFFFFFFull, 8);
    be(o, 0xFFFF, 2);
    n_leaves_out = n_leaves;
    return true;
}

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
    if (pp->n_genomes >= (1u << 23) || pp->lmax > 255 || pp->h > 31 || pp->h > pp->k) return nullptr;
    World *w = new World();
    w->p = Params{pp->seed, pp->n_genomes, pp->genome_len, pp->k, pp->h, pp->lmax, pp->marker_every,
                  pp->block ? pp->block : 2048, pp->frac_deep, pp->pair_share};
    return w;
}

void cqs_free(void *h) { delete (World *)h; }

/* Bases [start, start+n) of genome g (0-based), ASCII: what a FASTA of the world would hold. */
int cqs_genome_bases(void *hh, uint32_t g, uint64_t start, uint32_t n, uint8_t *out)
{
    World &w = *(World *)hh;
    if (g >= w.p.n_genomes || start + n > w.p.genome_len) return -1;
    genome_fetch(w.p, g, start, n, out);
    return 0;
}

/* A second, much smaller index beside the full one: exactly the markers of the full index whose h-mer is in the
 * sorted set hv[0..n_hv), in the full index's file order, plus the position of each in the full index's decode
 * order (ids: sub_n[0] positions of the unique table, then sub_n[1] of the doubly-unique one).  A lookup of an
 * h-mer of the set gives the same answer in both indices (find64_p starts with map64.find(h-mer),
 * hashtrie.cpp:350-352, and both hold the same bucket for it), so reads whose every window is in the set classify
 * identically against either: an oracle that cannot hold 10^9 markers checks a slice of reads against the
 * sub-index and maps rcount through ids. */
struct SubSpec { const uint64_t *hv; uint64_t n_hv; const char *path_u, *path_d; uint64_t *ids; uint64_t ids_cap; uint64_t n[2]; };

static int write_index_impl(void *hh, const char *path_u, const char *path_d, uint64_t *n_leaves_u, uint64_t *n_leaves_d, SubSpec *sub);

/* Writes index_u (and index_d when path_d != NULL and pair_share > 0).  Returns 0. */
int cqs_write_index(void *hh, const char *path_u, const char *path_d, uint64_t *n_leaves_u, uint64_t *n_leaves_d)
{
    return write_index_impl(hh, path_u, path_d, n_leaves_u, n_leaves_d, nullptr);
}

int cqs_write_index_with_sub(void *hh, const char *path_u, const char *path_d, uint64_t *n_leaves_u, uint64_t *n_leaves_d,
                             const uint64_t *hv_sorted, uint64_t n_hv, const char *sub_path_u, const char *sub_path_d,
                             uint64_t *ids, uint64_t ids_cap, uint64_t *sub_n_u, uint64_t *sub_n_d)
{
    SubSpec sub{hv_sorted, n_hv, sub_path_u, sub_path_d, ids, ids_cap, {0, 0}};
    const int rc = write_index_impl(hh, path_u, path_d, n_leaves_u, n_leaves_d, &sub);
    if (sub_n_u) *sub_n_u = sub.n[0];
    if (sub_n_d) *sub_n_d = sub.n[1];
    return rc;
}

static int write_index_impl(void *hh, const char *path_u, const char *path_d, uint64_t *n_leaves_u, uint64_t *n_leaves_d, SubSpec *sub)
{
    World &w = *(World *)hh;
    const Params &p = w.p;
    const bool both = path_d && path_d[0] && p.pair_share > 0;
    const unsigned nt = n_workers();
    const bool timing = getenv("CQS_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char *what) { if (timing) { const double t1 = now(); fprintf(stderr, "[cq_synth] %-28s %7.2f s\n", what, t1 - t0); t0 = t1; } };
    // ---- 1. enumerate: worker t keeps kParts bins per table; a genome is re-created, scanned and forgotten
    std::vector<std::vector<std::vector<Marker>>> bins[2];
    for (int t = 0; t < 2; t++) bins[t].assign(nt, std::vector<std::vector<Marker>>(kParts));
    {   // reserve what a bin is expected to take (no doubling copies of multi-GB totals)
        const double total = 2.0 * p.genome_len / std::max(1u, p.marker_every) * p.n_genomes;
        const double sh = both ? p.pair_share : 0.0;
        const size_t r0 = (size_t)(total * (1.0 - sh) / nt / kParts * 1.2) + 16, r1 = (size_t)(total * sh / 2 / nt / kParts * 1.2) + 16;
        thread_pool(nt, [&](size_t t, unsigned) {
            for (unsigned q = 0; q < kParts; q++) { bins[0][t][q].reserve(r0); if (sh > 0) bins[1][t][q].reserve(r1); }
        });
    }
    std::vector<std::vector<uint8_t>> gbuf(nt), gshared(nt);
    thread_pool(p.n_genomes, [&](size_t gi, unsigned t) {
        const uint32_t g = (uint32_t)gi;
        if (gbuf[t].size() < p.genome_len) gbuf[t].resize(p.genome_len);
        genome_materialise(p, g, gbuf[t].data(), gshared[t]);
        const uint8_t *G = gbuf[t].data();
        const std::vector<uint8_t> &shared = gshared[t];
        Rng r(mix(p.seed ^ 0x5151ull, g));
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
                    const bool sh = shared[b0] != 0;
                    // a shared block is indexed once, from the even genome of the pair
                    if (!sh || (both && (g & 1u) == 0)) {
                        uint64_t hv = 0;
                        if (!strand) for (uint32_t i = 0; i < p.h; i++) hv = (hv << 2) | sym(G[pos + i]);
                        else for (uint32_t i = 0; i < p.h; i++) hv = (hv << 2) | (3u - sym(G[pos + len - 1 - i]));
                        Marker m{hv, (uint32_t)pos, (g << 9) | ((uint32_t)strand << 8) | len};
                        bins[sh ? 1 : 0][t][order_key(hv) >> (64 - kPartBits)].push_back(m);
                    }
                }
                pos += gap;
            }
        }
    });
    gbuf.clear(); gshared.clear();
    lap("enumerate markers");

    for (int table = 0; table < (both ? 2 : 1); table++) {
        // ---- 2. per partition: gather the workers' bins, sort by order_key, drop every h-mer that occurs more than once
        //         (one key per bucket: two keys with one h-mer prefix would need a branching trie)
        std::vector<std::vector<Marker>> part(kParts);
        thread_pool(kParts, [&](size_t q, unsigned) {
            std::vector<Marker> &all = part[q];
            size_t tot = 0;
            for (unsigned t = 0; t < nt; t++) tot += bins[table][t][q].size();
            all.reserve(tot);
            for (unsigned t = 0; t < nt; t++) {
                std::vector<Marker> &b = bins[table][t][q];
                all.insert(all.end(), b.begin(), b.end());
                std::vector<Marker>().swap(b);
            }
            std::sort(all.begin(), all.end(), [](const Marker &a, const Marker &b) {
                const uint64_t ka = order_key(a.hv), kb = order_key(b.hv);
                return ka != kb ? ka < kb : (a.gsl != b.gsl ? a.gsl < b.gsl : a.pos < b.pos);
            });
            size_t wr = 0;
            for (size_t i = 0; i < all.size();) {
                size_t j = i + 1;
                while (j < all.size() && all[j].hv == all[i].hv) j++;
                if (j == i + 1) all[wr++] = all[i];
                i = j;
            }
            all.resize(wr);
        });
        lap(table ? "sort + dedupe (d)" : "sort + dedupe (u)");
        uint64_t n_leaves = 0;
        if (!emit_table(p, table, part, table ? path_d : path_u, n_leaves)) return -1;
        if (sub && (table ? sub->path_d : sub->path_u)) {
            // the sub-index: markers whose h-mer is in the set (a 2^27-bit filter in front of the binary search)
            std::vector<uint64_t> filt(1u << 21, 0);
            for (uint64_t i = 0; i < sub->n_hv; i++) { const uint64_t k = order_key(sub->hv[i]) >> 37; filt[k >> 6] |= 1ull << (k & 63); }
            std::vector<std::vector<Marker>> spart(kParts);
            std::vector<std::vector<uint64_t>> spos(kParts);
            std::vector<uint64_t> base(kParts + 1, 0);
            for (unsigned q = 0; q < kParts; q++) base[q + 1] = base[q] + part[q].size();
            thread_pool(kParts, [&](size_t q, unsigned) {
                for (size_t i = 0; i < part[q].size(); i++) {
                    const uint64_t hv = part[q][i].hv, k = order_key(hv) >> 37;
                    if (!((filt[k >> 6] >> (k & 63)) & 1u)) continue;
                    if (!std::binary_search(sub->hv, sub->hv + sub->n_hv, hv)) continue;
                    spart[q].push_back(part[q][i]);
                    spos[q].push_back(base[q] + i);   // one leaf per bucket: position in file order == leaf id in decode order
                }
            });
            uint64_t ns = 0;
            if (!emit_table(p, table, spart, table ? sub->path_d : sub->path_u, ns)) return -1;
            uint64_t at = table ? sub->n[0] : 0;
            for (unsigned q = 0; q < kParts; q++)
                for (uint64_t v : spos[q]) { if (at < sub->ids_cap) sub->ids[at] = v; at++; }
            sub->n[table] = ns;
            if (at > sub->ids_cap) return -2;
        }
        lap(table ? "emit (d)" : "emit (u)");
        if (table == 0 && n_leaves_u) *n_leaves_u = n_leaves;
        if (table == 1 && n_leaves_d) *n_leaves_d = n_leaves;
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
    thread_pool(chunks, [&](size_t c, unsigned) {
        uint64_t lo = n * c / chunks, hi = n * (c + 1) / chunks;
        for (uint64_t i = lo; i < hi; i++) {
            Rng r(mix(seed, first + i));
            uint8_t *out = bases_out + i * len;
            if (r.unit() < frac_random) { fill_random(out, len, r); continue; }
            uint32_t g = (uint32_t)r.below(p.n_genomes);
            uint64_t st = r.below((uint64_t)p.genome_len - len + 1);
            uint8_t G[256 + 8];
            const bool big = len > 256;
            std::vector<uint8_t> gl;
            uint8_t *Gp = G;
            if (big) { gl.resize(len); Gp = gl.data(); }
            genome_fetch(p, g, st, len, Gp);
            if (r.next() & 1) for (uint32_t j = 0; j < len; j++) out[j] = comp(Gp[len - 1 - j]);
            else memcpy(out, Gp, len);
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
