// cq_layout.cpp -- turn two decoded tables into the flat HBM image.
//
// Replaces the reference's in-memory index (robin_hood::unordered_map<uint64_t, trieNode*>
// + heap pointer trie, /root/reference/src/hashtrie.hpp:8-13,49; hashtrie.cpp:8-13) by:
//
//   * ONE merged open-addressing table for ht_u and ht_d: a slot is {key = hv, val_u, val_d},
//     4 slots form a 64-byte bucket = one HBM access, stored as a structure of arrays
//     (cq_device.h) so that the probe's detect step reads only the four key_lo words; the two
//     find64_p calls the reference makes per window (query.cpp:487-492) cost one probe.
//     A key's home bucket is chosen by its canonical MINIMIZER (cq_device.h), not by the
//     key itself: both strands of a window, and runs of neighbouring windows, then share one
//     bucket, which cuts the random HBM accesses per read by an order of magnitude.
//     Collisions stay inside the home bucket; a full bucket sets an overflow bit and spills
//     to the next bucket(s).  A lookup that does not see the overflow bit stops after one
//     bucket -- which is what most of the (miss-dominated) probes do.
//   * a linked, path-compressed array trie (16-byte nodes: 4 child codes, or a chain of up to
//     32 symbols + next) for keys longer than h.
//   * leaf refIDs as two flat uint32 arrays indexed by global leaf id (u leaves first).
//
// Placement is a deterministic sort + linear sweep (no hashing races, sequential writes):
// keys sorted by home bucket are dealt into buckets in order; what does not fit is carried
// to the next bucket, whose predecessor gets the overflow bit.  The sweep never wraps: the
// table has CQ_SPILL_TAIL buckets past the hash range.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

#include "cq_index.hpp"

#include <sys/mman.h>
#include <new>

namespace cq {

void advise_huge(const void *p, size_t bytes)
{
    const uintptr_t two_mb = 2u << 20;
    if (!p || bytes < (32u << 20)) return;
    const uintptr_t lo = ((uintptr_t)p + two_mb - 1) / two_mb * two_mb, hi = ((uintptr_t)p + bytes) / two_mb * two_mb;
    if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);   // whole 2 MiB frames inside the block; a hint only
}

// Give the pages of a large block that is about to be freed back to the OS in 64 MiB steps.  madvise(MADV_DONTNEED) zaps
// pages under the READ side of the process's mmap lock, the side page faults take too; the munmap / free that follows finds
// nothing left to zap and holds the WRITE side for microseconds.  (One munmap of several GB holds the write side for its whole
// duration: every page fault of the process -- a query's first touch of its output arrays -- waits behind it.)
void release_pages(const void *p, size_t bytes)
{
    const uintptr_t page = 4096, step = 64u << 20;
    if (!p || bytes < (32u << 20)) return;
    const uintptr_t lo = ((uintptr_t)p + page - 1) / page * page, hi = ((uintptr_t)p + bytes) / page * page;
    for (uintptr_t a = lo; a < hi; a += step) (void)madvise((void *)a, std::min<uintptr_t>(step, hi - a), MADV_DONTNEED);
}

void HugeWords::alloc(size_t words)
{
    reset();
    const size_t bytes = (words ? words : 1) * sizeof(uint32_t);
    if (bytes < (32u << 20)) {
        p_ = new uint32_t[words ? words : 1];
        return;
    }
    const size_t two_mb = 2u << 20;
    const size_t len = ((bytes + two_mb - 1) / two_mb + 1) * two_mb;   // room to align the start
    void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw std::bad_alloc();
    base_ = m;
    map_len_ = len;
    p_ = (uint32_t *)(((uintptr_t)m + two_mb - 1) / two_mb * two_mb);
    (void)madvise(p_, (bytes + two_mb - 1) / two_mb * two_mb, MADV_HUGEPAGE);   // a hint: plain pages work too
}

void HugeWords::reset()
{
    // A mapping of tens of GB goes back in pieces: one munmap of configs[4]'s 80 GB image holds the process's
    // mmap lock for ~5 s, during which every other thread's mmap / hipMalloc / stream creation waits (measured:
    // "streams 4831 ms" in the load timing, while the release thread was running); 256 MiB pieces hold it for
    // milliseconds each.
    if (base_) {
        release_pages(base_, map_len_);
        const size_t piece = 256u << 20;
        size_t left = map_len_;
        while (left > piece) { left -= piece; (void)munmap((char *)base_ + left, piece); }
        (void)munmap(base_, left);
    } else delete[] p_;
    base_ = nullptr;
    map_len_ = 0;
    p_ = nullptr;
}

namespace {

struct Entry {
    uint32_t home;
    uint32_t seq;      // file order inside (table, file): later duplicate wins (map64[b] = root)
    uint64_t key;
    uint32_t val_u, val_d;
};

// Order by (home bucket, key).  Equal keys (the same h-mer in both tables, or twice in one file)
// end up adjacent in any order: the merge step looks at the whole group.
inline bool entry_less(const Entry &a, const Entry &b)
{
    if (a.home != b.home) return a.home < b.home;
    return a.key < b.key;
}

unsigned worker_count(size_t n_items)
{
    if (const char *v = getenv("CAMMIQ_LAYOUT_THREADS")) return (unsigned)std::max(1, atoi(v));   // tests: 1 = serial
    unsigned hw = std::thread::hardware_concurrency();
    return n_items < (1u << 16) ? 1u : std::max(1u, std::min(hw ? hw : 1u, 32u));   // (64 workers measured at 1.26e9 keys: no faster)
}

template <class F>
void parallel_for(unsigned nt, F f)
{
    if (nt <= 1) { f(0u); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(f, t);
    for (auto &x : th) x.join();
}

// Sort by (home, key): partition by the top bits of `home` (parallel count + parallel scatter),
// then sort the parts on all cores.  part_begin[p] .. part_begin[p+1] are the parts of the result.
constexpr unsigned kParts = 256;

// The result lives in a fresh uninitialised buffer (first touched by the scattering threads); `e` is released.
std::unique_ptr<Entry[]> sort_entries(std::unique_ptr<Entry[]> &e, size_t n, uint32_t n_buckets, std::vector<size_t> &part_begin)
{
    const unsigned nt = worker_count(n);
    auto part_of = [&](uint32_t home) { return (unsigned)(((uint64_t)home * kParts) / n_buckets); };
    std::vector<size_t> cnt((size_t)nt * kParts, 0);
    parallel_for(nt, [&](unsigned t) {
        const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
        size_t *c = &cnt[(size_t)t * kParts];
        for (size_t i = lo; i < hi; i++) c[part_of(e[i].home)]++;
    });
    // exclusive prefix over (part, thread): thread t's entries of part p go after threads < t's
    part_begin.assign(kParts + 1, 0);
    size_t run = 0;
    for (unsigned p = 0; p < kParts; p++) {
        part_begin[p] = run;
        for (unsigned t = 0; t < nt; t++) { const size_t c = cnt[(size_t)t * kParts + p]; cnt[(size_t)t * kParts + p] = run; run += c; }
    }
    part_begin[kParts] = run;
    std::unique_ptr<Entry[]> out(new Entry[n ? n : 1]);
    advise_huge(out.get(), n * sizeof(Entry));
    Entry *tmp = out.get();
    parallel_for(nt, [&](unsigned t) {
        const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
        size_t *c = &cnt[(size_t)t * kParts];
        for (size_t i = lo; i < hi; i++) tmp[c[part_of(e[i].home)]++] = e[i];
    });
    e.reset();
    std::atomic<unsigned> next{0};
    parallel_for(nt, [&](unsigned) {
        for (;;) {
            const unsigned p = next.fetch_add(1);
            if (p >= kParts) break;
            std::sort(tmp + part_begin[p], tmp + part_begin[p + 1], entry_less);
        }
    });
    return out;
}

// Path compression of the array trie.  The reference walks one heap node per base
// (hashtrie.cpp:350-369); on the GPU every level is a dependent ~1 us memory round trip, and a
// key longer than h is almost always the only key below its bucket, i.e. a single-child chain.
// A chain of up to 32 single-child inner nodes becomes ONE 16-byte node
//     { CQ_CHAIN_BIT | len, label_hi, label_lo, next }      label = len symbols, top aligned
// so a 50-base key below a 26-base bucket costs one node read instead of 24.  Branch nodes keep
// the 4-children form.  find64_p's semantics carry over: inner chain nodes are never leaves
// (keys are prefix-free and leaves have no children), so "consume len matching symbols or fail"
// is exactly what the symbol-by-symbol walk does.
// The two tables' tries seen as ONE node array without building it: indices 1 .. nnu are ht_u's nodes as decoded, nnu + 1 ..
// nnu + nnd ht_d's with their references moved into the common id space (d nodes after u nodes, d leaves after u leaves).
struct LinkedTries {
    const Node *u, *d;        // decoded node arrays (index 0 = dummy)
    uint32_t nnu, leaf_off;
    uint32_t relink_d(uint32_t code) const
    {
        if (code == 0) return 0;
        if (code & CQ_LEAF_BIT) return CQ_LEAF_BIT | ((code & ~CQ_LEAF_BIT) + leaf_off);
        return code + nnu;
    }
    Node operator[](uint32_t i) const
    {
        if (i <= nnu) return u[i];
        Node n = d[i - nnu];
        for (int c = 0; c < 4; c++) n.child[c] = relink_d(n.child[c]);
        return n;
    }
};

struct Compressor {
    const LinkedTries &in;
    std::vector<Node> &out;
    const uint32_t *r1, *r2;   // refIDs by global leaf id: a chain that ends at a unique leaf carries the refID inline
    uint32_t run(uint32_t code)
    {
        if (code == 0 || (code & CQ_LEAF_BIT)) return code;
        uint32_t head = 0;
        int64_t link = -1;   // chain node whose `next` still has to be filled in (-1: head)
        auto set_link = [&](uint32_t v) { if (link < 0) head = v; else out[(size_t)link].child[3] = v; };
        auto push_chain = [&](uint32_t len, uint64_t label) {
            const uint64_t top = label << (64u - 2u * len);
            out.push_back(Node{{CQ_CHAIN_BIT | len, (uint32_t)(top >> 32), (uint32_t)top, 0}});
            const uint32_t idx = (uint32_t)(out.size() - 1);
            set_link(idx);
            link = idx;
        };
        uint32_t cur = code, len = 0;
        uint64_t label = 0;
        for (;;) {
            const Node n = in[cur];
            int cnt = 0, only = -1;
            for (int c = 0; c < 4; c++) if (n.child[c]) { cnt++; only = c; }
            if (cnt == 0) { set_link(0); return head; }   // cannot happen: childless nodes are leaves
            if (cnt == 1) {
                label = (label << 2) | (uint64_t)only;
                len++;
                const uint32_t nxt = n.child[only];
                const bool leaf = (nxt & CQ_LEAF_BIT) != 0;
                if (len == 32 || leaf) { push_chain(len, label); len = 0; label = 0; }
                if (leaf) {
                    set_link(nxt);
                    // the usual deep key: alone below its bucket, one chain, then its leaf.  A unique leaf's refID rides in the
                    // chain node's spare bits (cq_device.h CQ_CHAIN_RID_*): the walk ends without the leaf_rids read, one
                    // dependent HBM round trip less on the exact path
                    const uint32_t g = nxt & ~CQ_LEAF_BIT;
                    if (r1 && r2[g] == 0 && r1[g] != 0 && r1[g] <= CQ_CHAIN_RID_MAX) out[(size_t)link].child[0] |= r1[g] << CQ_CHAIN_RID_SHIFT;
                    return head;
                }
                cur = nxt;
                continue;
            }
            if (len) push_chain(len, label);
            out.push_back(Node{{0, 0, 0, 0}});
            const uint32_t b = (uint32_t)(out.size() - 1);
            set_link(b);
            for (int c = 0; c < 4; c++)
                if (n.child[c]) { const uint32_t v = run(n.child[c]); out[b].child[c] = v; }
            return head;
        }
    }
};

}  // namespace

namespace {
struct StageTimer {   // CAMMIQ_LOAD_TIMING=1: stage timings on stderr (diagnostic)
    bool on = getenv("CAMMIQ_LOAD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        const auto n = std::chrono::steady_clock::now();
        if (on) fprintf(stderr, "[build_image]   %-20s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
}  // namespace

// Stage 1 of the layout, always on the host: link the two tries, give leaves global ids, path-compress, and produce
// every bucket's final trie code.  vals[i] is entry i's code in ITS table (entries [0, nb_u) are ht_u's buckets in file
// order, [nb_u, nb_u + nb_d) ht_d's); the table itself is laid out from (bucket_key, vals) either by finish_image_host
// below or on the device (cq_layout_gpu.hip) -- byte-identical results.
int prepare_image(const DecodedTable &u, const DecodedTable &d, double keys_per_bucket, uint32_t minimizer_len,
                  FlatImage &img, RawVec<uint32_t> &vals, std::string &err)
{
    StageTimer st;
    img = FlatImage();
    if (u.hash_len != d.hash_len) { err = "hash lengths of the two index files differ"; return CQ_ERR_HASHLEN; }
    img.hash_len = u.hash_len;
    if (minimizer_len == 0) minimizer_len = cq_choose_minimizer_len(u.hash_len, u.bucket_key.size() + d.bucket_key.size());
    if (minimizer_len > CQ_MAX_MINIMIZER || minimizer_len > u.hash_len) { err = "minimizer length outside [1, min(h, 21)]"; return CQ_ERR_ARG; }
    img.minimizer_len = minimizer_len;
    const uint64_t nu = u.leaves.size(), nd = d.leaves.size();
    if (nu + nd >= 0x7FFFFFFFull) { err = "more than 2^31-1 leaves in total"; return CQ_ERR_LIMIT; }
    img.n_leaves[0] = nu;
    img.n_leaves[1] = nd;

    // ---- link the two tries into one node array; give leaves global ids (u first)
    const uint64_t nnu = u.nodes.size() - 1, nnd = d.nodes.size() - 1;  // real nodes (index 0 = dummy)
    if (nnu + nnd + 1 >= 0x7FFFFFFFull) { err = "more than 2^31-1 trie nodes in total"; return CQ_ERR_LIMIT; }
    // (no linked copy of the node arrays: at configs[4]'s size they are 13 GB, and the only reader is the path compression)
    const LinkedTries linked{u.nodes.data(), d.nodes.data(), (uint32_t)nnu, (uint32_t)nu};
    auto relink_d = [&](uint32_t code) { return linked.relink_d(code); };
    // the device gets the path-compressed form (bucket roots are compressed where they are used)
    img.nodes.clear();
    img.nodes.push_back(Node{{0, 0, 0, 0}});

    img.leaf_r1.reserve(nu + nd);
    img.leaf_r2.reserve(nu + nd);
    advise_huge(img.leaf_r1.data(), (nu + nd) * sizeof(uint32_t));
    advise_huge(img.leaf_r2.data(), (nu + nd) * sizeof(uint32_t));
    img.leaf_r1.resize(nu + nd);
    img.leaf_r2.resize(nu + nd);
    uint32_t maxr = 0;
    {
        const unsigned ntl = worker_count(nu + nd);
        std::vector<uint32_t> mx(ntl, 0);
        parallel_for(ntl, [&](unsigned t) {
            uint32_t m = 0;
            for (uint64_t i = (nu + nd) * t / ntl, e = (nu + nd) * (t + 1) / ntl; i < e; i++) {
                const cq_leaf &lf = i < nu ? u.leaves[i] : d.leaves[i - nu];
                img.leaf_r1[i] = lf.refID1; img.leaf_r2[i] = lf.refID2;
                m = std::max(m, std::max(lf.refID1, lf.refID2));
            }
            mx[t] = m;
        });
        for (uint32_t m : mx) maxr = std::max(maxr, m);
    }
    img.max_refid = maxr;
    st.lap("link + leaf refIDs");

    // ---- entries of both tables, sorted by (home bucket, key, file order)
    const uint64_t nb_u = u.bucket_key.size(), nb_d = d.bucket_key.size();
    if (!(keys_per_bucket > 0.0)) keys_per_bucket = 1.5;
    const double want = (double)(nb_u + nb_d) / keys_per_bucket;
    if (!(want < 4294967295.0)) { err = "table would exceed 2^32 buckets"; return CQ_ERR_LIMIT; }
    uint64_t nbk = (uint64_t)want + 1;
    if (nbk < 16) nbk = 16;
    if (nbk + CQ_SPILL_TAIL >= 0xFFFFFFFFull) { err = "table would exceed 2^32 buckets"; return CQ_ERR_LIMIT; }
    img.n_buckets = nbk;
    const uint32_t n_buckets = (uint32_t)nbk;

    const size_t n_ent = nb_u + nb_d;
    vals.resize(n_ent ? n_ent : 1);
    advise_huge(vals.data(), n_ent * sizeof(uint32_t));
    // trie codes first: a bucket root that is a leaf (the usual case) is its own code; the others go
    // through path compression, which appends to one node array and therefore runs serially, in file order
    {
        const unsigned nte = worker_count(nb_u + nb_d);
        std::vector<std::vector<uint64_t>> deep(nte);
        parallel_for(nte, [&](unsigned t) {
            for (uint64_t i = (nb_u + nb_d) * t / nte, e = (nb_u + nb_d) * (t + 1) / nte; i < e; i++) {
                const uint32_t code = i < nb_u ? u.bucket_code[i] : relink_d(d.bucket_code[i - nb_u]);
                vals[i] = code;
                if (code && !(code & CQ_LEAF_BIT)) deep[t].push_back(i);
            }
        });
        // Path compression, every worker on its own slice of the deep roots into its own node array
        // (local index 0 = dummy, like the final array); the arrays are then concatenated in slice order --
        // exactly the order a serial pass over the roots appends in -- and the local indices relocated.
        std::vector<std::vector<Node>> local(nte);
        parallel_for(nte, [&](unsigned t) {
            local[t].push_back(Node{{0, 0, 0, 0}});
            Compressor comp{linked, local[t], img.leaf_r1.data(), img.leaf_r2.data()};
            for (uint64_t i : deep[t]) vals[i] = comp.run(vals[i]);
        });
        std::vector<uint64_t> base(nte + 1, 1);   // global index of worker t's local node 1
        for (unsigned t = 0; t < nte; t++) base[t + 1] = base[t] + (local[t].size() - 1);
        if (base[nte] >= (1ull << 30)) { err = "more than 2^30 trie nodes after path compression"; return CQ_ERR_LIMIT; }
        img.nodes.resize(base[nte]);
        parallel_for(nte, [&](unsigned t) {
            const uint32_t shift = (uint32_t)(base[t] - 1);   // local index k -> global index k + shift
            auto reloc = [&](uint32_t code) { return (code == 0 || (code & CQ_LEAF_BIT)) ? code : code + shift; };
            for (size_t k = 1; k < local[t].size(); k++) {
                Node n = local[t][k];
                if ((n.child[0] >> 30) == 1u) n.child[3] = reloc(n.child[3]);          // chain node: only `next` is a reference
                else for (int c = 0; c < 4; c++) n.child[c] = reloc(n.child[c]);
                img.nodes[base[t] - 1 + k] = n;
            }
            for (uint64_t i : deep[t]) vals[i] = reloc(vals[i]);
            std::vector<Node>().swap(local[t]);
        });
    }
    st.lap("compress tries");
    if (img.nodes.size() >= (1ull << 30)) { err = "more than 2^30 trie nodes after path compression"; return CQ_ERR_LIMIT; }
    (void)n_buckets;
    return CQ_OK;
}

// Stage 2 on the host: home buckets, sort, merge, placement sweep -> img.table (the reference image of the layout;
// what handles without a device, the image cache and the tests of the device layout use).
int finish_image_host(const DecodedTable &u, const DecodedTable &d, const RawVec<uint32_t> &vals, FlatImage &img, std::string &err)
{
    StageTimer st;
    const uint64_t nb_u = u.bucket_key.size(), nb_d = d.bucket_key.size();
    const size_t n_ent = nb_u + nb_d;
    const uint64_t nbk = img.n_buckets;
    const uint32_t n_buckets = (uint32_t)nbk;
    std::unique_ptr<Entry[]> ent(new Entry[n_ent ? n_ent : 1]);   // uninitialised: filled (first touched) in parallel
    advise_huge(ent.get(), n_ent * sizeof(Entry));
    // the home buckets (a minimizer scan per key: the expensive part) on all cores
    {
        const unsigned nt = worker_count(n_ent);
        const uint32_t hl = img.hash_len, ml = img.minimizer_len;
        parallel_for(nt, [&](unsigned t) {
            const size_t lo = n_ent * t / nt, hi = n_ent * (t + 1) / nt;
            for (size_t i = lo; i < hi; i++) {
                const uint64_t key = i < nb_u ? u.bucket_key[i] : d.bucket_key[i - nb_u];
                ent[i] = i < nb_u ? Entry{0, (uint32_t)i, key, vals[i], 0} : Entry{0, (uint32_t)(i - nb_u), key, 0, vals[i]};
                ent[i].home = cq_home_bucket(key, hl, ml, n_buckets);
            }
        });
    }
    st.lap("home buckets");
    std::vector<size_t> part_begin;
    const std::unique_ptr<Entry[]> sorted = sort_entries(ent, n_ent, n_buckets, part_begin);
    Entry *const E = sorted.get();
    st.lap("sort");

    // ---- merge duplicates (same key in both tables, or repeated within one file: the later
    //      bucket wins, as map64[bucket] = root overwrites -- hashtrie.cpp:500).  Equal keys share
    //      their home bucket, hence their part: every part is compacted on its own.
    const unsigned nt = worker_count(n_ent);
    std::vector<size_t> part_end(kParts, 0);
    {
        std::atomic<unsigned> nextp{0};
        parallel_for(nt, [&](unsigned) {
            for (;;) {
                const unsigned p = nextp.fetch_add(1);
                if (p >= kParts) break;
                size_t w = part_begin[p];
                const size_t hi = part_begin[p + 1];
                for (size_t i = part_begin[p]; i < hi;) {
                    Entry m = E[i];
                    size_t j = i + 1;
                    while (j < hi && E[j].key == m.key) j++;
                    // per table, the entry with the highest file position wins
                    uint32_t su = 0, sd = 0; bool hu = false, hd = false;
                    for (size_t k = i; k < j; k++) {
                        if (E[k].val_u && (!hu || E[k].seq >= su)) { m.val_u = E[k].val_u; su = E[k].seq; hu = true; }
                        if (E[k].val_d && (!hd || E[k].seq >= sd)) { m.val_d = E[k].val_d; sd = E[k].seq; hd = true; }
                    }
                    // a unique depth-0 leaf with a free val_d slot carries its refID inline (cq_device.h)
                    if ((m.val_u & CQ_LEAF_BIT) && m.val_d == 0) {
                        const uint32_t g = m.val_u & ~CQ_LEAF_BIT;
                        if (img.leaf_r2[g] == 0 && img.leaf_r1[g] < CQ_INLINE_RID_BIT) m.val_d = CQ_INLINE_RID_BIT | img.leaf_r1[g];
                    }
                    // likewise a depth-0 leaf of ht_d alone: its two refIDs ride in the free val_u word
                    if ((m.val_d & CQ_LEAF_BIT) && m.val_u == 0) {
                        const uint32_t g = m.val_d & ~CQ_LEAF_BIT;
                        if (img.leaf_r1[g] < (1u << 15) && img.leaf_r2[g] < (1u << 15))
                            m.val_u = CQ_INLINE_PAIR_BIT | (img.leaf_r1[g] << 15) | img.leaf_r2[g];
                    }
                    E[w++] = m;
                    i = j;
                }
                part_end[p] = w;
            }
        });
    }
    img.n_keys = 0;
    for (unsigned p = 0; p < kParts; p++) img.n_keys += part_end[p] - part_begin[p];
    st.lap("merge duplicates");

    // ---- placement: a linear sweep over the buckets.  Keys sorted by home bucket are dealt into
    //      buckets in order; what does not fit is carried to the next bucket, whose predecessor
    //      gets the overflow bit.  Part p owns the buckets its keys are homed in, so all parts
    //      sweep at once (each starting with an empty carry and writing every one of its buckets
    //      exactly once: no separate initialisation pass); what a part could not place inside its
    //      own range is then carried, in order, into the next part, whose first buckets are
    //      redone until the carry has drained -- from there on the independent sweep was right.
    img.n_buckets_alloc = nbk + CQ_SPILL_TAIL;
    img.table_words = img.n_buckets_alloc * CQ_BUCKET_WORDS;
    img.table.alloc(img.table_words);   // uninitialised: the sweeps below first-touch it
    auto first_bucket = [&](unsigned p) -> uint64_t {   // smallest home with part_of(home) == p
        return p >= kParts ? img.n_buckets_alloc : ((uint64_t)p * nbk + kParts - 1) / kParts;
    };
    struct SweepOut { std::vector<Entry> left; uint64_t overflowed = 0; uint32_t max_chain = 1; uint64_t stopped_at = 0; };
    // Fill buckets [b0, b1) of part p from its keys, `carry` first.  until_drained: stop after the first
    // bucket that leaves nothing waiting (and report where); fix: subtract the flags the redone buckets had.
    auto sweep = [&](unsigned p, uint64_t b0, uint64_t b1, const std::vector<Entry> &carry, bool until_drained, SweepOut &o) {
        size_t qpos = 0, cl = part_begin[p], nx = part_begin[p];
        const size_t hi = part_end[p];
        o.stopped_at = b1;
        for (uint64_t b = b0; b < b1; b++) {
            while (nx < hi && E[nx].home <= b) nx++;
            uint32_t *bw = &img.table[b * CQ_BUCKET_WORDS];
            if (until_drained && (bw[CQ_BW_KEY_LO] & 1u) && bw[CQ_BW_KEY_HI] != 0xFFFFFFFFu) o.overflowed--;   // flag of the sweep being redone
            for (int k = 0; k < CQ_SLOTS_PER_BUCKET; k++) {
                bw[CQ_BW_KEY_LO + k] = bw[CQ_BW_KEY_HI + k] = 0xFFFFFFFFu;
                bw[CQ_BW_VAL_U + k] = bw[CQ_BW_VAL_D + k] = 0u;
            }
            bw[CQ_BW_KEY_LO] = 0xFFFFFFFEu;   // empty slot 0: overflow flag (bit 0) must read 0
            const size_t avail = (carry.size() - qpos) + (nx - cl);
            const size_t take = std::min<size_t>(avail, CQ_SLOTS_PER_BUCKET);
            for (size_t k = 0; k < take; k++) {
                const Entry &e = qpos < carry.size() ? carry[qpos++] : E[cl++];
                uint32_t lo = (uint32_t)e.key, hi32 = (uint32_t)(e.key >> 32);
                if (k == 0) {   // slot 0 lends bit 0 of key_lo to the overflow flag
                    if (lo & 1u) hi32 |= CQ_SLOT0_BIT0_IN_HI;
                    lo &= ~1u;
                }
                bw[CQ_BW_KEY_LO + k] = lo;
                bw[CQ_BW_KEY_HI + k] = hi32;
                bw[CQ_BW_VAL_U + k] = e.val_u;
                bw[CQ_BW_VAL_D + k] = e.val_d;
                const uint32_t chain = (uint32_t)(b - e.home) + 1;
                if (chain > o.max_chain) o.max_chain = chain;
            }
            if (avail > take) {   // something is still waiting: this bucket is full and spilled
                bw[CQ_BW_KEY_LO] |= 1u;
                o.overflowed++;
            } else if (until_drained) { o.stopped_at = b + 1; return; }
        }
        // leftovers, oldest first
        o.left.assign(carry.begin() + (ptrdiff_t)qpos, carry.end());
        o.left.insert(o.left.end(), E + cl, E + nx);
    };
    std::vector<SweepOut> outs(kParts);
    {
        std::atomic<unsigned> nextp{0};
        const std::vector<Entry> none;
        parallel_for(worker_count(img.n_buckets_alloc), [&](unsigned) {
            for (;;) {
                const unsigned p = nextp.fetch_add(1);
                if (p >= kParts) break;
                sweep(p, first_bucket(p), first_bucket(p + 1), none, false, outs[p]);
            }
        });
    }
    uint64_t overflowed = 0;
    for (unsigned p = 0; p < kParts; p++) {
        if (p + 1 < kParts && !outs[p].left.empty()) {   // redo the head of part p+1 with the carry of part p
            SweepOut redo;
            redo.overflowed = 0;
            const std::vector<Entry> carry = std::move(outs[p].left);
            sweep(p + 1, first_bucket(p + 1), first_bucket(p + 2), carry, true, redo);
            outs[p + 1].overflowed += redo.overflowed;   // may be "negative": flags of the redone buckets are taken back
            outs[p + 1].max_chain = std::max(outs[p + 1].max_chain, redo.max_chain);
            if (redo.stopped_at == first_bucket(p + 2)) outs[p + 1].left = std::move(redo.left);   // ran through: its carry replaces the old one
        }
        overflowed += outs[p].overflowed;
        img.max_chain = std::max(img.max_chain, outs[p].max_chain);
    }
    if (!outs[kParts - 1].left.empty()) {
        // The carry outran the CQ_SPILL_TAIL buckets past the hash range (a very dense table whose last homes
        // are crowded): the tail grows by what is still waiting.  Rare, so the table is simply copied.
        const std::vector<Entry> &left = outs[kParts - 1].left;
        const uint64_t extra = (left.size() + CQ_SLOTS_PER_BUCKET - 1) / CQ_SLOTS_PER_BUCKET;
        const uint64_t old_alloc = img.n_buckets_alloc;
        if (old_alloc + extra >= 0xFFFFFFFFull) { err = "table would exceed 2^32 buckets"; return CQ_ERR_LIMIT; }
        HugeWords grown;
        grown.alloc((old_alloc + extra) * CQ_BUCKET_WORDS);
        memcpy(grown.get(), img.table.get(), old_alloc * CQ_BUCKET_WORDS * sizeof(uint32_t));
        img.table = std::move(grown);
        img.n_buckets_alloc = old_alloc + extra;
        img.table_words = img.n_buckets_alloc * CQ_BUCKET_WORDS;
        size_t q = 0;
        for (uint64_t b = old_alloc; b < img.n_buckets_alloc; b++) {   // the last old bucket already carries its overflow flag
            uint32_t *bw = &img.table[b * CQ_BUCKET_WORDS];
            for (int k = 0; k < CQ_SLOTS_PER_BUCKET; k++) {
                bw[CQ_BW_KEY_LO + k] = bw[CQ_BW_KEY_HI + k] = 0xFFFFFFFFu;
                bw[CQ_BW_VAL_U + k] = bw[CQ_BW_VAL_D + k] = 0u;
            }
            bw[CQ_BW_KEY_LO] = 0xFFFFFFFEu;
            for (int k = 0; k < CQ_SLOTS_PER_BUCKET && q < left.size(); k++, q++) {
                const Entry &e = left[q];
                uint32_t lo = (uint32_t)e.key, hi32 = (uint32_t)(e.key >> 32);
                if (k == 0) {
                    if (lo & 1u) hi32 |= CQ_SLOT0_BIT0_IN_HI;
                    lo &= ~1u;
                }
                bw[CQ_BW_KEY_LO + k] = lo;
                bw[CQ_BW_KEY_HI + k] = hi32;
                bw[CQ_BW_VAL_U + k] = e.val_u;
                bw[CQ_BW_VAL_D + k] = e.val_d;
                img.max_chain = std::max(img.max_chain, (uint32_t)(b - e.home) + 1);
            }
            if (q < left.size()) { bw[CQ_BW_KEY_LO] |= 1u; overflowed++; }
        }
    }
    img.n_overflowed = overflowed;
    st.lap("placement sweep");
    return CQ_OK;
}

int build_image(const DecodedTable &u, const DecodedTable &d, double keys_per_bucket, uint32_t minimizer_len,
                FlatImage &img, std::string &err)
{
    RawVec<uint32_t> vals;
    int rc = prepare_image(u, d, keys_per_bucket, minimizer_len, img, vals, err);
    if (rc != CQ_OK) return rc;
    return finish_image_host(u, d, vals, img, err);
}

void image_lookup(const FlatImage &img, uint64_t key, uint32_t &val_u, uint32_t &val_d, uint32_t *chain_len)
{
    val_u = val_d = 0;
    uint64_t b = cq_home_bucket(key, img.hash_len, img.minimizer_len, (uint32_t)img.n_buckets);
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    uint32_t chain = 0;
    for (;;) {
        const uint32_t *bw = &img.table[b * CQ_BUCKET_WORDS];
        chain++;
        for (int k = 0; k < CQ_SLOTS_PER_BUCKET; k++) {
            uint32_t lo = bw[CQ_BW_KEY_LO + k], hi = bw[CQ_BW_KEY_HI + k];
            if (k == 0) {   // undo the bit-0 loan (an empty slot 0 comes out as hi = 0xBFFFFFFF: no match)
                lo = (lo & ~1u) | ((hi >> 30) & 1u);
                hi &= ~CQ_SLOT0_BIT0_IN_HI;
            }
            if (lo == klo && hi == khi) {
                val_u = bw[CQ_BW_VAL_U + k]; val_d = bw[CQ_BW_VAL_D + k];
                if ((val_d >> 30) == 1u) val_d = 0;   // inline refID of the u leaf, not an ht_d entry
                if ((val_u >> 30) == 1u) val_u = 0;   // inline refID pair of the d leaf, not an ht_u entry
                if (chain_len) *chain_len = chain;
                return;
            }
        }
        const bool ovf = (bw[CQ_BW_KEY_LO] & 1u) != 0;
        if (!ovf || b + 1 >= img.n_buckets_alloc) break;
        b++;
    }
    if (chain_len) *chain_len = chain;
}

}  // namespace cq
