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
#include <cstring>
#include <thread>

#include "cq_index.hpp"

namespace cq {

namespace {

struct Entry {
    uint32_t home;
    uint32_t seq;      // file order inside (table, file): later duplicate wins (map64[b] = root)
    uint64_t key;
    uint32_t val_u, val_d;
};

inline bool entry_less(const Entry &a, const Entry &b)
{
    if (a.home != b.home) return a.home < b.home;
    if (a.key != b.key) return a.key < b.key;
    return a.seq < b.seq;
}

// Sort by (home, key, seq).  Parallel: partition by the top bits of `home`, sort the parts.
void sort_entries(std::vector<Entry> &e, uint32_t n_buckets)
{
    const size_t n = e.size();
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = std::min(hw ? hw : 1u, 32u);
    if (n < (1u << 18) || nt < 2) { std::sort(e.begin(), e.end(), entry_less); return; }
    const unsigned parts = 256;
    std::vector<size_t> cnt(parts + 1, 0);
    auto part_of = [&](uint32_t home) { return (unsigned)(((uint64_t)home * parts) / n_buckets); };
    for (size_t i = 0; i < n; i++) cnt[part_of(e[i].home) + 1]++;
    for (unsigned p = 0; p < parts; p++) cnt[p + 1] += cnt[p];
    std::vector<Entry> tmp(n);
    {
        std::vector<size_t> cur(cnt.begin(), cnt.end() - 1);
        for (size_t i = 0; i < n; i++) tmp[cur[part_of(e[i].home)]++] = e[i];
    }
    e.swap(tmp);
    std::vector<std::thread> th;
    std::atomic<unsigned> next{0};
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&] {
            for (;;) {
                unsigned p = next.fetch_add(1);
                if (p >= parts) break;
                std::sort(e.begin() + cnt[p], e.begin() + cnt[p + 1], entry_less);
            }
        });
    for (auto &x : th) x.join();
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
struct Compressor {
    const std::vector<Node> &in;
    std::vector<Node> &out;
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
                if (leaf) { set_link(nxt); return head; }
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

int build_image(const DecodedTable &u, const DecodedTable &d, double keys_per_bucket,
                FlatImage &img, std::string &err)
{
    img = FlatImage();
    if (u.hash_len != d.hash_len) { err = "hash lengths of the two index files differ"; return CQ_ERR_HASHLEN; }
    img.hash_len = u.hash_len;
    const uint64_t nu = u.leaves.size(), nd = d.leaves.size();
    if (nu + nd >= 0x7FFFFFFFull) { err = "more than 2^31-1 leaves in total"; return CQ_ERR_LIMIT; }
    img.n_leaves[0] = nu;
    img.n_leaves[1] = nd;

    // ---- link the two tries into one node array; give leaves global ids (u first)
    const uint64_t nnu = u.nodes.size() - 1, nnd = d.nodes.size() - 1;  // real nodes (index 0 = dummy)
    if (nnu + nnd + 1 >= 0x7FFFFFFFull) { err = "more than 2^31-1 trie nodes in total"; return CQ_ERR_LIMIT; }
    std::vector<Node> linked(1 + nnu + nnd);
    linked[0] = Node{{0, 0, 0, 0}};
    for (uint64_t i = 1; i <= nnu; i++) linked[i] = u.nodes[i];  // u codes are already global
    const uint32_t node_off = (uint32_t)nnu, leaf_off = (uint32_t)nu;
    auto relink_d = [&](uint32_t code) -> uint32_t {
        if (code == 0) return 0;
        if (code & CQ_LEAF_BIT) return CQ_LEAF_BIT | ((code & ~CQ_LEAF_BIT) + leaf_off);
        return code + node_off;
    };
    for (uint64_t i = 1; i <= nnd; i++) {
        Node n = d.nodes[i];
        for (int c = 0; c < 4; c++) n.child[c] = relink_d(n.child[c]);
        linked[nnu + i] = n;
    }
    // the device gets the path-compressed form (bucket roots are compressed where they are used)
    img.nodes.clear();
    img.nodes.push_back(Node{{0, 0, 0, 0}});
    Compressor comp{linked, img.nodes};

    img.leaf_r1.resize(nu + nd);
    img.leaf_r2.resize(nu + nd);
    uint32_t maxr = 0;
    for (uint64_t i = 0; i < nu; i++) {
        img.leaf_r1[i] = u.leaves[i].refID1; img.leaf_r2[i] = u.leaves[i].refID2;
        maxr = std::max(maxr, std::max(u.leaves[i].refID1, u.leaves[i].refID2));
    }
    for (uint64_t i = 0; i < nd; i++) {
        img.leaf_r1[nu + i] = d.leaves[i].refID1; img.leaf_r2[nu + i] = d.leaves[i].refID2;
        maxr = std::max(maxr, std::max(d.leaves[i].refID1, d.leaves[i].refID2));
    }
    img.max_refid = maxr;

    // ---- entries of both tables, sorted by (home bucket, key, file order)
    const uint64_t nb_u = u.bucket_key.size(), nb_d = d.bucket_key.size();
    if (keys_per_bucket <= 0.1) keys_per_bucket = 1.5;
    uint64_t nbk = (uint64_t)((double)(nb_u + nb_d) / keys_per_bucket) + 1;
    if (nbk < 16) nbk = 16;
    if (nbk + CQ_SPILL_TAIL >= 0xFFFFFFFFull) { err = "table would exceed 2^32 buckets"; return CQ_ERR_LIMIT; }
    img.n_buckets = nbk;
    const uint32_t n_buckets = (uint32_t)nbk;

    std::vector<Entry> ent(nb_u + nb_d);
    // trie codes first (path compression appends to one node array: sequential) ...
    for (uint64_t i = 0; i < nb_u; i++)
        ent[i] = Entry{0, (uint32_t)i, u.bucket_key[i], comp.run(u.bucket_code[i]), 0};
    for (uint64_t i = 0; i < nb_d; i++)
        ent[nb_u + i] = Entry{0, (uint32_t)i, d.bucket_key[i], 0, comp.run(relink_d(d.bucket_code[i]))};
    // ... then the home buckets (a minimizer scan per key: the expensive part) on all cores
    {
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned nt = ent.size() < (1u << 16) ? 1u : std::max(1u, std::min(hw ? hw : 1u, 32u));
        const uint32_t hl = img.hash_len;
        auto work = [&](unsigned t) {
            const size_t lo = ent.size() * t / nt, hi = ent.size() * (t + 1) / nt;
            for (size_t i = lo; i < hi; i++) ent[i].home = cq_home_bucket(ent[i].key, hl, n_buckets);
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
    }
    if (img.nodes.size() >= (1ull << 30)) { err = "more than 2^30 trie nodes after path compression"; return CQ_ERR_LIMIT; }
    std::vector<Node>().swap(linked);
    sort_entries(ent, n_buckets);

    // ---- merge duplicates (same key in both tables, or repeated within one file: the later
    //      bucket wins, as map64[bucket] = root overwrites -- hashtrie.cpp:500)
    size_t w = 0;
    for (size_t i = 0; i < ent.size();) {
        Entry m = ent[i];
        size_t j = i + 1;
        while (j < ent.size() && ent[j].key == m.key) j++;
        // per table, the entry with the highest file position wins
        uint32_t su = 0, sd = 0; bool hu = false, hd = false;
        for (size_t k = i; k < j; k++) {
            if (ent[k].val_u && (!hu || ent[k].seq >= su)) { m.val_u = ent[k].val_u; su = ent[k].seq; hu = true; }
            if (ent[k].val_d && (!hd || ent[k].seq >= sd)) { m.val_d = ent[k].val_d; sd = ent[k].seq; hd = true; }
        }
        // a unique depth-0 leaf with a free val_d slot carries its refID inline (cq_device.h)
        if ((m.val_u & CQ_LEAF_BIT) && m.val_d == 0) {
            const uint32_t g = m.val_u & ~CQ_LEAF_BIT;
            if (img.leaf_r2[g] == 0 && img.leaf_r1[g] < CQ_INLINE_RID_BIT) m.val_d = CQ_INLINE_RID_BIT | img.leaf_r1[g];
        }
        ent[w++] = m;
        i = j;
    }
    ent.resize(w);
    img.n_keys = w;

    // ---- linear sweep placement
    img.n_buckets_alloc = nbk + CQ_SPILL_TAIL;
    img.table_words = img.n_buckets_alloc * CQ_BUCKET_WORDS;
    img.table.reset(new uint32_t[img.table_words]);   // uninitialised: the threads below first-touch it
    {
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned nt = img.n_buckets_alloc < (1u << 16) ? 1u : std::max(1u, std::min(hw ? hw : 1u, 32u));
        auto init = [&](unsigned t) {
            const uint64_t lo = img.n_buckets_alloc * t / nt, hi = img.n_buckets_alloc * (t + 1) / nt;
            for (uint64_t b = lo; b < hi; b++) {
                uint32_t *bw = &img.table[b * CQ_BUCKET_WORDS];
                for (int k = 0; k < CQ_SLOTS_PER_BUCKET; k++) {
                    bw[CQ_BW_KEY_LO + k] = bw[CQ_BW_KEY_HI + k] = 0xFFFFFFFFu;
                    bw[CQ_BW_VAL_U + k] = bw[CQ_BW_VAL_D + k] = 0u;
                }
                bw[CQ_BW_KEY_LO] = 0xFFFFFFFEu;   // empty slot 0: overflow flag (bit 0) must read 0
            }
        };
        if (nt == 1) init(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) th.emplace_back(init, t);
            for (auto &x : th) x.join();
        }
    }
    size_t next = 0;         // next entry not yet pulled into the carry
    size_t carry_lo = 0;     // entries [carry_lo, next) are waiting for a slot, oldest first
    uint64_t overflowed = 0;
    for (uint64_t b = 0; b < img.n_buckets_alloc; b++) {
        while (next < w && ent[next].home <= b) next++;
        size_t avail = next - carry_lo;
        if (avail == 0) {
            if (next >= w) break;
            continue;
        }
        size_t take = std::min<size_t>(avail, CQ_SLOTS_PER_BUCKET);
        uint32_t *bw = &img.table[b * CQ_BUCKET_WORDS];
        for (size_t k = 0; k < take; k++) {
            const Entry &e = ent[carry_lo + k];
            uint32_t lo = (uint32_t)e.key, hi = (uint32_t)(e.key >> 32);
            if (k == 0) {   // slot 0 lends bit 0 of key_lo to the overflow flag
                if (lo & 1u) hi |= CQ_SLOT0_BIT0_IN_HI;
                lo &= ~1u;
            }
            bw[CQ_BW_KEY_LO + k] = lo;
            bw[CQ_BW_KEY_HI + k] = hi;
            bw[CQ_BW_VAL_U + k] = e.val_u;
            bw[CQ_BW_VAL_D + k] = e.val_d;
            uint32_t chain = (uint32_t)(b - e.home) + 1;
            if (chain > img.max_chain) img.max_chain = chain;
        }
        carry_lo += take;
        if (carry_lo < next) {  // something is still waiting: this bucket is full and spilled
            bw[CQ_BW_KEY_LO] |= 1u;
            overflowed++;
        }
    }
    if (carry_lo < w) { err = "spill tail exhausted while laying out the table"; return CQ_ERR_LIMIT; }
    img.n_overflowed = overflowed;
    return CQ_OK;
}

void image_lookup(const FlatImage &img, uint64_t key, uint32_t &val_u, uint32_t &val_d, uint32_t *chain_len)
{
    val_u = val_d = 0;
    uint64_t b = cq_home_bucket(key, img.hash_len, (uint32_t)img.n_buckets);
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
