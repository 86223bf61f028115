// cq_cache.cpp -- optional on-disk cache of the flat HBM image ("<index_u>.cqimg").
//
// The reference decodes its index into a pointer trie on every start ("a few minutes",
// /root/reference/README.md:187; Hash::loadIdx64_p, hashtrie.cpp:486-507).  Decoding and laying
// out are fast here, but they are still repeated work: with CAMMIQ_IMAGE_CACHE=1 the finished
// image (leaves in decode order, merged table, compressed trie) is written next to index_u once
// and mapped back on later loads.  The original .bin1/.bin2 files stay authoritative: the cache
// records their sizes and mtimes and the layout revision, and is ignored -- then rewritten --
// whenever anything differs or the file is damaged.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <algorithm>
#include <thread>
#include <vector>

#include "cq_index.hpp"

namespace cq {

namespace {

constexpr uint32_t kLayoutRev = 13;   // bump whenever cq_device.h's table / trie / minimizer layout changes

struct Header {
    char magic[8];              // "CQIMG\0\0\0"
    uint32_t layout_rev, slots_per_bucket, bucket_words, minimizer_len;   // minimizer_len: m of THIS image
    SourceStamp src;
    uint32_t hash_len, doubly[2], max_chain, max_refid, m_override;   // m_override: CAMMIQ_MINIMIZER_LEN the image was built with, 0 = automatic
    uint64_t n_file_buckets[2], n_leaves[2];
    uint64_t n_buckets, n_buckets_alloc, n_keys, n_overflowed, n_nodes, table_words;
    double kpb_override;        // CAMMIQ_KEYS_PER_BUCKET the image was built with, 0 = automatic
    uint64_t payload_bytes, checksum;   // checksum over the header fields above + a sample of the payload
};

uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

// Cheap integrity check: the header plus 4 KiB from the start, middle and end of every section.
uint64_t sample_sum(const Header &h, const void *const sec[4], const size_t len[4])
{
    uint64_t s = fnv(&h, offsetof(Header, checksum));
    for (int i = 0; i < 4; i++) {
        const uint8_t *p = (const uint8_t *)sec[i];
        const size_t n = len[i], k = n < 4096 ? n : 4096;
        if (!n) continue;
        s = fnv(p, k, s);
        s = fnv(p + (n - k) / 2, k, s);
        s = fnv(p + n - k, k, s);
    }
    return s;
}

bool write_all(int fd, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    while (n) {
        ssize_t w = ::write(fd, b, n > (1u << 30) ? (1u << 30) : n);
        if (w <= 0) return false;
        b += w; n -= (size_t)w;
    }
    return true;
}

bool pread_all(int fd, void *p, size_t n, uint64_t off)
{
    uint8_t *b = (uint8_t *)p;
    while (n) {
        ssize_t r = ::pread(fd, b, n > (1u << 30) ? (1u << 30) : n, (off_t)off);
        if (r <= 0) return false;
        b += r; n -= (size_t)r; off += (uint64_t)r;
    }
    return true;
}

// A multi-GB section comes out of the page cache at memcpy speed: use all cores (first touch of the
// destination included).
bool read_section(int fd, void *p, size_t n, uint64_t off)
{
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned nt = n < (64u << 20) ? 1u : std::max(1u, std::min(hw ? hw : 1u, 16u));
    if (nt == 1) return pread_all(fd, p, n, off);
    std::vector<std::thread> th;
    std::vector<char> ok(nt, 0);
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            const size_t lo = (n / nt) * t, hi = t + 1 == nt ? n : (n / nt) * (t + 1);
            ok[t] = pread_all(fd, (uint8_t *)p + lo, hi - lo, off + lo) ? 1 : 0;
        });
    for (auto &x : th) x.join();
    for (unsigned t = 0; t < nt; t++) if (!ok[t]) return false;
    return true;
}

}  // namespace

bool stamp_sources(const std::string &path_u, const std::string &path_d, SourceStamp &s)
{
    memset(&s, 0, sizeof s);
    const std::string f[4] = {path_u, path_u + ".aux", path_d, path_d.empty() ? std::string() : path_d + ".aux"};
    for (int i = 0; i < 4; i++) {
        if (f[i].empty()) continue;
        struct stat st;
        if (stat(f[i].c_str(), &st) != 0) return false;
        s.size[i] = (uint64_t)st.st_size;
        s.mtime_ns[i] = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
    }
    return true;
}

bool save_image(const std::string &file, const SourceStamp &src, double kpb_override, uint32_t m_override,
                const DecodedTable tab[2], const FlatImage &img)
{
    Header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "CQIMG", 5);
    h.layout_rev = kLayoutRev; h.slots_per_bucket = CQ_SLOTS_PER_BUCKET; h.bucket_words = CQ_BUCKET_WORDS;
    h.minimizer_len = img.minimizer_len; h.m_override = m_override;
    h.src = src;
    h.kpb_override = kpb_override;
    h.hash_len = img.hash_len; h.max_chain = img.max_chain; h.max_refid = img.max_refid;
    for (int t = 0; t < 2; t++) {
        h.doubly[t] = tab[t].doubly; h.n_file_buckets[t] = tab[t].n_file_buckets; h.n_leaves[t] = tab[t].leaves.size();
    }
    h.n_buckets = img.n_buckets; h.n_buckets_alloc = img.n_buckets_alloc; h.n_keys = img.n_keys;
    h.n_overflowed = img.n_overflowed; h.n_nodes = img.nodes.size(); h.table_words = img.table_words;
    const void *sec[4] = {tab[0].leaves.data(), tab[1].leaves.data(), img.table.get(), img.nodes.data()};
    const size_t len[4] = {tab[0].leaves.size() * sizeof(cq_leaf), tab[1].leaves.size() * sizeof(cq_leaf),
                           img.table_words * sizeof(uint32_t), img.nodes.size() * sizeof(Node)};
    h.payload_bytes = len[0] + len[1] + len[2] + len[3];
    h.checksum = sample_sum(h, sec, len);
    const std::string tmp = file + ".tmp" + std::to_string((long)getpid());
    int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;
    bool ok = write_all(fd, &h, sizeof h);
    for (int i = 0; i < 4 && ok; i++) ok = write_all(fd, sec[i], len[i]);
    ok = (::close(fd) == 0) && ok;
    if (ok) ok = ::rename(tmp.c_str(), file.c_str()) == 0;   // atomic: readers see the old file or the new one
    if (!ok) ::unlink(tmp.c_str());
    return ok;
}

bool load_image(const std::string &file, const SourceStamp &src, double kpb_override, uint32_t m_override,
                uint64_t max_table_bytes, DecodedTable tab[2], FlatImage &img)
{
    int fd = ::open(file.c_str(), O_RDONLY);
    if (fd < 0) return false;
    Header h;
    bool ok = pread_all(fd, &h, sizeof h, 0) && memcmp(h.magic, "CQIMG\0\0\0", 8) == 0 && h.layout_rev == kLayoutRev &&
              h.slots_per_bucket == CQ_SLOTS_PER_BUCKET && h.bucket_words == CQ_BUCKET_WORDS &&
              h.minimizer_len >= 1 && h.minimizer_len <= CQ_MAX_MINIMIZER && h.minimizer_len <= h.hash_len && h.m_override == m_override &&
              memcmp(&h.src, &src, sizeof src) == 0 && h.kpb_override == kpb_override &&
              h.table_words == h.n_buckets_alloc * CQ_BUCKET_WORDS && h.n_nodes >= 1 &&
              h.table_words * sizeof(uint32_t) <= max_table_bytes && h.hash_len >= 1 && h.hash_len <= 31;
    struct stat st;
    const size_t len[4] = {(size_t)h.n_leaves[0] * sizeof(cq_leaf), (size_t)h.n_leaves[1] * sizeof(cq_leaf),
                           (size_t)h.table_words * sizeof(uint32_t), (size_t)h.n_nodes * sizeof(Node)};
    ok = ok && fstat(fd, &st) == 0 && h.payload_bytes == len[0] + len[1] + len[2] + len[3] &&
         (uint64_t)st.st_size == sizeof h + h.payload_bytes;
    if (ok) {
        try {
            for (int t = 0; t < 2; t++) {
                tab[t] = DecodedTable();
                tab[t].hash_len = h.hash_len; tab[t].doubly = h.doubly[t]; tab[t].n_file_buckets = h.n_file_buckets[t];
                tab[t].leaves.resize(h.n_leaves[t]);
            }
            img = FlatImage();
            img.table.alloc(h.table_words);
            img.table_words = h.table_words;
            img.nodes.resize(h.n_nodes);
        } catch (const std::bad_alloc &) { ok = false; }
    }
    if (ok) {
        void *sec[4] = {tab[0].leaves.data(), tab[1].leaves.data(), img.table.get(), img.nodes.data()};
        uint64_t off = sizeof h;
        for (int i = 0; i < 4 && ok; i++) { ok = read_section(fd, sec[i], len[i], off); off += len[i]; }
        if (ok) ok = sample_sum(h, (const void *const *)sec, len) == h.checksum;
    }
    ::close(fd);
    if (!ok) return false;
    img.hash_len = h.hash_len; img.minimizer_len = h.minimizer_len; img.max_chain = h.max_chain; img.max_refid = h.max_refid;
    img.n_leaves[0] = h.n_leaves[0]; img.n_leaves[1] = h.n_leaves[1];
    img.n_buckets = h.n_buckets; img.n_buckets_alloc = h.n_buckets_alloc; img.n_keys = h.n_keys;
    img.n_overflowed = h.n_overflowed;
    // leaf refIDs by global leaf id (u first) come straight from the leaves
    const uint64_t nu = h.n_leaves[0], nd = h.n_leaves[1];
    img.leaf_r1.resize(nu + nd); img.leaf_r2.resize(nu + nd);
    for (uint64_t i = 0; i < nu; i++) { img.leaf_r1[i] = tab[0].leaves[i].refID1; img.leaf_r2[i] = tab[0].leaves[i].refID2; }
    for (uint64_t i = 0; i < nd; i++) { img.leaf_r1[nu + i] = tab[1].leaves[i].refID1; img.leaf_r2[nu + i] = tab[1].leaves[i].refID2; }
    return true;
}

}  // namespace cq
