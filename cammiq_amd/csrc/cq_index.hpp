// cq_index.hpp -- internal host-side structures of libcammiq_hip.so.
//
// Data flow (all new code; the reference's counterpart is the pointer trie + robin_hood map
// built by Hash::loadIdx64_p, /root/reference/src/hashtrie.cpp:425-507):
//
//   .bin1/.bin2 + .aux  --cq_decode_table-->  DecodedTable (leaves in file order, one root
//                                             code per bucket, 16-byte array-trie nodes)
//   2 x DecodedTable    --cq_build_image--->  FlatImage   (merged u+d open-addressing table
//                                             of 64-byte buckets, linked trie, leaf refIDs)
//   FlatImage           --upload----------->  HBM
#ifndef CQ_INDEX_HPP_
#define CQ_INDEX_HPP_

#include <cstdint>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/cammiq_hip.h"
#include "cq_device.h"

namespace cq {

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the multi-GB arrays of a decoded
// table are sized once and first touched by the threads that fill them, not zeroed by the thread that sizes them.
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
    DefaultInitAlloc() = default;
    template <class U> DefaultInitAlloc(const DefaultInitAlloc<U> &) {}
    template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
template <class T> using RawVec = std::vector<T, DefaultInitAlloc<T>>;

// A trie reference ("code"): 0 = absent, CQ_LEAF_BIT|leaf id = leaf, otherwise node index (>=1).
struct Node { uint32_t child[4]; };  // A,C,G,T -- one 16-byte load per level on the GPU

struct DecodedTable {
    uint32_t hash_len = 0;
    uint32_t doubly = 0;
    uint64_t n_file_buckets = 0;
    RawVec<cq_leaf> leaves;        // decode order (== map_sp fill order)
    RawVec<uint64_t> bucket_key;   // hv of each bucket, file order
    RawVec<uint32_t> bucket_code;  // root code of each bucket (table-local ids)
    RawVec<Node> nodes;            // nodes[0] is a reserved dummy
};

// The table image on the host: an uninitialised uint32 array, first touched by all cores at once.  From 32 MB on it
// is an anonymous mapping aligned to 2 MiB with MADV_HUGEPAGE -- where the host allows transparent huge pages the
// layout's first touch takes 512 x fewer page faults, the upload pins fewer pages, and giving a multi-GB table back
// is a few thousand pages instead of ~10^6 (0.35 s for configs[1]'s 3.2 GB, during which every other mmap of the
// process waits).  Smaller tables come from the heap (and stay visible to the address sanitizer in the tests).
class HugeWords {
    void *base_ = nullptr;     // mapping (nullptr: heap)
    size_t map_len_ = 0;
    uint32_t *p_ = nullptr;

public:
    HugeWords() = default;
    HugeWords(const HugeWords &) = delete;
    HugeWords &operator=(const HugeWords &) = delete;
    HugeWords(HugeWords &&o) noexcept : base_(o.base_), map_len_(o.map_len_), p_(o.p_) { o.base_ = nullptr; o.map_len_ = 0; o.p_ = nullptr; }
    HugeWords &operator=(HugeWords &&o) noexcept
    {
        if (this != &o) { reset(); base_ = o.base_; map_len_ = o.map_len_; p_ = o.p_; o.base_ = nullptr; o.map_len_ = 0; o.p_ = nullptr; }
        return *this;
    }
    ~HugeWords() { reset(); }
    void alloc(size_t words);   // throws std::bad_alloc
    void reset();
    uint32_t *get() const { return p_; }
    uint32_t &operator[](size_t i) const { return p_[i]; }
    explicit operator bool() const { return p_ != nullptr; }
};

// Hint for a large, not yet touched heap block (a reserve()d vector, a new[] array): back it with transparent huge
// pages where the host allows them.  The decoder alone first-touches 1.4 GB on one thread for configs[1]'s index.
void advise_huge(const void *p, size_t bytes);
// Before freeing a large block: its pages go back in steps that do not keep the rest of the process from page faulting.
void release_pages(const void *p, size_t bytes);

// Device-ready image.  Everything is plain arrays so upload is a handful of memcpys.
struct FlatImage {
    uint32_t hash_len = 0;
    uint32_t minimizer_len = 0;   // m of this image (cq_device.h: 16, or 18 for large tables)
    uint64_t n_leaves[2] = {0, 0};
    uint64_t n_buckets = 0;       // hash range: a key's home bucket is in [0, n_buckets)
    uint64_t n_buckets_alloc = 0; // n_buckets + spill tail
    uint64_t n_keys = 0;
    uint64_t n_overflowed = 0;
    uint32_t max_chain = 1;
    uint32_t max_refid = 0;
    // CQ_BUCKET_WORDS words per bucket (layout in cq_device.h).  Not a std::vector: a multi-GB table
    // is allocated uninitialised and first touched by all cores at once.
    HugeWords table;
    size_t table_words = 0;
    std::vector<Node> nodes;      // linked: d-table indices/leaf ids already offset
    std::vector<uint32_t> leaf_r1, leaf_r2;  // global leaf id -> refIDs (u leaves first)
};

// Decode one index file pair.  Returns CQ_OK or a negative cq_status and fills err.
int decode_table(const std::string &path, DecodedTable &out, std::string &err);
void make_empty_table(uint32_t hash_len, DecodedTable &out);

// Merge + lay out.  load_factor = average keys per 4-slot bucket (default 1.5).
// minimizer_len: 0 = cq_choose_minimizer_len(h, keys).
int build_image(const DecodedTable &u, const DecodedTable &d, double keys_per_bucket, uint32_t minimizer_len,
                FlatImage &img, std::string &err);

// The two stages of build_image.  prepare_image: everything but the table (linked + path-compressed trie, leaf refIDs,
// hash range, minimizer length) and vals[i] = final trie code of entry i in its own table -- entries [0, nb_u) are
// ht_u's buckets in file order, the rest ht_d's.  finish_image_host lays the table out on the host from
// (bucket_key, vals); cq_layout_gpu.hip does the same on the device, byte for byte.
int prepare_image(const DecodedTable &u, const DecodedTable &d, double keys_per_bucket, uint32_t minimizer_len,
                  FlatImage &img, RawVec<uint32_t> &vals, std::string &err);
int finish_image_host(const DecodedTable &u, const DecodedTable &d, const RawVec<uint32_t> &vals, FlatImage &img, std::string &err);

// Optional on-disk cache of the finished image (cq_cache.cpp).
struct SourceStamp { uint64_t size[4]; int64_t mtime_ns[4]; };   // index_u, .aux, index_d, .aux (0 = absent)
bool stamp_sources(const std::string &path_u, const std::string &path_d, SourceStamp &s);
bool save_image(const std::string &file, const SourceStamp &src, double kpb_override, uint32_t m_override,
                const DecodedTable tab[2], const FlatImage &img);
bool load_image(const std::string &file, const SourceStamp &src, double kpb_override, uint32_t m_override,
                uint64_t max_table_bytes, DecodedTable tab[2], FlatImage &img);

// Where a host-fed query's ASCII reads lie (cq_pack.cpp): one buffer + offsets, or -- ptrs != nullptr -- one pointer and one
// length byte per read (FqReader's reads[f] / rlengths[f], query.hpp:35-36).
struct ReadSource {
    const uint8_t *bases = nullptr;
    const uint64_t *offsets = nullptr;
    const uint8_t *const *ptrs = nullptr;
    const uint8_t *lens = nullptr;
};
void pack_tight_slice(const ReadSource &src, uint64_t lo, uint64_t hi, uint32_t h, uint32_t sb, uint8_t *dst, uint8_t *lens_out,
                      uint64_t *skipped, uint32_t *mn, uint32_t *mx);
uint32_t longest_read(const ReadSource &src, uint64_t lo, uint64_t hi);

// Host mirror of the device lookup (used by tests of the layout through the C ABI and by
// build_image's self-check).  Returns the slot values for `key` (0,0 when absent).
void image_lookup(const FlatImage &img, uint64_t key, uint32_t &val_u, uint32_t &val_d,
                  uint32_t *chain_len = nullptr);

}  // namespace cq
#endif
