// cq_decode.cpp -- reader for CAMMiQ's on-disk index (index_u.bin1 / index_d.bin2 + .aux).
//
// Consumes, unchanged, the format Hash::encodeIdx64[_d] writes and Hash::loadIdx64_p /
// decodeTrie_p read back (/root/reference/src/hashtrie.cpp:425-507, 625-699) through
// BitWriter/BitReader (/root/reference/src/binaryio.cpp):
//
//   X.aux  MSB-first bit stream: [1b doubly_unique][7b = 64][8b hash_len], then per bucket
//          the trie shape in pre-order: present = 1 followed by its 4 children (A,C,G,T),
//          absent = 0.  A node with no children is a leaf.  Tail: >= 72 one-bits; bits past
//          EOF read as 1 (binaryio.cpp:146-149).
//   X      big-endian byte stream: per bucket u64 hv, then one record per leaf in pre-order:
//          unique  u32 refID, u16 ucount                        (hashtrie.cpp:471-474)
//          doubly  u32 refID1, u32 refID2, u16 uc1, u16 uc2     (hashtrie.cpp:443-450)
//          terminator u64 0xFFFF...FF (+ u16 0xFFFF)             (binaryio.cpp:120-123)
//
// Unlike the reference (recursive, two heap allocations per node, ~100 B/node) this is a
// single forward pass with an explicit stack that emits flat arrays: leaves in decode
// order, one root code per bucket and 16-byte array-trie nodes for the (rare) deep keys.
#include <cstdio>
#include <cstring>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include "cq_index.hpp"

namespace cq {

namespace {

struct Mapped {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const std::string &path)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = nullptr; return true; }
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        madvise(m, n, MADV_SEQUENTIAL);
        p = (const uint8_t *)m;
        return true;
    }
    ~Mapped()
    {
        if (p) munmap((void *)p, n);
        if (fd >= 0) ::close(fd);
    }
};

// MSB-first bit cursor over the .aux stream.  Bits beyond the file read as 1, as the
// reference's reader does, but `past_eof` lets the decoder reject a truncated stream
// instead of recursing forever.
struct Bits {
    const uint8_t *p;
    size_t nbits;
    size_t pos = 0;
    bool past_eof = false;
    inline uint32_t bit()
    {
        if (pos >= nbits) { past_eof = true; pos++; return 1u; }
        uint32_t v = (p[pos >> 3] >> (7 - (pos & 7))) & 1u;
        pos++;
        return v;
    }
    inline uint32_t bits(int n)
    {
        uint32_t v = 0;
        for (int i = 0; i < n; i++) v = (v << 1) | bit();
        return v;
    }
    // Next 5 bits without consuming (only when fully inside the file), else 0xFFFFFFFF.
    inline uint32_t peek5() const
    {
        if (pos + 5 > nbits) return 0xFFFFFFFFu;
        size_t byte = pos >> 3;
        uint32_t w = (uint32_t)p[byte] << 8;
        if (byte + 1 < ((nbits + 7) >> 3)) w |= p[byte + 1];
        return (w >> (11 - (pos & 7))) & 31u;
    }
};

struct Ints {
    const uint8_t *p;
    size_t n;
    size_t pos = 0;
    inline bool has(size_t k) const { return pos + k <= n; }
    inline uint16_t u16() { uint16_t v = (uint16_t)((p[pos] << 8) | p[pos + 1]); pos += 2; return v; }
    inline uint32_t u32()
    {
        uint32_t v = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3];
        pos += 4;
        return v;
    }
    inline uint64_t u64() { uint64_t hi = u32(); uint64_t lo = u32(); return (hi << 32) | lo; }
};

struct Frame {
    uint32_t node;   // index in out.nodes of the tentative node
    uint8_t next;    // next child to read (0..4)
    uint8_t kids;    // number of present children so far
};

}  // namespace

void make_empty_table(uint32_t hash_len, DecodedTable &out)
{
    out = DecodedTable();
    out.hash_len = hash_len;
    out.doubly = 1;
    out.nodes.push_back(Node{{0, 0, 0, 0}});
}

int decode_table(const std::string &path, DecodedTable &out, std::string &err)
{
    out = DecodedTable();
    Mapped fi, fa;
    if (!fi.open(path)) { err = "Cannot open file: " + path + "."; return CQ_ERR_IO; }
    if (!fa.open(path + ".aux")) { err = "Cannot open file: " + path + ".aux."; return CQ_ERR_IO; }
    Bits aux{fa.p, fa.n * 8};
    Ints ints{fi.p, fi.n};

    out.doubly = aux.bit();
    uint32_t option = aux.bits(7);
    if (aux.past_eof || option != 64) { err = path + ".aux: header option != 64"; return CQ_ERR_FORMAT; }
    out.hash_len = aux.bits(8);
    if (out.hash_len < 1 || out.hash_len > 31) { err = path + ".aux: hash length outside [1,31]"; return CQ_ERR_FORMAT; }
    const uint32_t h = out.hash_len;
    const size_t rec = out.doubly ? 12 : 6;
    const uint64_t hv_limit = (h == 32) ? ~0ull : (1ull << (2 * h));

    // Rough reservation: a depth-0 bucket costs 8 + rec bytes of the byte stream.
    size_t guess = fi.n / (8 + rec) + 16;
    out.leaves.reserve(guess);
    out.bucket_key.reserve(guess);
    out.bucket_code.reserve(guess);
    advise_huge(out.leaves.data(), guess * sizeof(cq_leaf));        // one thread first-touches all of these
    advise_huge(out.bucket_key.data(), guess * sizeof(uint64_t));
    advise_huge(out.bucket_code.data(), guess * sizeof(uint32_t));
    out.nodes.push_back(Node{{0, 0, 0, 0}});  // index 0 reserved: code 0 means "absent"

    std::vector<Frame> stack;
    stack.reserve(256);

    auto read_leaf = [&](uint32_t trie_depth) -> int {
        if (!ints.has(rec)) { err = path + ": byte stream truncated inside a leaf record"; return CQ_ERR_FORMAT; }
        if (h + trie_depth > 255) { err = path + ": key longer than 255"; return CQ_ERR_LIMIT; }
        cq_leaf lf;
        memset(&lf, 0, sizeof lf);
        lf.depth = (uint8_t)(h + trie_depth);
        if (out.doubly) {
            lf.refID1 = ints.u32();
            lf.refID2 = ints.u32();
            // "Doubly-unique substring must have two RIDs" (hashtrie.cpp:445-446)
            if (lf.refID1 == 0 || lf.refID2 == 0) { err = path + ": doubly-unique leaf with a zero refID"; return CQ_ERR_FORMAT; }
            lf.ucount1 = ints.u16();
            lf.ucount2 = ints.u16();
        } else {
            lf.refID1 = ints.u32();
            lf.ucount1 = ints.u16();
            // refID2 == 0 is how the classifier tells unique from doubly-unique (query.cpp:532);
            // a unique leaf with refID 0 would index genomes[0] == NULL in the reference.
            if (lf.refID1 == 0) { err = path + ": unique leaf with refID 0"; return CQ_ERR_FORMAT; }
        }
        if (out.leaves.size() >= 0x7FFFFFFFull) { err = path + ": more than 2^31-1 leaves"; return CQ_ERR_LIMIT; }
        out.leaves.push_back(lf);
        return CQ_OK;
    };

    for (;;) {
        if (!ints.has(8)) { err = path + ": byte stream ends without the END64 terminator"; return CQ_ERR_FORMAT; }
        uint64_t hv = ints.u64();
        if (hv == CQ_EMPTY_KEY) break;  // END64
        if (hv >= hv_limit) { err = path + ": bucket value does not fit 2*hash_len bits"; return CQ_ERR_FORMAT; }
        out.n_file_buckets++;

        // Fast path: "1 0000" = the bucket root is itself a leaf (key length == h).
        if (aux.peek5() == 0x10u) {
            aux.pos += 5;
            int rc = read_leaf(0);
            if (rc != CQ_OK) return rc;
            out.bucket_key.push_back(hv);
            out.bucket_code.push_back(CQ_LEAF_BIT | (uint32_t)(out.leaves.size() - 1));
            continue;
        }

        if (aux.bit() != 1u || aux.past_eof) { err = path + ".aux: bucket without a root node"; return CQ_ERR_FORMAT; }
        // General path: explicit-stack pre-order walk.  A tentative node is appended when a
        // '1' is read; if it turns out to have no children it is the most recently appended
        // node (nothing can follow a childless node), so it is popped and replaced by a leaf.
        stack.clear();
        out.nodes.push_back(Node{{0, 0, 0, 0}});
        stack.push_back(Frame{(uint32_t)(out.nodes.size() - 1), 0, 0});
        uint32_t root_code = 0;
        while (!stack.empty()) {
            Frame &f = stack.back();
            if (f.next < 4) {
                uint32_t c = f.next++;
                uint32_t b = aux.bit();
                if (aux.past_eof) { err = path + ".aux: bit stream truncated inside a trie"; return CQ_ERR_FORMAT; }
                if (!b) continue;
                f.kids++;
                if (stack.size() + h > 256) { err = path + ": key longer than 255"; return CQ_ERR_LIMIT; }
                if (out.nodes.size() >= 0x7FFFFFFFull) { err = path + ": more than 2^31-1 trie nodes"; return CQ_ERR_LIMIT; }
                uint32_t parent = f.node;
                out.nodes.push_back(Node{{0, 0, 0, 0}});
                uint32_t me = (uint32_t)(out.nodes.size() - 1);
                out.nodes[parent].child[c] = me;
                stack.push_back(Frame{me, 0, 0});  // invalidates f
                continue;
            }
            // all four children consumed
            uint32_t me = f.node;
            uint32_t kids = f.kids;
            uint32_t trie_depth = (uint32_t)stack.size() - 1;
            stack.pop_back();
            uint32_t code = me;
            if (kids == 0) {
                out.nodes.pop_back();  // `me` is the last node by construction
                int rc = read_leaf(trie_depth);
                if (rc != CQ_OK) return rc;
                code = CQ_LEAF_BIT | (uint32_t)(out.leaves.size() - 1);
            }
            if (stack.empty()) root_code = code;
            else {
                Frame &pf = stack.back();
                out.nodes[pf.node].child[pf.next - 1] = code;
            }
        }
        out.bucket_key.push_back(hv);
        out.bucket_code.push_back(root_code);
    }
    return CQ_OK;
}

}  // namespace cq
