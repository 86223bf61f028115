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
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
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

namespace {

// ---- parallel decode --------------------------------------------------------------------------------------------
// The two streams of a table can only be read front to back: where bucket i starts in the bit stream depends on the
// shapes of all tries before it, and where it starts in the byte stream on the number of leaves before it.  A
// first pass over the bit stream alone finds that out -- it walks the shapes without building anything (a '1'
// followed by '0000' is a leaf, any other '1' opens four child slots) and notes, every `step` buckets, the bit
// position and the bucket / leaf / node counts so far (prescan: the serial form, kept for CAMMIQ_DECODE_STEP and for
// byte streams the parallel form's counts do not explain; parallel_scan below: the same notes from a prefix sum and a
// prefix minimum over segments of the stream, on all cores).  The chunks between two notes are then decoded by all cores
// at once, each writing straight into its slice of the final arrays (whose exact sizes the first pass also gave):
// ids are global from the start, nothing is relocated or copied.  The scan accepts exactly the well-formed
// streams; anything else (truncation, a bucket without a root, a key longer than 255, a bad leaf record) sends the
// file through the serial decoder below, which reports the first error the way it always did.
struct Mark { size_t aux_pos; uint64_t buckets, leaves, nodes; };

bool prescan(const uint8_t *ap, size_t nbits, size_t aux_start, const uint8_t *ip, size_t in, size_t rec, uint32_t h,
             uint64_t step, std::vector<Mark> &marks)
{
    Bits aux{ap, nbits};
    aux.pos = aux_start;
    uint64_t nb = 0, nl = 0, nn = 0;
    marks.clear();
    for (;;) {
        if (nb % step == 0) marks.push_back(Mark{aux.pos, nb, nl, nn});
        // "1 0000": the root is a leaf (93 % of the buckets).  It cannot be the terminator -- that is a run of one-bits --
        // so the byte stream is not looked at: the scan of a 15 GB table touches its 0.8 GB of shape bits only.  (A byte
        // stream that disagrees -- END64 or another impossible key inside a chunk, too few bytes -- is rejected by
        // decode_chunk's own checks and by the bounds test at the end, and the file goes to the serial decoder.)
        if (aux.peek5() == 0x10u) { aux.pos += 5; nb++; nl++; continue; }
        const uint64_t ipos = 8 * nb + rec * nl;
        if (ipos + 8 > in) return false;
        uint64_t hv = 0;
        for (int k = 0; k < 8; k++) hv = (hv << 8) | ip[ipos + k];
        if (hv == CQ_EMPTY_KEY) {
            if (marks.empty() || marks.back().buckets != nb) marks.push_back(Mark{aux.pos, nb, nl, nn});
            return true;   // every byte position before this one is smaller: the chunks stay inside the byte stream
        }
        if (aux.pos + 5 > nbits) return false;
        if (aux.bit() != 1u) return false;
        // depth bookkeeping: slots[d] = child slots of the open node at depth d still to be read
        uint8_t slots[260];
        uint32_t depth = 0;
        slots[0] = 4;
        nn++;   // the root is an inner node (it was not "1 0000")
        for (;;) {
            if (slots[depth] == 0) { if (depth == 0) break; depth--; continue; }
            slots[depth]--;
            if (aux.pos >= nbits) return false;
            if (!aux.bit()) continue;
            if (h + depth + 1 > 255) return false;
            const size_t q = aux.pos;
            if (q + 4 > nbits) return false;
            // next four bits all zero: a leaf
            uint32_t w = ((uint32_t)ap[q >> 3] << 8) | ((q >> 3) + 1 < ((nbits + 7) >> 3) ? ap[(q >> 3) + 1] : 0xFFu);
            if (((w >> (12 - (q & 7))) & 15u) == 0) { aux.pos += 4; nl++; continue; }
            nn++;
            slots[++depth] = 4;
        }
        nb++;
    }
}

// ---- the shape scan on all cores ----------------------------------------------------------------------------------
// Every bit of the shape stream is one "child present?" answer of a pre-order walk; reading a bit b changes the number
// of answers still owed by 4b - 1 (a present node owes four more, every answer settles one), and a bucket's trie is
// complete when the count, started at 1, reaches 0.  So with S(p) = sum of (4b - 1) over the first p bits, bucket k ends
// at the first p with S(p) = -k: the bucket boundaries are exactly the positions where S reaches a NEW MINIMUM -- a
// prefix sum and a prefix minimum, both of which split over segments of the stream.  Leaves and inner nodes need no
// context either: a leaf is a 1 followed by 0000 wherever it stands (every 1 in the stream is a node, the four bits
// after a node are its children), an inner node any other 1.  This holds for ANY bit string, well formed or not: a walk
// that starts at a boundary (decode_chunk) ends its buckets where S says and meets the leaves and nodes counted here,
// so the chunks stay inside their slices of the output whatever the file holds; what the walk itself rejects (a bucket
// whose first bit is 0, a key past 255, a bad record) and a byte stream whose END64 is not where the counts put it send
// the file to the serial decoder, as before.
struct ScanTables {
    int8_t sum[256], minp[256];   // of one byte, MSB first: S after it, lowest S inside it (relative to S before it)
    uint8_t ones[256];
    uint8_t leaves[4096];         // leaves ("10000") that START in a byte, by the byte and the four bits after it
    ScanTables()
    {
        for (int b = 0; b < 256; b++) {
            int r = 0, mn = 127, o = 0;
            for (int i = 7; i >= 0; i--) { const int bit = (b >> i) & 1; r += 4 * bit - 1; mn = std::min(mn, r); o += bit; }
            sum[b] = (int8_t)r; minp[b] = (int8_t)mn; ones[b] = (uint8_t)o;
        }
        for (int w = 0; w < 4096; w++) {
            int n = 0;
            for (int j = 0; j < 8; j++) n += ((w >> (7 - j)) & 31) == 16;   // bits j .. j+4 of the 12-bit window
            leaves[w] = (uint8_t)n;
        }
    }
};
const ScanTables kScan;

struct SegStat { int64_t sum = 0, minp = 0; uint64_t ones = 0, leaves = 0; };

// Byte i of the stream and the four bits after it (bits past the end read as 1, as everywhere).
inline uint32_t window12(const uint8_t *ap, size_t nbytes, size_t i)
{
    return ((uint32_t)ap[i] << 4) | (i + 1 < nbytes ? (uint32_t)(ap[i + 1] >> 4) : 15u);
}

// The boundaries inside bytes [b0, b1), S = s before byte b0, lowest S so far = lo, ones / leaves before b0 given:
// the FIRST one (last == false) or the LAST one.  Returns false when there is none.
bool find_boundary(const uint8_t *ap, size_t nbytes, size_t b0, size_t b1, int64_t s, int64_t lo, uint64_t ones, uint64_t leaves,
                   bool last, Mark &out)
{
    bool found = false;
    for (size_t i = b0; i < b1; i++) {
        const uint32_t b = ap[i];
        if (s + kScan.minp[b] < lo) {   // a new minimum inside this byte: bit by bit
            const uint32_t w = window12(ap, nbytes, i);
            uint64_t o = ones, l = leaves;
            int64_t r = s;
            for (int j = 0; j < 8; j++) {
                const uint32_t bit = (b >> (7 - j)) & 1u;
                o += bit;
                l += ((w >> (7 - j)) & 31u) == 16u;
                r += 4 * (int64_t)bit - 1;
                if (r < lo) {
                    lo = r;
                    out = Mark{8 * i + (size_t)j + 1, (uint64_t)(-r), l, o - l};
                    found = true;
                    if (!last) return true;
                }
            }
        }
        s += kScan.sum[b];
        ones += kScan.ones[b];
        leaves += kScan.leaves[window12(ap, nbytes, i)];
    }
    return found;
}

// marks: the start of the stream, one boundary per segment that has one, the end of the last bucket.
bool parallel_scan(const uint8_t *ap, size_t nbytes, size_t start_byte, unsigned nt, size_t seg_bytes, std::vector<Mark> &marks)
{
    marks.clear();
    marks.push_back(Mark{8 * start_byte, 0, 0, 0});
    if (nbytes <= start_byte) return true;
    const size_t nseg = (nbytes - start_byte + seg_bytes - 1) / seg_bytes;
    std::vector<SegStat> st(nseg);
    auto seg_lo = [&](size_t k) { return start_byte + k * seg_bytes; };
    auto seg_hi = [&](size_t k) { return std::min(nbytes, start_byte + (k + 1) * seg_bytes); };
    auto on_all = [&](size_t n, auto &&fn) {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        const unsigned workers = (unsigned)std::min<size_t>(nt, n);
        for (unsigned t = 1; t < workers; t++) th.emplace_back([&] { for (size_t k; (k = next.fetch_add(1)) < n;) fn(k); });
        for (size_t k; (k = next.fetch_add(1)) < n;) fn(k);
        for (auto &x : th) x.join();
    };
    on_all(nseg, [&](size_t k) {
        SegStat a;
        int64_t s = 0, mn = INT64_MAX;
        for (size_t i = seg_lo(k), e = seg_hi(k); i < e; i++) {
            const uint32_t b = ap[i];
            mn = std::min(mn, s + kScan.minp[b]);
            s += kScan.sum[b];
            a.ones += kScan.ones[b];
            a.leaves += kScan.leaves[window12(ap, nbytes, i)];
        }
        a.sum = s; a.minp = mn;
        st[k] = a;
    });
    // prefix over the segments: S, lowest S, ones, leaves before each
    std::vector<int64_t> s0(nseg), lo0(nseg);
    std::vector<uint64_t> ones0(nseg), leaves0(nseg);
    std::vector<size_t> with;   // segments that hold a boundary
    {
        int64_t s = 0, lo = 0;
        uint64_t o = 0, l = 0;
        for (size_t k = 0; k < nseg; k++) {
            s0[k] = s; lo0[k] = lo; ones0[k] = o; leaves0[k] = l;
            if (s + st[k].minp < lo) { with.push_back(k); lo = s + st[k].minp; }
            s += st[k].sum; o += st[k].ones; l += st[k].leaves;
        }
    }
    if (with.empty()) return true;   // no bucket at all
    std::vector<Mark> first(with.size());
    on_all(with.size(), [&](size_t j) {
        const size_t k = with[j];
        (void)find_boundary(ap, nbytes, seg_lo(k), seg_hi(k), s0[k], lo0[k], ones0[k], leaves0[k], false, first[j]);
    });
    for (const Mark &m : first) marks.push_back(m);
    Mark end{};
    const size_t k = with.back();
    if (!find_boundary(ap, nbytes, seg_lo(k), seg_hi(k), s0[k], lo0[k], ones0[k], leaves0[k], true, end)) return false;
    if (end.aux_pos != marks.back().aux_pos) marks.push_back(end);
    return true;
}

}  // namespace

static int decode_serial(const std::string &path, DecodedTable &out, std::string &err);

// Chunk [m0, m1) of a well-formed table, into the final arrays at the offsets the scan found.  Same walk as
// decode_serial with cursors instead of push_back.  Returns false on a leaf record the serial decoder rejects.
static bool decode_chunk(const uint8_t *ap, size_t nbits, const uint8_t *ip, size_t in, size_t rec, uint32_t h, bool doubly,
                         const Mark &m0, const Mark &m1, DecodedTable &out)
{
    Bits aux{ap, nbits};
    aux.pos = m0.aux_pos;
    Ints ints{ip, in};
    ints.pos = 8 * m0.buckets + rec * m0.leaves;
    uint64_t nl = m0.leaves, nn = 1 + m0.nodes;   // nodes[0] is the reserved dummy
    cq_leaf *leaves = out.leaves.data();
    Node *nodes = out.nodes.data();
    auto read_leaf = [&](uint32_t trie_depth) -> bool {
        cq_leaf lf;
        memset(&lf, 0, sizeof lf);
        lf.depth = (uint8_t)(h + trie_depth);
        if (doubly) {
            lf.refID1 = ints.u32(); lf.refID2 = ints.u32();
            if (lf.refID1 == 0 || lf.refID2 == 0) return false;
            lf.ucount1 = ints.u16(); lf.ucount2 = ints.u16();
        } else {
            lf.refID1 = ints.u32(); lf.ucount1 = ints.u16();
            if (lf.refID1 == 0) return false;
        }
        leaves[nl++] = lf;
        return true;
    };
    const uint64_t hv_limit = 1ull << (2 * h);
    std::vector<Frame> stack;
    stack.reserve(256);
    for (uint64_t b = m0.buckets; b < m1.buckets; b++) {
        const uint64_t hv = ints.u64();
        if (hv >= hv_limit) return false;
        if (aux.peek5() == 0x10u) {
            aux.pos += 5;
            if (!read_leaf(0)) return false;
            out.bucket_key[b] = hv;
            out.bucket_code[b] = CQ_LEAF_BIT | (uint32_t)(nl - 1);
            continue;
        }
        if (aux.bit() != 1u) return false;   // a bucket without a root node; otherwise not a leaf (that was the fast path): an inner node
        stack.clear();
        nodes[nn] = Node{{0, 0, 0, 0}};
        const uint32_t root_code = (uint32_t)nn;
        stack.push_back(Frame{(uint32_t)nn++, 0, 0});
        // Unlike the serial walk no tentative node is ever taken: a chunk owns exactly the node slots the scan counted
        // for it, so a '1' is classified (leaf: next four bits zero) BEFORE a slot is used.
        while (!stack.empty()) {
            Frame &f = stack.back();
            if (f.next == 4) { stack.pop_back(); continue; }
            const uint32_t c = f.next++;
            if (!aux.bit()) continue;
            const size_t q = aux.pos;
            const uint32_t w = ((uint32_t)ap[q >> 3] << 8) | ((q >> 3) + 1 < ((nbits + 7) >> 3) ? ap[(q >> 3) + 1] : 0xFFu);
            if (h + stack.size() > 255) return false;   // a key longer than 255
            if (((w >> (12 - (q & 7))) & 15u) == 0) {
                aux.pos += 4;
                if (!read_leaf((uint32_t)stack.size())) return false;
                nodes[f.node].child[c] = CQ_LEAF_BIT | (uint32_t)(nl - 1);
                continue;
            }
            const uint32_t parent = f.node;
            nodes[nn] = Node{{0, 0, 0, 0}};
            const uint32_t me = (uint32_t)nn++;
            nodes[parent].child[c] = me;
            stack.push_back(Frame{me, 0, 0});   // invalidates f
        }
        out.bucket_key[b] = hv;
        out.bucket_code[b] = root_code;
    }
    return nl == m1.leaves && nn == 1 + m1.nodes && aux.pos == m1.aux_pos;
}

int decode_table(const std::string &path, DecodedTable &out, std::string &err)
{
    // CAMMIQ_DECODE_THREADS: 1 = always the serial decoder; CAMMIQ_DECODE_STEP: buckets per chunk (tests)
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    const char *tenv = getenv("CAMMIQ_DECODE_THREADS");
    if (tenv) nt = (unsigned)std::max(1, atoi(tenv));
    Mapped fi, fa;
    const bool opened = nt > 1 && fi.open(path) && fa.open(path + ".aux");
    if (opened && fa.n >= 2 && (getenv("CAMMIQ_DECODE_STEP") || getenv("CAMMIQ_DECODE_SEG") || fi.n >= (64u << 20))) {
        const uint32_t doubly = fa.p[0] >> 7, option = fa.p[0] & 127u, h = fa.p[1];
        if (option == 64 && h >= 1 && h <= 31) {
            const size_t rec = doubly ? 12 : 6;
            uint64_t step = std::max<uint64_t>(1, fi.n / (8 + rec) / ((uint64_t)nt * 8));
            if (const char *v = getenv("CAMMIQ_DECODE_STEP")) step = (uint64_t)std::max(1, atoi(v));
            std::vector<Mark> marks;
            const bool timing = getenv("CAMMIQ_LOAD_TIMING") != nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            auto lap = [&](const char *what) {
                if (timing) fprintf(stderr, "[decode_table]   %-8s %-18s %8.1f ms\n", doubly ? "(d)" : "(u)", what,
                                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            };
            // the scan: on all cores (parallel_scan; CAMMIQ_DECODE_SEG = bytes per segment, tests), or -- CAMMIQ_DECODE_STEP set,
            // or a byte stream that does not end where the parallel scan's counts put its END64 -- the serial walk
            bool scanned = false;
            if (!getenv("CAMMIQ_DECODE_STEP")) {
                size_t seg = std::max<size_t>(4096, (fa.n + (size_t)nt * 16 - 1) / ((size_t)nt * 16));
                if (const char *v = getenv("CAMMIQ_DECODE_SEG")) seg = (size_t)std::max(1, atoi(v));
                if (parallel_scan(fa.p, fa.n, 2, nt, seg, marks)) {
                    const Mark &e = marks.back();
                    const uint64_t ipos = 8 * e.buckets + rec * e.leaves;
                    uint64_t hv = 0;
                    if (ipos + 8 <= fi.n) for (int k = 0; k < 8; k++) hv = (hv << 8) | fi.p[ipos + k];
                    scanned = ipos + 8 <= fi.n && hv == CQ_EMPTY_KEY;
                }
            }
            if ((scanned || prescan(fa.p, fa.n * 8, 16, fi.p, fi.n, rec, h, step, marks)) && marks.back().leaves < 0x7FFFFFFFull &&
                marks.back().nodes + 1 < 0x7FFFFFFFull) {
                const Mark &end = marks.back();
                lap(scanned ? "scan (all cores) at" : "scan (serial) at");
                out = DecodedTable();
                out.doubly = doubly;
                out.hash_len = h;
                out.n_file_buckets = end.buckets;
                // exact sizes, uninitialised storage (RawVec): first touched by the workers that fill it
                out.leaves.resize(end.leaves); out.bucket_key.resize(end.buckets); out.bucket_code.resize(end.buckets);
                out.nodes.resize(end.nodes + 1);
                advise_huge(out.leaves.data(), end.leaves * sizeof(cq_leaf));
                advise_huge(out.bucket_key.data(), end.buckets * sizeof(uint64_t));
                advise_huge(out.bucket_code.data(), end.buckets * sizeof(uint32_t));
                advise_huge(out.nodes.data(), (end.nodes + 1) * sizeof(Node));
                out.nodes[0] = Node{{0, 0, 0, 0}};
                const size_t nchunks = marks.size() - 1;
                std::atomic<size_t> next{0};
                std::atomic<bool> bad{false};
                {
                    std::vector<std::thread> th;
                    for (unsigned t = 0; t < nt; t++)
                        th.emplace_back([&] {
                            for (;;) {
                                const size_t c = next.fetch_add(1);
                                if (c >= nchunks || bad.load()) break;
                                if (!decode_chunk(fa.p, fa.n * 8, fi.p, fi.n, rec, h, doubly != 0, marks[c], marks[c + 1], out)) bad = true;
                            }
                        });
                    for (auto &x : th) x.join();
                }
                lap("chunks done at");
                if (!bad.load()) return CQ_OK;
            }
        }
    }
    return decode_serial(path, out, err);   // small files, and every file the scan did not accept: errors are reported here
}

static int decode_serial(const std::string &path, DecodedTable &out, std::string &err)
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
