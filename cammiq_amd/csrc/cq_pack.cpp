// cq_pack.cpp -- ASCII reads -> 2-bit rows (host, multi-threaded).
//
// The reference keeps reads as one heap block of ASCII per read plus a uint8_t length
// (FqReader::readFastq, /root/reference/src/query.cpp:371-393; query.hpp:35-36) and maps
// bytes through symbolIdx on every access (query.cpp:1860-1873).  Here the mapping happens
// once: base j of a read lands in word j/16 at bits [31-2(j%16) : 30-2(j%16)], so an h-mer
// starting at base p is a big-endian bit-field of the row and the kernel extracts it with
// two shifts.  Rows have a fixed stride of 1..16 words (ceil(longest read / 16), no padding: the
// rows are what travels over PCIe), so a tile of reads is one contiguous, coalesced load.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/cammiq_hip.h"

namespace {

struct SymTable {
    uint8_t t[256];
    SymTable()
    {
        memset(t, 0xFF, sizeof t);
        t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
        // The reference's table also maps 230/232/236/249 (build-side base+165 codes,
        // build.hpp:60), but its reverse-complement table has only 128 entries
        // (query.cpp:1875-1881): such bytes are outside the parity domain -> skipped.
    }
};
const SymTable kSym;

// Scalar row packer (reference for the fast path; also the fallback without BMI2).
inline bool pack_row_scalar(const uint8_t *s, uint32_t len, uint32_t *row)
{
    uint32_t bad = 0;
    for (uint32_t j = 0; j < len; j++) {
        const uint32_t c = kSym.t[s[j]];
        bad |= c;
        row[j >> 4] |= (c & 3u) << (30u - 2u * (j & 15u));
    }
    return !(bad & 0x80u);
}

#if defined(__x86_64__)
#include <immintrin.h>
// Eight ASCII bases -> 16 bits (first base in the top two bits), 64 bits at a time:
// (c >> 1) & 3 maps A,C,G,T (either case) to 0,1,3,2; x ^= x >> 1 turns that into 0,1,2,3;
// byte swap + PEXT gathers the eight 2-bit fields.  `bad` collects bytes that are not ACGTacgt
// (checked by rebuilding the upper-case letter from the code and comparing).
__attribute__((target("bmi2"))) inline uint32_t pack8(uint64_t w, uint64_t &bad)
{
    const uint64_t K1 = 0x0101010101010101ull;
    const uint64_t u = w & (0xDFull * K1);                  // fold to upper case
    const uint64_t raw = (w >> 1) & (3 * K1);                // A0 C1 G3 T2
    const uint64_t r0 = raw & K1, r1 = (raw >> 1) & K1, r01 = r0 & r1;
    // expected letter: 'A' + {0, 2, 0x13, 6}[raw]  ('A' 0x41, 'C' 0x43, 'T' 0x54, 'G' 0x47)
    const uint64_t add = (r0 << 1) + (r1 << 4) + (r1 << 1) + r1 - (r01 << 4) + r01;
    bad |= (0x41 * K1 + add) ^ u;
    const uint64_t code = raw ^ r1;                          // A0 C1 G2 T3
    return (uint32_t)_pext_u64(__builtin_bswap64(code), 3 * K1);
}

__attribute__((target("bmi2"))) bool pack_row_bmi2(const uint8_t *s, uint32_t len, uint32_t *row)
{
    uint64_t bad = 0;
    uint32_t j = 0;
    for (; j + 16 <= len; j += 16) {
        uint64_t a, b;
        memcpy(&a, s + j, 8);
        memcpy(&b, s + j + 8, 8);
        row[j >> 4] = (pack8(a, bad) << 16) | pack8(b, bad);
    }
    if (bad) return false;
    uint32_t sb = 0;
    for (; j < len; j++) {
        const uint32_t c = kSym.t[s[j]];
        sb |= c;
        row[j >> 4] |= (c & 3u) << (30u - 2u * (j & 15u));
    }
    return !(sb & 0x80u);
}
#endif

void pack_range(const uint8_t *bases, const uint64_t *offsets, uint64_t lo, uint64_t hi, uint32_t h,
                uint32_t sw, uint32_t *packed, uint8_t *lens, uint64_t *skipped)
{
#if defined(__x86_64__)
    const bool fast = __builtin_cpu_supports("bmi2");
#else
    const bool fast = false;
#endif
    uint64_t sk = 0;
    for (uint64_t r = lo; r < hi; r++) {
        uint32_t *row = packed + r * sw;
        memset(row, 0, (size_t)sw * 4);
        const uint64_t len = offsets[r + 1] - offsets[r];
        if (len < h || len > 255 || len > (uint64_t)sw * 16) { lens[r] = 0; sk++; continue; }
        const uint8_t *s = bases + offsets[r];
        bool ok;
#if defined(__x86_64__)
        ok = fast ? pack_row_bmi2(s, (uint32_t)len, row) : pack_row_scalar(s, (uint32_t)len, row);
#else
        ok = pack_row_scalar(s, (uint32_t)len, row);
#endif
        if (!ok) { memset(row, 0, (size_t)sw * 4); lens[r] = 0; sk++; }
        else lens[r] = (uint8_t)len;
    }
    *skipped = sk;
}

// Tight rows: the same 2-bit string at a stride of whole bytes (base j in byte j/4, first base in the top bits):
// a word row written most significant byte first, cut after stride_bytes.
void pack_range_tight(const uint8_t *bases, const uint64_t *offsets, uint64_t lo, uint64_t hi, uint32_t h,
                      uint32_t sb, uint8_t *packed, uint8_t *lens, uint64_t *skipped)
{
    uint64_t sk = 0;
    const uint32_t sw = (sb + 3) / 4;
    for (uint64_t r = lo; r < hi; r++) {
        uint32_t row[16];
        const uint64_t one[2] = {offsets[r], offsets[r + 1]};
        uint64_t s1 = 0;
        const uint64_t len = one[1] - one[0];
        if (len > (uint64_t)sb * 4) { memset(row, 0, sizeof row); lens[r] = 0; s1 = 1; }   // does not fit the tight row
        else pack_range(bases, one, 0, 1, h, sw, row, lens + r, &s1);   // writes row[0 .. sw), lens[r]
        sk += s1;
        uint8_t *dst = packed + r * (uint64_t)sb;
        for (uint32_t k = 0; k < sb; k++) dst[k] = (uint8_t)(row[k >> 2] >> (24u - 8u * (k & 3u)));
    }
    *skipped = sk;
}

template <class Fn>
uint64_t run_threads(uint64_t n_reads, Fn &&fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = std::max(1u, std::min(hw ? hw : 1u, 32u));
    if (n_reads < 65536) nt = 1;
    std::vector<uint64_t> sk(nt, 0);
    if (nt == 1) fn(0, n_reads, &sk[0]);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] { fn(n_reads * t / nt, n_reads * (t + 1) / nt, &sk[t]); });
        for (auto &x : th) x.join();
    }
    uint64_t total = 0;
    for (auto v : sk) total += v;
    return total;
}

}  // namespace

extern "C" uint32_t cq_pack_stride_bytes(uint32_t max_len)
{
    if (max_len > 255) max_len = 255;
    const uint32_t b = (max_len + 3) / 4;
    return b ? b : 1;
}

extern "C" int cq_pack_read_tight(const uint8_t *seq, uint32_t len, uint32_t hash_len, uint32_t stride_bytes, uint8_t *row,
                                  uint8_t *len_out)
{
    if (!row || !len_out || stride_bytes == 0 || stride_bytes > 64 || (!seq && len)) return CQ_ERR_ARG;
    const uint64_t off[2] = {0, len};
    uint64_t sk = 0;
    pack_range_tight(seq, off, 0, 1, hash_len, stride_bytes, row, len_out, &sk);
    return CQ_OK;
}

extern "C" int cq_pack_reads_tight(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, uint32_t hash_len,
                                   uint32_t stride_bytes, uint8_t *packed, uint8_t *lens, uint64_t *n_skipped)
{
    if ((!bases && n_reads && offsets && offsets[n_reads] != 0) || !offsets || !packed || !lens ||
        stride_bytes == 0 || stride_bytes > 64)
        return CQ_ERR_ARG;
    const uint64_t total = run_threads(n_reads, [&](uint64_t lo, uint64_t hi, uint64_t *sk) {
        pack_range_tight(bases, offsets, lo, hi, hash_len, stride_bytes, packed, lens, sk);
    });
    if (n_skipped) *n_skipped = total;
    return CQ_OK;
}

extern "C" uint32_t cq_pack_stride_words(uint32_t max_len)
{
    // 16 bases per word, no padding to 16 bytes: rows are what travels over PCIe (100 bp: 28 bytes, not 32)
    if (max_len > 255) max_len = 255;
    const uint32_t w = (max_len + 15) / 16;
    return w ? w : 1;
}

extern "C" int cq_pack_read(const uint8_t *seq, uint32_t len, uint32_t hash_len, uint32_t stride_words, uint32_t *row,
                            uint8_t *len_out)
{
    if (!row || !len_out || stride_words == 0 || stride_words > 16 || (!seq && len)) return CQ_ERR_ARG;
    const uint64_t off[2] = {0, len};
    uint64_t sk = 0;
    pack_range(seq, off, 0, 1, hash_len, stride_words, row, len_out, &sk);
    return CQ_OK;
}

extern "C" int cq_pack_reads(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                             uint32_t hash_len, uint32_t stride_words, uint32_t *packed, uint8_t *lens,
                             uint64_t *n_skipped)
{
    if ((!bases && n_reads && offsets && offsets[n_reads] != 0) || !offsets || !packed || !lens ||
        stride_words == 0 || stride_words > 16)
        return CQ_ERR_ARG;
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = std::max(1u, std::min(hw ? hw : 1u, 32u));
    if (n_reads < 65536) nt = 1;
    std::vector<uint64_t> sk(nt, 0);
    if (nt == 1) pack_range(bases, offsets, 0, n_reads, hash_len, stride_words, packed, lens, &sk[0]);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) {
            uint64_t lo = n_reads * t / nt, hi = n_reads * (t + 1) / nt;
            th.emplace_back(pack_range, bases, offsets, lo, hi, hash_len, stride_words, packed, lens, &sk[t]);
        }
        for (auto &x : th) x.join();
    }
    uint64_t total = 0;
    for (auto v : sk) total += v;
    if (n_skipped) *n_skipped = total;
    return CQ_OK;
}
