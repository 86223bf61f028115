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
#include "cq_index.hpp"

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

// AVX2: 32 ASCII bases -> 8 bytes of the 2-bit string per step (first base in the top bits of the first byte).
// (c & 0xDF) >> 1 & 3 maps A,C,G,T (either case) to 0,1,3,2; x ^ (x >> 1) turns that into 0,1,2,3; a byte that is
// not ACGTacgt is found by looking the letter up again from its code.  Two multiply-adds put four codes into one
// byte (b0*64 + b1*16 + b2*4 + b3), a byte shuffle collects the eight bytes.  `out` must be zeroed for
// ceil(len / 4) bytes; the bases behind the last full block of 32 go through the scalar table.
__attribute__((target("avx2"))) bool pack_bytes_avx2(const uint8_t *s, uint32_t len, uint8_t *out)
{
    const __m256i kDF = _mm256_set1_epi8((char)0xDF), k3 = _mm256_set1_epi8(3), k1 = _mm256_set1_epi8(1);
    const __m256i lut = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                         'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i m4_1 = _mm256_set1_epi16(0x0104), m16_1 = _mm256_set1_epi32(0x00010010);
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    __m256i bad = _mm256_setzero_si256();
    uint32_t j = 0;
    for (; j + 32 <= len; j += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(s + j));
        const __m256i u = _mm256_and_si256(v, kDF);
        const __m256i r = _mm256_and_si256(_mm256_srli_epi16(u, 1), k3);
        const __m256i code = _mm256_xor_si256(r, _mm256_and_si256(_mm256_srli_epi16(r, 1), k1));
        bad = _mm256_or_si256(bad, _mm256_xor_si256(_mm256_shuffle_epi8(lut, code), u));
        const __m256i m = _mm256_madd_epi16(_mm256_maddubs_epi16(code, m4_1), m16_1);
        const __m256i p = _mm256_shuffle_epi8(m, pick);
        const uint32_t lo = (uint32_t)_mm_cvtsi128_si32(_mm256_castsi256_si128(p));
        const uint32_t hi = (uint32_t)_mm_cvtsi128_si32(_mm256_extracti128_si256(p, 1));
        memcpy(out + (j >> 2), &lo, 4);
        memcpy(out + (j >> 2) + 4, &hi, 4);
    }
    if (len - j >= 8) {   // a longer tail: one more block on a copy padded with 'A' (code 0: no bits, always valid)
        alignas(32) uint8_t pad[32];
        memset(pad, 'A', 32);
        memcpy(pad, s + j, len - j);
        const __m256i v = _mm256_load_si256((const __m256i *)pad);
        const __m256i u = _mm256_and_si256(v, kDF);
        const __m256i r = _mm256_and_si256(_mm256_srli_epi16(u, 1), k3);
        const __m256i code = _mm256_xor_si256(r, _mm256_and_si256(_mm256_srli_epi16(r, 1), k1));
        bad = _mm256_or_si256(bad, _mm256_xor_si256(_mm256_shuffle_epi8(lut, code), u));
        const __m256i m = _mm256_madd_epi16(_mm256_maddubs_epi16(code, m4_1), m16_1);
        const __m256i p = _mm256_shuffle_epi8(m, pick);
        uint8_t o8[8];
        const uint32_t lo = (uint32_t)_mm_cvtsi128_si32(_mm256_castsi256_si128(p));
        const uint32_t hi = (uint32_t)_mm_cvtsi128_si32(_mm256_extracti128_si256(p, 1));
        memcpy(o8, &lo, 4);
        memcpy(o8 + 4, &hi, 4);
        memcpy(out + (j >> 2), o8, (len - j + 3) / 4);   // only the bytes this read owns (a tight row ends there)
        j = len;
    }
    if (!_mm256_testz_si256(bad, bad)) return false;
    uint32_t sb = 0;
    for (; j < len; j++) {
        const uint32_t c = kSym.t[s[j]];
        sb |= c;
        out[j >> 2] |= (uint8_t)((c & 3u) << (6u - 2u * (j & 3u)));
    }
    return !(sb & 0x80u);
}
#endif

// Which packer runs: 2 = AVX2, 1 = BMI2 (PEXT), 0 = scalar table; CAMMIQ_PACK_ISA=scalar|bmi2|avx2 caps it (tests).
int pack_isa()
{
    static const int isa = [] {
        int best = 0;
#if defined(__x86_64__)
        if (__builtin_cpu_supports("bmi2")) best = 1;
        if (__builtin_cpu_supports("avx2")) best = 2;
#endif
        if (const char *v = getenv("CAMMIQ_PACK_ISA")) {
            const int want = !strcmp(v, "scalar") ? 0 : !strcmp(v, "bmi2") ? 1 : 2;
            if (want < best) best = want;
        }
        return best;
    }();
    return isa;
}

// One read -> the 2-bit string as bytes (most significant base first) in out[0 .. ceil(len/4)), zero behind it up to
// n_out bytes.  false: a byte that is not ACGTacgt.
bool pack_bytes(const uint8_t *s, uint32_t len, uint8_t *out, uint32_t n_out, int isa)
{
    memset(out, 0, n_out);
#if defined(__x86_64__)
    if (isa == 2) return pack_bytes_avx2(s, len, out);
#endif
    uint32_t row[16] = {0};
    bool ok;
#if defined(__x86_64__)
    ok = isa == 1 ? pack_row_bmi2(s, len, row) : pack_row_scalar(s, len, row);
#else
    ok = pack_row_scalar(s, len, row);
#endif
    for (uint32_t k = 0; k < (len + 3) / 4; k++) out[k] = (uint8_t)(row[k >> 2] >> (24u - 8u * (k & 3u)));
    return ok;
}

void pack_range(const uint8_t *bases, const uint64_t *offsets, uint64_t lo, uint64_t hi, uint32_t h,
                uint32_t sw, uint32_t *packed, uint8_t *lens, uint64_t *skipped)
{
    const int isa = pack_isa();
    uint64_t sk = 0;
    for (uint64_t r = lo; r < hi; r++) {
        uint32_t *row = packed + r * sw;
        memset(row, 0, (size_t)sw * 4);
        const uint64_t len = offsets[r + 1] - offsets[r];
        if (len < h || len > 255 || len > (uint64_t)sw * 16) { lens[r] = 0; sk++; continue; }
        const uint8_t *s = bases + offsets[r];
        bool ok;
        if (isa == 2) {   // bytes first (the AVX2 packer's output order), then words: most significant byte first
            uint8_t tmp[64];
            ok = pack_bytes(s, (uint32_t)len, tmp, 64, isa);
            for (uint32_t w = 0; w < sw; w++) {
                uint32_t v;
                memcpy(&v, tmp + 4 * w, 4);
                row[w] = __builtin_bswap32(v);
            }
        }
#if defined(__x86_64__)
        else if (isa == 1) ok = pack_row_bmi2(s, (uint32_t)len, row);
#endif
        else ok = pack_row_scalar(s, (uint32_t)len, row);
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
    cq::ReadSource src;
    src.bases = bases;
    src.offsets = offsets;
    uint32_t mn, mx;
    cq::pack_tight_slice(src, lo, hi, h, sb, packed + lo * (uint64_t)sb, lens + lo, skipped, &mn, &mx);   // below
}

}  // namespace

// Reads [lo, hi) of a source -- one buffer + offsets (cq_query), or one pointer + one length byte per read, the arrays
// FqReader::readFastq leaves (query.hpp:35-36; cq_query_reads) -- as tight rows dst[(r - lo) * sb ..], lens_out[r - lo].
// *mn / *mx: the smallest and the largest length written (skipped reads count as 0).  For the host-fed pipeline's slices.
#if defined(__x86_64__)
namespace {
// pack_tight_slice for AVX2 hosts: the packer's constants stay in registers across the reads of a slice, a read's last
// partial block is one more 32-base block over the 32 bases that end at its last whole output byte (it overlaps what the
// full blocks already wrote with the same bytes: 100 bp = blocks at 0, 32, 64 and 68; 150 bp = 0 .. 96 and 116, two bases
// through the table), no zero-fill of bytes that are written anyway.  35 -> 19 ns per 100-bp read on one core of a 2.1 GHz Xeon; same bytes as pack_bytes.
__attribute__((target("avx2"))) void pack_tight_slice_avx2(const cq::ReadSource &src, uint64_t lo, uint64_t hi, uint32_t h, uint32_t sb,
                                                             uint8_t *dst, uint8_t *lens_out, uint64_t *skipped, uint32_t *mn, uint32_t *mx)
{
    const __m256i kDF = _mm256_set1_epi8((char)0xDF), k3 = _mm256_set1_epi8(3), k1 = _mm256_set1_epi8(1);
    const __m256i lut = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                         'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i m4_1 = _mm256_set1_epi16(0x0104), m16_1 = _mm256_set1_epi32(0x00010010);
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    // 32 bases at P -> the 8 bytes (lo8, hi8 as two 32-bit halves); every byte's validity is or-ed into `bad`
#define CQ_PACK_BLOCK(P, LO, HI)                                                                              \
    {                                                                                                         \
        const __m256i v_ = _mm256_loadu_si256((const __m256i *)(P));                                          \
        const __m256i u_ = _mm256_and_si256(v_, kDF);                                                         \
        const __m256i r_ = _mm256_and_si256(_mm256_srli_epi16(u_, 1), k3);                                    \
        const __m256i c_ = _mm256_xor_si256(r_, _mm256_and_si256(_mm256_srli_epi16(r_, 1), k1));              \
        bad = _mm256_or_si256(bad, _mm256_xor_si256(_mm256_shuffle_epi8(lut, c_), u_));                       \
        const __m256i m_ = _mm256_madd_epi16(_mm256_maddubs_epi16(c_, m4_1), m16_1);                          \
        const __m256i p_ = _mm256_shuffle_epi8(m_, pick);                                                     \
        LO = (uint32_t)_mm_cvtsi128_si32(_mm256_castsi256_si128(p_));                                         \
        HI = (uint32_t)_mm_cvtsi128_si32(_mm256_extracti128_si256(p_, 1));                                    \
    }
    uint64_t sk = 0;
    uint32_t lo_len = 255, hi_len = 0;
    for (uint64_t r = lo; r < hi; r++) {
        uint8_t *row = dst + (r - lo) * (uint64_t)sb;
        const uint8_t *s;
        uint64_t len64;
        if (src.ptrs) { s = src.ptrs[r]; len64 = src.lens[r]; }
        else { s = src.bases + src.offsets[r]; len64 = src.offsets[r + 1] - src.offsets[r]; }
        uint32_t out_len = 0;
        if (len64 < h || len64 > 255 || len64 > (uint64_t)sb * 4 || (!s && len64)) { memset(row, 0, sb); sk++; }
        else {
            const uint32_t len = (uint32_t)len64, nb = (len + 3) / 4;
            __m256i bad = _mm256_setzero_si256();
            uint32_t j = 0, a, b;
            for (; j + 32 <= len; j += 32) {
                CQ_PACK_BLOCK(s + j, a, b);
                memcpy(row + (j >> 2), &a, 4);
                memcpy(row + (j >> 2) + 4, &b, 4);
            }
            uint32_t sbad = 0;
            if (j < len && len >= 32) {
                // the 32 bases that END at the last whole output byte (e = len rounded down to a multiple of four): byte-aligned
                // in the row, overlapping equal bytes of the full blocks (100 bp: blocks at 0, 32, 64 and 68; 150 bp: 0 .. 96, 116)
                const uint32_t e = len & ~3u;
                if (e > j) {
                    CQ_PACK_BLOCK(s + e - 32, a, b);
                    memcpy(row + ((e - 32) >> 2), &a, 4);
                    memcpy(row + ((e - 32) >> 2) + 4, &b, 4);
                    j = e;
                }
                if (j < len) {                          // one to three bases into the last byte
                    uint32_t v = 0;
                    for (uint32_t k = j; k < len; k++) { const uint32_t c = kSym.t[s[k]]; sbad |= c; v |= (c & 3u) << (6u - 2u * (k - j)); }
                    row[j >> 2] = (uint8_t)v;
                    j = len;
                }
            }
            if (j < len) {
                {                                       // a read shorter than 32 bases: a copy padded with 'A' (code 0: no bits, always valid)
                    alignas(32) uint8_t pad[32];
                    memset(pad, 'A', 32);
                    memcpy(pad, s + j, len - j);
                    CQ_PACK_BLOCK(pad, a, b);
                    uint8_t o8[8];
                    memcpy(o8, &a, 4);
                    memcpy(o8 + 4, &b, 4);
                    memcpy(row + (j >> 2), o8, nb - (j >> 2));   // only the bytes this read owns
                }
            }
            if (nb < sb) memset(row + nb, 0, sb - nb);
            if (!_mm256_testz_si256(bad, bad) || (sbad & 0x80u)) { memset(row, 0, sb); sk++; }
            else out_len = len;
        }
        lens_out[r - lo] = (uint8_t)out_len;
        lo_len = out_len < lo_len ? out_len : lo_len;
        hi_len = out_len > hi_len ? out_len : hi_len;
    }
#undef CQ_PACK_BLOCK
    *skipped = sk;
    *mn = lo_len;
    *mx = hi_len;
}
}  // namespace
#endif

namespace cq {
void pack_tight_slice(const ReadSource &src, uint64_t lo, uint64_t hi, uint32_t h, uint32_t sb, uint8_t *dst, uint8_t *lens_out,
                      uint64_t *skipped, uint32_t *mn, uint32_t *mx)
{
    const int isa = pack_isa();
#if defined(__x86_64__)
    if (isa == 2) { pack_tight_slice_avx2(src, lo, hi, h, sb, dst, lens_out, skipped, mn, mx); return; }
#endif
    uint64_t sk = 0;
    uint32_t lo_len = 255, hi_len = 0;
    for (uint64_t r = lo; r < hi; r++) {
        uint8_t *row = dst + (r - lo) * (uint64_t)sb;
        const uint8_t *s;
        uint64_t len;
        if (src.ptrs) { s = src.ptrs[r]; len = src.lens[r]; }
        else { s = src.bases + src.offsets[r]; len = src.offsets[r + 1] - src.offsets[r]; }
        uint32_t out_len = 0;
        if (len < h || len > 255 || len > (uint64_t)sb * 4 || (!s && len)) { memset(row, 0, sb); sk++; }
        else if (!pack_bytes(s, (uint32_t)len, row, sb, isa)) { memset(row, 0, sb); sk++; }
        else out_len = (uint32_t)len;
        lens_out[r - lo] = (uint8_t)out_len;
        lo_len = out_len < lo_len ? out_len : lo_len;
        hi_len = out_len > hi_len ? out_len : hi_len;
    }
    *skipped = sk;
    *mn = lo_len;
    *mx = hi_len;
}

// The longest length <= 255 among reads [lo, hi) (sizes the rows of a chunk).
uint32_t longest_read(const ReadSource &src, uint64_t lo, uint64_t hi)
{
    uint64_t m = 0;
    if (src.ptrs) { for (uint64_t r = lo; r < hi; r++) m = src.lens[r] > m ? src.lens[r] : m; }
    else for (uint64_t r = lo; r < hi; r++) { const uint64_t l = src.offsets[r + 1] - src.offsets[r]; if (l <= 255 && l > m) m = l; }
    return (uint32_t)m;
}
}  // namespace cq

namespace {

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
