// cq_pack.cpp -- ASCII reads -> 2-bit rows (host, multi-threaded).
//
// The reference keeps reads as one heap block of ASCII per read plus a uint8_t length
// (FqReader::readFastq, /root/reference/src/query.cpp:371-393; query.hpp:35-36) and maps
// bytes through symbolIdx on every access (query.cpp:1860-1873).  Here the mapping happens
// once: base j of a read lands in word j/16 at bits [31-2(j%16) : 30-2(j%16)], so an h-mer
// starting at base p is a big-endian bit-field of the row and the kernel extracts it with
// two shifts.  Rows have a fixed stride (multiple of 16 bytes) so a tile of reads is one
// contiguous, coalesced load.
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

void pack_range(const uint8_t *bases, const uint64_t *offsets, uint64_t lo, uint64_t hi, uint32_t h,
                uint32_t sw, uint32_t *packed, uint8_t *lens, uint64_t *skipped)
{
    uint64_t sk = 0;
    for (uint64_t r = lo; r < hi; r++) {
        uint32_t *row = packed + r * sw;
        memset(row, 0, (size_t)sw * 4);
        const uint64_t len = offsets[r + 1] - offsets[r];
        if (len < h || len > 255 || len > (uint64_t)sw * 16) { lens[r] = 0; sk++; continue; }
        const uint8_t *s = bases + offsets[r];
        uint32_t bad = 0;
        for (uint32_t j = 0; j < (uint32_t)len; j++) {
            uint32_t c = kSym.t[s[j]];
            bad |= c;
            row[j >> 4] |= (c & 3u) << (30u - 2u * (j & 15u));
        }
        if (bad & 0x80u) { memset(row, 0, (size_t)sw * 4); lens[r] = 0; sk++; }
        else lens[r] = (uint8_t)len;
    }
    *skipped = sk;
}

}  // namespace

extern "C" uint32_t cq_pack_stride_words(uint32_t max_len)
{
    if (max_len > 255) max_len = 255;
    uint32_t w = (max_len + 15) / 16;
    w = (w + 3) & ~3u;
    return w ? w : 4;
}

extern "C" int cq_pack_reads(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                             uint32_t hash_len, uint32_t stride_words, uint32_t *packed, uint8_t *lens,
                             uint64_t *n_skipped)
{
    if ((!bases && n_reads && offsets && offsets[n_reads] != 0) || !offsets || !packed || !lens ||
        stride_words == 0 || (stride_words & 3u))
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
