// cq_device.h -- definitions shared by the host layout builder and the HIP kernels.
// The two sides MUST agree bit for bit: the host places keys, the GPU finds them.
#ifndef CQ_DEVICE_H_
#define CQ_DEVICE_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define CQ_HD __host__ __device__ __forceinline__
#else
#define CQ_HD inline
#endif

#define CQ_LEAF_BIT 0x80000000u
#define CQ_CHAIN_BIT 0x40000000u          /* word 0 of a path-compressed trie node: CQ_CHAIN_BIT | refID << 6 | length */
#define CQ_CHAIN_RID_SHIFT 6u             /* bits 6..29 of that word: refID1 of the UNIQUE leaf the chain ends at (0: none --
                                             the chain continues, the leaf is doubly unique, or its refID needs more than 24 bits) */
#define CQ_CHAIN_RID_MAX 0xFFFFFFu
#define CQ_INLINE_RID_BIT 0x40000000u     /* slot val_d when ht_d has no such key and val_u is a unique leaf:
                                             CQ_INLINE_RID_BIT | refID1 of that leaf (saves the leaf_rids read) */
#define CQ_INLINE_PAIR_BIT 0x40000000u    /* slot val_u when ht_u has no such key and val_d is a leaf whose two refIDs
                                             are below 2^15: CQ_INLINE_PAIR_BIT | refID1 << 15 | refID2 */
#define CQ_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define CQ_KEY_MASK (~(3ull << 62))       /* h <= 31  =>  hv < 2^62 (query.cpp:482-485) */
#define CQ_SLOTS_PER_BUCKET 4             /* 4 slots x 16 B = one 64-byte HBM access */
#define CQ_SPILL_TAIL 64                  /* extra buckets past the hash range, no wrap-around */

/* One bucket of the merged unique + doubly-unique table: 64 bytes, 4 slots, stored as a
 * structure of arrays so that the probe loop's DETECT step is ONE 16-byte load per lane:
 *
 *   word  0.. 3  key_lo[4]  low 32 bits of the 2h-bit packed h-mer (the reference's map64 key);
 *                           bit 0 of key_lo[0] is the bucket's OVERFLOW flag instead
 *   word  4.. 7  key_hi[4]  high 32 bits (< 2^30); bit 30 of key_hi[0] keeps slot 0's true bit 0
 *   word  8..11  val_u[4]   trie code of the bucket root in ht_u: 0 absent (or CQ_INLINE_PAIR_BIT|refIDs, see above),
 *   word 12..15  val_d[4]   CQ_LEAF_BIT|global leaf id, or trie node index (same for ht_d;
 *                           or CQ_INLINE_RID_BIT|refID1, see above)
 *
 * An empty slot has key_hi = 0xFFFFFFFF (no valid key) and key_lo = 0xFFFFFFFF, except slot 0
 * whose key_lo is 0xFFFFFFFE so that the overflow flag reads 0.  A bucket whose overflow flag is
 * set is full and at least one key homed at or before it lives further on. */
#define CQ_BUCKET_WORDS 16
#define CQ_BW_KEY_LO 0
#define CQ_BW_KEY_HI 4
#define CQ_BW_VAL_U 8
#define CQ_BW_VAL_D 12
#define CQ_SLOT0_BIT0_IN_HI (1u << 30)

/* ---- minimizer addressing -------------------------------------------------------------
 * A window's forward h-mer F and its reverse complement R are BOTH looked up (the reference
 * scans both strands, query.cpp:480-527), and neighbouring windows overlap in h-1 bases.
 * Keys are therefore not placed by hash(key) but by hash(minimizer(key)): the canonical
 * m-mer (m = min(h, 16), or 18 for large tables: below) that minimises a 32-bit hash phi over all m-mers of the
 * key and of its reverse complement (only the minimum phi itself is used).  Consequences:
 *   * F and R have the same minimizer -> ONE bucket chain serves both strand lookups;
 *   * consecutive windows share their minimizer for ~(h-m+2)/2 positions -> adjacent lanes
 *     read the SAME 64-byte bucket and the memory system serves them with one HBM access.
 * Random HBM accesses per read drop from 2(rl-h+1) to about 2(rl-h+1)/(h-m+2)  (150 -> ~12
 * for rl=100, h=26); lookups stay exact because every slot still holds the full key. */
/* The minimizer length is a property of the INDEX, chosen when it is laid out (cq_choose_minimizer_len):
 *   m = 16  an m-mer is one 32-bit word and phi a bijection on it; windows share their minimizer for ~(h-m+2)/2 = 6
 *           positions (h = 26).  The right choice up to a few 10^8 keys.
 *   m = 18  (17..21 possible) for large tables.  The bucket address is a function of the minimizer's 32-bit phi
 *           value, and the minima of the keys concentrate on the SMALL values: N keys with w = h-m+1 m-mers each put
 *           ~N w / 2^32 keys on every small value.  At 1.26e9 keys and m = 16 (w = 11) that is 3.2 keys per address
 *           where the table has 4 slots: 9.6 % of the buckets overflowed, chains up to 9, and the kernel took
 *           39.0 ms per 20 M x 150-bp reads; m = 17: 4.6 %, 21.3 ms; m = 18 (w = 9; 36-bit m-mers folded to 32 bits by
 *           cq_phi_wide -- not a bijection, which the scheme does not need: only the minimum VALUE is used, and host
 *           and device take it over the same m-mers): 3.7 %, 20.0 ms with a uniform fold, 1.2 % and 17.5 ms with the
 *           fold that gives the small values more room (cq_phi_wide; DESIGN 3.2, profiles/r03_cfg4_*).  Shorter
 *           runs (5 positions) are the price: on configs[2]'s 84 M keys m = 18 is slower than m = 16. */
#define CQ_MAX_MINIMIZER 21               /* k + m <= 32 for the pre-pass's 11 positions per 64-bit piece */
#define CQ_MINIMIZER_SMALL 16
#define CQ_MINIMIZER_LARGE 18
#define CQ_MINIMIZER_LARGE_FROM 250000000ull   /* keys.  Measured (kernel ms, m = 16 / 17 / 18): 84 M keys 21.0 / 22.2 / 23.8;
                                                  420 M keys 18.2 / 16.8 / 16.9; 1.26e9 keys 39.0 / 21.3 / 20.0 (19: 20.6, 20: 21.2) */

CQ_HD uint32_t cq_minimizer_len(uint32_t h, uint32_t want) { return h < want ? h : want; }

/* The layout's choice for a table of n_keys keys (CAMMIQ_MINIMIZER_LEN overrides it on the host). */
CQ_HD uint32_t cq_choose_minimizer_len(uint32_t h, uint64_t n_keys)
{
    return cq_minimizer_len(h, n_keys >= CQ_MINIMIZER_LARGE_FROM ? CQ_MINIMIZER_LARGE : CQ_MINIMIZER_SMALL);
}

/* Reverse the order of the 32 two-bit symbols of x. */
CQ_HD uint64_t cq_rev2(uint64_t x)
{
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
}

/* Reverse complement of the h-symbol string in the low 2h bits of x (A<->T, C<->G = 3-s). */
CQ_HD uint64_t cq_revcomp(uint64_t x, uint32_t h) { return (~cq_rev2(x)) >> (64u - 2u * h); }

/* Bijection on 32 bits (odd multiply + xorshift): equal phi <=> equal m-mer, so the minimum
 * is unambiguous and identical for a key and its reverse complement.  One multiply only:
 * 32-bit integer multiplies are quarter rate on CDNA and this runs once per base. */
CQ_HD uint32_t cq_phi32(uint32_t c)
{
    c *= 0x9E3779B1u;
    c ^= c >> 15;
    return c;
}

/* Reverse the order of the 16 two-bit symbols of a 32-bit word. */
CQ_HD uint32_t cq_rev2_32(uint32_t x)
{
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}

/* phi of the canonical form of the m-mer f (m <= 16, f in the low 2m bits). */
CQ_HD uint32_t cq_mmer_phi(uint32_t f, uint32_t m)
{
    const uint32_t r = (~cq_rev2_32(f)) >> (32u - 2u * m);
    return cq_phi32(f < r ? f : r);
}

/* The same for 16 < m <= 21: the canonical m-mer is up to 42 bits; its low word goes through phi's multiply, the
 * high bits (< 2^10) through a second, 24-bit one (full rate on CDNA), then one xorshift: x, a uniform 32-bit value.
 *
 * x alone is not what is returned.  The bucket address is a function of the MINIMUM of a key's w = h-m+1 values, and
 * minima crowd the small values: N keys put N w (1-x)^(w-1) / 2^32 keys on value x -- 2.6 on every small one for
 * configs[4]'s 1.26e9 keys, whatever the number of buckets (3.7 % of them overflowed).  So the small values get more
 * room: two further bits e2 of the m-mer's identity extend x to 34 bits, and a monotone piecewise-linear map takes
 * (x, e2) back to 32 bits with slope 4 below 1/8, 2 up to 1/4, 1/2 up to 1/2 and 1/4 above -- a coarse version of
 * 1 - (1-x)^8, the map that would make the minimum of nine values uniform.  Being monotone it selects the same
 * minimizer as x would (up to ties that e2 now breaks); the keys per address drop from 2.6 to 0.66 where they were
 * densest (Monte Carlo of the placement: 3.2 % -> 0.65 % overflowed buckets; uniform addresses: 0.4 %). */
CQ_HD uint32_t cq_phi_wide(uint64_t c)
{
    const uint32_t lo = (uint32_t)c, hi = ((uint32_t)(c >> 32) + 1u) & 0xFFFFFFu;
    uint32_t x = lo * 0x9E3779B1u + hi * 0x85EBCBu;
    x ^= x >> 15;
    const uint32_t e2 = (lo * 0xC2B2AE35u + hi * 0x27D4EBu) >> 30;
    /* the four segments, each with its offset folded in (branch-free: data-dependent branches cost the host's layout
     * pass three times the arithmetic): [0, 2^29) -> 4x + e2; [2^29, 2^30) -> 2^31 + 2(x - 2^29) + e2/2;
     * [2^30, 2^31) -> 3 2^30 + (x - 2^30)/2; [2^31, 2^32) -> 7 2^29 + (x - 2^31)/4 */
    const uint32_t s0 = (x << 2) | e2, s1 = 0x40000000u + (x << 1) + (e2 >> 1), s2 = 0xA0000000u + (x >> 1), s3 = 0xC0000000u + (x >> 2);
    const uint32_t z = x >> 29;
    return z == 0 ? s0 : z == 1 ? s1 : z < 4 ? s2 : s3;
}

CQ_HD uint32_t cq_mmer_phi_wide(uint64_t f, uint32_t m)
{
    const uint64_t r = (~cq_rev2(f)) >> (64u - 2u * m);
    return cq_phi_wide(f < r ? f : r);
}

/* Minimizer hash of an h-mer: min of cq_mmer_phi over its h-m+1 m-mers.  Strand symmetric:
 * the m-mers of the reverse complement are the reverse complements of these m-mers. */
CQ_HD uint32_t cq_min_phi(uint64_t hmer, uint32_t h, uint32_t m)
{
    uint32_t best = 0xFFFFFFFFu;
    if (m > 16) {
        const uint64_t wmask = (1ull << (2u * m)) - 1ull;
        for (uint32_t j = 0; j + m <= h; j++) {
            const uint32_t p = cq_mmer_phi_wide((hmer >> (2u * (h - m - j))) & wmask, m);
            best = p < best ? p : best;
        }
        return best;
    }
    const uint32_t mask = m >= 16 ? 0xFFFFFFFFu : (1u << (2u * m)) - 1u;
    for (uint32_t j = 0; j + m <= h; j++) {
        const uint32_t p = cq_mmer_phi((uint32_t)(hmer >> (2u * (h - m - j))) & mask, m);
        best = p < best ? p : best;
    }
    return best;
}

/* Home bucket of a minimizer hash.  min_phi is biased towards small values (it is a
 * minimum), so it goes through a second mix; the range reduction is a multiply-high, so
 * n_buckets need not be a power of two. */
CQ_HD uint32_t cq_hash32(uint64_t k)
{
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t x = (lo * 0x9E3779B1u) ^ ((hi + 0x7F4A7C15u) * 0x85EBCA77u);
    x ^= x >> 15; x *= 0x2C1B3C6Du;
    x ^= x >> 12; x *= 0x297A2D39u;
    x ^= x >> 15;
    return x;
}

CQ_HD uint32_t cq_bucket_of_minimizer(uint32_t min_phi, uint32_t n_buckets)
{
    /* min_phi is already a mixed value, only biased towards small numbers (it is a minimum):
     * one more odd multiply spreads it over the high bits the range reduction uses. */
    const uint32_t x = (min_phi ^ 0x5BD1E995u) * 0x85EBCA6Bu;
    return (uint32_t)(((uint64_t)(x ^ (x >> 16)) * (uint64_t)n_buckets) >> 32);
}

/* Home bucket of an h-mer key (host side; the kernel has fw and rc at hand already). */
CQ_HD uint32_t cq_home_bucket(uint64_t key, uint32_t h, uint32_t m, uint32_t n_buckets)
{
    return cq_bucket_of_minimizer(cq_min_phi(key, h, m), n_buckets);
}

/* Layout of the device counter block (uint64 words) for G = n_genomes:
 *   [0, G]            cnt_u          [G+1, 2G+1]  cnt_d
 *   2G+2 nundet   2G+3 nconf   2G+4 nskipped   2G+5 flags (bit0: pair table full)
 *   2G+6 n_overflow_reads (reads that took the exact slow path) */
#define CQ_CTR_EXTRA 8
#define CQ_CTR_NUNDET(G) (2ull * ((G) + 1) + 0)
#define CQ_CTR_NCONF(G) (2ull * ((G) + 1) + 1)
#define CQ_CTR_NSKIP(G) (2ull * ((G) + 1) + 2)
#define CQ_CTR_FLAGS(G) (2ull * ((G) + 1) + 3)
#define CQ_CTR_NSLOW(G) (2ull * ((G) + 1) + 4)

#endif
