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
#define CQ_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define CQ_OVERFLOW_BIT (1ull << 62)      /* set on slot 0's key of a bucket that spilled */
#define CQ_KEY_MASK (~(3ull << 62))       /* h <= 31  =>  hv < 2^62 (query.cpp:482-485) */
#define CQ_SLOTS_PER_BUCKET 4             /* 4 x 16 B = one 64-byte HBM access */
#define CQ_SPILL_TAIL 64                  /* extra buckets past the hash range, no wrap-around */

/* One slot of the merged unique + doubly-unique table.  key = the 2h-bit packed h-mer (the
 * reference's map64 key).  val_u / val_d = trie code of the bucket root in ht_u / ht_d:
 * 0 absent, CQ_LEAF_BIT|global leaf id, or trie node index. */
typedef struct cq_slot {
    uint64_t key;
    uint32_t val_u;
    uint32_t val_d;
} cq_slot;

/* Home bucket of a key.  32-bit mixing only (64-bit multiplies are multi-instruction on
 * CDNA); the range reduction is a multiply-high, so n_buckets need not be a power of two. */
CQ_HD uint32_t cq_hash32(uint64_t k)
{
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t x = (lo * 0x9E3779B1u) ^ ((hi + 0x7F4A7C15u) * 0x85EBCA77u);
    x ^= x >> 15; x *= 0x2C1B3C6Du;
    x ^= x >> 12; x *= 0x297A2D39u;
    x ^= x >> 15;
    return x;
}

CQ_HD uint32_t cq_home_bucket(uint64_t k, uint32_t n_buckets)
{
    return (uint32_t)(((uint64_t)cq_hash32(k) * (uint64_t)n_buckets) >> 32);
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
