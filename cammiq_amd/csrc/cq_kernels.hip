// cq_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for CAMMiQ's classify step.
//
// What one launch computes is what one call of FqReader::query64_p / query64mt_p /
// query64_sc computes (/root/reference/src/query.cpp:458-648, 650-889, 891-1080) with
// Hash::find64_p (/root/reference/src/hashtrie.cpp:350-369) underneath -- but organised
// for a 64-wide, memory-latency-bound machine instead of a pointer-chasing CPU loop:
//
//   * every WAVE takes a private sub-tile of R reads; their 2-bit rows are loaded coalesced
//     and staged in LDS (the "sliding window" lives there, not in registers of one thread);
//     waves never synchronise with each other inside the main loop;
//   * a pre-pass hashes every m-mer position of the tile once (canonical form, bijective
//     32-bit hash) into LDS -- eleven adjacent positions per lane, whose m-mers and reverse
//     complements are bit-fields of one 64-bit piece of the row and of its reverse complement;
//     a window's MINIMIZER is then the minimum over h-m+1 adjacent LDS words instead of
//     h-m+1 hash evaluations per window;
//   * every lane owns a GROUP of consecutive window positions of one read and handles BOTH
//     strands of each: the forward h-mer is a bit-field of the row, the reverse-complement
//     h-mer is ~bitreverse of it, so no reverse-complement read is ever materialised
//     (reference: getRC + a second scan, query.cpp:447-450,503-527); the group's minima share
//     one pass over the hash words and its low words two 64-bit pieces of the row;
//   * the merged unique+doubly-unique table is addressed by the minimizer (cq_device.h):
//     forward and reverse h-mer of a window share one 64-byte bucket, and so do runs of
//     neighbouring windows -- a lane loads ONE 16-byte key_lo[4] per run of equal minimizer
//     among its windows, all of its loads issued before the single wait; across lanes the
//     memory system serves identical loads with one HBM access (reference: four robin_hood
//     lookups per window position, query.cpp:487-492,513-518);
//   * the hot loop only DETECTS (low-word compare of the four slots); the few windows
//     that may hit are compacted with ballot + prefix popcount into a per-wave LDS work
//     list, which the wave drains with full lanes: exact 64-bit compare, bucket chain,
//     trie walk -- the divergent, dependent work never stalls the probe stream;
//   * hits are gathered per read in LDS; one lane per read then de-duplicates them and
//     applies the decision rule; per-genome counters are reduced in LDS (cnt_u | cnt_d in
//     the halves of one word, when that costs no resident workgroup) and flushed with one
//     global atomic per touched genome per workgroup; rcount uses global atomics.
//
// Integer only; MFMA is deliberately unused.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "cq_device.h"
#include "cq_kernels.h"

namespace cq {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
#ifndef CQ_STAMPS
#define CQ_STAMPS 0   /* diagnostic build only: per-phase s_memtime sums go to a.stamps (own buffer) */
#endif
#if CQ_STAMPS
#define CQ_STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define CQ_STAMP(i) do {} while (0)
#endif
#ifndef CQ_WORK_DRAIN
#define CQ_WORK_DRAIN 64
#endif
#ifndef CQ_HIT_PAD
#define CQ_HIT_PAD 0   /* extra words per read in the hit lists: 1 makes the stride odd (the deciding lanes of a sub-tile on distinct banks) */
#endif
#ifndef CQ_NT_ROWS
#define CQ_NT_ROWS 1   /* read rows are a read-once stream: nontemporal loads keep them from displacing buckets in L2 */
#endif
#ifndef CQ_EXP
#define CQ_EXP 0   /* diagnostic builds only (wrong results): 1 no exact lookups, 2 no bucket chains, 3 no decision, 4 lookups stop after the first bucket, 5 no rcount atomics, 6 hits resolved but not recorded, 8 no bucket loads / compares in the probe loop, 9 no probe loop at all, 10 no pre-pass hashing */
#endif
#ifndef CQ_MAX_BLOCKS_PER_CU
#define CQ_MAX_BLOCKS_PER_CU 6
#endif
#ifndef CQ_BIG_R
#define CQ_BIG_R 8   /* reads per wave sub-tile of the "eight reads" instantiations (experiment builds: 16, with CQ_WAVES_PER_EU=3) */
#endif
#ifndef CQ_WIN_PER_LANE
#define CQ_WIN_PER_LANE 5  /* consecutive windows one lane probes per pass: 100-bp reads, 8 per wave -> 120 lanes' worth = two passes */
#endif
constexpr uint32_t kWinPerLane = CQ_WIN_PER_LANE;
#ifndef CQ_RUN_SLOTS
#define CQ_RUN_SLOTS (CQ_WIN_PER_LANE <= 6 ? 3 : 4)   /* minimizer runs per lane that get a bucket load; later runs go to the exact path */
#endif
constexpr uint32_t kRunSlots = CQ_RUN_SLOTS;
#ifndef CQ_PRE_POS
#define CQ_PRE_POS 11      /* adjacent m-mer positions one lane hashes in the pre-pass (<= 16): 100-bp reads, 8 per wave -> 8 x 8 groups = one full iteration */
#endif
constexpr uint32_t kPrePos = CQ_PRE_POS;
constexpr int kWorkDrain = CQ_WORK_DRAIN;     // drain a wave's list once it holds this many items
constexpr int kWorkCap = kWorkDrain + 64;     // an iteration appends at most one item per lane
constexpr uint32_t kNoBucket = 0xFFFFFFFFu;   // work item without a bucket: lookup_window recomputes it

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Reverse the order of the 32 two-bit symbols of x.
__device__ __forceinline__ uint64_t rev2(uint64_t x)
{
    uint64_t y = __brevll(x);
    return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}

// One 64-byte bucket, structure of arrays (cq_device.h): lo = key_lo[4], hi = key_hi[4],
// vu = val_u[4], vd = val_d[4].
struct Bucket { uint4 lo, hi, vu, vd; };

__device__ __forceinline__ Bucket load_bucket(const uint4 *__restrict__ slots, uint32_t b)
{
    const uint4 *p = slots + (size_t)b * 4;
    Bucket r;
    r.lo = p[0]; r.hi = p[1]; r.vu = p[2]; r.vd = p[3];
    return r;
}

// Compare the four slots with the forward h-mer and with its reverse complement (both live
// in the same bucket chain: same minimizer).  vf/vr = (val_u, val_d) of the slot holding
// fw / rc.  Returns true when the chain continues: overflow flag set and not both found.
__device__ __forceinline__ bool match_bucket(const Bucket &bk, uint64_t fw, uint64_t rc, uint2 &vf, uint2 &vr,
                                             bool &ff, bool &fr)
{
    const uint32_t flo = (uint32_t)fw, fhi = (uint32_t)(fw >> 32);
    const uint32_t rlo = (uint32_t)rc, rhi = (uint32_t)(rc >> 32);
    // slot 0 lends bit 0 of its key_lo to the overflow flag; the true bit sits in key_hi[0] bit 30
    // (an empty slot 0 comes out with hi = 0xBFFFFFFF and never matches)
    const uint32_t lo[4] = {(bk.lo.x & ~1u) | ((bk.hi.x >> 30) & 1u), bk.lo.y, bk.lo.z, bk.lo.w};
    const uint32_t hi[4] = {bk.hi.x & ~CQ_SLOT0_BIT0_IN_HI, bk.hi.y, bk.hi.z, bk.hi.w};
    const uint32_t vu[4] = {bk.vu.x, bk.vu.y, bk.vu.z, bk.vu.w};
    const uint32_t vd[4] = {bk.vd.x, bk.vd.y, bk.vd.z, bk.vd.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (lo[k] == flo && hi[k] == fhi) { vf.x = vu[k]; vf.y = vd[k]; ff = true; }
        if (lo[k] == rlo && hi[k] == rhi) { vr.x = vu[k]; vr.y = vd[k]; fr = true; }
    }
    return !(ff && fr) && (bk.lo.x & 1u);
}

// Base q of a staged row (A=0..T=3).
__device__ __forceinline__ uint32_t row_base(const uint32_t *row, uint32_t q)
{
    return (row[q >> 4] >> (30u - 2u * (q & 15u))) & 3u;
}

// LDS layout.  Every WAVE owns a private sub-tile of R reads (rows, m-mer hashes, hit lists,
// work list): waves never wait for each other -- no __syncthreads in the main loop.  Only
// the per-genome histogram and four scalar counters are shared by the workgroup (atomics).
struct SmemLayout { uint32_t rows, phi, len, hitcnt, hit_gid, hit_r1, hit_r2, nwork, work, per_wave, scal, hist, total; };

__host__ __device__ inline SmemLayout smem_layout(int R, int CAP, uint32_t sw, uint32_t pstride, uint32_t n_genomes, bool hist)
{
    SmemLayout L;
    uint32_t o = 0;
    L.work = o;    o += 2u * kWorkCap;                 // uint2 list first: keeps it 8-byte aligned
    L.rows = o;    o += (uint32_t)R * (sw | 1u) + 3u;  // odd row stride (banks); +3: a ragged tail is stored as whole 16-byte pieces
    L.phi = o;     o += (uint32_t)R * pstride;
    L.len = o;     o += (uint32_t)R;
    L.hitcnt = o;  o += (uint32_t)R;
    L.hit_gid = o; o += (uint32_t)R * (uint32_t)(CAP + CQ_HIT_PAD);
    L.hit_r1 = o;  o += (uint32_t)R * (uint32_t)(CAP + CQ_HIT_PAD);
    L.hit_r2 = o;  o += (uint32_t)R * (uint32_t)(CAP + CQ_HIT_PAD);
    L.nwork = o;   o += 1;
    o = (o + 3u) & ~3u;                                // 16-byte multiple per wave
    L.per_wave = o;
    o *= kWaves;
    L.scal = o;    o += 4;
    L.hist = o;    if (hist) o += n_genomes + 1u;       // one word per genome: cnt_u in the low half, cnt_d in the high half
    L.total = o;
    return L;
}

struct Tile {             // one wave's view of LDS
    uint32_t *rows;      // [R][sw|1]  2-bit rows; bases past a read's end are whatever follows (never used)
    uint32_t *phi;       // [R][pstride]  hash of the canonical m-mer starting at each base (pstride: odd, >= pmax rounded up to 8)
    uint32_t *len;       // [R]        read length (0 = skip)
    uint32_t *hitcnt;    // [R]
    uint32_t *hit_gid;   // [R][CAP]
    uint32_t *hit_r1;    // [R][CAP]
    uint32_t *hit_r2;    // [R][CAP]
    uint32_t *nwork;     // [1]        length of this wave's work list
    uint2 *work;         // [kWorkCap] .x = trie code, .y = read | strand<<8 | pos<<9
    uint32_t *hist;      // [G+1] or null: cnt_u | cnt_d << 16 (workgroup-shared; see kMaxReadsPerGroup)
    uint32_t *scal;      // [4] nundet nconf nskipped nslow   (workgroup-shared)
};

// floor(idx / d) for idx < 2^11, d < 2^8 with magic = ceil(2^19 / d): exact in that range, and
// a 24-bit multiply (full rate) instead of a 32-bit multiply-high (quarter rate).
__device__ __forceinline__ uint32_t div_small(uint32_t idx, uint32_t d, uint32_t magic)
{
    return __umul24(idx, magic) >> 19;
}

// The 64 bits (32 bases) of a staged row that start at base `off`: two funnel shifts over three
// words.  `off` may be negative or run past the read: a row is surrounded by other LDS words of
// the same wave (rows carry two zero pad words), and callers never USE bases outside the read.
__device__ __forceinline__ uint64_t row_bits64(const uint32_t *row, int off)
{
    const int q = off >> 4;
    const uint32_t s = ((uint32_t)off & 15u) * 2u;
    const uint32_t w0 = row[q], w1 = row[q + 1], w2 = row[q + 2];
    // v_alignbit takes a right-shift count in [0,31]; "left by 0" is the one case it cannot express
    const uint32_t n = (32u - s) & 31u;
    const uint32_t hi = s ? __builtin_amdgcn_alignbit(w0, w1, n) : w0;
    const uint32_t lo = s ? __builtin_amdgcn_alignbit(w1, w2, n) : w1;
    return ((uint64_t)hi << 32) | lo;
}

// hashtrie.cpp:350-369 on the path-compressed array trie.  `code` is the bucket root; returns
// the global leaf id or 0xFFFFFFFF.  The forward strand consumes bases p+h, p+h+1, ...; the
// reverse strand consumes the complements of p-1, p-2, ... (that IS rc_read[i+h+j],
// query.cpp:447-450).  A chain node stands for `len` single-child inner nodes: the walk
// either matches all of its symbols or ends without a leaf, exactly like the symbol-by-symbol
// loop (inner nodes are never leaves; running out of read inside the chain returns NULL).
// rid_inline: refID1 of the leaf when the walk ended through a chain node that carries it (a unique leaf), else 0.
__device__ __forceinline__ uint32_t walk_trie(const DevIndex &ix, uint32_t h, const uint32_t *row, uint32_t len,
                                              uint32_t code, uint32_t strand, uint32_t p, uint32_t &rid_inline)
{
    const uint32_t rem = strand ? p : (len - h - p);
    uint32_t j = 0;
    rid_inline = 0;
    for (;;) {
        if (code & CQ_LEAF_BIT) return code & ~CQ_LEAF_BIT;   // cur->isEnd
        if (j == rem) return 0xFFFFFFFFu;                       // read exhausted on an inner node
        const uint4 n = ix.nodes[code];
        if ((n.x >> 30) == 1u) {                                // chain: n.x = CQ_CHAIN_BIT | L
            const uint32_t L = n.x & 63u;
            if (rem - j < L) return 0xFFFFFFFFu;                // the read ends inside the chain
            uint64_t syms;
            if (!strand) syms = row_bits64(row, (int)(p + h + j)) >> (64u - 2u * L);
            else syms = (~rev2(row_bits64(row, (int)(p - j - L)) >> (64u - 2u * L))) >> (64u - 2u * L);
            const uint64_t label = (((uint64_t)n.y << 32) | n.z) >> (64u - 2u * L);
            if (syms != label) return 0xFFFFFFFFu;              // some children[index] == NULL
            j += L;
            code = n.w;
            rid_inline = (n.x >> CQ_CHAIN_RID_SHIFT) & CQ_CHAIN_RID_MAX;   // non-zero only when n.w is a unique leaf
        } else {
            const uint32_t sym = strand ? (3u - row_base(row, p - 1u - j)) : row_base(row, p + h + j);
            const uint32_t c = sym == 0 ? n.x : sym == 1 ? n.y : sym == 2 ? n.z : n.w;
            if (c == 0) return 0xFFFFFFFFu;                      // children[index] == NULL
            code = c;
            rid_inline = 0;
            j++;
        }
    }
}

// Hand-off between lanes of ONE wave through LDS.  The hardware executes a wave's LDS
// operations in order; the wait + compiler barrier keep hipcc from moving or caching LDS
// accesses across the hand-off.  Deliberately NOT a fence: a fence also waits for the
// wave's outstanding global loads and atomics (vmcnt), which have nothing to do with LDS.
__device__ __forceinline__ void wave_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Append one hit (global leaf id + its refIDs) to the read's hit list.
template <int CAP>
__device__ __forceinline__ void append_hit(const Tile &t, uint32_t rl, uint32_t gid, uint32_t r1, uint32_t r2)
{
    if (CQ_EXP == 6) { if (gid == 0xFFFFFFF0u) t.hitcnt[rl] = r1 + r2; return; }   // diagnostic: resolve, but record nothing
    uint32_t k;
    if (CQ_EXP == 7) { k = lane_id() & 7u; t.hitcnt[rl] = k + 1u; }   // diagnostic: no LDS atomic
    else k = atomicAdd(&t.hitcnt[rl], 1u);
    if (k < (uint32_t)CAP) {
        t.hit_gid[rl * (CAP + CQ_HIT_PAD) + k] = gid;
        t.hit_r1[rl * (CAP + CQ_HIT_PAD) + k] = r1;
        t.hit_r2[rl * (CAP + CQ_HIT_PAD) + k] = r2;
    }
}

// Resolve one candidate: walk the trie below the bucket root (most codes are depth-0 leaves
// already), fetch the leaf's refIDs, append to the read's hit list.
template <int CAP>
__device__ __forceinline__ void resolve(const DevIndex &ix, uint32_t h, const Tile &t, const uint32_t *row, uint32_t len,
                                        uint32_t rl, uint32_t code, uint32_t strand, uint32_t p)
{
    uint32_t rid_inline;
    const uint32_t gid = walk_trie(ix, h, row, len, code, strand, p, rid_inline);
    if (gid != 0xFFFFFFFFu) {
        if (rid_inline) { append_hit<CAP>(t, rl, gid, rid_inline, 0u); return; }   // the chain node brought the refID along
        const uint2 rr = ix.leaf_rids[gid];
        append_hit<CAP>(t, rl, gid, rr.x, rr.y);
    }
}

// A slot's (val_u, val_d) pair for one strand.  The common case -- a unique marker of length
// h, nothing in ht_d -- has the leaf's refID inline in val_d: no trie walk, no leaf_rids read.
template <int CAP>
__device__ __forceinline__ void resolve_pair(const DevIndex &ix, uint32_t h, const Tile &t, const uint32_t *row, uint32_t len,
                                             uint32_t rl, uint2 v, uint32_t strand, uint32_t p)
{
    if ((v.y >> 30) == 1u) {                                       // CQ_INLINE_RID_BIT
        append_hit<CAP>(t, rl, v.x & ~CQ_LEAF_BIT, v.y & ~CQ_INLINE_RID_BIT, 0u);
        return;
    }
    if ((v.x >> 30) == 1u) {                                       // CQ_INLINE_PAIR_BIT: a doubly-unique leaf, ht_d only
        append_hit<CAP>(t, rl, v.y & ~CQ_LEAF_BIT, (v.x >> 15) & 0x7FFFu, v.x & 0x7FFFu);
        return;
    }
    if (v.x) resolve<CAP>(ix, h, t, row, len, rl, v.x, strand, p);   // ht_u
    if (v.y) resolve<CAP>(ix, h, t, row, len, rl, v.y, strand, p);   // ht_d
}

// The exact lookup of one window (both strands): full 64-bit key compare along the bucket
// chain, then trie walk + hit append for every table that holds the h-mer.  Runs only for
// the few windows the probe loop flagged, one window per lane, all lanes busy.
template <int CAP>
__device__ __forceinline__ void lookup_window(const DevIndex &ix, uint32_t h, uint32_t m, const Tile &t, uint32_t swp,
                                              uint32_t pstride, uint32_t rl, uint32_t pw, uint32_t b)
{
    const uint32_t *row = t.rows + rl * swp;
    const uint32_t len = t.len[rl];
    if (b == kNoBucket) {   // the probe loop kept no bucket for this window (fourth minimizer of its lane): look it up again
        const uint32_t *ph = t.phi + rl * pstride + pw;
        uint32_t v = ph[0];
        for (uint32_t i = 1; i + m <= h; i++) v = min(v, ph[i]);
        b = cq_bucket_of_minimizer(v, ix.n_buckets);
    }
    // forward h-mer = bit-field of the row; reverse complement = ~bitreverse
    const uint32_t q = pw >> 4, s = (pw & 15u) * 2u;
    const uint64_t x = ((uint64_t)row[q] << 32) | row[q + 1];
    const uint64_t top = (x << s) | (((uint64_t)row[q + 2] << s) >> 32);
    const uint64_t fw = top >> (64u - 2u * h);
    const uint64_t rc = (~rev2(fw)) >> (64u - 2u * h);
    uint2 vf = make_uint2(0, 0), vr = make_uint2(0, 0);
    bool ff = false, fr = false, more;
    do {
        const Bucket bk = load_bucket(ix.slots, b++);
        more = match_bucket(bk, fw, rc, vf, vr, ff, fr);
    } while (more && CQ_EXP != 2 && CQ_EXP != 4);
    if (CQ_EXP == 4) return;
    if (vf.x | vf.y) resolve_pair<CAP>(ix, h, t, row, len, rl, vf, 0, pw);   // forward strand
    if (vr.x | vr.y) resolve_pair<CAP>(ix, h, t, row, len, rl, vr, 1, pw);   // reverse strand
}

// Drain this wave's work list (n <= kWorkCap items: .x = bucket, .y = read | window << 8).
template <int CAP>
__device__ __forceinline__ void drain_work(const DevIndex &ix, uint32_t h, uint32_t m, const Tile &t, uint32_t swp,
                                           uint32_t pstride, uint32_t n)
{
    wave_sync();
    for (uint32_t i = lane_id(); i < n; i += 64) {
        const uint2 it = t.work[i];
        lookup_window<CAP>(ix, h, m, t, swp, pstride, it.y & 255u, it.y >> 8, it.x);
    }
    wave_sync();
}

__device__ __forceinline__ void add_cnt(const QueryArgs &a, const Tile &t, uint32_t which, uint32_t rid)
{
    // which: 0 = cnt_u, 1 = cnt_d
    if (rid > a.n_genomes) return;  // guarded on the host (CQ_ERR_RANGE); never index out of bounds
    if (t.hist) atomicAdd(&t.hist[rid], which ? 0x10000u : 1u);
    else atomicAdd((unsigned long long *)&a.counters[which * (a.n_genomes + 1) + rid], 1ull);
}

__device__ __forceinline__ void pair_add(const QueryArgs &a, uint32_t pa, uint32_t pb)
{
    // FqReader::read_cnts_b[(a,b)]++ (query.cpp:994-997) as an open-addressing device map.
    const uint64_t key = ((uint64_t)pa << 32) | pb;
    uint32_t i = cq_hash32(key) & (a.pair_cap - 1);
    for (uint32_t probe = 0; probe < a.pair_cap; probe++) {
        unsigned long long old = atomicCAS((unsigned long long *)&a.pair_keys[i], (unsigned long long)CQ_EMPTY_KEY,
                                           (unsigned long long)key);
        if (old == CQ_EMPTY_KEY || old == key) {
            atomicAdd((unsigned long long *)&a.pair_cnts[i], 1ull);
            return;
        }
        i = (i + 1) & (a.pair_cap - 1);
    }
    atomicOr((unsigned long long *)&a.counters[CQ_CTR_FLAGS(a.n_genomes)], 1ull);
}

// Decision rule for one read: query.cpp:529-636 (P mode) / :962-1071 (SC mode), in the
// closed form of SURVEY.md 8(a) row a10.  U = distinct refID1 of unique hits, P = distinct
// (min,max) pairs of doubly-unique hits, I = intersection of all pairs.
template <int CAP>
__device__ __forceinline__ void decide(const QueryArgs &a, const Tile &t, uint32_t rl, uint32_t n)
{
    const uint32_t *gid = t.hit_gid + rl * (CAP + CQ_HIT_PAD), *r1 = t.hit_r1 + rl * (CAP + CQ_HIT_PAD), *r2 = t.hit_r2 + rl * (CAP + CQ_HIT_PAD);
    uint32_t nU = 0, u0 = 0, nP = 0, pa = 0, pb = 0, ia = 0, ib = 0;
    bool va = false, vb = false;
    uint32_t dupmask = 0;                   // CAP <= 32: bit i = hit i repeats an earlier leaf
    uint64_t seen = 0;                      // 64-bit filter over the leaf ids seen so far: the scan for an
                                            // earlier copy runs only when the filter says there may be one
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t g = gid[i];
        const uint64_t bit = 1ull << (g & 63u);
        bool dup = false;
        if (seen & bit) for (uint32_t j = 0; j < i; j++) dup |= (gid[j] == g);   // pnodes is a set (query.cpp:465)
        seen |= bit;
        if (dup) { if (CAP <= 32) dupmask |= 1u << i; continue; }
        const uint32_t x = r1[i], y = r2[i];
        if (y == 0) {                       // rids.insert(refID1)
            if (nU == 0) { u0 = x; nU = 1; }
            else if (x != u0) nU = 2;
        } else {                            // rid_pairs.insert(sorted pair)
            const uint32_t lo = x < y ? x : y, hi = x < y ? y : x;
            if (nP == 0) { pa = lo; pb = hi; nP = 1; ia = lo; ib = hi; va = true; vb = (hi != lo); }
            else {
                if (lo != pa || hi != pb) nP = 2;
                va = va && (ia == lo || ia == hi);
                vb = vb && (ib == lo || ib == hi);
            }
        }
    }
    bool counted = false;
    if (nU >= 2) atomicAdd(&t.scal[1], 1u);
    else if (nU == 1) {
        if (nP == 0) { add_cnt(a, t, 0, u0); counted = true; }
        else if ((va && ia == u0) || (vb && ib == u0)) { add_cnt(a, t, 0, u0); add_cnt(a, t, 1, u0); counted = true; }
        else atomicAdd(&t.scal[1], 1u);
    } else {
        if (nP == 0) atomicAdd(&t.scal[0], 1u);
        else if (nP == 1) {
            add_cnt(a, t, 1, pa); add_cnt(a, t, 1, pb); counted = true;
            if (a.mode == CQ_MODE_SC) pair_add(a, pa, pb);
        } else {
            const uint32_t ni = (va ? 1u : 0u) + (vb ? 1u : 0u);
            if (ni == 1) {
                const uint32_t x = va ? ia : ib;
                add_cnt(a, t, 1, x);
                if (a.mode == CQ_MODE_SC) add_cnt(a, t, 0, x);   // query.cpp:1055-1059
                counted = true;
            } else atomicAdd(&t.scal[1], 1u);
        }
    }
    if (counted && a.mode == CQ_MODE_P && a.rcount && CQ_EXP != 5) {
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t g = gid[i];
            bool dup = CAP <= 32 && ((dupmask >> (i & 31u)) & 1u);
            if (CAP > 32) for (uint32_t j = 0; j < i; j++) dup |= (gid[j] == g);
            if (!dup) atomicAdd(&a.rcount[g], 1u);               // pn->rcount += 1
        }
    }
}

}  // namespace

// R reads per wave sub-tile, CAP hit slots per read.  SLOW = exact path for reads whose hit
// list overflowed CAP in the fast kernel: reads come from a device-side list, one per wave,
// CAP covers the worst case (2 strands x 2 tables x 251 windows).
#ifndef CQ_WAVES_PER_EU
#define CQ_WAVES_PER_EU 6   /* register budget: 512 / 6 -> 80 VGPRs */
#endif
//
// H, RL, M: 0 = hash length, batch shape and minimizer length are launch arguments (any index, any read length).
// H = 26, RL = 100 / 150, M = 16 / 18 is the SAME code with CAMMiQ's default hash length (main.cpp:335-346), the
// benchmark read lengths and the two minimizer lengths the layout chooses between (cq_device.h) folded in as
// constants: the row stride, the windows / m-mer positions per read, the four small-division magics, every 2h shift
// and the minimizer network's shape stop being SGPR-resident launch arguments.  Chosen by launch_classify when index
// and batch match; results are those of the generic instantiation (tests/test_gpu_parity.py runs both).
constexpr uint32_t magic_of_c(uint32_t d) { return d == 0 ? 0u : (uint32_t)(((1u << 19) + d - 1) / d); }
// TIGHT: the rows arrive as they crossed the link (a.tight: tight_sb BYTES per read) and the staging widens them itself --
// the host-fed door's instantiations.  Its own template argument, not a launch-time branch: with the branch inside, the
// HBM-resident instantiations (TIGHT = false: the headline path) picked up 3 VGPR spills and 16 B of scratch.
template <int R, int CAP, bool SLOW, int H, int RL, int M, bool TIGHT = false>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(CQ_WAVES_PER_EU, CQ_WAVES_PER_EU)))
classify_kernel(DevIndex ix, QueryArgs a)
{
    static_assert(SLOW || R * 4 <= 64, "a sub-tile's rows (R x <= 16 words) must fit one 16-byte load per lane");
    // RL > 0: the batch's longest read is exactly RL (row stride, windows and positions per read, all magics constant);
    // RL < 0: only the row stride is fixed, -RL words (any batch of that stride: 101-, 125-, 151-bp reads -- what real
    //         FASTQs hold -- keep h, m and the stride as constants and take the window counts as launch arguments).
    static_assert((H == 0) == (RL == 0) && (H == 0) == (M == 0) &&
                      (H == 0 || (H >= 5 && H <= 31 && M <= H && M <= CQ_MAX_MINIMIZER && ((RL >= H && RL <= 255) || (RL <= -1 && RL >= -16)))),
                  "H, RL and M are fixed together");
    constexpr bool FX = H != 0;       // hash and minimizer length fixed
    constexpr bool FXL = RL > 0;      // ... and the batch's exact shape
    constexpr uint32_t cSW = RL > 0 ? (uint32_t)(RL + 15) / 16 : (RL < 0 ? (uint32_t)(-RL) : 0u);
    constexpr uint32_t cM = M, cW = FXL ? RL - H + 1 : 0, cP = cW + (H - cM);
    constexpr uint32_t cN = H - cM + 1;   // m-mers per window
    extern __shared__ __align__(16) uint32_t smem[];
    const uint32_t sw = FX ? cSW : a.stride_words, swp = sw | 1u;
    const uint32_t G1 = a.n_genomes + 1;
    const uint32_t h = FX ? (uint32_t)H : ix.hash_len, m = FX ? cM : ix.minimizer_len, nphi = h - m + 1;
    const uint32_t wmax = FXL ? cW : a.wmax, pmax = FXL ? cP : a.pmax;
    const uint32_t pstride = FXL ? (((cP + kPrePos - 1u) / kPrePos * kPrePos) | 1u) : a.pstride;
    const uint32_t magic_s = FX ? magic_of_c(cSW) : a.magic_s;
    const uint32_t magic_pp = FXL ? magic_of_c((cP + kPrePos - 1u) / kPrePos) : a.magic_pp;
    const uint32_t magic_w = FXL ? magic_of_c((cW + kWinPerLane - 1u) / kWinPerLane) : a.magic_w;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const SmemLayout L = smem_layout(R, CAP, sw, pstride, a.n_genomes, a.use_lds_hist != 0);
    uint32_t *mine = smem + wave * L.per_wave;
    Tile t;
    t.work = (uint2 *)(mine + L.work);
    t.rows = mine + L.rows;
    t.phi = mine + L.phi;
    t.len = mine + L.len;
    t.hitcnt = mine + L.hitcnt;
    t.hit_gid = mine + L.hit_gid;
    t.hit_r1 = mine + L.hit_r1;
    t.hit_r2 = mine + L.hit_r2;
    t.nwork = mine + L.nwork;
    t.scal = smem + L.scal;
    t.hist = a.use_lds_hist ? smem + L.hist : nullptr;

    if (tid < 4) t.scal[tid] = 0;
    if (t.hist) for (uint32_t i = tid; i < G1; i += kBlock) t.hist[i] = 0;
    __syncthreads();   // the only workgroup barrier before the final flush

    uint64_t n_reads = a.n_reads;
    if (SLOW) n_reads = *a.ovf_count < a.ovf_cap ? *a.ovf_count : a.ovf_cap;
    const uint64_t n_sub = (n_reads + R - 1) / R;
    const uint64_t wave_gid = (uint64_t)blockIdx.x * kWaves + wave, n_waves = (uint64_t)gridDim.x * kWaves;

    // rows + lengths of the NEXT sub-tile, requested one sub-tile ahead (fast path only)
    uint4 pf_row = make_uint4(0, 0, 0, 0);
    uint32_t pf_len = 0;
    auto prefetch = [&](uint64_t sub) {
        if (sub >= n_sub) return;
        const uint64_t r0 = sub * R;          // relative to this launch's first read
        const uint32_t nr = (uint32_t)((n_reads - r0) < (uint64_t)R ? (n_reads - r0) : (uint64_t)R);
        // the sub-tile's rows are nr * sw consecutive words (any stride 1..16: rows are not padded to 16 bytes,
        // they travel over PCIe); lane l takes words 4l .. 4l+3, the last lane of a ragged tail word by word
        if (TIGHT) {
            // tight rows (sb bytes per read, what crossed the link): the sub-tile is nr * sb consecutive bytes, lane l takes
            // bytes 16 l .. 16 l + 15 as four dwords (r0 is a multiple of R, so the sub-tile starts dword-aligned; the last lane
            // may read up to 15 bytes past the sub-tile: inside the buffer's slack).  The staging below turns them into word rows.
            const uint32_t nbytes = nr * a.tight_sb, b0 = lane * 16;
            if (b0 < nbytes) {
                const uint32_t *srcb = (const uint32_t *)(a.tight + r0 * (uint64_t)a.tight_sb + b0);
                pf_row = make_uint4(srcb[0], srcb[1], srcb[2], srcb[3]);
            }
            if (lane < nr) pf_len = a.lens[r0 + lane];
            return;
        }
        const uint32_t nwords = nr * sw, w0 = lane * 4;
        const uint32_t *src = a.packed + r0 * sw + w0;
        if (w0 + 4 <= nwords) {
#if CQ_NT_ROWS
            pf_row = make_uint4(__builtin_nontemporal_load(src), __builtin_nontemporal_load(src + 1),
                                __builtin_nontemporal_load(src + 2), __builtin_nontemporal_load(src + 3));
#else
            pf_row = make_uint4(src[0], src[1], src[2], src[3]);
#endif
        } else if (w0 < nwords) {
            pf_row.x = src[0];
            pf_row.y = w0 + 1 < nwords ? src[1] : 0u;
            pf_row.z = w0 + 2 < nwords ? src[2] : 0u;
            pf_row.w = 0u;
        }
        if (lane < nr) pf_len = a.lens[r0 + lane];
    };
    // ---- which sub-tiles a wave takes.  Stride wave_gid, wave_gid + n_waves, ... for the first 3/4 of the rounds, then one
    // at a time off a (striped) device counter (a.work_counter, zeroed per launch).  With the stride alone the kernel ends when its
    // SLOWEST wave does: a wave's time is a sum of ~n_sub / n_waves sub-tile times that differ with the reads' hits, and the
    // maximum over 6 144 waves of such sums lies several standard deviations above their mean -- measured: a 2 M-read launch
    // (41 sub-tiles per wave) takes 0.897 ms where 1/25 of a 50 M-read launch (1 017 per wave) takes 0.818.  The tail off the
    // counter evens the waves out.  max_sub (LDS histogram: 15-bit halves, launch_fast) still bounds what one wave takes.
    // (32-bit arithmetic: n_reads < 2^31, so every sub-tile index fits; the wave's loop is short of scalar registers)
    // The counter is STRIPED: kWorkStripes words, each in a cache line of its own; stripe g serves the waves with
    // wave_gid % stripes == g and hands out the sub-tiles dyn_base + g, + stripes, + 2 stripes ...  (One counter for all
    // 6 144 waves serialises on its address: the waves leave the static rounds together, 6 144 atomics on one line cost a
    // 2 M-read launch 0.26 ms -- 1.10 against 0.84 ms.)
    const bool dyn = !SLOW && a.work_counter != nullptr;
    const uint32_t static_rounds = dyn ? ((uint32_t)n_sub / (uint32_t)n_waves) * a.static_16ths / 16u : 0u;
    const uint32_t dyn_base = static_rounds * (uint32_t)n_waves;
    const uint32_t stripes = (uint32_t)n_waves < kWorkStripes ? (uint32_t)n_waves : kWorkStripes;
    const uint32_t stripe = (uint32_t)wave_gid % stripes;
    auto grab = [&]() -> uint64_t {   // the next sub-tile of this wave's stripe (wave-uniform)
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(a.work_counter + stripe * kWorkStripeWords, 1u);
        return (uint64_t)(dyn_base + stripe + stripes * (uint32_t)__builtin_amdgcn_readfirstlane(k));
    };
    uint64_t sub = (dyn && static_rounds == 0) ? grab() : wave_gid;
    uint32_t rounds_done = 0;
    if (!SLOW) prefetch(sub);
#if CQ_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory");
    // in-kernel clock (MI355X_MICROARCH.md, DVFS item 6): shader cycles (s_memtime) over the 100 MHz constant
    // clock (s_memrealtime) across this wave's whole main loop
    const unsigned long long ck_c0 = st_last, ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    while (sub < n_sub) {
        const uint64_t r0 = sub * R;
        const uint32_t nr = (uint32_t)((n_reads - r0) < (uint64_t)R ? (n_reads - r0) : (uint64_t)R);

        // ---- stage the sub-tile: 2-bit rows -> LDS.  The fast path fetched them one sub-tile
        // ahead (16 B per lane, contiguous: R reads x <= 4 vectors fit one wave instruction).
        if (!SLOW && TIGHT) {
            // the byte image of the sub-tile goes through LDS (the hash words' area: free until the pre-pass), then every lane
            // assembles one word of one row: bytes 4w .. 4w+3 of the read's tight row, first byte on top, zero past the row
            const uint32_t sb = a.tight_sb, nbytes = nr * sb;
            uint32_t *raw = t.phi;
            if (lane * 16 < nbytes) { uint32_t *d = raw + lane * 4; d[0] = pf_row.x; d[1] = pf_row.y; d[2] = pf_row.z; d[3] = pf_row.w; }
            wave_sync();
            for (uint32_t j = lane; j < nr * sw; j += 64) {
                const uint32_t r = div_small(j, sw, magic_s), w = j - r * sw, o = r * sb + 4u * w;
                const uint32_t d0 = raw[o >> 2], d1 = raw[(o >> 2) + 1], sh = 8u * (o & 3u);
                uint32_t v = sh ? (d0 >> sh) | (d1 << (32u - sh)) : d0;   // bytes o .. o+3 as they lie in memory
                v = __builtin_bswap32(v);                                  // first byte on top
                const uint32_t valid = sb - 4u * w;                        // bytes of the row this word still covers (>= 1)
                if (valid < 4u) v &= 0xFFFFFFFFu << (8u * (4u - valid));
                t.rows[r * swp + w] = v;
            }   // (the wave_sync below, before the pre-pass, also covers the byte image)
        } else if (!SLOW) {
            const uint32_t nwords = nr * sw, w = lane * 4;
            if (w < nwords) {
                if (swp == sw) {        // odd stride: the sub-tile's image in LDS is its image in HBM, one 16-byte store per lane
                    uint32_t *dst = t.rows + w;
                    dst[0] = pf_row.x; dst[1] = pf_row.y; dst[2] = pf_row.z; dst[3] = pf_row.w;
                } else {                // even stride: rows are spread to stride sw + 1, word by word
                    const uint32_t v[4] = {pf_row.x, pf_row.y, pf_row.z, pf_row.w};
                    uint32_t wv = w;
                    asm volatile("" : "+v"(wv));   // not loop-invariant for the compiler: four hoisted addresses would cost registers on the odd-stride path
#pragma unroll
                    for (uint32_t i = 0; i < 4; i++)
                        if (wv + i < nwords) t.rows[wv + i + div_small(wv + i, sw, magic_s)] = v[i];
                }
            }
        } else {
            for (uint32_t i = lane; i < nr * sw; i += 64) {
                const uint32_t rl = i / sw, c = i - rl * sw;
                if (a.tight) {   // the exact path reads its few rows byte by byte
                    const uint8_t *p = a.tight + (uint64_t)a.ovf_list[r0 + rl] * a.tight_sb + 4u * c;
                    uint32_t v = 0;
                    for (uint32_t k = 0; k < 4u; k++)
                        if (4u * c + k < a.tight_sb) v |= (uint32_t)p[k] << (24u - 8u * k);
                    t.rows[rl * swp + c] = v;
                } else
                    t.rows[rl * swp + c] = a.packed[(uint64_t)a.ovf_list[r0 + rl] * sw + c];
            }
        }
        if (lane < (uint32_t)R) {
            uint32_t len = 0;
            if (lane < nr) len = SLOW ? a.lens[a.ovf_list[r0 + lane]] : pf_len;
            t.len[lane] = len;
            t.hitcnt[lane] = 0;
        }
        // the sub-tile after this one: by stride, or off the counter once the static rounds are through
        rounds_done++;
        uint64_t sub_next = sub + n_waves;
        if (dyn && rounds_done >= static_rounds) sub_next = (a.max_sub && rounds_done >= a.max_sub) ? n_sub : grab();
        if (!SLOW) prefetch(sub_next);   // in flight while this sub-tile is processed
        wave_sync();
        CQ_STAMP(0);   // staging

        // ---- pre-pass: hash the canonical m-mer at every base position, once.  A lane takes
        // kPrePos adjacent positions: their m-mers are bit-fields of one 64-bit piece of the row,
        // their reverse complements bit-fields of that piece's reverse complement.
        {
            const uint32_t gpr = (pmax + kPrePos - 1u) / kPrePos;  // position groups per read
            const uint32_t total = nr * gpr;
            const uint32_t mmask = m >= 16 ? 0xFFFFFFFFu : (1u << (2u * m)) - 1u;
            for (uint32_t it = lane; it < (CQ_EXP == 10 ? 0u : total); it += 64) {
                const uint32_t rl = div_small(it, gpr, magic_pp), j = (it - __umul24(rl, gpr)) * kPrePos;
                const uint32_t len = t.len[rl];
                if (j + m > len) continue;
                const uint32_t *row = t.rows + __umul24(rl, swp);
                const uint64_t w64 = row_bits64(row, (int)j);      // 32 bases starting at base j
                const uint64_t rc64 = ~rev2(w64);
                uint32_t *dst = t.phi + __umul24(rl, pstride) + j;
                // all kPrePos positions of the group are hashed and stored, valid or not (a row of phi holds whole
                // groups): positions past len - m get the hash of whatever follows the read and are never read by a
                // valid window -- no per-position branch
                if (m > 16) {   // large tables: 64-bit m-mers folded to 32 bits (cq_device.h)
                    const uint64_t wmask = (1ull << (2u * m)) - 1ull;
#pragma unroll
                    for (uint32_t k = 0; k < kPrePos; k++) {
                        const uint64_t f = (w64 << (2u * k)) >> (64u - 2u * m);
                        const uint64_t r = (rc64 >> (2u * k)) & wmask;
                        dst[k] = cq_phi_wide(f < r ? f : r);
                    }
                    continue;
                }
#pragma unroll
                for (uint32_t k = 0; k < kPrePos; k++) {
                    const uint32_t f = (uint32_t)((w64 << (2u * k)) >> (64u - 2u * m));
                    const uint32_t r = (uint32_t)(rc64 >> (2u * k)) & mmask;
                    dst[k] = cq_phi32(f < r ? f : r);
                }
            }
        }
        wave_sync();
        CQ_STAMP(1);   // pre-pass

        // ---- probe phase: lane = (read, group of KW consecutive windows).  The hot loop only
        // DETECTS: it reads a window's bucket and compares the low 32 key bits of the four slots
        // with the low words of the forward and the reverse-complement h-mer.  Nearly every
        // window stops here.  The windows of a lane share almost everything: their minima come
        // out of one pass over h-m+KW hash words, neighbouring windows usually have the SAME
        // minimizer, hence the same bucket (loaded once), and their low words are bit-fields of
        // two 64-bit pieces of the row.  A window with a low-word match (a hit, up to a 2^-32
        // fluke) or with an overflowed bucket is appended to the wave's work list -- ballot +
        // prefix popcount, the list length is wave-uniform and lives in a register -- and gets
        // the exact treatment (lookup_window) once 64 of them are waiting.
        constexpr uint32_t KW = kWinPerLane;
        constexpr uint32_t NS = kRunSlots;                      // bucket slots (minimizer runs) a lane keeps per pass
        static_assert(32u + 2u * (KW - 1u) <= 64u && KW <= 32u, "a lane's windows must fit one 64-bit piece of the row");
        const uint32_t gpr = (wmax + KW - 1u) / KW;             // window groups per read
        const uint32_t total = nr * gpr;
        const uint32_t hmask = h < 16 ? (0xFFFFFFFFu >> (32u - 2u * h)) : 0xFFFFFFFFu;   // keys shorter than 32 bits
        uint32_t nw = 0;
        for (uint32_t base = 0; base < (CQ_EXP == 9 ? 0u : total); base += 64) {
            const uint32_t idx = base + lane;
            bool act = idx < total;
            uint32_t rl = 0, pw0 = 0, len = 0, fl = 0;
            // A lane's KW consecutive windows fall into RUNS of equal minimizer; run r (r < NS) gets bucket slot r.
            // bnd[r] = first window of run r (r >= 1; >= KW when the lane has no such run), bkt[r] = its bucket.
            uint32_t bnd[NS + 1], bkt[NS];
#pragma unroll
            for (uint32_t r = 0; r <= NS; r++) bnd[r] = KW;
#pragma unroll
            for (uint32_t r = 0; r < NS; r++) bkt[r] = 0;
            if (act) {
                rl = div_small(idx, gpr, magic_w);
                pw0 = (idx - __umul24(rl, gpr)) * KW;
                len = t.len[rl];
                act = (len >= h) && (pw0 + h <= len);
            }
            if (act) {
                const uint32_t nwin = len - h + 1u - pw0;         // valid windows from pw0 on (>= 1)
                // minimizer hash of every window = min over its h-m+1 m-mers
                const uint32_t *ph = t.phi + __umul24(rl, pstride) + pw0;
                uint32_t mp[KW];
                if (nphi == 11 && KW == 4) {   // h = 26 (CAMMiQ's default): fourteen LDS reads serve four windows
                    const uint32_t v0 = ph[0], v1 = ph[1], v2 = ph[2], v3 = ph[3], v4 = ph[4], v5 = ph[5], v6 = ph[6],
                                   v7 = ph[7], v8 = ph[8], v9 = ph[9], v10 = ph[10], v11 = ph[11], v12 = ph[12], v13 = ph[13];
                    const uint32_t core = min(min(min(v3, v4), min(v5, v6)), min(min(v7, v8), min(v9, v10)));
                    const uint32_t x = min(v1, v2), y = min(v11, v12);
                    mp[0] = min(min(core, v0), x);
                    mp[1 % KW] = min(min(core, x), v11);
                    mp[2 % KW] = min(min(core, v2), y);
                    mp[3 % KW] = min(min(core, y), v13);
                } else if (nphi == 11 && KW == 5) {   // fifteen LDS reads serve five windows
                    const uint32_t v0 = ph[0], v1 = ph[1], v2 = ph[2], v3 = ph[3], v4 = ph[4], v5 = ph[5], v6 = ph[6], v7 = ph[7],
                                   v8 = ph[8], v9 = ph[9], v10 = ph[10], v11 = ph[11], v12 = ph[12], v13 = ph[13], v14 = ph[14];
                    const uint32_t core = min(min(min(v4, v5), min(v6, v7)), min(min(v8, v9), v10));
                    const uint32_t s3 = min(v3, core), s2 = min(v2, s3), s1 = min(v1, s2);
                    const uint32_t p12 = min(v11, v12), p13 = min(p12, v13);
                    mp[0] = min(v0, s1);
                    mp[1 % KW] = min(s1, v11);
                    mp[2 % KW] = min(s2, p12);
                    mp[3 % KW] = min(s3, p13);
                    mp[4 % KW] = min(core, min(p13, v14));
                } else if (nphi == 11 && KW <= 11) {
                    // every window contains m-mer 10: window k = min(suffix minimum of v[k..10], prefix minimum of v[10..k+10])
                    uint32_t v[KW + 10];
#pragma unroll
                    for (uint32_t i = 0; i < KW + 10; i++) v[i] = ph[i];
                    uint32_t suf[11];
                    suf[10] = v[10];
#pragma unroll
                    for (int i = 9; i >= 0; i--) suf[i] = min(v[i], suf[i + 1]);
                    uint32_t pre = v[10];
                    mp[0] = suf[0];
#pragma unroll
                    for (uint32_t k = 1; k < KW; k++) {
                        pre = min(pre, v[k + 10]);
                        mp[k] = min(suf[k < 11 ? k : 10], pre);
                    }
                } else if (FX && cN >= KW) {
                    // any fixed nphi >= KW (minimizer-length experiments): every window contains m-mer cN-1, so
                    // window k = min(suffix minimum of v[k..cN-1], prefix minimum of v[cN-1..k+cN-1])
                    constexpr uint32_t N = FX ? (cN >= KW ? cN : KW) : KW;
                    uint32_t v[KW + N - 1];
#pragma unroll
                    for (uint32_t i = 0; i < KW + N - 1; i++) v[i] = ph[i];
                    uint32_t suf[N];
                    suf[N - 1] = v[N - 1];
#pragma unroll
                    for (int i = (int)N - 2; i >= 0; i--) suf[i] = min(v[i], suf[i + 1]);
                    uint32_t pre = v[N - 1];
                    mp[0] = suf[0];
#pragma unroll
                    for (uint32_t k = 1; k < KW; k++) {
                        pre = min(pre, v[k + N - 1]);
                        mp[k] = min(suf[k], pre);
                    }
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < KW; k++) {
                        uint32_t v = ph[k];
                        for (uint32_t i = 1; i < nphi; i++) v = min(v, ph[k + i]);
                        mp[k] = v;
                    }
                }
                // run boundaries: bit k of cm = window k starts a new run
                uint32_t cm = 0;
#pragma unroll
                for (uint32_t k = 1; k < KW; k++) cm |= ((k < nwin) && (mp[k] != mp[k - 1])) ? (1u << k) : 0u;
                uint32_t mr[NS];
                mr[0] = mp[0];
#pragma unroll
                for (uint32_t r = 1; r <= NS; r++) {
                    bnd[r] = cm ? (uint32_t)__ffs(cm) - 1u : KW;
                    cm &= cm - 1u;
                    if (r < NS) {
                        mr[r] = 0;
#pragma unroll
                        for (uint32_t k = 1; k < KW; k++) mr[r] = (bnd[r] == k) ? mp[k] : mr[r];
                    }
                }
                // buckets: key_lo[4] is the only bucket read of the hot loop -- ONE 16-byte load per minimizer run,
                // all of a lane's loads issued back to back and waited for once (no loaded value is copied or
                // selected before the last load has been issued: a dependent use would put a wait between them)
                uint4 kk[NS];
                bkt[0] = cq_bucket_of_minimizer(mr[0], ix.n_buckets);
#pragma unroll
                for (uint32_t r = 1; r < NS; r++)
                    if (bnd[r] < KW) bkt[r] = cq_bucket_of_minimizer(mr[r], ix.n_buckets);
                kk[0] = ix.slots[(size_t)bkt[0] * 4];
#pragma unroll
                for (uint32_t r = 1; r < NS; r++) {
                    kk[r] = make_uint4(0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);   // never selected unless loaded
                    if (bnd[r] < KW) kk[r] = ix.slots[(size_t)bkt[r] * 4];
                }
                // low words.  Forward h-mer: the 32 bits that END at the window's end -- all KW of them
                // inside the 64 bits that end at the last window's end.  Reverse complement: ~reverse of
                // the 32 bits that START the window -- bit-fields of the reverse complement of the 64
                // bits that start at the first window (reversal puts a window's first base into the
                // lowest symbol, so the low 2h bits are the whole key when h < 16).
                const uint32_t *row = t.rows + __umul24(rl, swp);
                const uint64_t t64 = row_bits64(row, (int)(pw0 + (KW - 1u) + h) - 32);
                const uint64_t rh64 = ~rev2(row_bits64(row, (int)pw0));
#pragma unroll
                for (uint32_t k = 0; k < KW; k++) {
                    const uint32_t flo = (uint32_t)(t64 >> (2u * (KW - 1u - k))) & hmask;
                    const uint32_t rlo = (uint32_t)(rh64 >> (2u * k)) & hmask;
                    uint4 kl = kk[0];
                    if (k >= 1) {
#pragma unroll
                        for (uint32_t r = 1; r < NS; r++) {
                            const bool in = k >= bnd[r];
                            kl.x = in ? kk[r].x : kl.x; kl.y = in ? kk[r].y : kl.y;
                            kl.z = in ? kk[r].z : kl.z; kl.w = in ? kk[r].w : kl.w;
                        }
                    }
                    // slot 0's bit 0 is the overflow flag: compare it without that bit
                    const bool f = ((kl.x ^ flo) < 2u) | (kl.y == flo) | (kl.z == flo) | (kl.w == flo) |
                                   ((kl.x ^ rlo) < 2u) | (kl.y == rlo) | (kl.z == rlo) | (kl.w == rlo) |
                                   ((kl.x & 1u) != 0) | (k >= bnd[NS]);   // run without a slot: no bucket was loaded, the exact path decides
                    if (f && k < nwin && CQ_EXP != 8) fl |= 1u << k;
                }
                if (CQ_EXP == 8) fl = (mr[0] == 0x12345u) ? 1u : 0u;   // keep the minimizer work alive
            }
            // the one place where windows enter the work list (and where it is drained): a lane hands
            // over one flagged window per trip -- most passes see one trip or none
            for (;;) {
                const bool has = fl != 0;
                const uint64_t mask = __ballot(has);
                if (!mask) break;
                if (has) {
                    const uint32_t k = (uint32_t)__ffs(fl) - 1u;
                    fl &= fl - 1u;
                    uint32_t b = bkt[0];
#pragma unroll
                    for (uint32_t r = 1; r < NS; r++) b = (k >= bnd[r]) ? bkt[r] : b;
                    b = (k >= bnd[NS]) ? kNoBucket : b;
                    const uint32_t off = nw + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                    t.work[off] = make_uint2(b, rl | ((pw0 + k) << 8));
                }
                nw += (uint32_t)__popcll(mask);
                if (nw >= (uint32_t)kWorkDrain) { CQ_STAMP(2); if (CQ_EXP != 1) drain_work<CAP>(ix, h, m, t, swp, pstride, nw); nw = 0; CQ_STAMP(3); }
            }
        }
        CQ_STAMP(2);   // probe loop
        if (nw && CQ_EXP != 1) drain_work<CAP>(ix, h, m, t, swp, pstride, nw);
        CQ_STAMP(3);   // exact lookups
        wave_sync();

        // ---- decision phase: one lane per read
        if (lane < nr) {
            const uint32_t len = t.len[lane];
            if (len < h) atomicAdd(&t.scal[2], 1u);   // outside the parity domain
            else {
                const uint32_t n = t.hitcnt[lane];
                if (n > (uint32_t)CAP) {
                    if (!SLOW) {   // hand the read to the exact slow path
                        const uint32_t k = atomicAdd(a.ovf_count, 1u);
                        if (k < a.ovf_cap) a.ovf_list[k] = (uint32_t)(a.read0 + r0 + lane);   // index into the CALL's rows
                        atomicAdd(&t.scal[3], 1u);
                    }
                } else if (CQ_EXP != 3) decide<CAP>(a, t, lane, n);
            }
        }
        wave_sync();   // hit lists fully consumed before the next sub-tile resets them
        CQ_STAMP(4);   // decision
        sub = sub_next;
    }
#if CQ_STAMPS
    if (!SLOW && lane == 0 && a.stamps) {
        for (int i = 0; i < 5; i++) atomicAdd((unsigned long long *)&a.stamps[i], st_acc[i]);
        const unsigned long long ck_r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd((unsigned long long *)&a.stamps[5], st_last - ck_c0);   // st_last = the last phase stamp of this wave
        atomicAdd((unsigned long long *)&a.stamps[6], ck_r1 - ck_r0);
    }
#endif

    // ---- flush workgroup-level counters
    __syncthreads();
    if (t.hist)
        for (uint32_t i = tid; i < G1; i += kBlock) {
            const uint32_t v = t.hist[i];
            if (v & 0xFFFFu) atomicAdd((unsigned long long *)&a.counters[i], (unsigned long long)(v & 0xFFFFu));
            if (v >> 16) atomicAdd((unsigned long long *)&a.counters[G1 + i], (unsigned long long)(v >> 16));
        }
    if (tid < 4 && t.scal[tid]) {
        const uint64_t off[4] = {CQ_CTR_NUNDET(a.n_genomes), CQ_CTR_NCONF(a.n_genomes), CQ_CTR_NSKIP(a.n_genomes),
                                 CQ_CTR_NSLOW(a.n_genomes)};
        atomicAdd((unsigned long long *)&a.counters[off[tid]], (unsigned long long)t.scal[tid]);
    }
}

#ifndef CQ_FAST_CAP
#define CQ_FAST_CAP 16
#endif
// Reads per wave sub-tile of the fast kernel: eight, or four when eight reads' rows and hash words take so much
// LDS that fewer workgroups stay resident (long reads) -- chosen per launch, see launch_classify.
constexpr int kFastCAP = CQ_FAST_CAP;
constexpr int kSlowR = 1, kSlowCAP = 1024;
constexpr uint64_t kMaxReadsPerGroup = 32767;   // per workgroup and launch: the LDS histogram's 15-bit halves (see launch_fast)

static size_t smem_bytes(int R, int CAP, const QueryArgs &a, bool hist)
{
    return (size_t)smem_layout(R, CAP, a.stride_words, a.pstride, a.n_genomes, hist).total * 4;
}

bool lds_hist_fits(uint32_t n_genomes)
{
    if (const char *v = getenv("CAMMIQ_LDS_HIST_MAX")) return ((size_t)n_genomes + 1) * 4 <= (size_t)atoi(v);   // tuning knob
    return ((size_t)n_genomes + 1) * 4 <= 32 * 1024;
}

static uint32_t magic_of(uint32_t d) { return d == 0 ? 0u : (uint32_t)(((1u << 19) + d - 1) / d); }

__global__ void __launch_bounds__(256) accumulate_kernel(uint64_t *dst64, const uint64_t *src64, uint64_t n64,
                                                         uint32_t *dst32, const uint32_t *src32, uint64_t n32)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += stride) dst64[i] += src64[i];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += stride) dst32[i] += src32[i];
}

// One thread per output word: word w of read r = bytes 4w .. 4w+3 of the read's tight row, first byte on top.
// The byte loads of neighbouring threads fall into the same cache lines (a wave covers 64 x 4 consecutive-ish
// bytes); at 2 M reads per chunk this is ~0.1 ms next to 0.9 ms of classify kernel.
__global__ void __launch_bounds__(256) widen_rows_kernel(const uint8_t *__restrict__ tight, uint32_t sb,
                                                         uint32_t *__restrict__ rows, uint32_t sw, uint64_t n_words)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
        const uint64_t r = i / sw;
        const uint32_t w = (uint32_t)(i - r * sw), b0 = w * 4u;
        const uint8_t *src = tight + r * sb + b0;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (b0 + k < sb) v |= (uint32_t)src[k] << (24u - 8u * k);
        rows[i] = v;
    }
}

hipError_t launch_widen_rows(const uint8_t *tight, uint32_t sb, uint32_t *rows, uint32_t sw, uint64_t n_reads,
                             hipStream_t stream)
{
    const uint64_t n_words = n_reads * sw;
    if (n_words == 0) return hipSuccess;
    uint64_t grid = (n_words + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(widen_rows_kernel, dim3((unsigned)grid), dim3(256), 0, stream, tight, sb, rows, sw, n_words);
    return hipGetLastError();
}

// rcount leaves the device NARROW: one byte per leaf (the count, saturated at 255) plus an escape list of
// (leaf, count) for the few entries of 255 and more; the host widens into the caller's uint32 arrays
// (pleafNode::rcount is a uint32, hashtrie.hpp:43) and overwrites the escaped entries -- bit-exact, a quarter of
// the bytes on the link behind the last classify kernel.
//
// The kernel IS the copy: out8 and flags are page-locked HOST memory, written by the lanes themselves (16 entries per lane
// and step: four 16-byte loads from HBM, one 16-byte store over the link; a wave's store is 1 KB contiguous).  A workgroup
// owns one segment of `seg` entries (kNarrowSeg); when its bytes are out it publishes flags[segment] = epoch (system-scope
// release), which the host's widening threads poll -- no event, no runtime call, no wake-up between the link and the
// threads (the first version copied pieces with hipMemcpyAsync + one hipEventSynchronize per piece: the hand-offs, not the
// bytes, made a 1.5 ms transfer take 4.4 ms).  The workgroup that finishes last publishes the number of escapes and the
// final flag flags[n_segments].  *esc_count may end above esc_cap: the host then falls back to the plain uint32 copy.
__global__ void __launch_bounds__(256) narrow_rcount_kernel(const uint32_t *__restrict__ rc, uint64_t n, uint8_t *__restrict__ out8,
                                                            uint64_t seg, uint64_t n_seg, uint32_t *__restrict__ flags, uint32_t epoch,
                                                            uint2 *__restrict__ esc, uint32_t *__restrict__ esc_count, uint32_t esc_cap,
                                                            uint32_t *__restrict__ blocks_done, uint32_t *__restrict__ host_esc_count)
{
    auto squeeze = [&](uint32_t v, uint64_t idx) -> uint32_t {
        if (v < 255u) return v;
        const uint32_t at = atomicAdd(esc_count, 1u);
        if (at < esc_cap) esc[at] = make_uint2((uint32_t)idx, v);
        return 255u;
    };
    // A FEW workgroups walk the segments in order (workgroup b takes b, b + grid, ...): the segments reach the host one
    // after the other at the link's rate, so the host's threads widen segment k while k + 1 .. are still on the wire.
    // (One workgroup per segment, all resident at once, shares the link among all of them: every flag came at the end.)
    for (uint64_t sg = blockIdx.x; sg < n_seg; sg += gridDim.x) {
        const uint64_t lo = sg * seg, hi = lo + seg < n ? lo + seg : n;   // seg is a multiple of 16: lo stays 16-entry aligned
        const uint64_t hi16 = lo + ((hi - lo) & ~(uint64_t)15);
        for (uint64_t i = lo + (uint64_t)threadIdx.x * 16; i < hi16; i += 256 * 16) {
            const uint4 *src = (const uint4 *)(rc + i);
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = src[q];
                const uint64_t i0 = i + (uint64_t)q * 4;
                w[q] = squeeze(v.x, i0) | squeeze(v.y, i0 + 1) << 8 | squeeze(v.z, i0 + 2) << 16 | squeeze(v.w, i0 + 3) << 24;
            }
            *(uint4 *)(out8 + i) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        for (uint64_t i = hi16 + threadIdx.x; i < hi; i += 256) out8[i] = (uint8_t)squeeze(rc[i], i);
        __threadfence_system();   // this lane's bytes are on the host before the flag can be
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + sg, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0 && atomicAdd(blocks_done, 1u) == gridDim.x - 1) {   // every other workgroup's escapes are counted (their atomics precede their ticket)
        __threadfence();
        *host_esc_count = __hip_atomic_load(esc_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *blocks_done = 0;
        __threadfence_system();
        __hip_atomic_store(flags + n_seg, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_narrow_rcount(const uint32_t *rc, uint64_t n, uint8_t *host_out8, uint64_t seg, uint32_t *host_flags, uint32_t epoch, uint2 *esc,
                                uint32_t *esc_count, uint32_t esc_cap, uint32_t *blocks_done, uint32_t *host_esc_count, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (seg == 0 || (seg & 15u)) return hipErrorInvalidValue;
    const uint64_t n_seg = (n + seg - 1) / seg;
    // workgroups in flight: enough 16-byte stores outstanding to keep the link busy (64 x 256 lanes x 16 B = 256 KB per step),
    // few enough that the segments arrive in order (CAMMIQ_NARROW_WGS: tuning knob)
    uint64_t grid = 64;
    if (const char *v = getenv("CAMMIQ_NARROW_WGS")) grid = (uint64_t)std::max(1, atoi(v));
    if (grid > n_seg) grid = n_seg;
    hipLaunchKernelGGL(narrow_rcount_kernel, dim3((unsigned)grid), dim3(256), 0, stream, rc, n, host_out8, seg, n_seg, host_flags, epoch, esc, esc_count,
                       esc_cap, blocks_done, host_esc_count);
    return hipGetLastError();
}

// ---- board calibrators (cq_calibrate; diagnostic, never on the classify path) ------------------------------------
// What the classify kernel is held against -- the chip's rate of random 16-byte loads from a table of THIS size -- and
// the shader clock the chip holds under such a load, measured on the board a run is on, in about a second.  The
// product kernel stays stamp-free; these kernels carry the s_memtime / s_memrealtime stamps.
//   MIX = false: every lane issues `iters` independent random 16-byte loads from the handle's own table (the probe
//                loop's access: one key_lo quad per minimizer run), four in flight per lane.
//   MIX = true:  the same loads with what the real kernel has beside them and a plain gather lacks: a returnless
//                global atomic to a random word of an rcount-sized array per 16 loads (rcount++), and LDS traffic
//                (per load: a 16-byte store of the loaded quad into a wave-private region and two 8-byte reads back).
// stamps[2 b], stamps[2 b + 1] = shader cycles / 100 MHz ticks workgroup b's first wave spent in the loop.
template <bool MIX>
__global__ void __launch_bounds__(256) calib_gather_kernel(const uint4 *__restrict__ tab, uint64_t n_units, int iters,
                                                           uint32_t *__restrict__ atom, uint64_t n_atom, uint64_t *__restrict__ stamps,
                                                           uint32_t *__restrict__ sink)
{
    __shared__ uint4 lds[MIX ? 256 * 4 : 1];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = (uint64_t)gid * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i += 4) {
        uint64_t idx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            idx[u] = (uint64_t)(((__uint128_t)s * n_units) >> 64);
        }
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = tab[idx[u]];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc += v[u].x ^ v[u].w;
            if (MIX) {
                lds[threadIdx.x * 4 + u] = v[u];
                const uint2 *back = (const uint2 *)&lds[(threadIdx.x ^ 1u) * 4 + u];   // the neighbour lane's quad: a real LDS round trip
                acc += back[0].x + back[1].y;
            }
        }
        if (MIX && (i & 12) == 0) {   // one returnless atomic per 16 loads
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            atomicAdd(atom + (uint64_t)(((__uint128_t)s * n_atom) >> 64), 0u);   // adds nothing: the array keeps its contents
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// The third calibrator: DEPENDENT random loads -- the address of a lane's next 16-byte load is a function of the quad
// the last one returned (one load in flight per lane) -- from ONE wave per CU (at the classify kernel's residency the
// chase saturates the memory system like the gather does: 40 G loads/s either way; one wave per CU is far below that,
// so lanes in flight / rate is what a lone request takes).  The classify kernel is a
// chain of such round trips per sub-tile (bucket quad -> compare -> bucket re-read -> trie node -> refIDs) hidden only
// by the number of resident waves, so what a board's memory system answers a lone request in decides its time where the
// saturated gather rate above is the same on every board.  rate = loads/s; latency = lanes in flight / rate.
__global__ void __launch_bounds__(256) calib_chase_kernel(const uint4 *__restrict__ tab, uint64_t n_units, int iters,
                                                          uint64_t *__restrict__ stamps, uint32_t *__restrict__ sink)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = (uint64_t)gid * 0x9E3779B97F4A7C15ull + 777;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint4 v = tab[(uint64_t)(((__uint128_t)s * n_units) >> 64)];
        s ^= ((uint64_t)v.x << 32) | v.y;   // the next address waits for this quad
        acc += v.w;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    if (acc == 0x12345678u) sink[0] = acc;
}

hipError_t launch_calib_chase(const uint4 *tab, uint64_t n_units, int iters, uint64_t *stamps, uint32_t *sink, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(calib_chase_kernel, dim3((unsigned)grid), dim3(64), 0, stream, tab, n_units, iters, stamps, sink);
    return hipGetLastError();
}

hipError_t launch_calib_gather(bool mix, const uint4 *tab, uint64_t n_units, int iters, uint32_t *atom, uint64_t n_atom,
                               uint64_t *stamps, uint32_t *sink, int grid, hipStream_t stream)
{
    if (mix) hipLaunchKernelGGL(calib_gather_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, tab, n_units, iters, atom, n_atom, stamps, sink);
    else hipLaunchKernelGGL(calib_gather_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, tab, n_units, iters, atom, n_atom, stamps, sink);
    return hipGetLastError();
}

hipError_t launch_accumulate(uint64_t *dst64, const uint64_t *src64, uint64_t n64, uint32_t *dst32,
                             const uint32_t *src32, uint64_t n32, hipStream_t stream)
{
    const uint64_t n = n64 > n32 ? n64 : n32;
    if (n == 0) return hipSuccess;
    uint64_t grid = (n + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)grid), dim3(256), 0, stream, dst64, src64, n64, dst32, src32, n32);
    return hipGetLastError();
}

namespace {

// ---- what is launched -----------------------------------------------------------------------------------------
// One fast-kernel instantiation = (reads per sub-tile, fixed shape).  The occupancy questions behind the choice
// cost HIP runtime calls (hipFuncSetAttribute + hipOccupancyMaxActiveBlocksPerMultiprocessor: tens of
// microseconds, 24 x per host-fed query when asked per chunk), so their answers are kept per (device, variant,
// LDS bytes): the first launch of a shape pays, later launches look up.
enum Variant { kV8 = 0, kV4 = 1, kV8h26r100 = 2, kV8h26r150 = 3, kV8h26r100m18 = 4, kV8h26r150m18 = 5,
               // h = 26, m and the row stride fixed, lengths at run time: batches of 97..112 / 113..128 / 145..160 bases
               kV8h26s7 = 6, kV8h26s8 = 7, kV8h26s10 = 8, kV8h26s7m18 = 9, kV8h26s8m18 = 10, kV8h26s10m18 = 11,
               kVSlow = 12, kNVariants = 13 };

template <bool T>
const void *variant_fn_t(int v)
{
    switch (v) {
    case kV8: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 0, 0, 0, T>;
    case kV4: return (const void *)classify_kernel<4, kFastCAP, false, 0, 0, 0, T>;
    case kV8h26r100: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, 100, 16, T>;
    case kV8h26r150: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, 150, 16, T>;
    case kV8h26r100m18: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, 100, 18, T>;
    case kV8h26r150m18: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, 150, 18, T>;
    case kV8h26s7: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, -7, 16, T>;
    case kV8h26s8: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, -8, 16, T>;
    case kV8h26s10: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, -10, 16, T>;
    case kV8h26s7m18: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, -7, 18, T>;
    case kV8h26s8m18: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, -8, 18, T>;
    case kV8h26s10m18: return (const void *)classify_kernel<CQ_BIG_R, kFastCAP, false, 26, -10, 18, T>;
    default: return (const void *)classify_kernel<kSlowR, kSlowCAP, true, 0, 0, 0>;   // the exact path reads tight rows byte by byte at run time
    }
}

// v in [0, kNVariants): rows as words; v + kNVariants: the same instantiation reading tight rows (host-fed door)
const void *variant_fn(int v) { return v >= kNVariants ? variant_fn_t<true>(v - kNVariants) : variant_fn_t<false>(v); }

struct OccKey { int dev, variant; size_t smem; };
struct OccEnt { OccKey k; int blocks; };
std::mutex g_occ_mu;
std::vector<OccEnt> g_occ;
std::atomic<bool> g_attr_done[64][2 * kNVariants];   // hipFuncSetAttribute(160 KB of LDS) once per device and kernel (set from several host threads: cq_multi)

hipError_t ensure_lds_optin(int dev, int v)
{
    if (dev >= 0 && dev < 64 && g_attr_done[dev][v].load(std::memory_order_acquire)) return hipSuccess;
    // LDS above the 64 KiB default needs an explicit opt-in (large G, long reads, the slow path's hit lists)
    hipError_t e = hipFuncSetAttribute(variant_fn(v), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && dev >= 0 && dev < 64) g_attr_done[dev][v].store(true, std::memory_order_release);
    return e;
}

// Resident workgroups per CU of a fast variant with `sm` bytes of LDS.
hipError_t fast_resident(int dev, int v, size_t sm, int &n)
{
    std::lock_guard<std::mutex> lk(g_occ_mu);
    for (const OccEnt &e : g_occ)
        if (e.k.dev == dev && e.k.variant == v && e.k.smem == sm) { n = e.blocks; return hipSuccess; }
    hipError_t e = ensure_lds_optin(dev, v);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, variant_fn(v), kBlock, sm);
    if (e != hipSuccess) return e;
    // The 160 KB are handed out in 1280-byte granules (measured: 5 x 32292 B = 161460 B fit by arithmetic and by the
    // occupancy query, yet only four such workgroups were resident; 5 x 31524 B were): a sub-tile or a histogram
    // that just fits can still cost a workgroup, so count in granules.
    const int by_granules = (int)((160u * 1024u) / ((sm + 1279u) / 1280u * 1280u));
    if (n > by_granules) n = by_granules;
    if (n > CQ_MAX_BLOCKS_PER_CU) n = CQ_MAX_BLOCKS_PER_CU;
    g_occ.push_back(OccEnt{OccKey{dev, v, sm}, n});
    return hipSuccess;
}

// every instantiation has the one signature (DevIndex, QueryArgs): launched through its pointer
void launch_one(int variant, const DevIndex &ix, const QueryArgs &a, unsigned grid, size_t sm, hipStream_t stream)
{
    DevIndex ixc = ix;
    QueryArgs ac = a;
    void *args[2] = {&ixc, &ac};
    (void)hipLaunchKernel(variant_fn(variant), dim3(grid), dim3(kBlock), args, sm, stream);
}

hipError_t launch_fast(int variant, const DevIndex &ix, QueryArgs &a, int n_cus, int per_cu, hipStream_t stream)
{
    const int R = (variant % kNVariants) == kV4 ? 4 : CQ_BIG_R;
    const size_t sm = smem_bytes(R, kFastCAP, a, a.use_lds_hist);
    if (const char *v = getenv("CAMMIQ_BLOCKS_PER_CU")) per_cu = atoi(v);   // tuning knob
    if (per_cu < 1) per_cu = 1;
    const uint64_t grid_full = (uint64_t)n_cus * per_cu;
    // The LDS histogram keeps cnt_u and cnt_d of a genome in the two halves of one word; one read adds at most
    // 1 to a genome's cnt_u and at most 2 to its cnt_d (a pair (g, g)), so a workgroup must not classify more than
    // 32767 reads per launch: a wave takes at most max_sub sub-tiles, longer inputs are cut into several
    // launches (configs[2]'s 50 M reads per launch still fit one: 1536 workgroups x 32736 reads).
    uint64_t max_sub = kMaxReadsPerGroup / (uint64_t)(kWaves * R);
    if (const char *v = getenv("CAMMIQ_MAX_SUB_PER_WAVE")) max_sub = (uint64_t)atoi(v) >= 1 && (uint64_t)atoi(v) < max_sub ? (uint64_t)atoi(v) : max_sub;   // test knob
    const uint64_t chunk = a.use_lds_hist ? grid_full * kWaves * max_sub * R : a.n_reads;
    const uint32_t *packed0 = a.packed;
    uint32_t *const work_counter0 = a.work_counter;
    const uint8_t *tight0 = a.tight;
    const uint8_t *lens0 = a.lens;
    const uint64_t n_total = a.n_reads;
    hipError_t e = hipSuccess;
    for (uint64_t c0 = 0; c0 < n_total; c0 += chunk) {
        a.read0 = c0;
        a.n_reads = n_total - c0 < chunk ? n_total - c0 : chunk;
        a.packed = packed0 ? packed0 + c0 * a.stride_words : nullptr;
        a.tight = tight0 ? tight0 + c0 * a.tight_sb : nullptr;
        a.lens = lens0 + c0;
        const uint64_t n_sub = (a.n_reads + R - 1) / R;
        uint64_t grid = grid_full;
        const uint64_t need = (n_sub + kWaves - 1) / kWaves;
        if (grid > need) grid = need;
        if (grid == 0) grid = 1;
        // the dynamic tail, unless the LDS histogram's bound on what ONE wave may take (max_sub) could leave a stripe's
        // sub-tiles unfinished: every wave of a stripe can still take max_sub - static_rounds of them, so the stripe with the
        // fewest waves must have room for the largest stripe's share (the kernel's arithmetic, repeated here).  configs[2]'s
        // 50 M-read launch: 1 017 of 1 023 per wave, 762 by stride, 96 waves x 261 = 25 056 >= 24 504 per stripe: on.
        a.max_sub = a.use_lds_hist ? (uint32_t)max_sub : 0u;
        a.static_16ths = kStaticSixteenths;
        if (const char *v = getenv("CAMMIQ_STATIC_16THS")) a.static_16ths = (uint32_t)std::min(16, std::max(0, atoi(v)));   // tuning knob
        uint32_t *wc = work_counter0;
        if (wc && a.use_lds_hist) {
            const uint64_t n_waves = grid * kWaves, stripes = n_waves < kWorkStripes ? n_waves : kWorkStripes;
            const uint64_t static_rounds = (n_sub / n_waves) * a.static_16ths / 16, n_dyn = n_sub - static_rounds * n_waves;
            const uint64_t demand = (n_dyn + stripes - 1) / stripes, room = (n_waves / stripes) * (max_sub > static_rounds ? max_sub - static_rounds : 0);
            if (room < demand) wc = nullptr;
        }
        a.work_counter = wc;
        if (wc) { e = hipMemsetAsync(wc, 0, (size_t)kWorkStripes * kWorkStripeWords * sizeof(uint32_t), stream); if (e != hipSuccess) break; }
        launch_one(variant, ix, a, (unsigned)grid, sm, stream);
        e = hipGetLastError();
        if (e != hipSuccess) break;
    }
    a.read0 = 0; a.n_reads = n_total; a.packed = packed0; a.tight = tight0; a.lens = lens0; a.work_counter = work_counter0;
    return e;
}

}  // namespace

hipError_t launch_classify(const DevIndex &ix, QueryArgs a, int n_cus, hipStream_t stream,
                           hipEvent_t ev_start, hipEvent_t ev_mid, hipEvent_t ev_stop, LaunchInfo *info)
{
    a.pmax = a.wmax + (ix.hash_len - ix.minimizer_len);   // m-mer positions: max_len - m + 1
    a.pstride = ((a.pmax + kPrePos - 1u) / kPrePos * kPrePos) | 1u;   // whole position groups (the pre-pass stores unconditionally), odd (banks)
    a.magic_w = magic_of((a.wmax + kWinPerLane - 1u) / kWinPerLane);
    a.magic_p = magic_of(a.pmax);
    a.magic_s = magic_of(a.stride_words);
    a.magic_pp = magic_of((a.pmax + kPrePos - 1u) / kPrePos);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = ensure_lds_optin(dev, kVSlow)) != hipSuccess) return e;
    // Three choices per launch, the first two by what stays resident (six workgroups per CU is what the registers allow):
    //  * reads per sub-tile: eight, unless eight leave three workgroups or fewer and four keep more (rows and hash
    //    words of long reads);
    //  * per-genome counters: an LDS histogram per workgroup when it costs no resident workgroup, global 64-bit
    //    atomics otherwise -- measured: the atomics cost 4 %, a lost workgroup 8 %;
    //  * the instantiation with h = 26 and the batch's shape folded in, when index and batch are that shape.
    const bool hist_ok = lds_hist_fits(a.n_genomes), hist_forced = getenv("CAMMIQ_LDS_HIST_MAX") != nullptr;
    int r8 = 0, r4 = 0, r8h = 0, r4h = 0;
    if ((e = fast_resident(dev, kV8, smem_bytes(CQ_BIG_R, kFastCAP, a, false), r8)) != hipSuccess ||
        (e = fast_resident(dev, kV4, smem_bytes(4, kFastCAP, a, false), r4)) != hipSuccess) return e;
    if (hist_ok && ((e = fast_resident(dev, kV8, smem_bytes(CQ_BIG_R, kFastCAP, a, true), r8h)) != hipSuccess ||
                    (e = fast_resident(dev, kV4, smem_bytes(4, kFastCAP, a, true), r4h)) != hipSuccess)) return e;
    const int tv = a.tight ? kNVariants : 0;   // the instantiations that read tight rows
    int R = (r8 <= 3 && r4 > r8) ? 4 : CQ_BIG_R;   // measured: four reads per sub-tile cost 14 % at equal residency and 8 % at 6 against 5 workgroups (150 bp), and win 13 % at 6 against 3 (250 bp)
    if (const char *v = getenv("CAMMIQ_FAST_R")) R = atoi(v) == 4 ? 4 : CQ_BIG_R;   // tuning knob
    const int plain = R == CQ_BIG_R ? r8 : r4, with = R == CQ_BIG_R ? r8h : r4h;
    a.use_lds_hist = hist_ok && (hist_forced || with >= plain) ? 1 : 0;
    int per_cu = a.use_lds_hist ? with : plain;
    int variant = R == 4 ? kV4 : kV8;
    const char *nofix = getenv("CAMMIQ_NO_FIXED_SHAPE");   // test / A-B knob: always the generic instantiation
    if (R == CQ_BIG_R && ix.hash_len == 26 && (ix.minimizer_len == 16 || ix.minimizer_len == 18) && !(nofix && atoi(nofix))) {
        int fx = -1;
        const bool m18 = ix.minimizer_len == 18;
        // any batch of a common row stride keeps h, m and the stride as constants (101-, 125-, 151-bp reads: what real
        // FASTQs hold); the two benchmark lengths have their whole shape folded in
        const char *noshape = getenv("CAMMIQ_NO_STRIDE_SHAPE");   // A/B knob: only the exact-length instantiations
        if (!(noshape && atoi(noshape))) {
            if (a.stride_words == 7) fx = m18 ? kV8h26s7m18 : kV8h26s7;
            if (a.stride_words == 8) fx = m18 ? kV8h26s8m18 : kV8h26s8;
            if (a.stride_words == 10) fx = m18 ? kV8h26s10m18 : kV8h26s10;
        }
        if (a.wmax == 100 - 26 + 1 && a.stride_words == 7) fx = m18 ? kV8h26r100m18 : kV8h26r100;
        if (a.wmax == 150 - 26 + 1 && a.stride_words == 10) fx = m18 ? kV8h26r150m18 : kV8h26r150;
        if (fx >= 0) {
            int n = 0;   // same LDS layout as the generic kernel of this shape; its own register count
            if ((e = fast_resident(dev, fx + tv, smem_bytes(CQ_BIG_R, kFastCAP, a, a.use_lds_hist != 0), n)) != hipSuccess) return e;
            if (n >= per_cu) { variant = fx; per_cu = n; }
        }
    }
    if (info) { info->reads_per_subtile = R; info->hit_slots = kFastCAP; info->lds_hist = (int)a.use_lds_hist; info->fixed_shape = (variant >= kV8h26r100 && variant != kVSlow) ? 1 : 0;
                info->fixed_h = (variant >= kV8h26r100 && variant != kVSlow) ? 26 : 0;
                // the template's RL argument: the exact read length, or minus the row stride in words when only that is fixed
                info->fixed_read_len = (variant == kV8h26r100 || variant == kV8h26r100m18) ? 100 : (variant == kV8h26r150 || variant == kV8h26r150m18) ? 150
                                       : (variant >= kV8h26s7 && variant <= kV8h26s10m18) ? -(int)a.stride_words : 0;
                info->minimizer_len = (int)ix.minimizer_len;
                info->blocks_per_cu = per_cu; }
    if (ev_start) { e = hipEventRecord(ev_start, stream); if (e != hipSuccess) return e; }
    if (tv) {   // the tight build of the chosen instantiation has its own register count: ask once, keep what stays resident
        int n = 0;
        if ((e = fast_resident(dev, variant + tv, smem_bytes(R, kFastCAP, a, a.use_lds_hist != 0), n)) != hipSuccess) return e;
        if (n < per_cu) per_cu = n;
        if (info) info->blocks_per_cu = per_cu;
    }
    e = launch_fast(variant + tv, ix, a, n_cus, per_cu, stream);
    if (e != hipSuccess) return e;
    if (ev_mid) { e = hipEventRecord(ev_mid, stream); if (e != hipSuccess) return e; }
    // exact slow path for reads with more than kFastCAP hits (usually none: the kernel reads the count from
    // device memory and exits at once).  One read per wave; enough workgroups that none takes more than 32767.
    {
        const size_t sm = smem_bytes(kSlowR, kSlowCAP, a, a.use_lds_hist);
        uint64_t grid = (uint64_t)n_cus;
        const uint64_t need = a.n_reads / 32000 + 1;
        if (a.use_lds_hist && grid < need) grid = need;
        hipLaunchKernelGGL((classify_kernel<kSlowR, kSlowCAP, true, 0, 0, 0>), dim3((unsigned)grid), dim3(kBlock), sm, stream, ix, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (ev_stop) e = hipEventRecord(ev_stop, stream);
    }
    return e;
}

}  // namespace cq
