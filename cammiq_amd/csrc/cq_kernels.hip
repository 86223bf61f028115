// cq_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for CAMMiQ's classify step.
//
// What one launch computes is what one call of FqReader::query64_p / query64mt_p /
// query64_sc computes (/root/reference/src/query.cpp:458-648, 650-889, 891-1080) with
// Hash::find64_p (/root/reference/src/hashtrie.cpp:350-369) underneath -- but organised
// for a 64-wide, memory-latency-bound machine instead of a pointer-chasing CPU loop:
//
//   * a workgroup takes a TILE of TR reads; their 2-bit rows are loaded coalesced and
//     staged in LDS (the "sliding window" lives there, not in registers of one thread);
//   * every lane owns ONE window position of one read and handles BOTH strands of it:
//     the forward h-mer is a bit-field of the row, the reverse-complement h-mer is
//     ~bitreverse of it, so no reverse-complement read is ever materialised
//     (reference: getRC + a second scan, query.cpp:447-450,503-527);
//   * each h-mer costs ONE 64-byte bucket read of the merged unique+doubly-unique table
//     (reference: two find64_p calls = two hash lookups, query.cpp:487-492);
//   * lanes that see a bucket hit (a few per cent) do not walk the trie in place: the
//     wave compacts them with ballot + prefix popcount into an LDS work list and drains
//     the list with full lanes, so divergent trie walks do not stall the probe stream;
//   * hits are gathered per read in LDS; one lane per read then de-duplicates them and
//     applies the decision rule; per-genome counters are reduced in LDS and flushed with
//     one global atomic per touched genome per workgroup; rcount uses global atomics.
//
// Integer only, HBM-latency/-bandwidth bound; MFMA is deliberately unused.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cq_device.h"
#include "cq_kernels.h"

namespace cq {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kWorkCap = 320;   // per-wave work list: drain threshold 64 + 4 appends x 64 lanes
constexpr int kWorkDrain = 64;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Reverse the order of the 32 two-bit symbols of x.
__device__ __forceinline__ uint64_t rev2(uint64_t x)
{
    uint64_t y = __brevll(x);
    return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}

// One 64-byte bucket.
struct Bucket { uint4 s[4]; };

__device__ __forceinline__ Bucket load_bucket(const uint4 *__restrict__ slots, uint32_t b)
{
    const uint4 *p = slots + (size_t)b * 4;
    Bucket r;
    r.s[0] = p[0]; r.s[1] = p[1]; r.s[2] = p[2]; r.s[3] = p[3];
    return r;
}

// Compare the four slots with `key`.  Returns true when the chain continues (overflow bit
// set and key not found here).  vals.x = val_u, vals.y = val_d (0,0 when not found).
__device__ __forceinline__ bool match_bucket(const Bucket &bk, uint64_t key, uint2 &vals)
{
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    bool found = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t shi = bk.s[k].y;
        if (k == 0) shi &= ~(1u << 30);  // strip CQ_OVERFLOW_BIT; an empty slot keeps bit 31
        if (bk.s[k].x == klo && shi == khi) { vals.x = bk.s[k].z; vals.y = bk.s[k].w; found = true; }
    }
    // overflow: bit 62 set, bit 63 clear on slot 0 (an EMPTY key has both set)
    return !found && ((bk.s[0].y >> 30) == 1u);
}

// Base q of a staged row (A=0..T=3).
__device__ __forceinline__ uint32_t row_base(const uint32_t *row, uint32_t q)
{
    return (row[q >> 4] >> (30u - 2u * (q & 15u))) & 3u;
}

// Word offsets of the per-workgroup LDS regions (shared by the kernel and its launcher).
struct SmemLayout { uint32_t rows, len, hitcnt, hit_gid, hit_r1, hit_r2, scal, work, hist, total; };

__host__ __device__ inline SmemLayout smem_layout(int TR, int CAP, uint32_t sw, uint32_t n_genomes, bool hist)
{
    SmemLayout L;
    uint32_t o = 0;
    L.rows = o;    o += (uint32_t)TR * (sw + 2);
    L.len = o;     o += (uint32_t)TR;
    L.hitcnt = o;  o += (uint32_t)TR;
    L.hit_gid = o; o += (uint32_t)TR * (uint32_t)CAP;
    L.hit_r1 = o;  o += (uint32_t)TR * (uint32_t)CAP;
    L.hit_r2 = o;  o += (uint32_t)TR * (uint32_t)CAP;
    L.scal = o;    o += 4;
    o += (o & 1u);                       // 8-byte align the uint2 work list
    L.work = o;    o += 2u * kWaves * kWorkCap;
    L.hist = o;    if (hist) o += 2u * (n_genomes + 1u);
    L.total = o;
    return L;
}

struct Tile {
    uint32_t *rows;      // [TR][swp]
    uint32_t *len;       // [TR]   read length (0 = skip)
    uint32_t *hitcnt;    // [TR]
    uint32_t *hit_gid;   // [TR][CAP]
    uint32_t *hit_r1;    // [TR][CAP]
    uint32_t *hit_r2;    // [TR][CAP]
    uint2 *work;         // [kWaves][kWorkCap]  .x = trie code, .y = read | strand<<8 | pos<<9
    uint32_t *hist;      // [2*(G+1)] or null
    uint32_t *scal;      // [4] nundet nconf nskipped nslow
};

// hashtrie.cpp:350-369 on the array trie.  `code` is the bucket root; returns the global
// leaf id or 0xFFFFFFFF.  Forward strand consumes bases p+h, p+h+1, ...; the reverse strand
// consumes the complements of p-1, p-2, ... (that IS rc_read[i+h+j], query.cpp:447-450).
__device__ __forceinline__ uint32_t walk_trie(const DevIndex &ix, const uint32_t *row, uint32_t len,
                                              uint32_t code, uint32_t strand, uint32_t p)
{
    const uint32_t h = ix.hash_len;
    const uint32_t rem = strand ? p : (len - h - p);
    uint32_t j = 0;
    for (;;) {
        if (code & CQ_LEAF_BIT) return code & ~CQ_LEAF_BIT;   // cur->isEnd
        if (j == rem) return 0xFFFFFFFFu;                       // read exhausted on an inner node
        uint32_t sym = strand ? (3u - row_base(row, p - 1u - j)) : row_base(row, p + h + j);
        uint4 n = ix.nodes[code];
        uint32_t c = sym == 0 ? n.x : sym == 1 ? n.y : sym == 2 ? n.z : n.w;
        if (c == 0) return 0xFFFFFFFFu;                          // children[index] == NULL
        code = c;
        j++;
    }
}

// Orders this wave's LDS traffic (hardware executes one wave's LDS ops in order; this keeps
// the compiler from moving them across the hand-off between lanes).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Resolve the n items of this wave's work list with all 64 lanes: walk the trie (most codes
// are depth-0 leaves already), fetch the leaf's refIDs, append to the read's hit list.
template <int CAP>
__device__ __forceinline__ void drain_work(const DevIndex &ix, const Tile &t, uint32_t swp, uint32_t wave, uint32_t n)
{
    wave_sync();
    const uint2 *wl = t.work + wave * kWorkCap;
    for (uint32_t i = lane_id(); i < n; i += 64) {
        uint2 it = wl[i];
        uint32_t rl = it.y & 255u, strand = (it.y >> 8) & 1u, p = it.y >> 9;
        uint32_t gid = walk_trie(ix, t.rows + rl * swp, t.len[rl], it.x, strand, p);
        if (gid != 0xFFFFFFFFu) {
            uint2 rr = ix.leaf_rids[gid];
            uint32_t k = atomicAdd(&t.hitcnt[rl], 1u);
            if (k < (uint32_t)CAP) {
                t.hit_gid[rl * CAP + k] = gid;
                t.hit_r1[rl * CAP + k] = rr.x;
                t.hit_r2[rl * CAP + k] = rr.y;
            }
        }
    }
    wave_sync();
}

// Wave-level compaction: every lane with `have` appends one item.  The list length `nw` is
// wave-uniform and lives in a register (ballot gives every lane the same mask), so the
// append needs no atomic at all.
__device__ __forceinline__ void push_work(const Tile &t, uint32_t wave, uint32_t &nw, bool have, uint32_t code, uint32_t meta)
{
    const uint64_t m = __ballot(have);
    if (m == 0) return;
    const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    if (have) t.work[wave * kWorkCap + nw + off] = make_uint2(code, meta);
    nw += (uint32_t)__popcll(m);
}

__device__ __forceinline__ void add_cnt(const QueryArgs &a, const Tile &t, uint32_t which, uint32_t rid)
{
    // which: 0 = cnt_u, 1 = cnt_d
    if (rid > a.n_genomes) return;  // guarded on the host (CQ_ERR_RANGE); never index out of bounds
    uint32_t i = which * (a.n_genomes + 1) + rid;
    if (t.hist) atomicAdd(&t.hist[i], 1u);
    else atomicAdd((unsigned long long *)&a.counters[i], 1ull);
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
    const uint32_t *gid = t.hit_gid + rl * CAP, *r1 = t.hit_r1 + rl * CAP, *r2 = t.hit_r2 + rl * CAP;
    uint32_t nU = 0, u0 = 0, nP = 0, pa = 0, pb = 0, ia = 0, ib = 0;
    bool va = false, vb = false;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t g = gid[i];
        bool dup = false;
        for (uint32_t j = 0; j < i; j++) dup |= (gid[j] == g);   // pnodes is a set (query.cpp:465)
        if (dup) continue;
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
    if (counted && a.mode == CQ_MODE_P && a.rcount) {
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t g = gid[i];
            bool dup = false;
            for (uint32_t j = 0; j < i; j++) dup |= (gid[j] == g);
            if (!dup) atomicAdd(&a.rcount[g], 1u);               // pn->rcount += 1
        }
    }
}

}  // namespace

// TR reads per tile, CAP hit slots per read.  SLOW = exact path for reads whose hit list
// overflowed CAP in the fast kernel: reads come from a device-side list, CAP covers the
// worst case (2 strands x 2 tables x 251 windows).
template <int TR, int CAP, bool SLOW>
__global__ void __launch_bounds__(kBlock) classify_kernel(DevIndex ix, QueryArgs a)
{
    extern __shared__ __align__(16) uint32_t smem[];
    const uint32_t sw = a.stride_words, swp = sw + 2;
    const uint32_t G1 = a.n_genomes + 1;
    Tile t;
    const SmemLayout L = smem_layout(TR, CAP, sw, a.n_genomes, a.use_lds_hist != 0);
    t.rows = smem + L.rows;
    t.len = smem + L.len;
    t.hitcnt = smem + L.hitcnt;
    t.hit_gid = smem + L.hit_gid;
    t.hit_r1 = smem + L.hit_r1;
    t.hit_r2 = smem + L.hit_r2;
    t.scal = smem + L.scal;
    t.work = (uint2 *)(smem + L.work);
    t.hist = a.use_lds_hist ? smem + L.hist : nullptr;

    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const uint32_t h = ix.hash_len;

    if (tid < 4) t.scal[tid] = 0;
    if (t.hist) for (uint32_t i = tid; i < 2 * G1; i += kBlock) t.hist[i] = 0;

    uint64_t n_reads = a.n_reads;
    if (SLOW) n_reads = *a.ovf_count < a.ovf_cap ? *a.ovf_count : a.ovf_cap;
    const uint64_t n_tiles = (n_reads + TR - 1) / TR;
    const uint32_t wmax = a.wmax;

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t r0 = tile * TR;
        const uint32_t nr = (uint32_t)((n_reads - r0) < (uint64_t)TR ? (n_reads - r0) : (uint64_t)TR);
        __syncthreads();   // previous tile fully consumed (and the prologue's zeroing visible)

        // ---- stage the tile: 2-bit rows -> LDS, coalesced (16 B per lane when direct)
        if (!SLOW) {
            const uint4 *src = (const uint4 *)(a.packed + r0 * sw);
            const uint32_t nvec = nr * (sw >> 2);
            for (uint32_t i = tid; i < nvec; i += kBlock) {
                uint4 v = src[i];
                uint32_t w = i * 4, rl = w / sw, c = w - rl * sw;
                uint32_t *dst = t.rows + rl * swp + c;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        } else {
            for (uint32_t i = tid; i < nr * sw; i += kBlock) {
                uint32_t rl = i / sw, c = i - rl * sw;
                t.rows[rl * swp + c] = a.packed[(uint64_t)a.ovf_list[r0 + rl] * sw + c];
            }
        }
        for (uint32_t i = tid; i < TR; i += kBlock) {
            uint32_t len = 0;
            if (i < nr) len = SLOW ? a.lens[a.ovf_list[r0 + i]] : a.lens[r0 + i];
            t.len[i] = len;
            t.hitcnt[i] = 0;
            t.rows[i * swp + sw] = 0; t.rows[i * swp + sw + 1] = 0;   // pad words read by the window extract
        }
        __syncthreads();

        // ---- probe phase: lane = (read, window), both strands
        const uint32_t total = nr * wmax;
        uint32_t nw = 0;   // wave-uniform length of this wave's work list
        for (uint32_t base = wave * 64; base < total; base += kBlock) {
            const uint32_t idx = base + lane_id();
            bool act = idx < total;
            uint32_t rl = 0, pw = 0, len = 0;
            if (act) {
                rl = idx / wmax; pw = idx - rl * wmax;
                len = t.len[rl];
                act = (len >= h) && (pw + h <= len);
            }
            uint2 vf = make_uint2(0, 0), vr = make_uint2(0, 0);
            if (act) {
                const uint32_t *row = t.rows + rl * swp;
                const uint32_t q = pw >> 4, s = (pw & 15u) * 2u;
                const uint64_t x = ((uint64_t)row[q] << 32) | row[q + 1];
                const uint64_t top = (x << s) | (((uint64_t)row[q + 2] << s) >> 32);
                const uint64_t fw = top >> (64u - 2u * h);
                const uint64_t rc = (~rev2(fw)) >> (64u - 2u * h);
                uint32_t bf = cq_home_bucket(fw, ix.n_buckets), br = cq_home_bucket(rc, ix.n_buckets);
                Bucket kf = load_bucket(ix.slots, bf);
                Bucket kr = load_bucket(ix.slots, br);
                bool cf = match_bucket(kf, fw, vf);
                bool cr = match_bucket(kr, rc, vr);
                while (cf) { kf = load_bucket(ix.slots, ++bf); cf = match_bucket(kf, fw, vf); }
                while (cr) { kr = load_bucket(ix.slots, ++br); cr = match_bucket(kr, rc, vr); }
            }
            // candidates -> wave work list (forward u, forward d, reverse u, reverse d)
            const uint32_t mf = rl | (0u << 8) | (pw << 9), mr = rl | (1u << 8) | (pw << 9);
            push_work(t, wave, nw, vf.x != 0, vf.x, mf);
            push_work(t, wave, nw, vf.y != 0, vf.y, mf);
            push_work(t, wave, nw, vr.x != 0, vr.x, mr);
            push_work(t, wave, nw, vr.y != 0, vr.y, mr);
            if (nw >= (uint32_t)kWorkDrain) { drain_work<CAP>(ix, t, swp, wave, nw); nw = 0; }
        }
        if (nw != 0) drain_work<CAP>(ix, t, swp, wave, nw);
        __syncthreads();

        // ---- decision phase: one lane per read
        for (uint32_t rl = tid; rl < nr; rl += kBlock) {
            const uint32_t len = t.len[rl];
            if (len < h) { atomicAdd(&t.scal[2], 1u); continue; }   // outside the parity domain
            const uint32_t n = t.hitcnt[rl];
            if (n > (uint32_t)CAP) {
                if (!SLOW) {   // hand the read to the exact slow path
                    uint32_t k = atomicAdd(a.ovf_count, 1u);
                    if (k < a.ovf_cap) a.ovf_list[k] = (uint32_t)(r0 + rl);
                    atomicAdd(&t.scal[3], 1u);
                }
                continue;
            }
            decide<CAP>(a, t, rl, n);
        }
    }

    // ---- flush workgroup-level counters
    __syncthreads();
    if (t.hist)
        for (uint32_t i = tid; i < 2 * G1; i += kBlock) {
            uint32_t v = t.hist[i];
            if (v) atomicAdd((unsigned long long *)&a.counters[i], (unsigned long long)v);
        }
    if (tid < 4 && t.scal[tid]) {
        const uint64_t off[4] = {CQ_CTR_NUNDET(a.n_genomes), CQ_CTR_NCONF(a.n_genomes), CQ_CTR_NSKIP(a.n_genomes),
                                 CQ_CTR_NSLOW(a.n_genomes)};
        atomicAdd((unsigned long long *)&a.counters[off[tid]], (unsigned long long)t.scal[tid]);
    }
}

constexpr int kFastTR = 64, kFastCAP = 16;
constexpr int kSlowTR = 4, kSlowCAP = 1024;

static size_t smem_bytes(int TR, int CAP, uint32_t sw, uint32_t n_genomes, bool hist)
{
    return (size_t)smem_layout(TR, CAP, sw, n_genomes, hist).total * 4;
}

bool lds_hist_fits(uint32_t n_genomes)
{
    return 2 * ((size_t)n_genomes + 1) * 4 <= 32 * 1024;
}

hipError_t launch_classify(const DevIndex &ix, QueryArgs a, int n_cus, hipStream_t stream,
                           hipEvent_t ev_start, hipEvent_t ev_stop)
{
    a.use_lds_hist = lds_hist_fits(a.n_genomes) ? 1 : 0;
    hipError_t e;
    // fast kernel
    {
        size_t sm = smem_bytes(kFastTR, kFastCAP, a.stride_words, a.n_genomes, a.use_lds_hist);
        uint64_t n_tiles = (a.n_reads + kFastTR - 1) / kFastTR;
        uint64_t grid = (uint64_t)n_cus * 4;
        if (grid > n_tiles) grid = n_tiles;
        if (grid == 0) grid = 1;
        if (ev_start) { e = hipEventRecord(ev_start, stream); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL((classify_kernel<kFastTR, kFastCAP, false>), dim3((unsigned)grid), dim3(kBlock), sm, stream, ix, a);
        if (ev_stop) { e = hipEventRecord(ev_stop, stream); if (e != hipSuccess) return e; }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // exact slow path for reads with more than kFastCAP hits (usually none: the kernel
    // reads the count from device memory and exits at once)
    {
        size_t sm = smem_bytes(kSlowTR, kSlowCAP, a.stride_words, a.n_genomes, a.use_lds_hist);
        hipLaunchKernelGGL((classify_kernel<kSlowTR, kSlowCAP, true>), dim3((unsigned)(n_cus)), dim3(kBlock), sm, stream, ix, a);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace cq
