// cq_layout_gpu.hip -- the table of the flat image, laid out ON THE DEVICE.
//
// Replaces stage 2 of the host layout (cq_layout.cpp finish_image_host: home buckets, sort by (home, key), merge of
// duplicates, placement sweep) for handles that live on a GPU; the reference's counterpart is the insertion of every
// bucket into robin_hood::unordered_map<uint64_t, trieNode*> by Hash::loadIdx64_p
// (/root/reference/src/hashtrie.cpp:486-507).  The result is BYTE-IDENTICAL to the host builder's image (the host
// builder stays the reference: CAMMIQ_GPU_LAYOUT=verify builds both and compares; tests/test_gpu_layout.py), so nothing
// the kernels or the tests know about the table changes.  What changes is where 1.26e9 keys are hashed and ordered:
// the host needed 22.6 s and 110 GB for it at configs[4]'s size, the GPU next to it has the bandwidth to do it in
// about a second, and the 80 GB table image never exists on the host.
//
// Input (uploaded by the caller): keys[n] (uint64, the reference's map64 keys), vals[n] (uint32, final trie code of each
// entry in ITS table); entries [0, nb_u) are ht_u's buckets in file order, [nb_u, n) ht_d's.  leaf_rids as the kernels
// read it.  Steps -- integer, bandwidth / atomic bound, no sort of the whole array:
//   1. home[i] = cq_home_bucket(key)  (the minimizer scan: the expensive part on the host), cnt[home]++
//   2. start = exclusive scan of cnt                                   (three-level block scan, below)
//   3. counting-sort scatter: rec[start[home] + ticket] = (key, i)     (order inside a home group: arbitrary)
//   4. one thread per home group: insertion sort by (key, i) -- groups hold ~1 key, rarely more than 8 --, merge equal
//      keys exactly as the host does (per table the entry latest in file order wins, as map64[b] = root overwrites,
//      hashtrie.cpp:500; inline refID rules of cq_device.h), compact in place, ucnt[home] = distinct keys
//   5. ustart = exclusive scan of ucnt: the rank j of every distinct key in (home, key) order
//   6. E[j] = (key, val_u, val_d), hj[j] = home, g[j] = 4 home - j
//   7. the placement sweep in closed form.  The host deals keys in order into 4-slot buckets, never before a key's
//      home; the global slot of key j is s_j = max(4 home_j, s_{j-1} + 1) = j + max_{i <= j} (4 home_i - i):
//      an inclusive prefix MAXIMUM of g.  Bucket b's overflow flag ("something is still waiting after b was filled")
//      is set iff the key in slot 0 of bucket b + 1 is homed at or before b.
//   8. fill the table with the empty pattern, scatter the keys into their slots, set the flags, reduce the statistics.
// A home group larger than kMaxGroup keys (pathological: thousands of keys sharing one minimizer) is left to the host
// builder: the driver returns `unsupported` and the caller falls back.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <string>
#include <vector>

#include "cq_device.h"
#include "cq_layout_gpu.h"

namespace cq {

namespace {

constexpr int kBlock = 256;
constexpr int kItems = 8;                 // per thread in the scans
constexpr int kTile = kBlock * kItems;    // 2048 elements per workgroup
constexpr uint32_t kMaxGroup = 4096;      // keys sharing one home bucket beyond which the host builder takes over

struct Rec { uint64_t key; uint32_t a, b; };   // (key, entry index, -) before the merge; (key, val_u, val_d) after
static_assert(sizeof(Rec) == 16, "one 16-byte record per key");

// ---- step 1 ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) home_count_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint32_t h, uint32_t m,
                                                            uint32_t n_buckets, uint32_t *__restrict__ home, uint32_t *__restrict__ cnt)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t hm = cq_home_bucket(keys[i], h, m, n_buckets);
        home[i] = hm;
        atomicAdd(cnt + hm, 1u);
    }
}

// ---- scans: three phases per level (tile totals -> scan of the totals, recursively -> tiles with their carry-in) -------
struct OpSum { using T = uint32_t; static __device__ __host__ T id() { return 0u; } static __device__ T op(T a, T b) { return a + b; } };
struct OpMax { using T = long long; static __device__ __host__ T id() { return (long long)0x8000000000000000ull; } static __device__ T op(T a, T b) { return a > b ? a : b; } };

template <class Op>
__device__ typename Op::T block_scan_exclusive(typename Op::T v, typename Op::T *lds, typename Op::T &total)
{
    // Hillis-Steele over the workgroup's 256 values (memory-bound kernels: the eight barriers do not show)
    using T = typename Op::T;
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int d = 1; d < kBlock; d <<= 1) {
        const T x = t >= d ? lds[t - d] : Op::id();
        __syncthreads();
        lds[t] = Op::op(x, lds[t]);
        __syncthreads();
    }
    total = lds[kBlock - 1];
    const T ex = t ? lds[t - 1] : Op::id();
    __syncthreads();
    return ex;
}

template <class Op>
__global__ void __launch_bounds__(kBlock) scan_totals_kernel(const typename Op::T *__restrict__ in, uint64_t n, typename Op::T *__restrict__ totals)
{
    using T = typename Op::T;
    __shared__ T lds[kBlock];
    const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kItems;
    T acc = Op::id();
#pragma unroll
    for (int k = 0; k < kItems; k++)
        if (base + k < n) acc = Op::op(acc, in[base + k]);
    T total;
    (void)block_scan_exclusive<Op>(acc, lds, total);
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// data[i] <- scan of in[0..i) (exclusive) or in[0..i] (inclusive), with carry[blockIdx] (the scan of the tile totals,
// exclusive) folded in; carry may be null on the top level (one tile).
template <class Op, bool INCLUSIVE>
__global__ void __launch_bounds__(kBlock) scan_apply_kernel(const typename Op::T *__restrict__ in, typename Op::T *__restrict__ out, uint64_t n,
                                                            const typename Op::T *__restrict__ carry)
{
    using T = typename Op::T;
    __shared__ T lds[kBlock];
    const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kItems;
    T v[kItems];
    T acc = Op::id();
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        v[k] = base + k < n ? in[base + k] : Op::id();
        acc = Op::op(acc, v[k]);
    }
    T total;
    T run = block_scan_exclusive<Op>(acc, lds, total);
    if (carry) run = Op::op(carry[blockIdx.x], run);
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        if (INCLUSIVE) { run = Op::op(run, v[k]); if (base + k < n) out[base + k] = run; }
        else { if (base + k < n) out[base + k] = run; run = Op::op(run, v[k]); }
    }
}

// Scan `n` elements of d_in into d_out (in place allowed).  Returns the reduction of everything in *total (host).
template <class Op, bool INCLUSIVE>
hipError_t device_scan(const typename Op::T *d_in, typename Op::T *d_out, uint64_t n, typename Op::T *total, hipStream_t st)
{
    using T = typename Op::T;
    if (n == 0) { if (total) *total = Op::id(); return hipSuccess; }
    // level sizes
    std::vector<uint64_t> sizes{n};
    while (sizes.back() > (uint64_t)kTile) sizes.push_back((sizes.back() + kTile - 1) / kTile);
    // totals[l] = per-tile totals of level l (size sizes[l+1]); the top level has one tile
    std::vector<T *> tot(sizes.size(), nullptr);
    hipError_t e = hipSuccess;
    auto cleanup = [&] { for (T *p : tot) if (p) (void)hipFree(p); };
    for (size_t l = 0; l + 1 < sizes.size() && e == hipSuccess; l++) e = hipMalloc((void **)&tot[l], sizes[l + 1] * sizeof(T));
    T *d_total = nullptr;
    if (e == hipSuccess) e = hipMalloc((void **)&d_total, sizeof(T));
    if (e != hipSuccess) { cleanup(); if (d_total) (void)hipFree(d_total); return e; }
    // up: tile totals of every level
    const T *src = d_in;
    for (size_t l = 0; l + 1 < sizes.size(); l++) {
        hipLaunchKernelGGL(scan_totals_kernel<Op>, dim3((unsigned)sizes[l + 1]), dim3(kBlock), 0, st, src, sizes[l], tot[l]);
        src = tot[l];
    }
    // the grand total: the top level (<= one tile) reduced by one workgroup
    hipLaunchKernelGGL(scan_totals_kernel<Op>, dim3(1), dim3(kBlock), 0, st, src, sizes.back(), d_total);
    // down: exclusive scan of every totals array in place, carrying the level above
    for (size_t l = sizes.size() - 1; l-- > 0;) {
        const T *carry = (l + 1 < sizes.size() - 1) ? tot[l + 1] : nullptr;   // tot[l] has sizes[l+1] elements; its tiles' carries are tot[l+1]
        hipLaunchKernelGGL((scan_apply_kernel<Op, false>), dim3((unsigned)((sizes[l + 1] + kTile - 1) / kTile)), dim3(kBlock), 0, st,
                           tot[l], tot[l], sizes[l + 1], carry);
    }
    hipLaunchKernelGGL((scan_apply_kernel<Op, INCLUSIVE>), dim3((unsigned)((n + kTile - 1) / kTile)), dim3(kBlock), 0, st, d_in, d_out, n,
                       sizes.size() > 1 ? tot[0] : nullptr);
    e = hipGetLastError();
    if (e == hipSuccess && total) e = hipMemcpyAsync(total, d_total, sizeof(T), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    (void)hipFree(d_total);
    return e;
}

// ---- step 3 ------------------------------------------------------------------------------------------------------
// cnt is consumed as the ticket counter (it counts down to zero); the group sizes stay available as start[b+1] - start[b].
__global__ void __launch_bounds__(kBlock) scatter_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ home, uint64_t n,
                                                         const uint32_t *__restrict__ start, uint32_t *__restrict__ cnt, Rec *__restrict__ rec)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t hm = home[i];
        const uint32_t t = atomicSub(cnt + hm, 1u) - 1u;
        rec[(uint64_t)start[hm] + t] = Rec{keys[i], (uint32_t)i, 0u};
    }
}

// ---- step 4 ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) canon_kernel(Rec *__restrict__ rec, const uint32_t *__restrict__ start, uint32_t n_buckets, uint64_t n,
                                                       const uint32_t *__restrict__ vals, uint32_t nb_u, const uint2 *__restrict__ leaf_rids,
                                                       uint32_t *__restrict__ ucnt, uint32_t *__restrict__ too_big)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += stride) {
        const uint64_t lo = start[b], hi = b + 1 < n_buckets ? (uint64_t)start[b + 1] : n;
        const uint32_t cnt = (uint32_t)(hi - lo);
        if (cnt == 0) { ucnt[b] = 0; continue; }
        if (cnt > kMaxGroup) { atomicMax(too_big, cnt); ucnt[b] = 0; continue; }
        Rec *g = rec + lo;
        // insertion sort by (key, entry index): the index is the file position (u entries before d entries)
        for (uint32_t i = 1; i < cnt; i++) {
            const Rec x = g[i];
            uint32_t j = i;
            while (j > 0 && (g[j - 1].key > x.key || (g[j - 1].key == x.key && g[j - 1].a > x.a))) { g[j] = g[j - 1]; j--; }
            g[j] = x;
        }
        // merge runs of equal keys; per table the entry with the highest file position that HAS a code wins
        uint32_t w = 0;
        for (uint32_t i = 0; i < cnt;) {
            const uint64_t key = g[i].key;
            uint32_t vu = 0, vd = 0, j = i;
            for (; j < cnt && g[j].key == key; j++) {
                const uint32_t idx = g[j].a, v = vals[idx];
                if (v) { if (idx < nb_u) vu = v; else vd = v; }
            }
            // a unique depth-0 leaf with a free val_d slot carries its refID inline (cq_device.h) ...
            if ((vu & CQ_LEAF_BIT) && vd == 0) {
                const uint2 r = leaf_rids[vu & ~CQ_LEAF_BIT];
                if (r.y == 0 && r.x < CQ_INLINE_RID_BIT) vd = CQ_INLINE_RID_BIT | r.x;
            }
            // ... likewise a depth-0 leaf of ht_d alone: its two refIDs ride in the free val_u word
            if ((vd & CQ_LEAF_BIT) && vu == 0) {
                const uint2 r = leaf_rids[vd & ~CQ_LEAF_BIT];
                if (r.x < (1u << 15) && r.y < (1u << 15)) vu = CQ_INLINE_PAIR_BIT | (r.x << 15) | r.y;
            }
            g[w++] = Rec{key, vu, vd};
            i = j;
        }
        ucnt[b] = w;
    }
}

// ---- step 6 ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) compact_kernel(const Rec *__restrict__ rec, const uint32_t *__restrict__ start, const uint32_t *__restrict__ ucnt,
                                                         const uint32_t *__restrict__ ustart, uint32_t n_buckets, Rec *__restrict__ E,
                                                         uint32_t *__restrict__ hj, long long *__restrict__ g)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += stride) {
        const uint32_t c = ucnt[b];
        const uint64_t src = start[b], dst = ustart[b];
        for (uint32_t q = 0; q < c; q++) {
            const uint64_t j = dst + q;
            E[j] = rec[src + q];
            hj[j] = (uint32_t)b;
            g[j] = 4ll * (long long)b - (long long)j;
        }
    }
}

// ---- step 8 ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) empty_table_kernel(uint4 *__restrict__ table, uint64_t n_buckets_alloc)
{
    // one lane per 16-byte quad of a bucket: key_lo | key_hi | val_u | val_d (cq_device.h)
    const uint64_t n_quads = n_buckets_alloc * 4, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += stride) {
        const uint32_t part = (uint32_t)(q & 3u);
        uint4 v;
        if (part == 0) v = make_uint4(0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);   // empty slot 0: the overflow flag (bit 0) reads 0
        else if (part == 1) v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        else v = make_uint4(0u, 0u, 0u, 0u);
        table[q] = v;
    }
}

__global__ void __launch_bounds__(kBlock) place_kernel(const Rec *__restrict__ E, const uint32_t *__restrict__ hj, const long long *__restrict__ M, uint64_t n,
                                                       uint32_t *__restrict__ table, unsigned long long *__restrict__ stats /* [0] overflowed, [1] max chain */)
{
    // The key in slot 0 of a bucket writes the WHOLE bucket -- its own slot and those of the up to three keys that follow
    // it into the same bucket (consecutive ranks, consecutive slots), empty pattern in the rest: four 16-byte stores of
    // one full 64-byte line instead of sixteen 4-byte stores from four lanes into a line the fill kernel has just written
    // (measured at 1.26e9 keys: the per-key scatter took ~1 s).  Keys in slots 1..3 only contribute their chain length.
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long my_ovf = 0;
    uint32_t my_chain = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint64_t s = (uint64_t)((long long)j + M[j]);
        const uint64_t b = s >> 2;
        const uint32_t chain = (uint32_t)(b - hj[j]) + 1u;
        my_chain = chain > my_chain ? chain : my_chain;
        if (s & 3u) continue;
        uint32_t lo[4] = {0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        uint32_t vu[4] = {0u, 0u, 0u, 0u}, vd[4] = {0u, 0u, 0u, 0u};
        uint32_t k = 0;
        for (; k < 4 && j + k < n && (uint64_t)((long long)(j + k) + M[j + k]) == s + k; k++) {
            const Rec e = E[j + k];
            lo[k] = (uint32_t)e.key;
            hi[k] = (uint32_t)(e.key >> 32);
            vu[k] = e.a;
            vd[k] = e.b;
        }
        // slot 0 lends bit 0 of key_lo to the overflow flag (its true bit 0 moves to bit 30 of key_hi)
        if (lo[0] & 1u) hi[0] |= CQ_SLOT0_BIT0_IN_HI;
        lo[0] &= ~1u;
        // the bucket overflowed iff it is full and the key after it (slot 0 of bucket b + 1) is homed at or before b
        if (k == 4 && j + 4 < n && (uint64_t)((long long)(j + 4) + M[j + 4]) == s + 4 && (uint64_t)hj[j + 4] <= b) { lo[0] |= 1u; my_ovf++; }
        uint4 *bw = (uint4 *)(table + b * CQ_BUCKET_WORDS);
        bw[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        bw[1] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        bw[2] = make_uint4(vu[0], vu[1], vu[2], vu[3]);
        bw[3] = make_uint4(vd[0], vd[1], vd[2], vd[3]);
    }
    if (my_ovf) atomicAdd(stats, my_ovf);
    if (my_chain) atomicMax(stats + 1, (unsigned long long)my_chain);
}

struct Timer {
    bool on = getenv("CAMMIQ_LOAD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[device layout] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

unsigned grid_for(uint64_t n, int n_cus)
{
    uint64_t g = (n + kBlock - 1) / kBlock;
    const uint64_t cap = (uint64_t)(n_cus > 0 ? n_cus : 256) * 16;
    return (unsigned)std::max<uint64_t>(1, std::min(g, cap));
}

}  // namespace

hipError_t layout_table_on_device(const uint64_t *d_keys, const uint32_t *d_vals, uint64_t nb_u, uint64_t nb_d, uint32_t h, uint32_t m,
                                  uint32_t n_buckets, const uint2 *d_leaf_rids, int n_cus, DeviceLayoutResult &out, bool &unsupported,
                                  void *prealloc, uint64_t prealloc_buckets)
{
    unsupported = false;
    out = DeviceLayoutResult();
    const uint64_t n = nb_u + nb_d;
    hipStream_t st = nullptr;   // the null stream: the caller is between uploads, nothing else runs
    uint32_t *home = nullptr, *cnt = nullptr, *start = nullptr, *ustart = nullptr, *hj = nullptr, *too_big = nullptr;
    Rec *rec = nullptr, *E = nullptr;
    long long *g = nullptr;
    unsigned long long *stats = nullptr;
    void *table = nullptr;
    Timer tm;
    hipError_t e = hipSuccess;
#define LG(call) do { e = (call); if (e != hipSuccess) goto done; } while (0)
    {
        LG(hipMalloc((void **)&home, std::max<uint64_t>(n, 1) * 4));
        LG(hipMalloc((void **)&cnt, ((uint64_t)n_buckets + 1) * 4));
        LG(hipMalloc((void **)&start, ((uint64_t)n_buckets + 1) * 4));
        LG(hipMemsetAsync(cnt, 0, ((uint64_t)n_buckets + 1) * 4, st));
        // 1. home buckets + group sizes
        if (n) hipLaunchKernelGGL(home_count_kernel, dim3(grid_for(n, n_cus)), dim3(kBlock), 0, st, d_keys, n, h, m, n_buckets, home, cnt);
        LG(hipGetLastError());
        tm.lap("home buckets + counts");
        // 2. group starts
        uint32_t total = 0;
        LG((device_scan<OpSum, false>(cnt, start, n_buckets, &total, st)));
        if ((uint64_t)total != n) { e = hipErrorUnknown; goto done; }
        // 3. scatter into home groups
        tm.lap("  scan (group starts)");
        LG(hipMalloc((void **)&rec, std::max<uint64_t>(n, 1) * sizeof(Rec)));
        tm.lap("  hipMalloc (records)");
        if (n) hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(n, n_cus)), dim3(kBlock), 0, st, d_keys, home, n, start, cnt, rec);
        LG(hipGetLastError());
        LG(hipStreamSynchronize(st));
        tm.lap("  scatter");
        (void)hipFree(home); home = nullptr;
        tm.lap("  hipFree (homes)");
        // 4. sort + merge inside every home group; cnt becomes the distinct-key count of the group
        LG(hipMalloc((void **)&too_big, 4));
        LG(hipMemsetAsync(too_big, 0, 4, st));
        hipLaunchKernelGGL(canon_kernel, dim3(grid_for(n_buckets, n_cus)), dim3(kBlock), 0, st, rec, start, n_buckets, n, d_vals, (uint32_t)nb_u, d_leaf_rids, cnt, too_big);
        LG(hipGetLastError());
        uint32_t big = 0;
        LG(hipMemcpy(&big, too_big, 4, hipMemcpyDeviceToHost));
        if (big) { unsupported = true; goto done; }
        tm.lap("sort + merge groups");
        // 5. ranks of the distinct keys
        LG(hipMalloc((void **)&ustart, ((uint64_t)n_buckets + 1) * 4));
        uint32_t n_keys = 0;
        LG((device_scan<OpSum, false>(cnt, ustart, n_buckets, &n_keys, st)));
        out.n_keys = n_keys;
        // 6. compact + the sweep's scan input
        tm.lap("  scan (ranks)");
        LG(hipMalloc((void **)&E, std::max<uint64_t>(n_keys, 1) * sizeof(Rec)));
        LG(hipMalloc((void **)&hj, std::max<uint64_t>(n_keys, 1) * 4));
        LG(hipMalloc((void **)&g, std::max<uint64_t>(n_keys, 1) * 8));
        tm.lap("  hipMalloc (ranked keys)");
        hipLaunchKernelGGL(compact_kernel, dim3(grid_for(n_buckets, n_cus)), dim3(kBlock), 0, st, rec, start, cnt, ustart, n_buckets, E, hj, g);
        LG(hipGetLastError());
        LG(hipStreamSynchronize(st));
        tm.lap("rank + compact");
        (void)hipFree(rec); rec = nullptr;
        (void)hipFree(start); start = nullptr;
        (void)hipFree(ustart); ustart = nullptr;
        (void)hipFree(cnt); cnt = nullptr;
        tm.lap("  hipFree (records, counts)");
        // 7. the placement sweep: s_j = j + max_{i <= j} (4 home_i - i)
        long long gmax = 0;
        LG((device_scan<OpMax, true>(g, g, n_keys, &gmax, st)));
        uint64_t n_alloc = (uint64_t)n_buckets + CQ_SPILL_TAIL;
        if (n_keys) {   // the last key's slot decides whether the spill tail has to grow (a dense table whose last homes are crowded)
            const uint64_t s_last = (uint64_t)((long long)(n_keys - 1) + gmax);
            n_alloc = std::max<uint64_t>(n_alloc, (s_last >> 2) + 1);
        }
        if (n_alloc >= 0xFFFFFFFFull) { out.limit = true; goto done; }
        tm.lap("prefix maximum");
        // 8. the table
        if (prealloc && prealloc_buckets >= n_alloc) { table = prealloc; prealloc = nullptr; }   // (a larger block than needed is kept: the tail is never addressed)
        else LG(hipMalloc(&table, n_alloc * CQ_BUCKET_WORDS * 4));
        LG(hipMalloc((void **)&stats, 16));
        LG(hipMemsetAsync(stats, 0, 16, st));
        hipLaunchKernelGGL(empty_table_kernel, dim3(grid_for(n_alloc * 4, n_cus)), dim3(kBlock), 0, st, (uint4 *)table, n_alloc);
        if (n_keys) hipLaunchKernelGGL(place_kernel, dim3(grid_for(n_keys, n_cus)), dim3(kBlock), 0, st, E, hj, g, (uint64_t)n_keys, (uint32_t *)table, stats);
        LG(hipGetLastError());
        unsigned long long hs[2] = {0, 0};
        LG(hipMemcpy(hs, stats, 16, hipMemcpyDeviceToHost));
        out.n_overflowed = hs[0];
        out.max_chain = (uint32_t)std::max<unsigned long long>(1, hs[1]);
        out.n_buckets_alloc = n_alloc;
        out.d_table = table;
        table = nullptr;
        tm.lap("fill + place");
    }
done:
#undef LG
    for (void *p : {(void *)home, (void *)cnt, (void *)start, (void *)ustart, (void *)hj, (void *)too_big, (void *)rec, (void *)E, (void *)g, (void *)stats, table, prealloc})
        if (p) (void)hipFree(p);
    return e;
}

}  // namespace cq
