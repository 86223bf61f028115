// partition_bench.hip -- prices a bucket-range-partitioned probe before anybody builds it.
//
// Today the classify kernel reads ~13.5 table buckets per 100-bp read at random from a multi-GB table and tracks
// the chip's random-line rate (~42 G lines/s at 5 GiB).  From a 2 MiB table the same loads run at ~250 G/s (L2).
// A partitioned design would (1) emit one tuple per minimizer run into P bucket-range partitions, (2) per partition
// stream its tuples and probe a table slice that fits L2, (3) decide per read from hit lists.  A tuple has to CARRY
// the bases its run covers (h + w - 1 = 36 bases = 72 bits: without them step 2 would fetch the read's row at random
// and nothing would be gained) plus read id (26 bits), position (7) and bucket-in-partition (15): 16 bytes.
//
// This bench measures the two primitives that design adds, at configs[2]'s scale (675 M tuples per 50 M reads;
// run here with N tuples and scaled):
//   scatter  N 16-byte tuples -> P partitions
//              direct : one global atomic cursor per partition, every lane appends its own tuple
//              lds    : a workgroup bins tuples in LDS (P bins x 4 tuples = one 64-byte line per bin) and writes
//                       whole lines; cursors advance by 4 tuples per flush
//   probe    per partition: stream the tuples (coalesced 16-B loads), one 16-byte load per tuple from that
//            partition's slice of a table of T bytes (slice = T / P), compare, count
// Output: GB/s of tuple traffic and G tuples/s for each, from which DESIGN.md section 6 prices the design.
//   hipcc --offload-arch=gfx950 -O3 tools/partition_bench.hip -o tools/partition_bench
//   tools/partition_bench [N_MTUPLES=256] [P=2048] [TABLE_MIB=5120]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// tuple i: .x = bucket within partition | partition << 16 (the "address"), .y .z .w = payload (sequence, read, pos)
__device__ __forceinline__ uint4 make_tuple(uint64_t i, uint32_t P, uint32_t bpp)
{
    const uint64_t r = mix64(i + 0x9E3779B97F4A7C15ull);
    const uint32_t part = (uint32_t)(((r >> 32) * (uint64_t)P) >> 32);
    const uint32_t b = (uint32_t)(((r & 0xFFFFFFFFu) * (uint64_t)bpp) >> 32);
    return make_uint4(b, part, (uint32_t)i, (uint32_t)(r >> 7));
}

// ---- scatter, direct: a lane claims one slot of its partition with a global atomic and stores 16 bytes
__global__ void __launch_bounds__(256) scatter_direct(uint64_t n, uint32_t P, uint32_t bpp, uint32_t cap, uint32_t *cursor, uint4 *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 t = make_tuple(i, P, bpp);
        const uint32_t k = atomicAdd(&cursor[t.y], 1u);
        if (k < cap) out[(uint64_t)t.y * cap + k] = t;
    }
}

// ---- scatter through LDS bins: P bins x 4 tuples (64 B).  A lane appends to its bin (LDS atomic on the bin's
// fill count); the lane that makes a bin full claims 4 slots of the partition with ONE global atomic and writes the
// whole 64-byte line.  Bins are flushed at the end.  160 KB of LDS holds 2560 x 64 B: one workgroup per CU.
template <int BIN>
__global__ void __launch_bounds__(1024) scatter_lds(uint64_t n, uint32_t P, uint32_t bpp, uint32_t cap, uint32_t *cursor, uint4 *out)
{
    extern __shared__ __align__(16) uint32_t sm[];
    uint32_t *fill = sm;                       // [P]
    uint4 *bins = (uint4 *)(sm + ((P + 3u) & ~3u));   // [P][BIN]
    for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) fill[i] = 0;
    __syncthreads();
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (uint64_t base = lo; base < hi; base += blockDim.x) {
        const uint64_t i = base + threadIdx.x;
        bool pending = i < hi;
        uint4 t = make_uint4(0, 0, 0, 0);
        if (pending) t = make_tuple(i, P, bpp);
        // a bin may be full while its flusher has not emptied it yet: retry after the barrier
        for (int round = 0; round < 64; round++) {
            bool full_owner = false;
            if (pending) {
                const uint32_t k = atomicAdd(&fill[t.y], 1u);
                if (k < (uint32_t)BIN) {
                    bins[t.y * BIN + k] = t;
                    pending = false;
                    full_owner = (k == (uint32_t)BIN - 1u);
                } else atomicSub(&fill[t.y], 1u);
            }
            __syncthreads();
            if (full_owner) {
                const uint32_t k0 = atomicAdd(&cursor[t.y], (uint32_t)BIN);
                if (k0 + BIN <= cap) {
                    uint4 *dst = out + (uint64_t)t.y * cap + k0;
#pragma unroll
                    for (int q = 0; q < BIN; q++) dst[q] = bins[t.y * BIN + q];
                }
                fill[t.y] = 0;
            }
            const int any = __syncthreads_or(pending ? 1 : 0);
            if (!any) break;
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) {
        const uint32_t f = fill[p];
        if (f) {
            const uint32_t k0 = atomicAdd(&cursor[p], f);
            for (uint32_t q = 0; q < f && k0 + q < cap; q++) out[(uint64_t)p * cap + k0 + q] = bins[p * BIN + q];
        }
    }
}

// ---- probe: workgroups walk partitions; tuples are streamed, each does one 16-byte load from the partition's slice
__global__ void __launch_bounds__(256) probe(uint32_t P, uint32_t bpp, uint32_t cap, const uint32_t *cursor, const uint4 *tuples,
                                             const uint4 *table, uint32_t *hits)
{
    uint32_t acc = 0;
    for (uint32_t p = blockIdx.x; p < P; p += gridDim.x) {
        const uint32_t n = cursor[p] < cap ? cursor[p] : cap;
        const uint4 *tp = tuples + (uint64_t)p * cap;
        const uint4 *slice = table + (uint64_t)p * bpp * 4u;   // 64-byte buckets, the detect word first
        for (uint32_t i = threadIdx.x; i < n; i += 4 * blockDim.x) {
            uint4 t[4], k[4];
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] = (i + u * blockDim.x < n) ? tp[i + u * blockDim.x] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; u++) k[u] = slice[(uint64_t)t[u].x * 4u];
#pragma unroll
            for (int u = 0; u < 4; u++) acc += (k[u].x == t[u].z) + (k[u].y == t[u].w) + (k[u].z == t[u].z) + (k[u].w == t[u].w);
        }
    }
    if (acc) atomicAdd(hits, acc);
}

int main(int argc, char **argv)
{
    const uint64_t N = (argc > 1 ? (uint64_t)atoll(argv[1]) : 256ull) << 20;
    const uint32_t P = argc > 2 ? (uint32_t)atoi(argv[2]) : 2048u;
    const size_t table_bytes = (size_t)(argc > 3 ? atoll(argv[3]) : 5120) << 20;
    const uint32_t bpp = (uint32_t)(table_bytes / 64 / P);   // buckets per partition
    const uint32_t cap = (uint32_t)(N / P + N / P / 8 + 4096) & ~3u;
    printf("tuples %.0f M x 16 B = %.2f GB, partitions %u, table %.1f MiB (slice %.2f MiB = %u buckets), capacity %u per partition\n",
           N / 1048576.0, N * 16 / 1e9, P, table_bytes / 1048576.0, bpp * 64.0 / 1048576.0, bpp, cap);
    uint32_t *cursor, *hits;
    uint4 *out, *table;
    CK(hipMalloc(&cursor, (size_t)P * 4));
    CK(hipMalloc(&hits, 4));
    CK(hipMalloc(&out, (size_t)P * cap * 16));
    CK(hipMalloc(&table, table_bytes));
    CK(hipMemset(table, 1, table_bytes));
    CK(hipMemset(hits, 0, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    auto report = [&](const char *what, float t) {
        printf("%-44s %8.3f ms  %7.2f G tuples/s  %7.1f GB/s of tuples   -> configs[2] (675 M tuples): %6.2f ms\n", what, t,
               N / t / 1e6, N * 16.0 / t / 1e6, 675e6 / (N / (double)t));
    };
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(cursor, 0, (size_t)P * 4));
        CK(hipEventRecord(a));
        scatter_direct<<<256 * 8, 256>>>(N, P, bpp, cap, cursor, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep) report("scatter direct (global atomic per tuple)", ms);
    }
    {
        // P bins of 64 bytes need P * 68 bytes of LDS: 2048 partitions = 136 KB (opt-in above 64 KB).  If the opt-in is
        // refused the same kernel runs with the 896 partitions that fit 64 KB -- the first level of a two-level scatter
        // (45 x 45 would do for 2 k partitions; each level moves all the tuples once).
        uint32_t Pl = P;
        size_t sm = ((Pl + 3u) & ~3u) * 4 + (size_t)Pl * 4 * 16;
        hipError_t ae = sm <= 160 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(&scatter_lds<4>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)
                                         : hipErrorInvalidValue;
        if (ae != hipSuccess) {
            (void)hipGetLastError();
            printf("LDS opt-in for %zu bytes refused (%s): LDS-binned scatter runs with 896 partitions (60 KB)\n", sm, hipGetErrorString(ae));
            Pl = 896;
            sm = ((Pl + 3u) & ~3u) * 4 + (size_t)Pl * 4 * 16;
        }
        const uint32_t capl = (uint32_t)(N / Pl + N / Pl / 8 + 4096) & ~3u;
        uint4 *outl = out;
        if ((size_t)Pl * capl > (size_t)P * cap) CK(hipMalloc(&outl, (size_t)Pl * capl * 16));
        uint32_t *cursl;
        CK(hipMalloc(&cursl, (size_t)Pl * 4));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(cursl, 0, (size_t)Pl * 4));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(scatter_lds<4>, dim3(256), dim3(1024), sm, 0, N, Pl, (uint32_t)(table_bytes / 64 / Pl), capl, cursl, outl);
            CK(hipGetLastError());
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep) { char w[96]; snprintf(w, sizeof w, "scatter via LDS bins (64-B lines, %u parts)", Pl); report(w, ms); }
        }
        // plain streaming write of the same bytes: what a perfect scatter could reach
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(a));
            CK(hipMemsetAsync(out, 0x5A, N * 16, 0));
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep) report("(reference: hipMemset of the same bytes)", ms);
        }
        // refill `out` / `cursor` for the probe pass below
        CK(hipMemset(cursor, 0, (size_t)P * 4));
        scatter_direct<<<256 * 8, 256>>>(N, P, bpp, cap, cursor, out);
        CK(hipDeviceSynchronize());
    }
    std::vector<uint32_t> hc(P);
    CK(hipMemcpy(hc.data(), cursor, (size_t)P * 4, hipMemcpyDeviceToHost));
    uint64_t tot = 0; uint32_t mx = 0;
    for (uint32_t v : hc) { tot += v; mx = v > mx ? v : mx; }
    printf("partition fill: total %llu of %llu, max %u (capacity %u)\n", (unsigned long long)tot, (unsigned long long)N, mx, cap);
    for (int occ : {4, 8}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(a));
            probe<<<256 * occ, 256>>>(P, bpp, cap, cursor, out, table, hits);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep) { char w[96]; snprintf(w, sizeof w, "probe from L2-sized slices (%d wg/CU)", occ); report(w, ms); }
        }
    }
    return 0;
}
