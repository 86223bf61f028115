// gather_bench.hip -- measured random-gather ceilings of one MI355X, the yardstick next to the
// 8 TB/s spec in DESIGN.md.  Each lane does `iters` dependent-free random loads of W bytes
// (W = 4..64, naturally aligned) from a table of S bytes; addresses come from a per-lane LCG.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W>  // bytes per access: 4, 8, 16, 32, 64
__global__ void __launch_bounds__(256) gather(const uint4 *__restrict__ tab, uint64_t n_units, int iters, uint32_t *out)
{
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = (uint64_t)gid * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    const char *base = (const char *)tab;
    for (int i = 0; i < iters; i += 4) {
        uint64_t idx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            idx[u] = (uint64_t)(((__uint128_t)s * n_units) >> 64);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const char *p = base + idx[u] * W;
            if (W == 4) acc += *(const uint32_t *)p;
            else if (W == 8) { uint2 v = *(const uint2 *)p; acc += v.x ^ v.y; }
            else if (W == 16) { uint4 v = *(const uint4 *)p; acc += v.x ^ v.w; }
            else if (W == 32) { uint4 a = ((const uint4 *)p)[0], b = ((const uint4 *)p)[1]; acc += a.x ^ b.w; }
            else { uint4 a = ((const uint4 *)p)[0], b = ((const uint4 *)p)[1], c = ((const uint4 *)p)[2], d = ((const uint4 *)p)[3]; acc += a.x ^ b.y ^ c.z ^ d.w; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int W>
void run(const uint4 *tab, size_t bytes, uint32_t *out, int blocks_per_cu)
{
    int iters = 256;
    int grid = 256 * blocks_per_cu;
    uint64_t n_units = bytes / W;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gather<W><<<grid, 256>>>(tab, n_units, 16, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    gather<W><<<grid, 256>>>(tab, n_units, iters, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    double n = (double)grid * 256 * iters;
    printf("table %8.1f MiB  W=%2d B  occupancy %d blk/CU : %7.2f G accesses/s  %7.1f GB/s useful  %7.1f GB/s at 64B-line\n",
           bytes / 1048576.0, W, blocks_per_cu, n / ms / 1e6, n * W / ms / 1e6, n * 64 / ms / 1e6);
}

// gather_bench            the round-1 sweep (2 MiB .. 8 GiB)
// gather_bench MIB        one table size (MiB), W = 16 and 64 at 4 / 6 / 8 blocks per CU: the ceiling
//                         bench.py's roofline.gather_ceiling_frac is measured against for that table
// Table from the virtual-memory API: physical chunks of chunk_mb MiB (each one allocation, hence contiguous), mapped
// back to back into one reserved range -- to see whether larger physically contiguous pieces buy TLB reach.
static uint4 *vmm_alloc(size_t bytes, size_t chunk_mb)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t chunk = chunk_mb << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    void *base = nullptr;
    CK(hipMemAddressReserve(&base, n * chunk, chunk, nullptr, 0));
    for (size_t i = 0; i < n; i++) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)base + i * chunk, chunk, 0, h, 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(base, n * chunk, &acc, 1));
    printf("vmm: granularity %zu KiB, %zu chunks of %zu MiB at %p\n", gran >> 10, n, chunk >> 20, base);
    return (uint4 *)base;
}

int main(int argc, char **argv)
{
    if (argc > 1) {
        const size_t bytes = (size_t)atoll(argv[1]) << 20;
        uint4 *tab;
        uint32_t *out;
        if (argc > 2) tab = vmm_alloc(bytes, (size_t)atoll(argv[2]));   // gather_bench MIB CHUNK_MIB
        else CK(hipMalloc(&tab, bytes));
        CK(hipMalloc(&out, 4));
        CK(hipMemset(tab, 1, bytes));
        for (int occ : {4, 6, 8}) {
            run<16>(tab, bytes, out, occ);
            run<64>(tab, bytes, out, occ);
        }
        return 0;
    }
    size_t maxb = 8ull << 30;
    uint4 *tab;
    uint32_t *out;
    CK(hipMalloc(&tab, maxb));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(tab, 1, maxb));
    size_t sizes[] = {2ull << 20, 24ull << 20, 128ull << 20, 2ull << 30, 8ull << 30};
    for (size_t s : sizes) {
        for (int occ : {4, 8}) {
            run<4>(tab, s, out, occ);
            run<8>(tab, s, out, occ);
            run<16>(tab, s, out, occ);
            run<32>(tab, s, out, occ);
            run<64>(tab, s, out, occ);
        }
    }
    return 0;
}
