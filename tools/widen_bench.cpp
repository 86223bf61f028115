// tools/widen_bench.cpp -- what the host gives rcount's narrow way back (cq_api.cpp narrow_start): bytes -> uint32
// with streaming stores, T threads on disjoint ranges of an 84 M-entry array (configs[2]'s leaf count), into plain and into
// page-locked memory is the same DRAM; GB/s WRITTEN.  The link delivers the bytes at ~55 GB/s, so the widening keeps up
// from ~220 GB/s written on.     g++ -O2 -o tools/widen_bench tools/widen_bench.cpp -lpthread
#include <immintrin.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

__attribute__((target("avx2"))) static void widen_nt(const uint8_t *src, uint32_t *dst, size_t n)
{
    for (size_t i = 0; i + 32 <= n; i += 32) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 16));
        _mm256_stream_si256((__m256i *)(dst + i), _mm256_cvtepu8_epi32(a));
        _mm256_stream_si256((__m256i *)(dst + i + 8), _mm256_cvtepu8_epi32(_mm_srli_si128(a, 8)));
        _mm256_stream_si256((__m256i *)(dst + i + 16), _mm256_cvtepu8_epi32(b));
        _mm256_stream_si256((__m256i *)(dst + i + 24), _mm256_cvtepu8_epi32(_mm_srli_si128(b, 8)));
    }
    _mm_sfence();
}

int main(int argc, char **argv)
{
    const size_t n = (size_t)(argc > 1 ? atoll(argv[1]) : 84) << 20;
    uint8_t *src = (uint8_t *)aligned_alloc(4096, n);
    uint32_t *dst = (uint32_t *)aligned_alloc(4096, n * 4);
    memset(src, 3, n);
    memset(dst, 0, n * 4);
    for (int T : {4, 8, 16, 24, 32, 48, 64, 96}) {
        double best = 0;
        for (int rep = 0; rep < 5; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] { const size_t a = (n * t / T) & ~(size_t)63, b = (n * (t + 1) / T) & ~(size_t)63; widen_nt(src + a, dst + a, b - a); });
            for (auto &x : th) x.join();
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            best = std::max(best, n * 4 / s / 1e9);
        }
        printf("widen %zu M entries, %3d threads: %7.1f GB/s written (best of 5, thread start included)\n", n >> 20, T, best);
    }
    return 0;
}
