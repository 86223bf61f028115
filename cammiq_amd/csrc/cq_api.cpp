// cq_api.cpp -- the C ABI of libcammiq_hip.so (see include/cammiq_hip.h).
//
// Host-side replacement of the seam the reference has at FqReader::loadIdx_p
// (/root/reference/src/query.cpp:109-123) and FqReader::query64_p / query64mt_p / query64_sc
// (query.cpp:458-1080): load + lay out + upload the index, then run the HIP classify
// kernels on reads and hand the counters back.  No CPU classify path exists here.
//
// Multi-GPU (no reference analogue -- the reference's only parallel axis is the
// OpenMP loop over reads, query.cpp:664-665): reads are sharded in contiguous ranges, the index is
// replicated in every GPU's HBM, and the one exchange step is an RCCL all-reduce(sum) of the
// counter block and of rcount at the end of a query, before the host hands the counts on
// (query.cpp:251-258).  Two shapes: cq_multi_* (one process, one host thread per device,
// ncclCommInitAll) and cq_comm_* (one process per GPU, ncclCommInitRank).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <immintrin.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <sys/stat.h>
#include <vector>

#include "cq_index.hpp"
#include "cq_kernels.h"
#include "cq_layout_gpu.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define CQ_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? CQ_ERR_NO_DEVICE \
                                                                             : CQ_ERR_HIP,      \
                        std::string(#call) + ": " + hipGetErrorString(e_));                     \
    } while (0)

#define CQ_NCCL(call)                                                                                       \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) return fail(CQ_ERR_COMM, std::string(#call) + ": " + ncclGetErrorString(r_)); \
    } while (0)

constexpr uint32_t kPairCapDefault = 1u << 20;   // slots of the SC-mode pair map (grown on demand)
constexpr uint64_t kChunk = 1ull << 21;          // reads per pipelined chunk of the host-fed paths (what the workspace is sized for)
uint64_t chunk_reads()                           // ... CAMMIQ_CHUNK_READS overrides (tuning knob, read per query so that one process can A/B)
{
    const char *v = getenv("CAMMIQ_CHUNK_READS");
    const unsigned long long n = v ? strtoull(v, nullptr, 10) : 0;
    return n >= 1024 && n <= (1ull << 28) ? (uint64_t)n : kChunk;
}
constexpr size_t kBounce = 16u << 20;            // pinned bounce buffer for D2H into pageable arrays
constexpr int kSlots = 4;                        // staging slots of the host-fed pipeline
// rcount comes back narrow (one byte per leaf + escapes): launch_narrow_rcount writes the bytes into page-locked host
// memory segment by segment (cq::kNarrowSeg entries each) and flags every segment; a few host threads poll the flags and
// widen the segments into the caller's uint32 arrays as they land
constexpr uint64_t kNarrowFrom = 1u << 20;       // leaves; below, the plain uint32 copy is a fraction of a millisecond anyway

// bytes -> uint32, entries [0, n): streaming stores where the destination allows (the 4 n bytes written are never
// read back by these threads, and a widened rcount of configs[2] is 336 MB: it must not go through the caches)
__attribute__((target("avx2"))) void widen_u8_avx2(const uint8_t *src, uint32_t *dst, size_t n)
{
    size_t i = 0;
    while (i < n && ((uintptr_t)(dst + i) & 31u)) { dst[i] = src[i]; i++; }
    for (; i + 32 <= n; i += 32) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 16));
        _mm256_stream_si256((__m256i *)(dst + i), _mm256_cvtepu8_epi32(a));
        _mm256_stream_si256((__m256i *)(dst + i + 8), _mm256_cvtepu8_epi32(_mm_srli_si128(a, 8)));
        _mm256_stream_si256((__m256i *)(dst + i + 16), _mm256_cvtepu8_epi32(b));
        _mm256_stream_si256((__m256i *)(dst + i + 24), _mm256_cvtepu8_epi32(_mm_srli_si128(b, 8)));
    }
    for (; i < n; i++) dst[i] = src[i];
    _mm_sfence();
}

void widen_u8(const uint8_t *src, uint32_t *dst, size_t n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return widen_u8_avx2(src, dst, n);
    for (size_t i = 0; i < n; i++) dst[i] = src[i];
}

// Narrow entries of GLOBAL leaf indices [a, b) (src8[0] = leaf a) into the two caller arrays: leaves below n_u are
// ht_u's (rcount_u[i]), the rest ht_d's (rcount_d[i - n_u]) -- the device keeps one id space, u first.
void widen_span(const uint8_t *src8, uint64_t a, uint64_t b, uint64_t n_u, uint32_t *dst_u, uint32_t *dst_d)
{
    if (a < n_u) { const uint64_t e = std::min(b, n_u); widen_u8(src8, dst_u + a, (size_t)(e - a)); }
    if (b > n_u) { const uint64_t s0 = std::max(a, n_u); widen_u8(src8 + (s0 - a), dst_d + (s0 - n_u), (size_t)(b - s0)); }
}

// A few host threads that sleep between queries and widen the segments of one narrow rcount as the GPU delivers them
// (narrow_start / narrow_finish).  flags[s] == epoch: segment s is in `narrow`; the threads take segments off a shared counter.
struct WidenPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t job_seq = 0;
    bool quit = false;
    // the current job
    const uint8_t *narrow = nullptr;
    const volatile uint32_t *flags = nullptr;
    uint32_t epoch = 0;
    uint64_t n = 0, n_u = 0, n_segments = 0, seg = 0;
    uint32_t *dst_u = nullptr, *dst_d = nullptr;
    std::atomic<uint64_t> next{0};
    std::atomic<uint32_t> workers_done{0};
    std::atomic<bool> abort{false};          // the kernel failed: stop waiting for flags

    void run()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || job_seq != seen; });
                if (quit) return;
                seen = job_seq;
            }
            for (;;) {
                const uint64_t k = next.fetch_add(1, std::memory_order_relaxed);
                if (k >= n_segments) break;
                while (flags[k] != epoch && !abort.load(std::memory_order_relaxed)) _mm_pause();
                std::atomic_thread_fence(std::memory_order_acquire);
                if (abort.load(std::memory_order_relaxed)) break;
                const uint64_t lo = k * seg, hi = std::min<uint64_t>(lo + seg, n);
                widen_span(narrow + lo, lo, hi, n_u, dst_u, dst_d);
            }
            workers_done.fetch_add(1, std::memory_order_release);
        }
    }
    void start(unsigned W)
    {
        for (unsigned t = 0; t < W; t++) th.emplace_back([this] { run(); });
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        for (auto &x : th) x.join();
        th.clear();
    }
};

// Host threads that sleep between chunks and pack ASCII reads into a chunk's page-locked rows (the ASCII doors of the host-fed
// pipeline).  Started once per handle: spawning 32 threads per 2 M-read chunk cost ~1 ms of the ~4 ms a chunk took.
// run(fn): fn(worker, n_workers) on every worker and on the calling thread (worker 0); returns when all are through.
struct PackPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t job_seq = 0;
    bool quit = false;
    const std::function<void(unsigned, unsigned)> *job = nullptr;
    std::atomic<uint32_t> done{0};

    void loop(unsigned me)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned, unsigned)> *j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || job_seq != seen; });
                if (quit) return;
                seen = job_seq;
                j = job;
            }
            (*j)(me, (unsigned)th.size() + 1u);
            done.fetch_add(1, std::memory_order_release);
        }
    }
    void start(unsigned workers)   // workers besides the caller
    {
        for (unsigned t = 0; t < workers; t++) th.emplace_back([this, t] { loop(t + 1u); });
    }
    void run(const std::function<void(unsigned, unsigned)> &fn)
    {
        done.store(0, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(mu); job = &fn; job_seq++; }
        cv.notify_all();
        fn(0u, (unsigned)th.size() + 1u);
        while (done.load(std::memory_order_acquire) != th.size()) _mm_pause();
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        for (auto &x : th) x.join();
        th.clear();
    }
};

// CAMMIQ_PIPE_TRACE=<file>: timeline of one host-fed query from the library's own HIP events (timing enabled) and host
// clock -- what rocprofv3 cannot give here: under its tracer the kernels of different streams no longer overlap and the
// bracket grows from 26 to 32 ms.  Every mark is (name, chunk, stream event | host time); the file lists them relative
// to the first device event.  Diagnostic: a traced query creates its events on the fly and is a little slower.
struct PipeTrace {
    struct Mark { std::string name; int chunk; hipEvent_t ev; double host_ms; };
    std::vector<Mark> marks;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    const char *file = getenv("CAMMIQ_PIPE_TRACE");
    bool on() const { return file && file[0]; }
    void dev(const char *name, int chunk, hipStream_t st)
    {
        if (!on()) return;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess || hipEventRecord(e, st) != hipSuccess) { (void)hipGetLastError(); return; }
        marks.push_back(Mark{name, chunk, e, 0.0});
    }
    void host(const char *name, int chunk)
    {
        if (!on()) return;
        marks.push_back(Mark{name, chunk, nullptr, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()});
    }
    void dump()
    {
        if (!on() || marks.empty()) return;
        FILE *f = fopen(file, "w");
        hipEvent_t first = nullptr;
        for (auto &m : marks) if (m.ev) { first = m.ev; break; }
        for (auto &m : marks) {
            if (m.ev) {
                float ms = 0.f;
                if (hipEventSynchronize(m.ev) == hipSuccess && first && hipEventElapsedTime(&ms, first, m.ev) == hipSuccess) { if (f) fprintf(f, "dev  %-14s %3d %10.3f\n", m.name.c_str(), m.chunk, ms); }
                else (void)hipGetLastError();
            } else if (f) fprintf(f, "host %-14s %3d %10.3f\n", m.name.c_str(), m.chunk, m.host_ms);
        }
        for (auto &m : marks) if (m.ev) (void)hipEventDestroy(m.ev);   // (after the loop: `first` is one of them)
        if (f) fclose(f);
        marks.clear();
    }
};

unsigned widen_threads()
{
    if (const char *v = getenv("CAMMIQ_WIDEN_THREADS")) return (unsigned)std::min(64, std::max(1, atoi(v)));
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(16u, hc / 2));
}

// What decode + layout leave on the host.  One copy serves every handle of a cq_multi (the
// index is replicated in HBM, not in host memory).
struct HostIndex {
    cq::DecodedTable tab[2];
    cq::FlatImage img;
    uint64_t n_file_buckets[2] = {0, 0};
    uint64_t n_trie_nodes = 0;
    bool from_cache = false;
    uint32_t doubly_flag[2] = {0, 1};
    // The table is laid out on the device (cq_layout_gpu.hip): img.table stays empty, the decoded keys and their final
    // trie codes wait here until every handle has uploaded them.  0 = host layout; 1 = device; 2 = device, and the host
    // builder's image is built too and compared byte for byte (CAMMIQ_GPU_LAYOUT=verify: tests).
    int gpu_layout = 0;
    cq::RawVec<uint32_t> vals;
    std::mutex mu;            // cq_multi: several uploading threads share this object
    bool host_table_built = false;
    // the table's block in HBM, allocated by a thread of its own while the host part of the layout runs (hipMalloc of
    // configs[4]'s 80 GB takes ~2 s): taken by the first handle on that device
    std::thread prealloc_thread;
    void *prealloc = nullptr;
    uint64_t prealloc_buckets = 0;
    int prealloc_device = -1;
    ~HostIndex()
    {
        if (prealloc_thread.joinable()) prealloc_thread.join();
        if (prealloc) { (void)hipSetDevice(prealloc_device); (void)hipFree(prealloc); }
    }
};

// CAMMIQ_GPU_LAYOUT: 0 = host builder, 1 = device (any size), verify = device + host, compared; unset = device from
// 65 536 entries on (below, the host builder is done before the first kernel would have launched).
int gpu_layout_mode(uint64_t n_entries)
{
    const char *v = getenv("CAMMIQ_GPU_LAYOUT");
    if (v && !strcmp(v, "verify")) return 2;
    if (v && v[0]) return atoi(v) != 0 ? 1 : 0;
    return n_entries >= (1u << 16) ? 1 : 0;
}

}  // namespace

struct cq_index {
    std::shared_ptr<HostIndex> H;
    int device = CQ_DEVICE_NONE;
    int n_cus = 0;
    uint64_t device_bytes = 0;
    // device image
    void *d_slots = nullptr, *d_nodes = nullptr, *d_leaf_rids = nullptr;
    cq::DevIndex dev{};
    // per-handle workspace (grown on demand, reused across calls): ONE query in flight per handle
    uint32_t *d_ovf_list = nullptr, *d_ovf_count = nullptr;
    uint64_t *d_stamps = nullptr;   // 8 words of their own for diagnostic (CQ_STAMPS) kernel builds
    uint64_t ovf_cap = 0;
    uint64_t *d_pair_keys = nullptr, *d_pair_cnts = nullptr;
    uint32_t pair_cap = 0;          // slots, power of two
    hipEvent_t ev0 = nullptr, ev_mid = nullptr, ev1 = nullptr;   // start | fast kernel done | slow kernel done
    bool ev_valid = false;
    cq::LaunchInfo last_launch{};   // what launch_classify chose for the most recent launch (cq_last_launch_info)
    // host-fed paths (cq_query, cq_query_packed): staging slots (kSlots) so that packing chunk c+1 on
    // the CPU, its H2D copy and the classify kernel of chunk c overlap with slack for host jitter
    struct Slot {
        uint32_t *h_packed = nullptr, *d_packed = nullptr;   // pinned host / device rows
        uint8_t *h_lens = nullptr, *d_lens = nullptr;
        uint8_t *d_tight = nullptr;                          // tight rows as they arrive (cq_query_packed_tight), widened into d_packed
        size_t cap_words_h = 0, cap_reads_h = 0, cap_words_d = 0, cap_reads_d = 0, cap_tight = 0;
        hipEvent_t copied = nullptr, copied_lens = nullptr, widened = nullptr, done = nullptr;
        uint32_t *d_ovf_list = nullptr, *d_ovf_count = nullptr;   // this slot's slow-path list: the kernels of consecutive chunks overlap
        uint64_t ovf_cap = 0;
    } slot[kSlots];   // the host may enqueue the copies of the next chunks while the kernels of the chunks before are still running
    hipStream_t s_copy = nullptr, s_copy2 = nullptr, s_comp = nullptr;   // rows | lengths (their own DMA queue) | kernels
    hipStream_t s_comp2 = nullptr;   // the kernels of odd chunks: chunk c + 1 fills the CUs while chunk c drains
    hipEvent_t ev_zeroed = nullptr, ev_comp2 = nullptr;
    hipStream_t s_widen = nullptr;   // tight rows -> word rows, beside the classify kernel of the chunk before (it leaves wave slots free)
    uint64_t *d_ctr = nullptr; size_t ctr_cap = 0;
    uint32_t *d_rc = nullptr; size_t rc_cap = 0;
    void *h_bounce[2] = {nullptr, nullptr};
    hipEvent_t ev_bounce[2] = {nullptr, nullptr};
    // rcount's narrow way back (narrow_start / narrow_finish)
    uint8_t *d_rc8 = nullptr; size_t rc8_cap = 0;
    uint2 *d_esc = nullptr; uint32_t *d_esc_count = nullptr; uint32_t esc_cap = 0;
    uint8_t *h_narrow = nullptr; size_t narrow_cap = 0;   // page-locked: the kernel writes rcount's bytes here
    uint32_t *h_flags = nullptr; size_t flags_cap = 0;    // page-locked: one word per segment + the final one
    uint32_t *d_blocks_done = nullptr;
    uint32_t narrow_epoch = 0;
    uint2 *h_esc = nullptr; uint32_t *h_esc_count = nullptr;
    hipEvent_t ev_narrow = nullptr;
    WidenPool *pool = nullptr;
    PackPool *pack_pool = nullptr;   // ASCII doors: packers of the host-fed pipeline (ensure_pack_pool)
    bool narrow_inflight = false;              // narrow_start has queued the kernel and woken the pool; narrow_finish collects
    uint64_t narrow_nseg = 0;
    PipeTrace *trace = nullptr;                // CAMMIQ_PIPE_TRACE: the query being traced (classify_range .. fetch_counts)
};

// One RCCL communicator bound to one handle's device.
struct cq_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, n_ranks = 1;
};

struct cq_multi {
    std::vector<cq_index *> ix;        // one per entry of the device list
    std::vector<int> leader_of;        // ix[i]'s device group: index of the first handle on that device
    std::vector<int> leaders;          // handles that take part in the RCCL all-reduce (distinct devices)
    std::vector<ncclComm_t> comms;     // one per leader, ncclCommInitAll
};

namespace {

void release_device(cq_index *ix)
{
    if (ix->device < 0) return;
    (void)hipSetDevice(ix->device);
    if (ix->d_slots) (void)hipFree(ix->d_slots);
    if (ix->d_nodes) (void)hipFree(ix->d_nodes);
    if (ix->d_leaf_rids) (void)hipFree(ix->d_leaf_rids);
    if (ix->d_ovf_list) (void)hipFree(ix->d_ovf_list);
    if (ix->d_ovf_count) (void)hipFree(ix->d_ovf_count);
    if (ix->d_stamps) {
        // diagnostic builds (-DCQ_STAMPS=1): per-phase cycle sums, dumped when the handle goes away
        if (const char *f = getenv("CAMMIQ_STAMPS_FILE")) {
            uint64_t v[8] = {0};
            if (hipMemcpy(v, ix->d_stamps, sizeof v, hipMemcpyDeviceToHost) == hipSuccess)
                if (FILE *fo = fopen(f, "w")) {
                    fprintf(fo, "staging %llu\nprepass %llu\nprobe %llu\nlookup %llu\ndecide %llu\n", (unsigned long long)v[0],
                            (unsigned long long)v[1], (unsigned long long)v[2], (unsigned long long)v[3], (unsigned long long)v[4]);
                    // shader cycles / 100 MHz ticks over the waves' main loops: the clock the chip held under this kernel
                    fprintf(fo, "shader_cycles %llu\nrealtime_ticks_100MHz %llu\nin_kernel_clock_MHz %.1f\n", (unsigned long long)v[5],
                            (unsigned long long)v[6], v[6] ? 100.0 * (double)v[5] / (double)v[6] : 0.0);
                    fclose(fo);
                }
        }
        (void)hipFree(ix->d_stamps);
    }
    if (ix->d_pair_keys) (void)hipFree(ix->d_pair_keys);
    if (ix->d_pair_cnts) (void)hipFree(ix->d_pair_cnts);
    for (hipEvent_t e : {ix->ev0, ix->ev_mid, ix->ev1, ix->ev_bounce[0], ix->ev_bounce[1]})
        if (e) (void)hipEventDestroy(e);
    for (auto &sl : ix->slot) {
        if (sl.h_packed) (void)hipHostFree(sl.h_packed);
        if (sl.h_lens) (void)hipHostFree(sl.h_lens);
        if (sl.d_packed) (void)hipFree(sl.d_packed);
        if (sl.d_lens) (void)hipFree(sl.d_lens);
        if (sl.d_tight) (void)hipFree(sl.d_tight);
        if (sl.copied) (void)hipEventDestroy(sl.copied);
        if (sl.copied_lens) (void)hipEventDestroy(sl.copied_lens);
        if (sl.widened) (void)hipEventDestroy(sl.widened);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.d_ovf_list) (void)hipFree(sl.d_ovf_list);
        if (sl.d_ovf_count) (void)hipFree(sl.d_ovf_count);
    }
    if (ix->s_comp2) (void)hipStreamDestroy(ix->s_comp2);
    if (ix->ev_zeroed) (void)hipEventDestroy(ix->ev_zeroed);
    if (ix->ev_comp2) (void)hipEventDestroy(ix->ev_comp2);
    for (void *b : ix->h_bounce) if (b) (void)hipHostFree(b);
    if (ix->s_copy) (void)hipStreamDestroy(ix->s_copy);
    if (ix->s_copy2) (void)hipStreamDestroy(ix->s_copy2);
    if (ix->s_comp) (void)hipStreamDestroy(ix->s_comp);
    if (ix->s_widen) (void)hipStreamDestroy(ix->s_widen);
    if (ix->d_ctr) (void)hipFree(ix->d_ctr);
    if (ix->d_rc) (void)hipFree(ix->d_rc);
    if (ix->pool) { ix->pool->stop(); delete ix->pool; ix->pool = nullptr; }
    if (ix->pack_pool) { ix->pack_pool->stop(); delete ix->pack_pool; ix->pack_pool = nullptr; }
    if (ix->d_rc8) (void)hipFree(ix->d_rc8);
    if (ix->d_esc) (void)hipFree(ix->d_esc);
    if (ix->d_esc_count) (void)hipFree(ix->d_esc_count);
    if (ix->h_narrow) (void)hipHostFree(ix->h_narrow);
    if (ix->h_flags) (void)hipHostFree(ix->h_flags);
    if (ix->d_blocks_done) (void)hipFree(ix->d_blocks_done);
    if (ix->h_esc) (void)hipHostFree(ix->h_esc);
    if (ix->h_esc_count) (void)hipHostFree(ix->h_esc_count);
    if (ix->ev_narrow) (void)hipEventDestroy(ix->ev_narrow);
}

// (Re)allocate the SC-mode pair map with `slots` slots (power of two), empty.
int pairs_alloc(cq_index *ix, uint32_t slots)
{
    if (ix->d_pair_keys) (void)hipFree(ix->d_pair_keys);
    if (ix->d_pair_cnts) (void)hipFree(ix->d_pair_cnts);
    ix->d_pair_keys = ix->d_pair_cnts = nullptr;
    ix->pair_cap = 0;
    CQ_HIP(hipMalloc((void **)&ix->d_pair_keys, (size_t)slots * 8));
    CQ_HIP(hipMalloc((void **)&ix->d_pair_cnts, (size_t)slots * 8));
    CQ_HIP(hipMemset(ix->d_pair_keys, 0xFF, (size_t)slots * 8));
    CQ_HIP(hipMemset(ix->d_pair_cnts, 0, (size_t)slots * 8));
    ix->pair_cap = slots;
    return CQ_OK;
}

int pairs_clear(cq_index *ix)
{
    CQ_HIP(hipMemset(ix->d_pair_keys, 0xFF, (size_t)ix->pair_cap * 8));
    CQ_HIP(hipMemset(ix->d_pair_cnts, 0, (size_t)ix->pair_cap * 8));
    return CQ_OK;
}

uint32_t pair_cap_from_env()
{
    // test knob: start the SC pair map small so that its growth path runs on small inputs
    if (const char *v = getenv("CAMMIQ_PAIR_SLOTS")) {
        uint32_t s = 16;
        const unsigned long want = strtoul(v, nullptr, 10);
        while (s < want && s < (1u << 30)) s <<= 1;
        return s;
    }
    return kPairCapDefault;
}

// Host (pageable) -> device copy of a large array.  A plain hipMemcpy of pageable memory is staged by the runtime at
// ~13 GB/s (configs[4]'s 80 GB table: 6 s); two page-locked 128 MiB pieces filled by eight memcpy threads while the
// previous piece is on the link run at what the link gives.  Falls back to hipMemcpy for small arrays and on any error.
// fill(piece, offset, n): writes bytes [offset, offset + n) of the source into `piece` (called from eight threads on
// disjoint ranges).  src != nullptr: the source is that array (fill may be empty) and small arrays / failures take a
// plain hipMemcpy; src == nullptr: the source exists only through fill (e.g. two arrays interleaved on the fly).
hipError_t upload_pieces(void *dst, const void *src, size_t bytes, const std::function<void(char *, size_t, size_t)> &fill)
{
    const size_t piece = 128u << 20;
    if (bytes == 0) return hipSuccess;
    auto staged = [&]() -> hipError_t {   // no contiguous source: through pageable memory, piece by piece
        std::vector<char> tmp(std::min(piece, bytes));
        for (size_t off = 0; off < bytes; off += tmp.size()) {
            const size_t n = std::min(tmp.size(), bytes - off);
            fill(tmp.data(), off, n);
            hipError_t e = hipMemcpy((char *)dst + off, tmp.data(), n, hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    if (bytes < 4 * piece) return src ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : staged();   // small: not worth two page-locked buffers
    const auto t_begin = std::chrono::steady_clock::now();
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    for (int b = 0; b < 2 && ok; b++)
        ok = hipHostMalloc(&pin[b], piece, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) == hipSuccess;
    size_t done = 0;
    for (size_t k = 0; ok && done < bytes; k++) {
        const size_t n = std::min(piece, bytes - done);
        const int b = (int)(k & 1);
        if (k >= 2) ok = hipEventSynchronize(ev[b]) == hipSuccess;   // the copy that last read this piece
        if (!ok) break;
        const unsigned nt = 8;
        std::vector<std::thread> th;
        char *d0 = (char *)pin[b];
        // sub-ranges on 64-byte boundaries (producers that interleave records must not split one)
        auto cut = [&](unsigned t) { return t >= nt ? n : (n * t / nt) & ~(size_t)63; };
        for (unsigned t = 1; t < nt; t++) th.emplace_back([&, t] { fill(d0 + cut(t), done + cut(t), cut(t + 1) - cut(t)); });
        fill(d0, done, cut(1));
        for (auto &x : th) x.join();
        ok = hipMemcpyAsync((char *)dst + done, pin[b], n, hipMemcpyHostToDevice, st) == hipSuccess && hipEventRecord(ev[b], st) == hipSuccess;
        if (ok) done += n;
    }
    if (st) ok = (hipStreamSynchronize(st) == hipSuccess) && ok;
    for (int b = 0; b < 2; b++) { if (pin[b]) (void)hipHostFree(pin[b]); if (ev[b]) (void)hipEventDestroy(ev[b]); }
    if (st) (void)hipStreamDestroy(st);
    if (getenv("CAMMIQ_LOAD_TIMING"))
        fprintf(stderr, "[cq_index_load]   upload_array %.1f GB in %.2f s (%s)\n", bytes / 1e9,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), ok && done == bytes ? "pinned pieces" : "fell back to hipMemcpy");
    if (ok && done == bytes) return hipSuccess;
    (void)hipGetLastError();
    return src ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : staged();   // whatever went wrong above: the plain copies decide
}

hipError_t upload_array(void *dst, const void *src, size_t bytes)
{
    const char *s0 = (const char *)src;
    return upload_pieces(dst, src, bytes, [s0](char *p, size_t off, size_t n) { memcpy(p, s0 + off, n); });
}

// Copy the flat image into the HBM of ix->device.
int upload(cq_index *ix)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(CQ_ERR_NO_DEVICE, "no HIP device available");
    if (ix->device >= ndev) return fail(CQ_ERR_NO_DEVICE, "device ordinal out of range");
    CQ_HIP(hipSetDevice(ix->device));
    hipDeviceProp_t prop;
    CQ_HIP(hipGetDeviceProperties(&prop, ix->device));
    ix->n_cus = prop.multiProcessorCount;
    HostIndex &H = *ix->H;
    cq::FlatImage &img = H.img;
    const size_t nb = img.nodes.size() * sizeof(cq::Node);
    const size_t nl = img.leaf_r1.size();
    const bool timing = getenv("CAMMIQ_LOAD_TIMING") != nullptr;
    auto t_up = std::chrono::steady_clock::now();
    auto up_lap = [&](const char *what) {
        const auto n = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "[cq_index_load]   %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t_up).count());
        t_up = n;
    };
    CQ_HIP(hipMalloc(&ix->d_nodes, nb));
    CQ_HIP(hipMalloc(&ix->d_leaf_rids, std::max<size_t>(nl, 1) * sizeof(uint2)));
    CQ_HIP(upload_array(ix->d_nodes, img.nodes.data(), nb));
    // leaf refIDs: (refID1, refID2) pairs interleaved straight into the page-locked pieces (no 8-byte-per-leaf staging array)
    CQ_HIP(upload_pieces(ix->d_leaf_rids, nullptr, nl * sizeof(uint2), [&img](char *p, size_t off, size_t n) {
        uint2 *o = (uint2 *)p;
        const size_t i0 = off / sizeof(uint2), cnt = n / sizeof(uint2);
        for (size_t i = 0; i < cnt; i++) o[i] = make_uint2(img.leaf_r1[i0 + i], img.leaf_r2[i0 + i]);
    }));
    up_lap("nodes + leaf refIDs");
    bool on_device = false;
    if (H.gpu_layout) {
        // ---- the table is laid out here, in HBM, from the decoded keys and their trie codes (cq_layout_gpu.hip)
        const uint64_t nb_u = H.tab[0].bucket_key.size(), nb_d = H.tab[1].bucket_key.size(), ne = nb_u + nb_d;
        uint64_t *d_keys = nullptr;
        uint32_t *d_vals = nullptr;
        cq::DeviceLayoutResult res;
        bool unsupported = false;
        hipError_t e = hipMalloc((void **)&d_keys, std::max<uint64_t>(ne, 1) * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&d_vals, std::max<uint64_t>(ne, 1) * 4);
        if (e == hipSuccess && nb_u) e = upload_array(d_keys, H.tab[0].bucket_key.data(), nb_u * 8);
        if (e == hipSuccess && nb_d) e = upload_array(d_keys + nb_u, H.tab[1].bucket_key.data(), nb_d * 8);
        if (e == hipSuccess && ne) e = upload_array(d_vals, H.vals.data(), ne * 4);
        up_lap("keys + codes");
        void *pre = nullptr;
        uint64_t pre_buckets = 0;
        {   // the block allocated beside the host part, if it is on this device and nobody took it yet
            std::lock_guard<std::mutex> lk(H.mu);
            if (H.prealloc_thread.joinable()) H.prealloc_thread.join();
            if (H.prealloc && H.prealloc_device == ix->device) { pre = H.prealloc; pre_buckets = H.prealloc_buckets; H.prealloc = nullptr; }
        }
        if (e == hipSuccess)
            e = cq::layout_table_on_device(d_keys, d_vals, nb_u, nb_d, img.hash_len, img.minimizer_len, (uint32_t)img.n_buckets,
                                           (const uint2 *)ix->d_leaf_rids, ix->n_cus, res, unsupported, pre, pre_buckets);
        else if (pre) (void)hipFree(pre);
        if (d_keys) (void)hipFree(d_keys);
        if (d_vals) (void)hipFree(d_vals);
        up_lap("device layout");
        if (e == hipSuccess && res.limit) return fail(CQ_ERR_LIMIT, "table would exceed 2^32 buckets");
        if (e == hipSuccess && !unsupported) {
            on_device = true;
            ix->d_slots = res.d_table;
            std::lock_guard<std::mutex> lk(H.mu);   // (every handle of a cq_multi computes the same numbers)
            img.n_buckets_alloc = res.n_buckets_alloc;
            img.table_words = res.n_buckets_alloc * CQ_BUCKET_WORDS;
            img.n_keys = res.n_keys;
            img.n_overflowed = res.n_overflowed;
            img.max_chain = res.max_chain;
        } else (void)hipGetLastError();   // out of device memory for the workspace, or a home group the device path leaves to the host
    }
    if (!on_device || H.gpu_layout == 2) {
        // the host builder's image: handles without the device layout, its fall-back, and the reference `verify` compares with
        if (H.gpu_layout) {
            std::lock_guard<std::mutex> lk(H.mu);
            if (!H.host_table_built) {
                const cq::DeviceLayoutResult dev_stats = [&] { cq::DeviceLayoutResult r; r.n_buckets_alloc = img.n_buckets_alloc; r.n_keys = img.n_keys; r.n_overflowed = img.n_overflowed; r.max_chain = img.max_chain; return r; }();
                std::string err;
                int rc = cq::finish_image_host(H.tab[0], H.tab[1], H.vals, img, err);
                if (rc != CQ_OK) return fail(rc, err);
                H.host_table_built = true;
                up_lap("host layout");
                if (on_device && (dev_stats.n_buckets_alloc != img.n_buckets_alloc || dev_stats.n_keys != img.n_keys ||
                                  dev_stats.n_overflowed != img.n_overflowed || dev_stats.max_chain != img.max_chain))
                    return fail(CQ_ERR_FORMAT, "device layout: statistics differ from the host builder's (buckets " + std::to_string(dev_stats.n_buckets_alloc) + "/" +
                                                   std::to_string(img.n_buckets_alloc) + ", keys " + std::to_string(dev_stats.n_keys) + "/" + std::to_string(img.n_keys) +
                                                   ", overflowed " + std::to_string(dev_stats.n_overflowed) + "/" + std::to_string(img.n_overflowed) + ", max chain " +
                                                   std::to_string(dev_stats.max_chain) + "/" + std::to_string(img.max_chain) + ")");
            }
        }
        const size_t sb = img.table_words * sizeof(uint32_t);
        if (on_device) {   // verify: the device's table against the host builder's, word for word
            std::vector<uint32_t> got(img.table_words);
            CQ_HIP(hipMemcpy(got.data(), ix->d_slots, sb, hipMemcpyDeviceToHost));
            for (size_t w = 0; w < img.table_words; w++)
                if (got[w] != img.table[w])
                    return fail(CQ_ERR_FORMAT, "device layout differs from the host builder's image at bucket " + std::to_string(w / CQ_BUCKET_WORDS) + " word " +
                                                   std::to_string(w % CQ_BUCKET_WORDS) + ": " + std::to_string(got[w]) + " vs " + std::to_string(img.table[w]));
            up_lap("verify (equal)");
        } else {
            CQ_HIP(hipMalloc(&ix->d_slots, sb));
            CQ_HIP(upload_array(ix->d_slots, img.table.get(), sb));
            up_lap("table");
        }
    }
    const size_t sb = img.table_words * sizeof(uint32_t);
    CQ_HIP(hipMalloc((void **)&ix->d_ovf_count, (size_t)(1 + cq::kWorkStripes) * cq::kWorkStripeWords * sizeof(uint32_t)));
    CQ_HIP(hipMalloc((void **)&ix->d_stamps, 8 * sizeof(uint64_t)));
    CQ_HIP(hipMemset(ix->d_stamps, 0, 8 * sizeof(uint64_t)));
    int rc = pairs_alloc(ix, pair_cap_from_env());
    if (rc != CQ_OK) return rc;
    CQ_HIP(hipEventCreate(&ix->ev0));
    CQ_HIP(hipEventCreate(&ix->ev_mid));
    CQ_HIP(hipEventCreate(&ix->ev1));
    ix->device_bytes = sb + nb + nl * sizeof(uint2) + (size_t)ix->pair_cap * 16;
    ix->dev.slots = (const uint4 *)ix->d_slots;
    ix->dev.nodes = (const uint4 *)ix->d_nodes;
    ix->dev.leaf_rids = (const uint2 *)ix->d_leaf_rids;
    ix->dev.n_buckets = (uint32_t)img.n_buckets;
    ix->dev.hash_len = img.hash_len;
    ix->dev.minimizer_len = img.minimizer_len;
    return CQ_OK;
}

// CAMMIQ_LOAD_TIMING=1: stage timings of cq_index_load on stderr (diagnostic)
struct LoadTimer {
    bool on = getenv("CAMMIQ_LOAD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        const auto n = std::chrono::steady_clock::now();
        if (on) fprintf(stderr, "[cq_index_load] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// Decode + lay out (or read the image cache): everything of cq_index_load that happens on the
// host.  budget = bytes of HBM the table may take (1e30 = unlimited).
int prepare_host(const char *path_u, const char *path_d, double budget, std::shared_ptr<HostIndex> &out, LoadTimer &lt, int device)
{
    const bool for_device = device >= 0;
    std::shared_ptr<HostIndex> H(new (std::nothrow) HostIndex());
    if (!H) return fail(CQ_ERR_NOMEM, "out of memory");
    const bool have_d = path_d && path_d[0];
    // Optional image cache next to index_u (cq_cache.cpp); the .bin files stay authoritative.
    const char *ce = getenv("CAMMIQ_IMAGE_CACHE");
    const bool cache_forced = ce && !strcmp(ce, "force");
    bool use_cache = ce && (ce[0] == '1' || cache_forced);
    // Estimate of the number of keys from the FILE SIZES: a bucket whose root is a leaf takes 8 + 6 (unique) or 8 + 12 (doubly
    // unique) bytes of the byte stream, any other bucket more, so size / 14 + size / 20 bounds the keys from above.
    double est_keys = 0.0;
    {
        struct stat su, sd;
        if (stat(path_u, &su) == 0) est_keys += (double)su.st_size / 14.0;
        if (have_d && stat(path_d, &sd) == 0) est_keys += (double)sd.st_size / 20.0;
    }
    // A handle on a GPU whose table is laid out THERE is ready sooner without the image than with it (configs[1]: 0.34-0.46 s
    // against 0.56-0.93 s for reading 4 GB back): the cache serves handles without a device and tables the host builder lays
    // out; CAMMIQ_IMAGE_CACHE=force keeps it for every handle.
    if (use_cache && !cache_forced && for_device && gpu_layout_mode((uint64_t)est_keys)) use_cache = false;
    const std::string cache_file = std::string(path_u) + ".cqimg";
    cq::SourceStamp stamp;
    const bool stamped = use_cache && cq::stamp_sources(path_u, have_d ? path_d : "", stamp);
    bool from_cache = false;
    const double kpb_override = getenv("CAMMIQ_KEYS_PER_BUCKET") ? atof(getenv("CAMMIQ_KEYS_PER_BUCKET")) : 0.0;
    // minimizer length of the table: automatic by its size (cq_device.h), CAMMIQ_MINIMIZER_LEN overrides (tuning knob)
    const uint32_t m_override = getenv("CAMMIQ_MINIMIZER_LEN") ? (uint32_t)std::max(0, atoi(getenv("CAMMIQ_MINIMIZER_LEN"))) : 0u;
    if (stamped)
        from_cache = cq::load_image(cache_file, stamp, kpb_override, m_override, budget >= 1e29 ? ~0ull : (uint64_t)budget, H->tab, H->img);
    if (from_cache) lt.lap("image cache read");
    // The reference loads the two files on two pthreads (query.cpp:112-116); same here.
    int rc_u = CQ_OK, rc_d = CQ_OK;
    std::string err_u, err_d;
    // The table's block of HBM is asked for on a thread of its own (hipMalloc of 80 GB takes ~2 s, and every other
    // allocation of the load queues behind it in the driver): before the decode, from the file sizes' bound on the number of
    // keys (est_keys above), so that the block is there when the layout wants it.
    // A block more than a tenth too large (an index of deep tries) is given back after the decode and asked for again.
    auto start_prealloc = [&](uint64_t n_table_buckets) {
        HostIndex *hp = H.get();
        hp->prealloc_device = device;
        hp->prealloc_buckets = n_table_buckets;
        hp->prealloc_thread = std::thread([hp] {
            if (hipSetDevice(hp->prealloc_device) != hipSuccess || hipMalloc(&hp->prealloc, hp->prealloc_buckets * 64) != hipSuccess) {
                (void)hipGetLastError();
                hp->prealloc = nullptr;
            }
        });
    };
    auto table_buckets_for = [&](double keys, double &kpb) {
        // Average keys per 4-slot bucket of the device table.  Emptier tables overflow less
        // (fewer windows take the exact path): 1.0 costs 64 B of HBM per key (measured: 0.5 -> +2 %,
        // 1.5 -> -3 %); when the table would not fit comfortably it is packed tighter.
        // CAMMIQ_KEYS_PER_BUCKET overrides (tuning knob, not part of the ABI).
        kpb = 1.0;
        if (keys / kpb * 64.0 > budget) kpb = std::min(3.2, keys * 64.0 / budget);
        if (kpb_override > 0.0) kpb = kpb_override;
        uint64_t nbk = (uint64_t)(keys / kpb) + 1;   // the hash range cq::prepare_image will choose ...
        if (nbk < 16) nbk = 16;
        return nbk + CQ_SPILL_TAIL;                  // ... + the spill tail
    };
    if (!from_cache && for_device && !stamped) {
        const double est = est_keys;
        double kpb_est;
        if (est >= 1e7 && gpu_layout_mode((uint64_t)est)) try { start_prealloc(table_buckets_for(est, kpb_est)); } catch (...) {}
    }
    if (!from_cache) try {
        std::thread td;
        if (have_d) td = std::thread([&] { rc_d = cq::decode_table(path_d, H->tab[1], err_d); });
        rc_u = cq::decode_table(path_u, H->tab[0], err_u);
        if (have_d) td.join();
        if (rc_u != CQ_OK) return fail(rc_u, err_u);
        if (rc_d != CQ_OK) return fail(rc_d, err_d);
        if (!have_d) cq::make_empty_table(H->tab[0].hash_len, H->tab[1]);
        lt.lap("decode");
        std::string err;
        double kpb = 1.0;
        const double keys = (double)(H->tab[0].bucket_key.size() + H->tab[1].bucket_key.size());
        const uint64_t n_table_buckets = table_buckets_for(keys, kpb);
        const uint32_t m_len = m_override ? cq_minimizer_len(H->tab[0].hash_len, std::min<uint32_t>(m_override, CQ_MAX_MINIMIZER)) : 0u;
        // a handle on a GPU has its table laid out THERE (upload(): cq_layout_gpu.hip); the image cache and handles without
        // a device need the host builder's image
        H->gpu_layout = (for_device && !stamped) ? gpu_layout_mode((uint64_t)keys) : 0;
        if (H->prealloc_thread.joinable() &&
            (!H->gpu_layout || H->prealloc_buckets < n_table_buckets || (double)H->prealloc_buckets > 1.1 * (double)n_table_buckets)) {
            // the estimate from the file sizes does not serve (host layout after all, or too small, or a tenth too large)
            H->prealloc_thread.join();
            if (H->prealloc) { (void)hipSetDevice(device); (void)hipFree(H->prealloc); H->prealloc = nullptr; }
            H->prealloc_buckets = 0;
        }
        // the table's size is known now: allocate it beside the host part of the layout, unless it is on its way already
        if (H->gpu_layout && keys >= 1e7 && !H->prealloc_thread.joinable()) start_prealloc(n_table_buckets);
        int rc = H->gpu_layout ? cq::prepare_image(H->tab[0], H->tab[1], kpb, m_len, H->img, H->vals, err)
                               : cq::build_image(H->tab[0], H->tab[1], kpb, m_len, H->img, err);
        if (rc != CQ_OK) return fail(rc, err);
        lt.lap(H->gpu_layout ? "layout (host part)" : "layout");
        // bucket/node arrays of the decode stage are no longer needed; leaves are (cq_index_leaves).  With the device
        // layout the keys wait for their upload (release_layout_inputs).
        for (int t = 0; t < 2; t++) {
            if (!H->gpu_layout) cq::RawVec<uint64_t>().swap(H->tab[t].bucket_key);
            cq::RawVec<uint32_t>().swap(H->tab[t].bucket_code);
            cq::RawVec<cq::Node>().swap(H->tab[t].nodes);
        }
        if (stamped) { (void)cq::save_image(cache_file, stamp, kpb_override, m_override, H->tab, H->img); lt.lap("image cache write"); }
    } catch (const std::bad_alloc &) {
        return fail(CQ_ERR_NOMEM, "out of memory while loading the index");
    }
    for (int t = 0; t < 2; t++) {
        H->n_file_buckets[t] = H->tab[t].n_file_buckets;
        H->doubly_flag[t] = H->tab[t].doubly;
    }
    H->from_cache = from_cache;
    H->n_trie_nodes = H->img.nodes.size() - 1;
    out = H;
    return CQ_OK;
}

// The image now lives in HBM: drop the host copy (cq_index_probe needs a CQ_DEVICE_NONE handle).
void drop_host_image(HostIndex &H)
{
    // Giving several GB back to the OS takes hundreds of ms (0.4 s for configs[1]'s 3.2 GB table): a thread of its
    // own does it while cq_index_load returns.
    // (with the device layout there is no table image; what goes back is what the layout was fed from: the decoded keys and
    // their trie codes, 15 GB at configs[4]'s size, 0.7 s when freed in line)
    struct Dead { cq::HugeWords table; std::vector<cq::Node> nodes; std::vector<uint32_t> r1, r2; cq::RawVec<uint64_t> k0, k1; cq::RawVec<uint32_t> vals; };
    std::shared_ptr<Dead> d(new (std::nothrow) Dead());
    if (!d) {   // no memory for the little carrier: free in place
        for (int t = 0; t < 2; t++) cq::RawVec<uint64_t>().swap(H.tab[t].bucket_key);
        cq::RawVec<uint32_t>().swap(H.vals);
        H.img.table.reset();
        std::vector<cq::Node>().swap(H.img.nodes);
        std::vector<uint32_t>().swap(H.img.leaf_r1);
        std::vector<uint32_t>().swap(H.img.leaf_r2);
        return;
    }
    d->table = std::move(H.img.table);
    d->k0.swap(H.tab[0].bucket_key);
    d->k1.swap(H.tab[1].bucket_key);
    d->vals.swap(H.vals);
    d->nodes.swap(H.img.nodes);
    d->r1.swap(H.img.leaf_r1);
    d->r2.swap(H.img.leaf_r2);
    try {
        std::thread([d = std::move(d)]() mutable {
            // pages first, in steps under the shared side of the mmap lock: a query that starts meanwhile keeps page faulting
            // (the cammiq shell's first query read 43 instead of 24 ms with one munmap per array running beside it)
            cq::release_pages(d->k0.data(), d->k0.capacity() * sizeof(uint64_t));
            cq::release_pages(d->k1.data(), d->k1.capacity() * sizeof(uint64_t));
            cq::release_pages(d->vals.data(), d->vals.capacity() * sizeof(uint32_t));
            cq::release_pages(d->nodes.data(), d->nodes.capacity() * sizeof(cq::Node));
            cq::release_pages(d->r1.data(), d->r1.capacity() * sizeof(uint32_t));
            cq::release_pages(d->r2.data(), d->r2.capacity() * sizeof(uint32_t));
            d.reset();
        }).detach();   // the thread holds the only reference
    } catch (...) {
        // could not start a thread: the closure died with the exception and freed everything here
    }
}

struct LoadTimer;
void warm_workspace(cq_index *ix, LoadTimer *lt = nullptr);   // below, next to the host-fed pipeline it prepares
int ensure_narrow(cq_index *ix, uint64_t nl);                 // below, with the narrow rcount fetch

double table_budget(int device)
{
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b)
        return 0.5 * (double)free_b;   // at most half of what is free
    return 1e30;
}

}  // namespace

extern "C" {

int cq_abi_version(void) { return CQ_ABI_VERSION; }

const char *cq_last_error(void) { return g_err.c_str(); }

int cq_index_load(const char *path_u, const char *path_d, int device, cq_index **out)
{
    if (!path_u || !out || device < CQ_DEVICE_NONE) return fail(CQ_ERR_ARG, "cq_index_load: bad argument");
    *out = nullptr;
    cq_index *ix = new (std::nothrow) cq_index();
    if (!ix) return fail(CQ_ERR_NOMEM, "out of memory");
    LoadTimer lt;
    int rc = prepare_host(path_u, path_d, device >= 0 ? table_budget(device) : 1e30, ix->H, lt, device);
    if (rc != CQ_OK) { delete ix; return rc; }
    ix->device = device;
    if (device >= 0) {
        rc = upload(ix);
        lt.lap("upload");
        if (rc != CQ_OK) { release_device(ix); delete ix; return rc; }
        warm_workspace(ix, &lt);            // before the host image goes back: its munmap and these allocations contend for the mmap lock
        lt.lap("workspace (rest)");
        drop_host_image(*ix->H);
        lt.lap("release host image");
    }
    *out = ix;
    return CQ_OK;
}

int cq_index_get_info(const cq_index *ix, cq_index_info *info)
{
    if (!ix || !info) return fail(CQ_ERR_ARG, "cq_index_get_info: NULL argument");
    memset(info, 0, sizeof *info);
    const HostIndex &H = *ix->H;
    info->abi_version = CQ_ABI_VERSION;
    info->hash_len = H.img.hash_len;
    info->max_refid = H.img.max_refid;
    info->device = ix->device;
    for (int t = 0; t < 2; t++) {
        info->doubly_flag[t] = H.doubly_flag[t];
        info->n_leaves[t] = H.img.n_leaves[t];
        info->n_file_buckets[t] = H.n_file_buckets[t];
    }
    info->n_trie_nodes = H.n_trie_nodes;
    info->n_keys = H.img.n_keys;
    info->n_table_buckets = H.img.n_buckets_alloc;
    info->n_overflowed = H.img.n_overflowed;
    info->max_chain = H.img.max_chain;
    info->device_bytes = ix->device_bytes;
    info->reserved_ = H.from_cache ? 1u : 0u;
    info->minimizer_len = H.img.minimizer_len;
    return CQ_OK;
}

int cq_index_leaves(const cq_index *ix, int table, cq_leaf *out)
{
    if (!ix || !out || table < 0 || table > 1) return fail(CQ_ERR_ARG, "cq_index_leaves: bad argument");
    const auto &lv = ix->H->tab[table].leaves;
    if (!lv.empty()) memcpy(out, lv.data(), lv.size() * sizeof(cq_leaf));
    return CQ_OK;
}

int cq_index_probe(const cq_index *ix, uint64_t hv, uint32_t *code_u, uint32_t *code_d, uint32_t *chain)
{
    if (!ix || !code_u || !code_d) return fail(CQ_ERR_ARG, "cq_index_probe: NULL argument");
    if (!ix->H->img.table) return fail(CQ_ERR_ARG, "cq_index_probe: host image was released");
    cq::image_lookup(ix->H->img, hv, *code_u, *code_d, chain);
    return CQ_OK;
}

void cq_index_free(cq_index *ix)
{
    if (!ix) return;
    release_device(ix);
    delete ix;
}

uint64_t cq_counter_words(uint32_t n_genomes) { return 2ull * ((uint64_t)n_genomes + 1) + CQ_CTR_EXTRA; }

int cq_host_alloc(void **p, size_t bytes)
{
    if (!p) return fail(CQ_ERR_ARG, "cq_host_alloc: NULL argument");
    *p = nullptr;
    CQ_HIP(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
    return CQ_OK;
}

void cq_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

}  // extern "C"

namespace {
// The slow-path list a launch hands its over-full reads to: the handle's own (cq_query_device: one launch after the
// other on the caller's stream), or a staging slot's (host-fed pipeline: the kernels of consecutive chunks overlap).
struct OvfRef { uint32_t **list; uint32_t **count; uint64_t *cap; };
int query_device_impl(cq_index *ix, int mode, const uint32_t *d_packed, const uint8_t *d_lens, uint64_t n_reads, uint32_t stride_words,
                      uint32_t max_len, uint32_t n_genomes, uint64_t *d_counters, uint32_t *d_rcount, hipStream_t st, OvfRef ovf,
                      const uint8_t *d_tight = nullptr, uint32_t tight_sb = 0);
}  // namespace

extern "C" {

int cq_query_device(cq_index *ix, int mode, const uint32_t *d_packed, const uint8_t *d_lens,
                    uint64_t n_reads, uint32_t stride_words, uint32_t max_len, uint32_t n_genomes,
                    uint64_t *d_counters, uint32_t *d_rcount, void *stream)
{
    if (!ix) return fail(CQ_ERR_ARG, "cq_query_device: NULL handle");
    return query_device_impl(ix, mode, d_packed, d_lens, n_reads, stride_words, max_len, n_genomes, d_counters, d_rcount, (hipStream_t)stream,
                             OvfRef{&ix->d_ovf_list, &ix->d_ovf_count, &ix->ovf_cap});
}

}  // extern "C"

namespace {
int query_device_impl(cq_index *ix, int mode, const uint32_t *d_packed, const uint8_t *d_lens, uint64_t n_reads, uint32_t stride_words,
                      uint32_t max_len, uint32_t n_genomes, uint64_t *d_counters, uint32_t *d_rcount, hipStream_t st, OvfRef ovf,
                      const uint8_t *d_tight, uint32_t tight_sb)
{
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only (CQ_DEVICE_NONE); no CPU classify path exists");
    if (mode != CQ_MODE_P && mode != CQ_MODE_SC) return fail(CQ_ERR_ARG, "cq_query_device: unknown mode");
    if (!d_counters || (n_reads && ((!d_packed && !d_tight) || !d_lens)) || stride_words == 0 || stride_words > 16)
        return fail(CQ_ERR_ARG, "cq_query_device: bad argument");
    if (n_reads > 0x7FFFFFFFull) return fail(CQ_ERR_ARG, "cq_query_device: more than 2^31-1 reads in one call");
    const cq::FlatImage &img = ix->H->img;
    if (img.max_refid > n_genomes)
        return fail(CQ_ERR_RANGE, "index holds refID " + std::to_string(img.max_refid) + " > n_genomes");
    if (n_reads == 0) return CQ_OK;
    CQ_HIP(hipSetDevice(ix->device));
    if (!*ovf.count)   // [0] reads on the slow-path list; from word kWorkStripeWords on: the kernel's striped work counter
        CQ_HIP(hipMalloc((void **)ovf.count, (size_t)(1 + cq::kWorkStripes) * cq::kWorkStripeWords * sizeof(uint32_t)));
    if (*ovf.cap < n_reads) {   // grow the slow-path list (outside steady state)
        CQ_HIP(hipDeviceSynchronize());   // an earlier launch may still be using the old list
        if (*ovf.list) CQ_HIP(hipFree(*ovf.list));
        *ovf.list = nullptr;
        CQ_HIP(hipMalloc((void **)ovf.list, n_reads * sizeof(uint32_t)));
        *ovf.cap = n_reads;
    }
    CQ_HIP(hipMemsetAsync(*ovf.count, 0, sizeof(uint32_t), st));
    const uint32_t h = img.hash_len;
    if (max_len == 0 || max_len > stride_words * 16) max_len = stride_words * 16;
    if (max_len > 255) max_len = 255;
    cq::QueryArgs a{};
    a.packed = d_tight ? nullptr : d_packed;
    a.tight = d_tight;          // rows as they crossed the link: the kernel's staging widens them (host-fed tight door)
    a.tight_sb = tight_sb;
    a.lens = d_lens;
    a.n_reads = n_reads;
    a.stride_words = stride_words;
    a.wmax = max_len >= h ? max_len - h + 1 : 1;
    a.n_genomes = n_genomes;
    a.mode = mode;
    a.counters = d_counters;
    a.rcount = (mode == CQ_MODE_P) ? d_rcount : nullptr;
    a.ovf_list = *ovf.list;
    a.ovf_count = *ovf.count;
    // the tail of every launch's sub-tiles is handed out dynamically (CAMMIQ_DYNAMIC=0: stride only, A/B knob)
    a.work_counter = (getenv("CAMMIQ_DYNAMIC") && atoi(getenv("CAMMIQ_DYNAMIC")) == 0) ? nullptr : *ovf.count + cq::kWorkStripeWords;
    a.ovf_cap = (uint32_t)*ovf.cap;
    a.pair_keys = ix->d_pair_keys;
    a.pair_cnts = ix->d_pair_cnts;
    a.pair_cap = ix->pair_cap;
    a.stamps = ix->d_stamps;   // only written by diagnostic (CQ_STAMPS) builds
    CQ_HIP(cq::launch_classify(ix->dev, a, ix->n_cus, st, ix->ev0, ix->ev_mid, ix->ev1, &ix->last_launch));
    ix->ev_valid = true;
    return CQ_OK;
}
}  // namespace

extern "C" {

int cq_last_launch_info(cq_index *ix, cq_launch_info *out)
{
    if (!ix || !out) return fail(CQ_ERR_ARG, "cq_last_launch_info: NULL argument");
    if (!ix->ev_valid) return fail(CQ_ERR_ARG, "cq_last_launch_info: no kernel has been launched on this handle");
    const cq::LaunchInfo &l = ix->last_launch;
    out->reads_per_subtile = l.reads_per_subtile;
    out->hit_slots = l.hit_slots;
    out->lds_hist = l.lds_hist;
    out->fixed_shape = l.fixed_shape;
    out->fixed_hash_len = l.fixed_h;
    out->fixed_read_len = l.fixed_read_len;
    out->blocks_per_cu = l.blocks_per_cu;
    out->minimizer_len = l.minimizer_len;
    return CQ_OK;
}

int cq_calibrate(cq_index *ix, cq_calibration *out)
{
    if (!ix || !out) return fail(CQ_ERR_ARG, "cq_calibrate: NULL argument");
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only");
    memset(out, 0, sizeof *out);
    const auto t_begin = std::chrono::steady_clock::now();
    CQ_HIP(hipSetDevice(ix->device));
    const uint64_t n_units = ix->H->img.n_buckets_alloc * 4;   // 16-byte units of the table
    if (n_units == 0) return fail(CQ_ERR_ARG, "cq_calibrate: empty table");
    const cq::FlatImage &img = ix->H->img;
    const uint64_t n_atom = std::max<uint64_t>(img.n_leaves[0] + img.n_leaves[1], 1u << 20);   // an rcount-sized target
    const int max_grid = ix->n_cus * 8;
    uint32_t *d_atom = nullptr, *d_sink = nullptr;
    uint64_t *d_stamps = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<uint64_t> stamps(2 * (size_t)max_grid);
    int rc = CQ_OK;
#define CQ_HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = fail(CQ_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); goto done; } } while (0)
    {
        CQ_HIPC(hipMalloc((void **)&d_atom, n_atom * 4));
        CQ_HIPC(hipMemset(d_atom, 0, n_atom * 4));
        CQ_HIPC(hipMalloc((void **)&d_sink, 4));
        CQ_HIPC(hipMalloc((void **)&d_stamps, stamps.size() * 8));
        CQ_HIPC(hipEventCreate(&e0));
        CQ_HIPC(hipEventCreate(&e1));
        for (int mix = 0; mix < 2; mix++) {
            double best = 0.0, best_clock = 0.0;
            int best_occ = 0;
            for (int occ : {4, 6, 8}) {
                const int grid = ix->n_cus * occ;
                const int iters = 2048;   // 256 x grid x 2048 loads: ~20 ms at 40 G loads/s
                CQ_HIPC(cq::launch_calib_gather(mix != 0, (const uint4 *)ix->d_slots, n_units, 64, d_atom, n_atom, d_stamps, d_sink, grid, nullptr));   // warm
                CQ_HIPC(hipEventRecord(e0, nullptr));
                CQ_HIPC(cq::launch_calib_gather(mix != 0, (const uint4 *)ix->d_slots, n_units, iters, d_atom, n_atom, d_stamps, d_sink, grid, nullptr));
                CQ_HIPC(hipEventRecord(e1, nullptr));
                CQ_HIPC(hipEventSynchronize(e1));
                float ms = 0.f;
                CQ_HIPC(hipEventElapsedTime(&ms, e0, e1));
                CQ_HIPC(hipMemcpy(stamps.data(), d_stamps, (size_t)grid * 16, hipMemcpyDeviceToHost));
                std::vector<double> mhz;
                for (int b = 0; b < grid; b++)
                    if (stamps[2 * b + 1]) mhz.push_back(100.0 * (double)stamps[2 * b] / (double)stamps[2 * b + 1]);
                std::sort(mhz.begin(), mhz.end());
                const double rate = (double)grid * 256.0 * iters / (ms * 1e-3) / 1e9;
                if (rate > best) { best = rate; best_occ = occ; best_clock = mhz.empty() ? 0.0 : mhz[mhz.size() / 2]; }
            }
            if (mix) { out->gather16_mix_Glines_s = best; out->clock_MHz_mix = best_clock; out->mix_blocks_per_cu = best_occ; }
            else { out->gather16_Glines_s = best; out->clock_MHz_gather = best_clock; out->gather_blocks_per_cu = best_occ; }
        }
        {   // dependent loads, one wave per CU: far below what saturates the memory system, so lanes / rate = latency
            const int grid = ix->n_cus, iters = 2048;
            CQ_HIPC(cq::launch_calib_chase((const uint4 *)ix->d_slots, n_units, 16, d_stamps, d_sink, grid, nullptr));   // warm
            double best = 0.0, clock = 0.0;
            for (int rep = 0; rep < 3; rep++) {
                CQ_HIPC(hipEventRecord(e0, nullptr));
                CQ_HIPC(cq::launch_calib_chase((const uint4 *)ix->d_slots, n_units, iters, d_stamps, d_sink, grid, nullptr));
                CQ_HIPC(hipEventRecord(e1, nullptr));
                CQ_HIPC(hipEventSynchronize(e1));
                float ms = 0.f;
                CQ_HIPC(hipEventElapsedTime(&ms, e0, e1));
                const double rate = (double)grid * 64.0 * iters / (ms * 1e-3) / 1e9;
                if (rate > best) {
                    best = rate;
                    CQ_HIPC(hipMemcpy(stamps.data(), d_stamps, (size_t)grid * 16, hipMemcpyDeviceToHost));
                    std::vector<double> mhz;
                    for (int b = 0; b < grid; b++)
                        if (stamps[2 * b + 1]) mhz.push_back(100.0 * (double)stamps[2 * b] / (double)stamps[2 * b + 1]);
                    std::sort(mhz.begin(), mhz.end());
                    clock = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
                }
            }
            out->chase16_Glines_s = best;
            out->chase_latency_ns = best > 0.0 ? (double)grid * 64.0 / best : 0.0;   // lanes in flight / (1e9 loads per second) = ns
            out->clock_MHz_chase = clock;
        }
    }
done:
#undef CQ_HIPC
    if (d_atom) (void)hipFree(d_atom);
    if (d_sink) (void)hipFree(d_sink);
    if (d_stamps) (void)hipFree(d_stamps);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    out->table_bytes = (double)n_units * 16.0;
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return rc;
}

int cq_last_kernel_times(cq_index *ix, float *fast_ms, float *slow_ms)
{
    if (!ix) return fail(CQ_ERR_ARG, "cq_last_kernel_times: NULL argument");
    if (!ix->ev_valid) return fail(CQ_ERR_ARG, "cq_last_kernel_times: no kernel has been launched on this handle");
    CQ_HIP(hipEventSynchronize(ix->ev1));
    float a = 0.f, b = 0.f;
    CQ_HIP(hipEventElapsedTime(&a, ix->ev0, ix->ev_mid));
    CQ_HIP(hipEventElapsedTime(&b, ix->ev_mid, ix->ev1));
    if (fast_ms) *fast_ms = a;
    if (slow_ms) *slow_ms = b;
    return CQ_OK;
}

int cq_last_kernel_ms(cq_index *ix, float *ms)
{
    if (!ms) return fail(CQ_ERR_ARG, "cq_last_kernel_ms: NULL argument");
    float a = 0.f, b = 0.f;
    int rc = cq_last_kernel_times(ix, &a, &b);
    if (rc != CQ_OK) return rc;
    *ms = a + b;
    return CQ_OK;
}

int cq_pairs_reserve(cq_index *ix, uint64_t n_slots)
{
    if (!ix) return fail(CQ_ERR_ARG, "cq_pairs_reserve: NULL handle");
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only");
    if (n_slots > (1ull << 30)) return fail(CQ_ERR_LIMIT, "cq_pairs_reserve: more than 2^30 slots");
    uint32_t s = 16;
    while (s < n_slots) s <<= 1;
    CQ_HIP(hipSetDevice(ix->device));
    CQ_HIP(hipDeviceSynchronize());
    if (s == ix->pair_cap) return pairs_clear(ix);
    return pairs_alloc(ix, s);
}

int cq_pairs_fetch(cq_index *ix, uint32_t *pair_a, uint32_t *pair_b, uint64_t *pair_cnt,
                   uint64_t pair_cap, uint64_t *n_pairs)
{
    if (!ix || !n_pairs) return fail(CQ_ERR_ARG, "cq_pairs_fetch: NULL argument");
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only");
    CQ_HIP(hipSetDevice(ix->device));
    CQ_HIP(hipDeviceSynchronize());
    const size_t cap = ix->pair_cap;
    std::vector<uint64_t> k(cap), c(cap);
    CQ_HIP(hipMemcpy(k.data(), ix->d_pair_keys, cap * 8, hipMemcpyDeviceToHost));
    CQ_HIP(hipMemcpy(c.data(), ix->d_pair_cnts, cap * 8, hipMemcpyDeviceToHost));
    const bool want = pair_a && pair_b && pair_cnt;
    uint64_t n = 0;
    for (size_t i = 0; i < cap; i++) {
        if (k[i] == CQ_EMPTY_KEY) continue;
        if (want && n < pair_cap) {
            pair_a[n] = (uint32_t)(k[i] >> 32);
            pair_b[n] = (uint32_t)k[i];
            pair_cnt[n] = c[i];
        }
        n++;
    }
    *n_pairs = n;
    if (!want) return CQ_OK;                     // count only: nothing copied, nothing cleared
    if (n > pair_cap)
        return fail(CQ_ERR_LIMIT, "more distinct pairs than pair_cap (nothing was cleared: call cq_pairs_fetch again with larger arrays)");
    return pairs_clear(ix);
}

}  // extern "C"

namespace {

// The widening kernel's queue: high priority, so that it gets a hardware queue of its own and runs in the wave
// slots the classify kernel of the chunk before leaves free, instead of queueing up behind it.
hipError_t make_widen_stream(cq_index *ix)
{
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void)hipGetLastError(); }
    return hipStreamCreateWithPriority(&ix->s_widen, hipStreamNonBlocking, hi);
}

bool is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    const bool pinned = p && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();   // plain malloc memory: the attribute query leaves a sticky error behind
    return pinned;
}

// Grow one staging slot: device rows for n reads of sw words, and (host_too) the pinned host side.
int slot_reserve(cq_index::Slot &sl, uint64_t n, uint32_t sw, bool host_too)
{
    const size_t words = (size_t)n * sw;
    if (sl.cap_words_d < words) {
        if (sl.d_packed) (void)hipFree(sl.d_packed);
        sl.d_packed = nullptr; sl.cap_words_d = 0;
        CQ_HIP(hipMalloc((void **)&sl.d_packed, words * 4));
        sl.cap_words_d = words;
    }
    if (sl.cap_reads_d < n) {
        if (sl.d_lens) (void)hipFree(sl.d_lens);
        sl.d_lens = nullptr; sl.cap_reads_d = 0;
        CQ_HIP(hipMalloc((void **)&sl.d_lens, n));
        sl.cap_reads_d = n;
    }
    if (host_too && sl.cap_words_h < words) {
        if (sl.h_packed) (void)hipHostFree(sl.h_packed);
        sl.h_packed = nullptr; sl.cap_words_h = 0;
        CQ_HIP(hipHostMalloc((void **)&sl.h_packed, words * 4, hipHostMallocDefault));
        sl.cap_words_h = words;
    }
    if (host_too && sl.cap_reads_h < n) {
        if (sl.h_lens) (void)hipHostFree(sl.h_lens);
        sl.h_lens = nullptr; sl.cap_reads_h = 0;
        CQ_HIP(hipHostMalloc((void **)&sl.h_lens, n, hipHostMallocDefault));
        sl.cap_reads_h = n;
    }
    // the slot's own slow-path list and work counters (query_device_impl would grow them on first use -- behind a
    // hipDeviceSynchronize, inside the first query's bracket: 34 instead of 22 ms "Time for query" in the cammiq shell)
    if (!sl.d_ovf_count) CQ_HIP(hipMalloc((void **)&sl.d_ovf_count, (size_t)(1 + cq::kWorkStripes) * cq::kWorkStripeWords * sizeof(uint32_t)));
    if (sl.ovf_cap < n) {
        if (sl.d_ovf_list) (void)hipFree(sl.d_ovf_list);
        sl.d_ovf_list = nullptr; sl.ovf_cap = 0;
        CQ_HIP(hipMalloc((void **)&sl.d_ovf_list, n * sizeof(uint32_t)));
        sl.ovf_cap = n;
    }
    if (!sl.copied) CQ_HIP(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
    if (!sl.copied_lens) CQ_HIP(hipEventCreateWithFlags(&sl.copied_lens, hipEventDisableTiming));
    if (!sl.widened) CQ_HIP(hipEventCreateWithFlags(&sl.widened, hipEventDisableTiming));
    if (!sl.done) CQ_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    return CQ_OK;
}

// Everything a first query would otherwise allocate while the clock of "Time for query" runs (about 25 ms of
// hipMalloc / hipHostMalloc / stream and event creation for a 10 M-read FASTQ): streams, events, the two bounce
// buffers, rcount, a counter block for up to 16 383 genomes, the slow-path list and the three device slots of one
// 2 M-read chunk of 128-base rows.  Larger needs grow on demand as before.  Failures here are not errors.
void warm_workspace(cq_index *ix, LoadTimer *lt)
{
    auto lap = [&](const char *w) { if (lt) lt->lap(w); };
    if (ix->device < 0 || hipSetDevice(ix->device) != hipSuccess) return;
    const cq::FlatImage &img = ix->H->img;
    if (!ix->s_copy) (void)hipStreamCreateWithFlags(&ix->s_copy, hipStreamNonBlocking);
    if (!ix->s_copy2) (void)hipStreamCreateWithFlags(&ix->s_copy2, hipStreamNonBlocking);
    if (!ix->s_comp) (void)hipStreamCreateWithFlags(&ix->s_comp, hipStreamNonBlocking);
    lap("  streams");
    if (!ix->s_widen) (void)make_widen_stream(ix);
    lap("  priority stream");
    for (auto &sl : ix->slot) {
        (void)slot_reserve(sl, kChunk, 8, true);   // with the page-locked host side ASCII reads are packed into (cq_query): 19.8 instead of 25.9 ms per 10 M reads
        if (!sl.d_tight && hipMalloc((void **)&sl.d_tight, kChunk * 32 + 16) == hipSuccess) sl.cap_tight = kChunk * 32 + 16;   // tight rows of up to 128 bases
    }
    lap("  staging slots");
    for (int b = 0; b < 2; b++) {
        if (!ix->h_bounce[b] && hipHostMalloc(&ix->h_bounce[b], kBounce, hipHostMallocDefault) != hipSuccess) ix->h_bounce[b] = nullptr;
        if (!ix->ev_bounce[b]) (void)hipEventCreateWithFlags(&ix->ev_bounce[b], hipEventDisableTiming);
    }
    lap("  bounce buffers");
    const uint64_t nl = img.n_leaves[0] + img.n_leaves[1], cw = cq_counter_words(16383);
    if (nl && !ix->d_rc && hipMalloc((void **)&ix->d_rc, nl * 4) == hipSuccess) ix->rc_cap = nl;
    if (!ix->d_ctr && hipMalloc((void **)&ix->d_ctr, cw * 8) == hipSuccess) ix->ctr_cap = cw;
    if (!ix->d_ovf_list && hipMalloc((void **)&ix->d_ovf_list, kChunk * sizeof(uint32_t)) == hipSuccess) ix->ovf_cap = kChunk;
    lap("  rcount, counters");
    (void)ensure_narrow(ix, nl);
    lap("  narrow rcount path");
    (void)hipGetLastError();
}

// Where the reads of a host-fed query come from: ASCII as FqReader::readFastq leaves them
// (bases + offsets), or rows already packed by cq_pack_reads (packed + lens).
struct Feed {
    const uint8_t *bases = nullptr;
    const uint64_t *offsets = nullptr;
    const uint8_t *const *ptrs = nullptr;   // ASCII reads as one pointer + one length byte (lens) per read: FqReader's own arrays
    const uint32_t *packed = nullptr;
    const uint8_t *tight = nullptr;   // rows at a byte stride of sb (cq_pack_reads_tight); sw = ceil(sb / 4)
    const uint8_t *lens = nullptr;
    uint32_t sw = 0, sb = 0, max_len = 0;
};

int query_checks(const cq_index *ix, int mode, uint32_t n_genomes, const cq_counts *out, const char *who)
{
    if (!out->cnt_u || !out->cnt_d) return fail(CQ_ERR_ARG, std::string(who) + ": cnt_u / cnt_d must be provided");
    const cq::FlatImage &img = ix->H->img;
    if (mode != CQ_MODE_P && mode != CQ_MODE_SC) return fail(CQ_ERR_ARG, std::string(who) + ": unknown mode");
    if (mode == CQ_MODE_P && ((!out->rcount_u && img.n_leaves[0]) || (!out->rcount_d && img.n_leaves[1])))
        return fail(CQ_ERR_ARG, std::string(who) + ": rcount_u / rcount_d are mandatory in CQ_MODE_P (the ILP reads them)");
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only (CQ_DEVICE_NONE); no CPU classify path exists");
    if (img.max_refid > n_genomes)
        return fail(CQ_ERR_RANGE, "index holds refID " + std::to_string(img.max_refid) + " > n_genomes");
    return CQ_OK;
}

int narrow_start(cq_index *ix, uint64_t n_u, uint64_t n_d, uint32_t *dst_u, uint32_t *dst_d, const uint32_t *src = nullptr);   // below (rcount's narrow way back)

unsigned pack_threads()
{
    if (const char *v = getenv("CAMMIQ_PACK_THREADS")) return (unsigned)std::min(64, std::max(1, atoi(v)));   // tuning knob (<= 64: per-worker result slots)
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hw ? hw : 1u, 32u));
}

// The handle's packers (nullptr: none to be had -- the caller packs on its own thread).
PackPool *ensure_pack_pool(cq_index *ix)
{
    const unsigned want = pack_threads();
    if (ix->pack_pool && ix->pack_pool->th.size() + 1 != want) { ix->pack_pool->stop(); delete ix->pack_pool; ix->pack_pool = nullptr; }   // knob changed
    if (!ix->pack_pool) {
        PackPool *p = new (std::nothrow) PackPool();
        if (!p) return nullptr;
        try { p->start(want - 1); } catch (...) { p->stop(); delete p; return nullptr; }
        ix->pack_pool = p;
    }
    return ix->pack_pool;
}

// Classify reads [lo, hi) of `f` on ix's device into the handle's own counter block (d_ctr) and
// rcount array (d_rc), both zeroed first: resetCounters + query64_* (query.cpp:1820-1840, 458-1080).
// Chunks of 2 M reads rotate over four staging slots:
//   CPU    pack(c+1) ........ pack(c+2) ........          (ASCII feed only)
//   copy            H2D(c+1) ...........H2D(c+2)
//   comp   kernel(c) ........          kernel(c+2) ......     even chunks
//   comp2            kernel(c+1) ........                      odd chunks: c + 1 fills the CUs while c drains
// out != nullptr (one device, CQ_MODE_P): rcount's narrow way back is started behind the last kernel BEFORE the host
// waits for the kernels (narrow_start), so no host round trip sits between the two; fetch_counts finishes it.
// Returns with every classify kernel complete on the device.
int classify_range(cq_index *ix, int mode, const Feed &f, uint64_t lo, uint64_t hi, uint32_t n_genomes, const cq_counts *out = nullptr)
{
    const cq::FlatImage &img = ix->H->img;
    CQ_HIP(hipSetDevice(ix->device));
    if (!ix->s_copy) CQ_HIP(hipStreamCreateWithFlags(&ix->s_copy, hipStreamNonBlocking));
    if (!ix->s_copy2) CQ_HIP(hipStreamCreateWithFlags(&ix->s_copy2, hipStreamNonBlocking));
    if (!ix->s_comp) CQ_HIP(hipStreamCreateWithFlags(&ix->s_comp, hipStreamNonBlocking));
    if (!ix->s_comp2) CQ_HIP(hipStreamCreateWithFlags(&ix->s_comp2, hipStreamNonBlocking));
    if (!ix->ev_zeroed) CQ_HIP(hipEventCreateWithFlags(&ix->ev_zeroed, hipEventDisableTiming));
    if (!ix->ev_comp2) CQ_HIP(hipEventCreateWithFlags(&ix->ev_comp2, hipEventDisableTiming));
    if (!ix->s_widen) CQ_HIP(make_widen_stream(ix));
    const uint64_t cw = cq_counter_words(n_genomes);
    const uint64_t nl = img.n_leaves[0] + img.n_leaves[1];
    if (ix->ctr_cap < cw) {
        if (ix->d_ctr) (void)hipFree(ix->d_ctr);
        ix->d_ctr = nullptr; ix->ctr_cap = 0;
        CQ_HIP(hipMalloc((void **)&ix->d_ctr, cw * 8));
        ix->ctr_cap = cw;
    }
    CQ_HIP(hipMemsetAsync(ix->d_ctr, 0, cw * 8, ix->s_comp));
    uint32_t *d_rc = nullptr;
    if (mode == CQ_MODE_P && nl) {
        if (ix->rc_cap < nl) {
            if (ix->d_rc) (void)hipFree(ix->d_rc);
            ix->d_rc = nullptr; ix->rc_cap = 0;
            CQ_HIP(hipMalloc((void **)&ix->d_rc, nl * 4));
            ix->rc_cap = nl;
        }
        d_rc = ix->d_rc;
        CQ_HIP(hipMemsetAsync(d_rc, 0, nl * 4, ix->s_comp));
    }
    // Two compute queues, even and odd chunks: a chunk's kernel is a persistent grid that ramps up and drains; run one
    // after the other on ONE queue, 24 chunks cost 8-15 % more than the same reads in one launch (measured: 0.93-1.0 ms
    // per 2 M-read chunk where 50 M reads take 19.2-21.6 ms).  On two queues the next chunk's workgroups move into the
    // CUs as the previous chunk's leave.  Every output is an atomic sum, each slot has its own slow-path list.
    // (CAMMIQ_TWO_STREAMS=0: one queue, A/B knob.)
    const bool two = !(getenv("CAMMIQ_TWO_STREAMS") && atoi(getenv("CAMMIQ_TWO_STREAMS")) == 0);
    CQ_HIP(hipEventRecord(ix->ev_zeroed, ix->s_comp));
    if (two) CQ_HIP(hipStreamWaitEvent(ix->s_comp2, ix->ev_zeroed, 0));
    const bool ascii = f.packed == nullptr && f.tight == nullptr;
    const cq::ReadSource ascii_src{f.bases, f.offsets, f.ptrs, f.ptrs ? f.lens : nullptr};
    PackPool *packers = ascii ? ensure_pack_pool(ix) : nullptr;
    auto on_packers = [&](const std::function<void(unsigned, unsigned)> &fn) { if (packers) packers->run(fn); else fn(0u, 1u); };
    int rc = CQ_OK;
    uint64_t c = 0;
    PipeTrace *tr = ix->trace;
    auto tdev = [&](const char *n, uint64_t ch, hipStream_t st) { if (tr) tr->dev(n, (int)ch, st); };
    auto thost = [&](const char *n, uint64_t ch) { if (tr) tr->host(n, (int)ch); };
    tdev("query_start", 0, ix->s_comp);
    // inside the loop a failed HIP call ends the loop instead of returning: copies from the caller's memory may be in flight
#define CQ_HIPB(call) if (hipError_t e_ = (call)) { rc = fail(CQ_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); break; } else (void)0
    // Chunk schedule: equal chunks, the LAST one cut finer (1/2, 1/4, 1/4: CAMMIQ_CHUNK_TAIL=0 for equal chunks) -- where the
    // link bounds the bracket the last chunk's kernel runs after the last copy with nothing beside it.  A finer HEAD was
    // measured too and buys nothing: the kernels consume reads about as fast as the link delivers them, so the first 2 M
    // reads are done ~2 ms into the query either way, and four more chunks cost the host four more rounds of enqueueing
    // (28.9-29.7 ms against 28.2-29.0 ms, A/B in one process; profiles/r04_hostfed_*).
    std::vector<uint64_t> sched;
    {
        const int tail_on = getenv("CAMMIQ_CHUNK_TAIL") ? atoi(getenv("CAMMIQ_CHUNK_TAIL")) : 1;
        const uint64_t chunk = chunk_reads();
        uint64_t left = hi - lo;
        std::vector<uint64_t> tail;
        // (A doubling head, 2-chunk pieces in the body and a halving tail -- 17 copies and launches instead of 27 -- was measured
        // too: 26.76-26.90 against 26.74-26.78 ms, six interleaved rounds; 4 M-read chunks 26.86-26.96: nothing, not kept.
        // profiles/r04_hostfed_ab_geometric_schedule.txt)
        {
            if (tail_on && left >= 3 * chunk && !ascii) {
                tail = {chunk / 2, chunk / 4, chunk - chunk / 2 - chunk / 4};
                left -= chunk;
            }
            while (left > 0) { const uint64_t n = std::min(chunk, left); sched.push_back(n); left -= n; }
        }
        sched.insert(sched.end(), tail.begin(), tail.end());
    }
    // Packed feeds: the shortest and the longest length of every chunk, found by a thread that runs ahead of the loop
    // (2 MB of lengths per chunk: ~0.2 ms of the ~0.5 ms the host spends per chunk when it scans them itself, on a loop
    // that is within a factor of two of bounding the bracket).
    struct Prescan {
        std::vector<uint32_t> lmin, lmax;
        std::atomic<size_t> ready{0};
        std::thread th;
        ~Prescan() { if (th.joinable()) th.join(); }
    } pre;
    if (!ascii && !sched.empty()) {
        pre.lmin.assign(sched.size(), 255);
        pre.lmax.assign(sched.size(), 0);
        const uint8_t *L = f.lens + lo;
        pre.th = std::thread([&pre, &sched, L] {
            uint64_t r0 = 0;
            for (size_t ci = 0; ci < sched.size(); r0 += sched[ci], ci++) {
                uint32_t mn = 255, mx = 0;
                for (uint64_t r = r0, e = r0 + sched[ci]; r < e; r++) { const uint32_t l = L[r]; mn = l < mn ? l : mn; mx = l > mx ? l : mx; }
                pre.lmin[ci] = mn; pre.lmax[ci] = mx;
                pre.ready.store(ci + 1, std::memory_order_release);
            }
        });
    }
    uint64_t c0 = lo;
    uint32_t ascii_next_longest = 0;   // ASCII feeds: the longest read of the chunk about to be packed
    for (size_t ci = 0; ci < sched.size() && rc == CQ_OK; c0 += sched[ci], ci++, c++) {
        cq_index::Slot &sl = ix->slot[c % kSlots];
        hipStream_t s_k = (two && (c & 1)) ? ix->s_comp2 : ix->s_comp;   // this chunk's kernels
        const uint64_t n = sched[ci];
        uint64_t max_len = f.max_len;
        uint32_t sw = f.sw;
        uint32_t sbc = f.sb;   // byte stride of this chunk's tight rows
        if (ascii) {   // longest read of the chunk (sizes the rows): on the packers, one thread would take ~1 ms per 2 M reads
            if (ci == 0) {   // (every later chunk's was found by the job that packed the chunk before it)
                uint32_t part[64] = {0};
                on_packers([&](unsigned t, unsigned nt) { part[t] = cq::longest_read(ascii_src, c0 + n * t / nt, c0 + n * (t + 1) / nt); });
                ascii_next_longest = *std::max_element(part, part + 64);
            }
            max_len = ascii_next_longest;
            sw = cq_pack_stride_words((uint32_t)max_len);
            sbc = cq_pack_stride_bytes((uint32_t)max_len);
        }
        thost("slot_wait", c);
        if (sl.done) { CQ_HIPB(hipEventSynchronize(sl.done)); }   // the kernel that last used this slot
        thost("slot_free", c);
        rc = slot_reserve(sl, n, sw, ascii);
        if (rc != CQ_OK) break;
        const uint32_t *src_rows = nullptr;
        const uint8_t *src_lens = nullptr;
        uint32_t ascii_mn = 255, ascii_mx = 0;
        if (ascii) {
            // ASCII -> TIGHT rows (ceil(longest / 4) bytes per read) straight into the slot's page-locked buffer, on the handle's
            // packers; from here on the chunk is a tight-row chunk (25 instead of 28 bytes per 100-bp read on the link, widened by
            // the classify kernel's own staging)
            uint8_t *rows = (uint8_t *)sl.h_packed;   // n * sw words were reserved: >= n * sbc bytes
            uint32_t mn[64], mx[64], nxt[64];
            for (int i = 0; i < 64; i++) { mn[i] = 255; mx[i] = 0; nxt[i] = 0; }
            const uint64_t n_next = ci + 1 < sched.size() ? sched[ci + 1] : 0, c_next = c0 + n;   // the same job sizes the next chunk's rows
            thost("pack_begin", c);
            on_packers([&](unsigned t, unsigned nt) {
                const uint64_t a = n * t / nt, b = n * (t + 1) / nt;
                uint64_t sk = 0;
                uint32_t lo_len = 255, hi_len = 0;
                cq::pack_tight_slice(ascii_src, c0 + a, c0 + b, img.hash_len, sbc, rows + a * sbc, sl.h_lens + a, &sk, &lo_len, &hi_len);
                if (a != b) { mn[t] = lo_len; mx[t] = hi_len; }
                if (n_next) nxt[t] = cq::longest_read(ascii_src, c_next + n_next * t / nt, c_next + n_next * (t + 1) / nt);
            });
            thost("pack_end", c);
            ascii_mn = *std::min_element(mn, mn + 64);
            ascii_mx = *std::max_element(mx, mx + 64);
            ascii_next_longest = *std::max_element(nxt, nxt + 64);
            src_lens = sl.h_lens;
        } else {
            src_rows = f.tight ? nullptr : f.packed + (size_t)c0 * sw;
            src_lens = f.lens + c0;
        }
        // (Rows in pageable memory are handed to the runtime as they are: staging them through the slot's page-locked
        // buffer with eight memcpy threads made the packed door 10 % faster and the cammiq shell's query 3-5 ms slower
        // -- same box, A/B/A/B -- so it is not done.)
        const uint8_t *src_tight = f.tight ? f.tight + (size_t)c0 * f.sb : (ascii ? (const uint8_t *)sl.h_packed : nullptr);
        const bool tight_chunk = src_tight != nullptr;
        if (tight_chunk) {   // fewer bytes over the link: the rows arrive tight and are widened on the device
            if (sl.cap_tight < (size_t)n * sbc + 16) {   // + 16: the kernel's last lane reads whole 16-byte pieces
                if (sl.d_tight) (void)hipFree(sl.d_tight);
                sl.d_tight = nullptr; sl.cap_tight = 0;
                CQ_HIPB(hipMalloc((void **)&sl.d_tight, (size_t)n * sbc + 16));
                sl.cap_tight = (size_t)n * sbc + 16;
            }
            tdev("h2d_begin", c, ix->s_copy);
            CQ_HIPB(hipMemcpyAsync(sl.d_tight, src_tight, (size_t)n * sbc, hipMemcpyHostToDevice, ix->s_copy));
        } else {
            tdev("h2d_begin", c, ix->s_copy);
            CQ_HIPB(hipMemcpyAsync(sl.d_packed, src_rows, (size_t)n * sw * 4, hipMemcpyHostToDevice, ix->s_copy));
        }
        CQ_HIPB(hipEventRecord(sl.copied, ix->s_copy));
        tdev("h2d_end", c, ix->s_copy);
        // The lengths.  The lane grid is sized by the longest read: take it from the lengths themselves (a max_len that
        // is too small would silently drop windows) and refuse lengths the rows cannot hold.  The row copy is in flight
        // by now -- in front of it a scan delayed every transfer (measured: 1 540 -> 1 325 Mreads/s).
        // A chunk whose reads all have ONE length (what a sequencing run delivers) does not send its lengths at all:
        // the device array is filled on the device (2 MB less per 50 MB chunk on the link that bounds the bracket).
        bool lens_uniform = false;
        uint32_t longest = 0;
        if (!ascii) {
            while (pre.ready.load(std::memory_order_acquire) <= ci) _mm_pause();
            longest = pre.lmax[ci];
            const uint32_t shortest = pre.lmin[ci];
            if (longest > (f.tight ? f.sb * 4u : sw * 16u)) { rc = fail(CQ_ERR_ARG, "cq_query_packed: a length exceeds what a row of this stride holds"); break; }
            max_len = std::max<uint64_t>(max_len, longest);
            const bool fill_off = getenv("CAMMIQ_LENS_FILL") && atoi(getenv("CAMMIQ_LENS_FILL")) == 0;   // A/B knob
            lens_uniform = shortest == longest && !fill_off;
        } else {
            longest = ascii_mx;
            const bool fill_off = getenv("CAMMIQ_LENS_FILL") && atoi(getenv("CAMMIQ_LENS_FILL")) == 0;   // A/B knob
            lens_uniform = ascii_mn == ascii_mx && !fill_off;
        }
        // the small copy of the lengths goes down its own queue: behind the rows it would put a bubble between
        // every two large transfers
        if (lens_uniform) CQ_HIPB(hipMemsetAsync(sl.d_lens, (int)longest, n, ix->s_copy2));
        else CQ_HIPB(hipMemcpyAsync(sl.d_lens, src_lens, n, hipMemcpyHostToDevice, ix->s_copy2));
        CQ_HIPB(hipEventRecord(sl.copied_lens, ix->s_copy2));
        // Tight rows are widened by the classify kernel's own staging (a byte image of the sub-tile through LDS): no widening
        // kernel sharing the GPU with the classify kernels, no second copy of the rows in HBM.  CAMMIQ_FUSED_WIDEN=0: the
        // separate widening kernel on its high-priority queue, as before (A/B knob).
        const bool fused = tight_chunk && !(getenv("CAMMIQ_FUSED_WIDEN") && atoi(getenv("CAMMIQ_FUSED_WIDEN")) == 0);
        if (fused) {
            CQ_HIPB(hipStreamWaitEvent(s_k, sl.copied, 0));
        } else if (tight_chunk) {   // widen on a queue of its own: in front of the classify kernel it would cost the chunk ~0.1 ms
            CQ_HIPB(hipStreamWaitEvent(ix->s_widen, sl.copied, 0));
            tdev("widen_begin", c, ix->s_widen);
            CQ_HIPB(cq::launch_widen_rows(sl.d_tight, sbc, sl.d_packed, sw, n, ix->s_widen));
            CQ_HIPB(hipEventRecord(sl.widened, ix->s_widen));
            tdev("widen_end", c, ix->s_widen);
            CQ_HIPB(hipStreamWaitEvent(s_k, sl.widened, 0));
        } else
            CQ_HIPB(hipStreamWaitEvent(s_k, sl.copied, 0));
        CQ_HIPB(hipStreamWaitEvent(s_k, sl.copied_lens, 0));
        tdev("kernel_begin", c, s_k);
        rc = query_device_impl(ix, mode, sl.d_packed, sl.d_lens, n, sw, (uint32_t)max_len, n_genomes, ix->d_ctr, d_rc, s_k,
                               OvfRef{&sl.d_ovf_list, &sl.d_ovf_count, &sl.ovf_cap}, fused ? sl.d_tight : nullptr, fused ? sbc : 0u);
        if (rc != CQ_OK) break;
        CQ_HIPB(hipEventRecord(sl.done, s_k));
        tdev("kernel_end", c, s_k);
        thost("enqueued", c);
    }
#undef CQ_HIPB
    if (rc == CQ_OK && two) {   // the even queue takes the odd one's kernels in: what follows on s_comp follows every kernel
        if (hipEventRecord(ix->ev_comp2, ix->s_comp2) != hipSuccess || hipStreamWaitEvent(ix->s_comp, ix->ev_comp2, 0) != hipSuccess)
            rc = fail(CQ_ERR_HIP, "joining the two compute queues failed");
    }
    const bool early = !(getenv("CAMMIQ_EARLY_NARROW") && atoi(getenv("CAMMIQ_EARLY_NARROW")) == 0);   // A/B / debugging knob
    if (rc == CQ_OK && out && mode == CQ_MODE_P && d_rc && early)
        rc = narrow_start(ix, img.n_leaves[0], img.n_leaves[1], out->rcount_u, out->rcount_d);   // queued behind the last kernel, no host wait in between
    if (!ix->narrow_inflight) {   // (with rcount on its way the wait is narrow_finish's: it ends behind every kernel)
        if (hipStreamSynchronize(ix->s_comp) != hipSuccess && rc == CQ_OK) rc = fail(CQ_ERR_HIP, "classify kernel failed");
        if (two && hipStreamSynchronize(ix->s_comp2) != hipSuccess && rc == CQ_OK) rc = fail(CQ_ERR_HIP, "classify kernel failed");
        thost("kernels_done", c);
    }
    if (rc != CQ_OK) {   // an error may have left copies from the caller's memory in flight with no kernel behind them
        (void)hipStreamSynchronize(ix->s_copy);
        (void)hipStreamSynchronize(ix->s_copy2);
        (void)hipStreamSynchronize(ix->s_widen);
    }
    return rc;
}

// D2H of `bytes` from device memory into a caller array that may be pageable.  Pinned arrays
// (cq_host_alloc) take one direct copy; pageable ones go through two pinned bounce buffers, the copy
// of piece k+1 overlapping the memcpy of piece k -- a pageable hipMemcpy of the few hundred MB of
// rcount of a 1000-genome index runs at a fraction of the link rate otherwise.
int copy_out(cq_index *ix, void *dst, const void *d_src, size_t bytes)
{
    if (!bytes) return CQ_OK;
    if (is_pinned(dst) || bytes <= (1u << 20)) {
        CQ_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
        return CQ_OK;
    }
    for (int b = 0; b < 2; b++) {
        if (!ix->h_bounce[b]) CQ_HIP(hipHostMalloc(&ix->h_bounce[b], kBounce, hipHostMallocDefault));
        if (!ix->ev_bounce[b]) CQ_HIP(hipEventCreateWithFlags(&ix->ev_bounce[b], hipEventDisableTiming));
    }
    const size_t np = (bytes + kBounce - 1) / kBounce;
    auto piece = [&](size_t k) { return std::min(kBounce, bytes - k * kBounce); };
    CQ_HIP(hipMemcpyAsync(ix->h_bounce[0], d_src, piece(0), hipMemcpyDeviceToHost, ix->s_copy));
    CQ_HIP(hipEventRecord(ix->ev_bounce[0], ix->s_copy));
    for (size_t k = 0; k < np; k++) {
        if (k + 1 < np) {
            CQ_HIP(hipMemcpyAsync(ix->h_bounce[(k + 1) & 1], (const char *)d_src + (k + 1) * kBounce, piece(k + 1),
                                  hipMemcpyDeviceToHost, ix->s_copy));
            CQ_HIP(hipEventRecord(ix->ev_bounce[(k + 1) & 1], ix->s_copy));
        }
        CQ_HIP(hipEventSynchronize(ix->ev_bounce[k & 1]));
        // out of the bounce buffer on a few threads: one thread's memcpy (~10 GB/s into first-touched pages) would be
        // several times slower than the link
        const size_t pb = piece(k);
        const unsigned nt = pb >= (4u << 20) ? 4u : 1u;
        char *d = (char *)dst + k * kBounce;
        const char *sbuf = (const char *)ix->h_bounce[k & 1];
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++)
            th.emplace_back([=] { memcpy(d + pb * t / nt, sbuf + pb * t / nt, pb * (t + 1) / nt - pb * t / nt); });
        memcpy(d, sbuf, pb / nt);
        for (auto &x : th) x.join();
    }
    return CQ_OK;
}

// entries per segment of rcount's narrow way back (CAMMIQ_NARROW_SEG: test knob -- small segments take a small rcount
// through many flags; a multiple of 64, read when a handle sets the path up)
uint64_t narrow_seg()
{
    if (const char *v = getenv("CAMMIQ_NARROW_SEG")) return std::max<uint64_t>(64, strtoull(v, nullptr, 10) & ~63ull);
    return cq::kNarrowSeg;
}

// Device buffers, page-locked landing area, flags and worker threads of rcount's narrow way back, for nl leaves.
// CQ_OK with ix->pool == nullptr afterwards means "not available": the caller takes the plain copy.
int ensure_narrow(cq_index *ix, uint64_t nl)
{
    const bool off = getenv("CAMMIQ_RCOUNT_NARROW") && atoi(getenv("CAMMIQ_RCOUNT_NARROW")) == 0;   // A/B knob
    const uint64_t from = getenv("CAMMIQ_NARROW_FROM") ? strtoull(getenv("CAMMIQ_NARROW_FROM"), nullptr, 10) : kNarrowFrom;   // test knob
    if (off || nl < from || nl == 0) {   // not this way (any more): the caller sees no pool and takes the plain copy
        if (ix->pool) { ix->pool->stop(); delete ix->pool; ix->pool = nullptr; }
        return CQ_OK;
    }
    const uint64_t nseg = (nl + narrow_seg() - 1) / narrow_seg();
    const bool timing = getenv("CAMMIQ_LOAD_TIMING") != nullptr;
    auto t_n = std::chrono::steady_clock::now();
    auto nlap = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "[cq_index_load]     narrow: %-18s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_n).count());
        t_n = now;
    };
    if (ix->narrow_cap < nl) {
        if (ix->h_narrow) (void)hipHostFree(ix->h_narrow);
        ix->h_narrow = nullptr; ix->narrow_cap = 0;
        CQ_HIP(hipHostMalloc((void **)&ix->h_narrow, (nl + 63) / 64 * 64, hipHostMallocDefault));
        ix->narrow_cap = nl;
        nlap("pinned landing area");
    }
    if (ix->flags_cap < nseg + 1) {
        if (ix->h_flags) (void)hipHostFree(ix->h_flags);
        ix->h_flags = nullptr; ix->flags_cap = 0;
        CQ_HIP(hipHostMalloc((void **)&ix->h_flags, (nseg + 1) * sizeof(uint32_t), hipHostMallocDefault));
        memset(ix->h_flags, 0, (nseg + 1) * sizeof(uint32_t));
        ix->flags_cap = nseg + 1;
        ix->narrow_epoch = 0;
    }
    if (!ix->d_esc) {
        uint32_t cap = 1u << 20;                                       // 8 MB of (leaf, count) pairs
        if (const char *v = getenv("CAMMIQ_ESC_CAP")) cap = (uint32_t)std::max(1, atoi(v));   // test knob: force the fall-back
        CQ_HIP(hipMalloc((void **)&ix->d_esc, (size_t)cap * sizeof(uint2)));
        CQ_HIP(hipMalloc((void **)&ix->d_esc_count, sizeof(uint32_t)));
        CQ_HIP(hipMalloc((void **)&ix->d_blocks_done, sizeof(uint32_t)));
        CQ_HIP(hipMemset(ix->d_blocks_done, 0, sizeof(uint32_t)));
        CQ_HIP(hipHostMalloc((void **)&ix->h_esc, (size_t)cap * sizeof(uint2), hipHostMallocDefault));
        CQ_HIP(hipHostMalloc((void **)&ix->h_esc_count, sizeof(uint32_t), hipHostMallocDefault));
        ix->esc_cap = cap;
    }
    if (!ix->ev_narrow) CQ_HIP(hipEventCreateWithFlags(&ix->ev_narrow, hipEventDisableTiming));
    if (ix->pool && ix->pool->th.size() != widen_threads()) { ix->pool->stop(); delete ix->pool; ix->pool = nullptr; }   // knob changed
    if (!ix->pool) {
        WidenPool *p = new (std::nothrow) WidenPool();
        if (!p) return fail(CQ_ERR_NOMEM, "out of memory");
        try { p->start(widen_threads()); } catch (...) { p->stop(); delete p; return CQ_OK; }   // no threads to be had: plain copy
        ix->pool = p;
    }
    nlap("flags, escapes, pool");
    return CQ_OK;
}

// rcount (ix->d_rc, nl = n_u + n_d leaves) -> the caller's two uint32 arrays, narrow over the link:
//   device + link   narrow_rcount_kernel: uint32 -> one byte per leaf (saturated at 255) + escape list; the kernel's lanes
//                   write the bytes straight into page-locked host memory and flag every finished segment
//   host            the pool's threads poll the flags and widen segment after segment into rcount_u / rcount_d (streaming
//                   stores); the escaped entries are written last.
// Bit-exact with the plain copy (tests: leaves forced past 255, the escape list overrun -> *fell_back = true, plain copy).
// What a query hands the ILP is unchanged: uint32 per leaf in decode order (query.cpp:1161,1176-1177).
// narrow_start queues the kernel on s_comp (behind whatever is queued there) and wakes the pool; narrow_finish waits for the
// threads and writes the escapes.  Not started (no pool, too few leaves): narrow_inflight stays false, the caller copies.
int narrow_start(cq_index *ix, uint64_t n_u, uint64_t n_d, uint32_t *dst_u, uint32_t *dst_d, const uint32_t *src)
{
    const uint64_t nl = n_u + n_d;
    if (!src) src = ix->d_rc;   // the handle's own array (host-fed doors), or the caller's (cq_rcount_fetch)
    ix->narrow_inflight = false;
    int rc = ensure_narrow(ix, nl);
    if (rc != CQ_OK) return rc;
    if (!ix->pool) return CQ_OK;
    WidenPool &P = *ix->pool;
    PipeTrace *tr = ix->trace;
    const uint64_t seg = narrow_seg(), nseg = (nl + seg - 1) / seg;
    if (ix->flags_cap < nseg + 1) return fail(CQ_ERR_ARG, "narrow rcount: segment size changed under a live handle");
    if (++ix->narrow_epoch == 0) {   // the epoch wrapped (2^32 queries): start the flags over
        memset(ix->h_flags, 0, ix->flags_cap * sizeof(uint32_t));
        ix->narrow_epoch = 1;
    }
    const uint32_t epoch = ix->narrow_epoch;
    CQ_HIP(hipMemsetAsync(ix->d_esc_count, 0, sizeof(uint32_t), ix->s_comp));
    if (tr) tr->dev("narrow_begin", 0, ix->s_comp);
    CQ_HIP(cq::launch_narrow_rcount(src, nl, ix->h_narrow, seg, ix->h_flags, epoch, ix->d_esc, ix->d_esc_count, ix->esc_cap, ix->d_blocks_done,
                                    ix->h_esc_count, ix->s_comp));
    CQ_HIP(hipEventRecord(ix->ev_narrow, ix->s_comp));
    if (tr) tr->dev("narrow_end", 0, ix->s_comp);
    {   // wake the workers: they poll the flags from here on
        std::lock_guard<std::mutex> lk(P.mu);
        P.narrow = ix->h_narrow; P.flags = ix->h_flags; P.epoch = epoch; P.n = nl; P.n_u = n_u; P.n_segments = nseg; P.seg = seg;
        P.dst_u = dst_u; P.dst_d = dst_d;
        P.next.store(0); P.workers_done.store(0); P.abort.store(false);
        P.job_seq++;
    }
    P.cv.notify_all();
    ix->narrow_inflight = true;
    ix->narrow_nseg = nseg;
    return CQ_OK;
}

int narrow_finish(cq_index *ix, uint64_t n_u, uint32_t *dst_u, uint32_t *dst_d, bool *fell_back)
{
    *fell_back = true;
    if (!ix->narrow_inflight) return CQ_OK;
    ix->narrow_inflight = false;
    WidenPool &P = *ix->pool;
    const unsigned W = (unsigned)P.th.size();
    PipeTrace *tr = ix->trace;
    const uint64_t nseg = ix->narrow_nseg;
    const uint32_t epoch = ix->narrow_epoch;
    // the kernel's completion is watched too: if it (or a classify kernel before it) failed no flag will ever come
    hipError_t bad = hipSuccess;
    const volatile uint32_t *flags = ix->h_flags;
    uint64_t spins = 0;
    while (P.workers_done.load(std::memory_order_acquire) < W) {
        _mm_pause();
        if ((++spins & 0xFFFF) == 0 && flags[nseg] != epoch) {
            const hipError_t q = hipEventQuery(ix->ev_narrow);
            if (q != hipSuccess && q != hipErrorNotReady) { bad = q; P.abort.store(true); }
        }
    }
    if (tr) tr->host("widened", (int)nseg);
    if (bad != hipSuccess) return fail(CQ_ERR_HIP, std::string("classify / narrow rcount kernel: ") + hipGetErrorString(bad));
    CQ_HIP(hipEventSynchronize(ix->ev_narrow));
    if (flags[nseg] != epoch) return fail(CQ_ERR_HIP, "narrow rcount kernel ended without its final flag");
    std::atomic_thread_fence(std::memory_order_acquire);
    const uint32_t n_esc = *(volatile uint32_t *)ix->h_esc_count;
    if (n_esc > ix->esc_cap) return CQ_OK;                               // more saturated leaves than the list holds: plain copy (fell_back stays true)
    if (n_esc) {
        CQ_HIP(hipMemcpy(ix->h_esc, ix->d_esc, (size_t)n_esc * sizeof(uint2), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n_esc; i++) {
            const uint64_t leaf = ix->h_esc[i].x;
            if (leaf < n_u) dst_u[leaf] = ix->h_esc[i].y; else dst_d[leaf - n_u] = ix->h_esc[i].y;
        }
    }
    *fell_back = false;
    return CQ_OK;
}

// Counter block + rcount of ix (device) -> the caller's cq_counts.  Pair counts are NOT handled here.
int fetch_counts(cq_index *ix, int mode, uint32_t n_genomes, cq_counts *out, uint64_t *flags)
{
    const cq::FlatImage &img = ix->H->img;
    const uint64_t G1 = (uint64_t)n_genomes + 1, cw = cq_counter_words(n_genomes);
    CQ_HIP(hipSetDevice(ix->device));
    std::vector<uint64_t> ctr(cw, 0);
    if (mode == CQ_MODE_P && ix->d_rc && !ix->narrow_inflight) {   // (one device: classify_range has started it behind its last kernel)
        int rc = narrow_start(ix, img.n_leaves[0], img.n_leaves[1], out->rcount_u, out->rcount_d);
        if (rc != CQ_OK) return rc;
    }
    if (mode == CQ_MODE_P && ix->d_rc) {
        bool plain = true;
        int rc = narrow_finish(ix, img.n_leaves[0], out->rcount_u, out->rcount_d, &plain);   // returns behind every kernel of the query
        if (rc != CQ_OK) return rc;
        if (plain) {
            rc = copy_out(ix, out->rcount_u, ix->d_rc, img.n_leaves[0] * 4);
            if (rc == CQ_OK) rc = copy_out(ix, out->rcount_d, ix->d_rc + img.n_leaves[0], img.n_leaves[1] * 4);
            if (rc != CQ_OK) return rc;
        }
    }
    CQ_HIP(hipStreamSynchronize(ix->s_comp));   // (SC mode, or rcount the plain way: nothing has waited for the kernels yet on the one-device path)
    CQ_HIP(hipMemcpy(ctr.data(), ix->d_ctr, cw * 8, hipMemcpyDeviceToHost));
    memcpy(out->cnt_u, ctr.data(), G1 * 8);
    memcpy(out->cnt_d, ctr.data() + G1, G1 * 8);
    out->nundet = ctr[CQ_CTR_NUNDET(n_genomes)];
    out->nconf = ctr[CQ_CTR_NCONF(n_genomes)];
    out->nskipped = ctr[CQ_CTR_NSKIP(n_genomes)];
    out->n_pairs = 0;
    *flags = ctr[CQ_CTR_FLAGS(n_genomes)];
    return CQ_OK;
}

// One host-fed query on one device.  In SC mode a full pair map (flags word != 0: some
// read_cnts_b increments were dropped) is grown x4 and the whole range is classified again.
int query_one(cq_index *ix, int mode, const Feed &f, uint64_t n_reads, uint32_t n_genomes, cq_counts *out)
{
    for (;;) {
        // "Counters are OVERWRITTEN": pairs an earlier query left behind (arrays too small, count-only fetch, a
        // failure half-way) must not add onto this one.  They are kept for a second cq_pairs_fetch only until here.
        int rc = CQ_OK;
        if (mode == CQ_MODE_SC) {
            CQ_HIP(hipSetDevice(ix->device));
            if ((rc = pairs_clear(ix)) != CQ_OK) return rc;
        }
        PipeTrace trace;
        ix->trace = trace.on() ? &trace : nullptr;
        rc = classify_range(ix, mode, f, 0, n_reads, n_genomes, out);
        uint64_t flags = 0;
        if (rc == CQ_OK) rc = fetch_counts(ix, mode, n_genomes, out, &flags);
        if (ix->trace) { trace.host("query_done", 0); trace.dump(); ix->trace = nullptr; }
        if (rc != CQ_OK) return rc;
        if (mode != CQ_MODE_SC) return CQ_OK;
        if (flags != 0) {
            if (ix->pair_cap >= (1u << 30)) { (void)pairs_clear(ix); return fail(CQ_ERR_LIMIT, "device pair table full at 2^30 slots"); }
            rc = pairs_alloc(ix, ix->pair_cap << 2);
            if (rc != CQ_OK) return rc;
            continue;
        }
        uint64_t np = 0;
        rc = cq_pairs_fetch(ix, out->pair_a, out->pair_b, out->pair_cnt, out->pair_cap, &np);
        out->n_pairs = np;
        // same contract as cq_multi_query: pairs that found no room in the caller's arrays are CQ_ERR_LIMIT with
        // n_pairs = the number needed (the map is kept for a cq_pairs_fetch with larger arrays until the next query)
        if (rc == CQ_OK && np && !(out->pair_a && out->pair_b && out->pair_cnt))
            return fail(CQ_ERR_LIMIT, "more distinct pairs than pair_cap (no pair arrays given: call cq_pairs_fetch with arrays of n_pairs entries)");
        return rc;
    }
}

}  // namespace

extern "C" {

int cq_query(cq_index *ix, int mode, const uint8_t *bases, const uint64_t *offsets,
             uint64_t n_reads, uint32_t n_genomes, cq_counts *out)
{
    if (!ix || !out || !offsets) return fail(CQ_ERR_ARG, "cq_query: NULL argument");
    int rc = query_checks(ix, mode, n_genomes, out, "cq_query");
    if (rc != CQ_OK) return rc;
    Feed f;
    f.bases = bases;
    f.offsets = offsets;
    return query_one(ix, mode, f, n_reads, n_genomes, out);
}

int cq_query_reads(cq_index *ix, int mode, const uint8_t *const *reads, const uint8_t *rlengths, uint64_t n_reads,
                   uint32_t n_genomes, cq_counts *out)
{
    if (!ix || !out || (n_reads && (!reads || !rlengths))) return fail(CQ_ERR_ARG, "cq_query_reads: NULL argument");
    int rc = query_checks(ix, mode, n_genomes, out, "cq_query_reads");
    if (rc != CQ_OK) return rc;
    Feed f;
    f.ptrs = reads;
    f.lens = rlengths;
    return query_one(ix, mode, f, n_reads, n_genomes, out);
}

int cq_query_packed(cq_index *ix, int mode, const uint32_t *packed, const uint8_t *lens, uint64_t n_reads,
                    uint32_t stride_words, uint32_t max_len, uint32_t n_genomes, cq_counts *out)
{
    if (!ix || !out || (n_reads && (!packed || !lens))) return fail(CQ_ERR_ARG, "cq_query_packed: NULL argument");
    if (stride_words == 0 || stride_words > 16) return fail(CQ_ERR_ARG, "cq_query_packed: bad stride");
    int rc = query_checks(ix, mode, n_genomes, out, "cq_query_packed");
    if (rc != CQ_OK) return rc;
    Feed f;
    f.packed = packed;
    f.lens = lens;
    f.sw = stride_words;
    f.max_len = max_len;
    return query_one(ix, mode, f, n_reads, n_genomes, out);
}

int cq_query_packed_tight(cq_index *ix, int mode, const uint8_t *packed, const uint8_t *lens, uint64_t n_reads,
                          uint32_t stride_bytes, uint32_t max_len, uint32_t n_genomes, cq_counts *out)
{
    if (!ix || !out || (n_reads && (!packed || !lens))) return fail(CQ_ERR_ARG, "cq_query_packed_tight: NULL argument");
    if (stride_bytes == 0 || stride_bytes > 64) return fail(CQ_ERR_ARG, "cq_query_packed_tight: bad stride");
    int rc = query_checks(ix, mode, n_genomes, out, "cq_query_packed_tight");
    if (rc != CQ_OK) return rc;
    Feed f;
    f.tight = packed;
    f.lens = lens;
    f.sb = stride_bytes;
    f.sw = (stride_bytes + 3) / 4;
    f.max_len = max_len;
    return query_one(ix, mode, f, n_reads, n_genomes, out);
}

int cq_rcount_fetch(cq_index *ix, const uint32_t *d_rcount, void *stream, uint32_t *rcount_u, uint32_t *rcount_d)
{
    if (!ix || !d_rcount) return fail(CQ_ERR_ARG, "cq_rcount_fetch: NULL argument");
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only");
    const cq::FlatImage &img = ix->H->img;
    const uint64_t n_u = img.n_leaves[0], n_d = img.n_leaves[1];
    if ((n_u && !rcount_u) || (n_d && !rcount_d)) return fail(CQ_ERR_ARG, "cq_rcount_fetch: rcount_u / rcount_d must be provided");
    if (n_u + n_d == 0) return CQ_OK;
    if ((uintptr_t)d_rcount & 15u) return fail(CQ_ERR_ARG, "cq_rcount_fetch: d_rcount must be 16-byte aligned");
    CQ_HIP(hipSetDevice(ix->device));
    if (!ix->s_comp) CQ_HIP(hipStreamCreateWithFlags(&ix->s_comp, hipStreamNonBlocking));
    if (!ix->ev_zeroed) CQ_HIP(hipEventCreateWithFlags(&ix->ev_zeroed, hipEventDisableTiming));
    // the narrow kernel runs on the handle's compute queue, behind everything queued on the caller's stream so far
    CQ_HIP(hipEventRecord(ix->ev_zeroed, (hipStream_t)stream));
    CQ_HIP(hipStreamWaitEvent(ix->s_comp, ix->ev_zeroed, 0));
    int rc = narrow_start(ix, n_u, n_d, rcount_u, rcount_d, d_rcount);
    if (rc != CQ_OK) return rc;
    bool plain = true;
    rc = narrow_finish(ix, n_u, rcount_u, rcount_d, &plain);
    if (rc != CQ_OK) return rc;
    if (plain) {   // few leaves, no worker threads, or more saturated leaves than the escape list holds: the plain uint32 copy
        CQ_HIP(hipStreamSynchronize((hipStream_t)stream));
        if (!ix->s_copy) CQ_HIP(hipStreamCreateWithFlags(&ix->s_copy, hipStreamNonBlocking));
        rc = copy_out(ix, rcount_u, d_rcount, n_u * 4);
        if (rc == CQ_OK) rc = copy_out(ix, rcount_d, d_rcount + n_u, n_d * 4);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Multi-GPU, one process per GPU: the caller (an MPI-style launcher, torchrun, ...) moves the
 * 128-byte id from rank 0 to the others by whatever means it has.
 * ------------------------------------------------------------------------------------------ */

int cq_comm_unique_id(uint8_t *id)
{
    if (!id) return fail(CQ_ERR_ARG, "cq_comm_unique_id: NULL argument");
    static_assert(CQ_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    CQ_NCCL(ncclGetUniqueId(&u));
    memcpy(id, u.internal, CQ_COMM_ID_BYTES);
    return CQ_OK;
}

int cq_comm_init_rank(cq_index *ix, const uint8_t *id, int rank, int n_ranks, cq_comm **out)
{
    if (!ix || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(CQ_ERR_ARG, "cq_comm_init_rank: bad argument");
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only");
    *out = nullptr;
    CQ_HIP(hipSetDevice(ix->device));
    ncclUniqueId u;
    memcpy(u.internal, id, CQ_COMM_ID_BYTES);
    cq_comm *c = new (std::nothrow) cq_comm();
    if (!c) return fail(CQ_ERR_NOMEM, "out of memory");
    ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) { delete c; return fail(CQ_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    c->device = ix->device;
    c->rank = rank;
    c->n_ranks = n_ranks;
    *out = c;
    return CQ_OK;
}

int cq_counts_allreduce(cq_comm *c, uint64_t *d_counters, uint64_t n_counter_words, uint32_t *d_rcount,
                        uint64_t n_rcount, void *stream)
{
    if (!c || !d_counters || !n_counter_words) return fail(CQ_ERR_ARG, "cq_counts_allreduce: bad argument");
    CQ_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    // every word of the block is a count (the flags word too: a count of lost increments), so one sum serves all
    CQ_NCCL(ncclGroupStart());
    ncclResult_t r1 = ncclAllReduce(d_counters, d_counters, n_counter_words, ncclUint64, ncclSum, c->comm, st);
    ncclResult_t r2 = (d_rcount && n_rcount) ? ncclAllReduce(d_rcount, d_rcount, n_rcount, ncclUint32, ncclSum, c->comm, st) : ncclSuccess;
    CQ_NCCL(ncclGroupEnd());
    if (r1 != ncclSuccess) return fail(CQ_ERR_COMM, std::string("ncclAllReduce(counters): ") + ncclGetErrorString(r1));
    if (r2 != ncclSuccess) return fail(CQ_ERR_COMM, std::string("ncclAllReduce(rcount): ") + ncclGetErrorString(r2));
    return CQ_OK;
}

int cq_comm_info(const cq_comm *c, int *rank, int *n_ranks)
{
    if (!c) return fail(CQ_ERR_ARG, "cq_comm_info: NULL argument");
    if (rank) *rank = c->rank;
    if (n_ranks) *n_ranks = c->n_ranks;
    return CQ_OK;
}

void cq_comm_free(cq_comm *c)
{
    if (!c) return;
    if (c->comm) { (void)hipSetDevice(c->device); (void)ncclCommDestroy(c->comm); }
    delete c;
}

int cq_shard_range(uint64_t n_reads, int rank, int n_ranks, uint64_t *lo, uint64_t *hi)
{
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(CQ_ERR_ARG, "cq_shard_range: rank out of range");
    // contiguous ranges [n*p/P, n*(p+1)/P); 128-bit product: n_reads * P may pass 2^64
    const unsigned __int128 n = n_reads;
    if (lo) *lo = (uint64_t)(n * (unsigned)rank / (unsigned)n_ranks);
    if (hi) *hi = (uint64_t)(n * ((unsigned)rank + 1u) / (unsigned)n_ranks);
    return CQ_OK;
}

/* ------------------------------------------------------------------------------------------
 * Multi-GPU, one process: one host thread per device.
 * ------------------------------------------------------------------------------------------ */

void cq_multi_free(cq_multi *m)
{
    if (!m) return;
    for (size_t i = 0; i < m->comms.size(); i++)
        if (m->comms[i]) { (void)hipSetDevice(m->ix[m->leaders[i]]->device); (void)ncclCommDestroy(m->comms[i]); }
    for (cq_index *ix : m->ix) cq_index_free(ix);
    delete m;
}

int cq_multi_load(const char *path_u, const char *path_d, const int *devices, int n_dev, cq_multi **out)
{
    if (!path_u || !devices || n_dev < 1 || n_dev > 64 || !out) return fail(CQ_ERR_ARG, "cq_multi_load: bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(CQ_ERR_NO_DEVICE, "no HIP device available");
    double budget = 1e30;
    for (int i = 0; i < n_dev; i++) {
        if (devices[i] < 0 || devices[i] >= ndev) return fail(CQ_ERR_NO_DEVICE, "cq_multi_load: device ordinal out of range");
        int same = 0;
        for (int j = 0; j < n_dev; j++) same += devices[j] == devices[i];
        budget = std::min(budget, table_budget(devices[i]) / same);
    }
    LoadTimer lt;
    std::shared_ptr<HostIndex> H;
    int rc = prepare_host(path_u, path_d, budget, H, lt, devices[0]);   // decode + (the host part of the) layout ONCE
    if (rc != CQ_OK) return rc;
    cq_multi *m = new (std::nothrow) cq_multi();
    if (!m) return fail(CQ_ERR_NOMEM, "out of memory");
    for (int i = 0; i < n_dev; i++) {
        cq_index *ix = new (std::nothrow) cq_index();
        if (!ix) { cq_multi_free(m); return fail(CQ_ERR_NOMEM, "out of memory"); }
        ix->H = H;
        ix->device = devices[i];
        m->ix.push_back(ix);
    }
    // upload to all devices at once (one host thread each; every GPU has its own PCIe link)
    std::vector<int> rcs(n_dev, CQ_OK);
    std::vector<std::string> errs(n_dev);
    {
        std::vector<std::thread> th;
        for (int i = 0; i < n_dev; i++)
            th.emplace_back([&, i] { rcs[i] = upload(m->ix[i]); if (rcs[i] != CQ_OK) errs[i] = g_err; });
        for (auto &x : th) x.join();
    }
    lt.lap("upload (all devices)");
    for (int i = 0; i < n_dev; i++)
        if (rcs[i] != CQ_OK) { cq_multi_free(m); return fail(rcs[i], errs[i]); }
    drop_host_image(*H);
    for (int i = 0; i < n_dev; i++) warm_workspace(m->ix[i]);
    // Handles on the same device form a group (rehearsal on a box with fewer GPUs than shards): they
    // are summed by a device kernel; the group leaders -- distinct devices -- meet in the RCCL all-reduce.
    m->leader_of.assign(n_dev, 0);
    std::vector<int> devs;
    for (int i = 0; i < n_dev; i++) {
        int l = i;
        for (int j = 0; j < i; j++) if (devices[j] == devices[i]) { l = j; break; }
        m->leader_of[i] = l;
        if (l == i) { m->leaders.push_back(i); devs.push_back(devices[i]); }
    }
    m->comms.assign(m->leaders.size(), nullptr);
    ncclResult_t r = ncclCommInitAll(m->comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) { cq_multi_free(m); return fail(CQ_ERR_COMM, std::string("ncclCommInitAll: ") + ncclGetErrorString(r)); }
    lt.lap("ncclCommInitAll");
    *out = m;
    return CQ_OK;
}

int cq_multi_size(const cq_multi *m) { return m ? (int)m->ix.size() : 0; }

cq_index *cq_multi_index(cq_multi *m, int i)
{
    if (!m || i < 0 || i >= (int)m->ix.size()) { fail(CQ_ERR_ARG, "cq_multi_index: bad argument"); return nullptr; }
    return m->ix[i];
}

}  // extern "C"

namespace {

int multi_query(cq_multi *m, int mode, const Feed &f, uint64_t n_reads, uint32_t n_genomes, cq_counts *out, const char *who)
{
    const int P = (int)m->ix.size();
    for (int i = 0; i < P; i++) {
        int rc = query_checks(m->ix[i], mode, n_genomes, out, who);
        if (rc != CQ_OK) return rc;
    }
    const cq::FlatImage &img = m->ix[0]->H->img;
    const uint64_t cw = cq_counter_words(n_genomes);
    const uint64_t nl = img.n_leaves[0] + img.n_leaves[1];
    const bool rc_on = mode == CQ_MODE_P && nl;
    for (;;) {
        if (mode == CQ_MODE_SC)   // as query_one: nothing an earlier query left in a shard's pair map may add onto this one
            for (int p = 0; p < P; p++) {
                CQ_HIP(hipSetDevice(m->ix[p]->device));
                int rc = pairs_clear(m->ix[p]);
                if (rc != CQ_OK) return rc;
            }
        // ---- shard p classifies reads [n*p/P, n*(p+1)/P) on its device (the loop of query.cpp:664-665)
        std::vector<int> rcs(P, CQ_OK);
        std::vector<std::string> errs(P);
        {
            std::vector<std::thread> th;
            for (int p = 0; p < P; p++)
                th.emplace_back([&, p] {
                    uint64_t lo = 0, hi = 0;
                    (void)cq_shard_range(n_reads, p, P, &lo, &hi);
                    rcs[p] = classify_range(m->ix[p], mode, f, lo, hi, n_genomes);
                    if (rcs[p] != CQ_OK) errs[p] = g_err;
                });
            for (auto &x : th) x.join();
        }
        for (int p = 0; p < P; p++) if (rcs[p] != CQ_OK) return fail(rcs[p], errs[p]);
        // ---- handles that share a device: add into their group leader
        for (int p = 0; p < P; p++) {
            const int l = m->leader_of[p];
            if (l == p) continue;
            cq_index *dst = m->ix[l], *src = m->ix[p];
            CQ_HIP(hipSetDevice(dst->device));
            CQ_HIP(cq::launch_accumulate(dst->d_ctr, src->d_ctr, cw, rc_on ? dst->d_rc : nullptr, rc_on ? src->d_rc : nullptr,
                                         rc_on ? nl : 0, dst->s_comp));
            CQ_HIP(hipStreamSynchronize(dst->s_comp));
        }
        // ---- the exchange step: one sum of the counter block and of rcount over the devices.  Only the host reads
        //      the totals, and it reads device 0's copy: a reduce to that root moves (P-1)/P of the bytes once, where
        //      the all-reduce moves them twice (SURVEY 5 suggests exactly this; CAMMIQ_MULTI_ALLREDUCE=1 restores the
        //      all-reduce for A/B timing on a multi-GPU node; read per query, so that one process can time both:
        //      tools/multi_leg.py does on the first node that shows it two GPUs).  cq_counts_allreduce -- one process
        //      per GPU, every rank wants the totals -- stays an all-reduce.
        //      AFTER THE REDUCE ONLY THE ROOT'S COUNTERS ARE TOTALS: the other shards' d_ctr / d_rc hold their partial
        //      sums (or RCCL scratch).  That is safe because every consumer below reads m->ix[0] and because
        //      classify_range zeroes d_ctr and d_rc of every shard at the start of every query.
        const bool all_reduce = getenv("CAMMIQ_MULTI_ALLREDUCE") && atoi(getenv("CAMMIQ_MULTI_ALLREDUCE")) != 0;
        CQ_NCCL(ncclGroupStart());
        ncclResult_t bad = ncclSuccess;
        for (size_t k = 0; k < m->leaders.size(); k++) {
            cq_index *ix = m->ix[m->leaders[k]];
            ncclResult_t r;
            if (all_reduce) {
                r = ncclAllReduce(ix->d_ctr, ix->d_ctr, cw, ncclUint64, ncclSum, m->comms[k], ix->s_comp);
                if (r == ncclSuccess && rc_on) r = ncclAllReduce(ix->d_rc, ix->d_rc, nl, ncclUint32, ncclSum, m->comms[k], ix->s_comp);
            } else {   // root = communicator rank 0 = leaders[0] = m->ix[0]'s device (ncclCommInitAll numbers ranks in list order)
                r = ncclReduce(ix->d_ctr, ix->d_ctr, cw, ncclUint64, ncclSum, 0, m->comms[k], ix->s_comp);
                if (r == ncclSuccess && rc_on) r = ncclReduce(ix->d_rc, ix->d_rc, nl, ncclUint32, ncclSum, 0, m->comms[k], ix->s_comp);
            }
            if (r != ncclSuccess) bad = r;
        }
        CQ_NCCL(ncclGroupEnd());
        if (bad != ncclSuccess) return fail(CQ_ERR_COMM, std::string("RCCL reduce of the counts: ") + ncclGetErrorString(bad));
        for (size_t k = 0; k < m->leaders.size(); k++) {
            cq_index *ix = m->ix[m->leaders[k]];
            CQ_HIP(hipSetDevice(ix->device));
            CQ_HIP(hipStreamSynchronize(ix->s_comp));
        }
        // ---- device 0 now holds the totals; the host takes them from there (query.cpp:251-258 hand-off)
        uint64_t flags = 0;
        int rc = fetch_counts(m->ix[0], mode, n_genomes, out, &flags);
        if (rc != CQ_OK) return rc;
        if (mode != CQ_MODE_SC) return CQ_OK;
        if (flags != 0) {   // some shard's pair map filled up: grow all of them, classify again
            for (int p = 0; p < P; p++) {
                cq_index *ix = m->ix[p];
                CQ_HIP(hipSetDevice(ix->device));
                if (ix->pair_cap >= (1u << 30)) return fail(CQ_ERR_LIMIT, "device pair table full at 2^30 slots");
                rc = pairs_alloc(ix, ix->pair_cap << 2);
                if (rc != CQ_OK) return rc;
            }
            continue;
        }
        // read_cnts_b: per-shard maps, merged on the host (SURVEY.md 8(e) "replicas + host merge")
        std::map<uint64_t, uint64_t> merged;
        for (int p = 0; p < P; p++) {
            uint64_t np = 0;
            rc = cq_pairs_fetch(m->ix[p], nullptr, nullptr, nullptr, 0, &np);
            if (rc != CQ_OK) return rc;
            std::vector<uint32_t> a(np + 1), b(np + 1);
            std::vector<uint64_t> c(np + 1);
            rc = cq_pairs_fetch(m->ix[p], a.data(), b.data(), c.data(), np + 1, &np);
            if (rc != CQ_OK) return rc;
            for (uint64_t i = 0; i < np; i++) merged[((uint64_t)a[i] << 32) | b[i]] += c[i];
        }
        out->n_pairs = merged.size();
        if (merged.size() > out->pair_cap || (merged.size() && (!out->pair_a || !out->pair_b || !out->pair_cnt)))
            return fail(CQ_ERR_LIMIT, "more distinct pairs than pair_cap");
        uint64_t i = 0;
        for (const auto &kv : merged) {
            out->pair_a[i] = (uint32_t)(kv.first >> 32);
            out->pair_b[i] = (uint32_t)kv.first;
            out->pair_cnt[i] = kv.second;
            i++;
        }
        return CQ_OK;
    }
}

}  // namespace

extern "C" {

int cq_multi_query(cq_multi *m, int mode, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                   uint32_t n_genomes, cq_counts *out)
{
    if (!m || !out || !offsets) return fail(CQ_ERR_ARG, "cq_multi_query: NULL argument");
    Feed f;
    f.bases = bases;
    f.offsets = offsets;
    return multi_query(m, mode, f, n_reads, n_genomes, out, "cq_multi_query");
}

int cq_multi_query_reads(cq_multi *m, int mode, const uint8_t *const *reads, const uint8_t *rlengths, uint64_t n_reads,
                         uint32_t n_genomes, cq_counts *out)
{
    if (!m || !out || (n_reads && (!reads || !rlengths))) return fail(CQ_ERR_ARG, "cq_multi_query_reads: NULL argument");
    Feed f;
    f.ptrs = reads;
    f.lens = rlengths;
    return multi_query(m, mode, f, n_reads, n_genomes, out, "cq_multi_query_reads");
}

int cq_multi_query_packed(cq_multi *m, int mode, const uint32_t *packed, const uint8_t *lens, uint64_t n_reads,
                          uint32_t stride_words, uint32_t max_len, uint32_t n_genomes, cq_counts *out)
{
    if (!m || !out || (n_reads && (!packed || !lens))) return fail(CQ_ERR_ARG, "cq_multi_query_packed: NULL argument");
    if (stride_words == 0 || stride_words > 16) return fail(CQ_ERR_ARG, "cq_multi_query_packed: bad stride");
    Feed f;
    f.packed = packed;
    f.lens = lens;
    f.sw = stride_words;
    f.max_len = max_len;
    return multi_query(m, mode, f, n_reads, n_genomes, out, "cq_multi_query_packed");
}

int cq_multi_query_packed_tight(cq_multi *m, int mode, const uint8_t *packed, const uint8_t *lens, uint64_t n_reads,
                                uint32_t stride_bytes, uint32_t max_len, uint32_t n_genomes, cq_counts *out)
{
    if (!m || !out || (n_reads && (!packed || !lens))) return fail(CQ_ERR_ARG, "cq_multi_query_packed_tight: NULL argument");
    if (stride_bytes == 0 || stride_bytes > 64) return fail(CQ_ERR_ARG, "cq_multi_query_packed_tight: bad stride");
    Feed f;
    f.tight = packed;
    f.lens = lens;
    f.sb = stride_bytes;
    f.sw = (stride_bytes + 3) / 4;
    f.max_len = max_len;
    return multi_query(m, mode, f, n_reads, n_genomes, out, "cq_multi_query_packed_tight");
}

}  // extern "C"
