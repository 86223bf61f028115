// cq_api.cpp -- the C ABI of libcammiq_hip.so (see include/cammiq_hip.h).
//
// Host-side replacement of the seam the reference has at FqReader::loadIdx_p
// (/root/reference/src/query.cpp:109-123) and FqReader::query64_p / query64mt_p / query64_sc
// (query.cpp:458-1080): load + lay out + upload the index, then run the HIP classify
// kernels on reads and hand the counters back.  No CPU classify path exists here.
#include <hip/hip_runtime.h>
#include <chrono>
#include <memory>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "cq_index.hpp"
#include "cq_kernels.h"

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

constexpr uint32_t kPairCap = 1u << 20;

}  // namespace

struct cq_index {
    cq::DecodedTable tab[2];
    cq::FlatImage img;
    uint64_t n_file_buckets[2] = {0, 0};
    uint64_t n_trie_nodes = 0;
    bool from_cache = false;
    uint32_t doubly_flag[2] = {0, 1};
    int device = CQ_DEVICE_NONE;
    int n_cus = 0;
    uint64_t device_bytes = 0;
    // device image
    void *d_slots = nullptr, *d_nodes = nullptr, *d_leaf_rids = nullptr;
    cq::DevIndex dev{};
    // per-handle workspace (grown on demand, reused across calls)
    uint32_t *d_ovf_list = nullptr, *d_ovf_count = nullptr;
    uint64_t *d_stamps = nullptr;   // 8 words of their own for diagnostic (CQ_STAMPS) kernel builds
    uint64_t ovf_cap = 0;
    uint64_t *d_pair_keys = nullptr, *d_pair_cnts = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    // host-buffer path (cq_query): two staging slots so that packing chunk c+1 on the CPU, its
    // H2D copy and the classify kernel of chunk c overlap
    struct Slot {
        uint32_t *h_packed = nullptr, *d_packed = nullptr;   // pinned host / device rows
        uint8_t *h_lens = nullptr, *d_lens = nullptr;
        size_t cap_words = 0, cap_reads = 0;
        hipEvent_t copied = nullptr, done = nullptr;
    } slot[2];
    hipStream_t s_copy = nullptr, s_comp = nullptr;
    uint64_t *d_ctr = nullptr; size_t ctr_cap = 0;
    uint32_t *d_rc = nullptr; size_t rc_cap = 0;
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
                    fclose(fo);
                }
        }
        (void)hipFree(ix->d_stamps);
    }
    if (ix->d_pair_keys) (void)hipFree(ix->d_pair_keys);
    if (ix->d_pair_cnts) (void)hipFree(ix->d_pair_cnts);
    if (ix->ev0) (void)hipEventDestroy(ix->ev0);
    if (ix->ev1) (void)hipEventDestroy(ix->ev1);
    for (auto &sl : ix->slot) {
        if (sl.h_packed) (void)hipHostFree(sl.h_packed);
        if (sl.h_lens) (void)hipHostFree(sl.h_lens);
        if (sl.d_packed) (void)hipFree(sl.d_packed);
        if (sl.d_lens) (void)hipFree(sl.d_lens);
        if (sl.copied) (void)hipEventDestroy(sl.copied);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (ix->s_copy) (void)hipStreamDestroy(ix->s_copy);
    if (ix->s_comp) (void)hipStreamDestroy(ix->s_comp);
    if (ix->d_ctr) (void)hipFree(ix->d_ctr);
    if (ix->d_rc) (void)hipFree(ix->d_rc);
}

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
    const cq::FlatImage &img = ix->img;
    const size_t sb = img.table_words * sizeof(uint32_t);
    const size_t nb = img.nodes.size() * sizeof(cq::Node);
    const size_t nl = img.leaf_r1.size();
    CQ_HIP(hipMalloc(&ix->d_slots, sb));
    CQ_HIP(hipMalloc(&ix->d_nodes, nb));
    CQ_HIP(hipMalloc(&ix->d_leaf_rids, std::max<size_t>(nl, 1) * sizeof(uint2)));
    CQ_HIP(hipMemcpy(ix->d_slots, img.table.get(), sb, hipMemcpyHostToDevice));
    CQ_HIP(hipMemcpy(ix->d_nodes, img.nodes.data(), nb, hipMemcpyHostToDevice));
    {
        std::unique_ptr<uint2[]> rr(new uint2[nl ? nl : 1]);
        const unsigned nt = nl < (1u << 20) ? 1u : std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                for (size_t i = nl * t / nt, e = nl * (t + 1) / nt; i < e; i++) rr[i] = make_uint2(img.leaf_r1[i], img.leaf_r2[i]);
            });
        for (auto &x : th) x.join();
        if (nl) CQ_HIP(hipMemcpy(ix->d_leaf_rids, rr.get(), nl * sizeof(uint2), hipMemcpyHostToDevice));
    }
    CQ_HIP(hipMalloc((void **)&ix->d_ovf_count, sizeof(uint32_t)));
    CQ_HIP(hipMalloc((void **)&ix->d_stamps, 8 * sizeof(uint64_t)));
    CQ_HIP(hipMemset(ix->d_stamps, 0, 8 * sizeof(uint64_t)));
    CQ_HIP(hipMalloc((void **)&ix->d_pair_keys, (size_t)kPairCap * 8));
    CQ_HIP(hipMalloc((void **)&ix->d_pair_cnts, (size_t)kPairCap * 8));
    CQ_HIP(hipMemset(ix->d_pair_keys, 0xFF, (size_t)kPairCap * 8));
    CQ_HIP(hipMemset(ix->d_pair_cnts, 0, (size_t)kPairCap * 8));
    CQ_HIP(hipEventCreate(&ix->ev0));
    CQ_HIP(hipEventCreate(&ix->ev1));
    ix->device_bytes = sb + nb + nl * sizeof(uint2) + (size_t)kPairCap * 16;
    ix->dev.slots = (const uint4 *)ix->d_slots;
    ix->dev.nodes = (const uint4 *)ix->d_nodes;
    ix->dev.leaf_rids = (const uint2 *)ix->d_leaf_rids;
    ix->dev.n_buckets = (uint32_t)img.n_buckets;
    ix->dev.hash_len = img.hash_len;
    ix->dev.minimizer_len = cq_minimizer_len(img.hash_len);
    return CQ_OK;
}

}  // namespace

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

extern "C" {

int cq_abi_version(void) { return CQ_ABI_VERSION; }

const char *cq_last_error(void) { return g_err.c_str(); }

int cq_index_load(const char *path_u, const char *path_d, int device, cq_index **out)
{
    if (!path_u || !out || device < CQ_DEVICE_NONE) return fail(CQ_ERR_ARG, "cq_index_load: bad argument");
    *out = nullptr;
    cq_index *ix = new (std::nothrow) cq_index();
    if (!ix) return fail(CQ_ERR_NOMEM, "out of memory");
    const bool have_d = path_d && path_d[0];
    // device-memory budget for the table: at most half of what is free
    double budget = 1e30;
    if (device >= 0) {
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b)
            budget = 0.5 * (double)free_b;
    }
    // Optional image cache next to index_u (cq_cache.cpp); the .bin files stay authoritative.
    const char *ce = getenv("CAMMIQ_IMAGE_CACHE");
    const bool use_cache = ce && ce[0] == '1';
    const std::string cache_file = std::string(path_u) + ".cqimg";
    cq::SourceStamp stamp;
    const bool stamped = use_cache && cq::stamp_sources(path_u, have_d ? path_d : "", stamp);
    bool from_cache = false;
    LoadTimer lt;
    const double kpb_override = getenv("CAMMIQ_KEYS_PER_BUCKET") ? atof(getenv("CAMMIQ_KEYS_PER_BUCKET")) : 0.0;
    if (stamped)
        from_cache = cq::load_image(cache_file, stamp, kpb_override, budget >= 1e29 ? ~0ull : (uint64_t)budget, ix->tab, ix->img);
    if (from_cache) lt.lap("image cache read");
    // The reference loads the two files on two pthreads (query.cpp:112-116); same here.
    int rc_u = CQ_OK, rc_d = CQ_OK;
    std::string err_u, err_d;
    if (!from_cache) try {
        std::thread td;
        if (have_d) td = std::thread([&] { rc_d = cq::decode_table(path_d, ix->tab[1], err_d); });
        rc_u = cq::decode_table(path_u, ix->tab[0], err_u);
        if (have_d) td.join();
        if (rc_u != CQ_OK) { delete ix; return fail(rc_u, err_u); }
        if (rc_d != CQ_OK) { delete ix; return fail(rc_d, err_d); }
        if (!have_d) cq::make_empty_table(ix->tab[0].hash_len, ix->tab[1]);
        lt.lap("decode");
        std::string err;
        // Average keys per 4-slot bucket of the device table.  Emptier tables overflow less
        // (fewer windows take the exact path): 1.0 costs 64 B of HBM per key (measured: 0.5 -> +2 %,
        // 1.5 -> -3 %); when the table would not fit comfortably it is packed tighter.
        // CAMMIQ_KEYS_PER_BUCKET overrides (tuning knob, not part of the ABI).
        double kpb = 1.0;
        const double keys = (double)(ix->tab[0].bucket_key.size() + ix->tab[1].bucket_key.size());
        if (keys / kpb * 64.0 > budget) kpb = std::min(3.2, keys * 64.0 / budget);
        if (kpb_override > 0.0) kpb = kpb_override;
        int rc = cq::build_image(ix->tab[0], ix->tab[1], kpb, ix->img, err);
        if (rc != CQ_OK) { delete ix; return fail(rc, err); }
        lt.lap("layout");
        // bucket/node arrays of the decode stage are no longer needed; leaves are (cq_index_leaves)
        for (int t = 0; t < 2; t++) {
            std::vector<uint64_t>().swap(ix->tab[t].bucket_key);
            std::vector<uint32_t>().swap(ix->tab[t].bucket_code);
            std::vector<cq::Node>().swap(ix->tab[t].nodes);
        }
        if (stamped) { (void)cq::save_image(cache_file, stamp, kpb_override, ix->tab, ix->img); lt.lap("image cache write"); }
    } catch (const std::bad_alloc &) {
        delete ix;
        return fail(CQ_ERR_NOMEM, "out of memory while loading the index");
    }
    for (int t = 0; t < 2; t++) {
        ix->n_file_buckets[t] = ix->tab[t].n_file_buckets;
        ix->doubly_flag[t] = ix->tab[t].doubly;
    }
    ix->from_cache = from_cache;
    ix->device = device;
    if (device >= 0) {
        int rc = upload(ix);
        lt.lap("upload");
        if (rc != CQ_OK) { release_device(ix); delete ix; return rc; }
        // the image now lives in HBM: drop the host copy (cq_index_probe needs a CQ_DEVICE_NONE handle)
        ix->n_trie_nodes = ix->img.nodes.size() - 1;
        ix->img.table.reset();
        std::vector<cq::Node>().swap(ix->img.nodes);
        std::vector<uint32_t>().swap(ix->img.leaf_r1);
        std::vector<uint32_t>().swap(ix->img.leaf_r2);
    }
    *out = ix;
    return CQ_OK;
}

int cq_index_get_info(const cq_index *ix, cq_index_info *info)
{
    if (!ix || !info) return fail(CQ_ERR_ARG, "cq_index_get_info: NULL argument");
    memset(info, 0, sizeof *info);
    info->abi_version = CQ_ABI_VERSION;
    info->hash_len = ix->img.hash_len;
    info->max_refid = ix->img.max_refid;
    info->device = ix->device;
    for (int t = 0; t < 2; t++) {
        info->doubly_flag[t] = ix->doubly_flag[t];
        info->n_leaves[t] = ix->img.n_leaves[t];
        info->n_file_buckets[t] = ix->n_file_buckets[t];
    }
    info->n_trie_nodes = ix->img.nodes.empty() ? ix->n_trie_nodes : ix->img.nodes.size() - 1;
    info->n_keys = ix->img.n_keys;
    info->n_table_buckets = ix->img.n_buckets_alloc;
    info->n_overflowed = ix->img.n_overflowed;
    info->max_chain = ix->img.max_chain;
    info->device_bytes = ix->device_bytes;
    info->reserved_ = ix->from_cache ? 1u : 0u;
    return CQ_OK;
}

int cq_index_leaves(const cq_index *ix, int table, cq_leaf *out)
{
    if (!ix || !out || table < 0 || table > 1) return fail(CQ_ERR_ARG, "cq_index_leaves: bad argument");
    const auto &lv = ix->tab[table].leaves;
    if (!lv.empty()) memcpy(out, lv.data(), lv.size() * sizeof(cq_leaf));
    return CQ_OK;
}

int cq_index_probe(const cq_index *ix, uint64_t hv, uint32_t *code_u, uint32_t *code_d, uint32_t *chain)
{
    if (!ix || !code_u || !code_d) return fail(CQ_ERR_ARG, "cq_index_probe: NULL argument");
    if (!ix->img.table) return fail(CQ_ERR_ARG, "cq_index_probe: host image was released");
    cq::image_lookup(ix->img, hv, *code_u, *code_d, chain);
    return CQ_OK;
}

void cq_index_free(cq_index *ix)
{
    if (!ix) return;
    release_device(ix);
    delete ix;
}

uint64_t cq_counter_words(uint32_t n_genomes) { return 2ull * ((uint64_t)n_genomes + 1) + CQ_CTR_EXTRA; }

int cq_query_device(cq_index *ix, int mode, const uint32_t *d_packed, const uint8_t *d_lens,
                    uint64_t n_reads, uint32_t stride_words, uint32_t max_len, uint32_t n_genomes,
                    uint64_t *d_counters, uint32_t *d_rcount, void *stream)
{
    if (!ix) return fail(CQ_ERR_ARG, "cq_query_device: NULL handle");
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only (CQ_DEVICE_NONE); no CPU classify path exists");
    if (mode != CQ_MODE_P && mode != CQ_MODE_SC) return fail(CQ_ERR_ARG, "cq_query_device: unknown mode");
    if (!d_counters || (n_reads && (!d_packed || !d_lens)) || stride_words == 0 || (stride_words & 3u) || stride_words > 16)
        return fail(CQ_ERR_ARG, "cq_query_device: bad argument");
    if (n_reads > 0x7FFFFFFFull) return fail(CQ_ERR_ARG, "cq_query_device: more than 2^31-1 reads in one call");
    if (ix->img.max_refid > n_genomes)
        return fail(CQ_ERR_RANGE, "index holds refID " + std::to_string(ix->img.max_refid) + " > n_genomes");
    if (n_reads == 0) return CQ_OK;
    hipStream_t st = (hipStream_t)stream;
    CQ_HIP(hipSetDevice(ix->device));
    if (ix->ovf_cap < n_reads) {   // grow the slow-path list (outside steady state)
        if (ix->d_ovf_list) CQ_HIP(hipFree(ix->d_ovf_list));
        ix->d_ovf_list = nullptr;
        CQ_HIP(hipMalloc((void **)&ix->d_ovf_list, n_reads * sizeof(uint32_t)));
        ix->ovf_cap = n_reads;
    }
    CQ_HIP(hipMemsetAsync(ix->d_ovf_count, 0, sizeof(uint32_t), st));
    const uint32_t h = ix->img.hash_len;
    if (max_len == 0 || max_len > stride_words * 16) max_len = stride_words * 16;
    if (max_len > 255) max_len = 255;
    cq::QueryArgs a{};
    a.packed = d_packed;
    a.lens = d_lens;
    a.n_reads = n_reads;
    a.stride_words = stride_words;
    a.wmax = max_len >= h ? max_len - h + 1 : 1;
    a.n_genomes = n_genomes;
    a.mode = mode;
    a.counters = d_counters;
    a.rcount = (mode == CQ_MODE_P) ? d_rcount : nullptr;
    a.ovf_list = ix->d_ovf_list;
    a.ovf_count = ix->d_ovf_count;
    a.ovf_cap = (uint32_t)ix->ovf_cap;
    a.pair_keys = ix->d_pair_keys;
    a.pair_cnts = ix->d_pair_cnts;
    a.pair_cap = kPairCap;
    a.stamps = ix->d_stamps;   // only written by diagnostic (CQ_STAMPS) builds
    CQ_HIP(cq::launch_classify(ix->dev, a, ix->n_cus, st, ix->ev0, ix->ev1));
    ix->ev_valid = true;
    return CQ_OK;
}

int cq_last_kernel_ms(cq_index *ix, float *ms)
{
    if (!ix || !ms) return fail(CQ_ERR_ARG, "cq_last_kernel_ms: NULL argument");
    if (!ix->ev_valid) return fail(CQ_ERR_ARG, "cq_last_kernel_ms: no kernel has been launched on this handle");
    CQ_HIP(hipEventSynchronize(ix->ev1));
    CQ_HIP(hipEventElapsedTime(ms, ix->ev0, ix->ev1));
    return CQ_OK;
}

int cq_pairs_fetch(cq_index *ix, uint32_t *pair_a, uint32_t *pair_b, uint64_t *pair_cnt,
                   uint64_t pair_cap, uint64_t *n_pairs)
{
    if (!ix || !n_pairs) return fail(CQ_ERR_ARG, "cq_pairs_fetch: NULL argument");
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only");
    CQ_HIP(hipSetDevice(ix->device));
    CQ_HIP(hipDeviceSynchronize());
    std::vector<uint64_t> k(kPairCap), c(kPairCap);
    CQ_HIP(hipMemcpy(k.data(), ix->d_pair_keys, (size_t)kPairCap * 8, hipMemcpyDeviceToHost));
    CQ_HIP(hipMemcpy(c.data(), ix->d_pair_cnts, (size_t)kPairCap * 8, hipMemcpyDeviceToHost));
    CQ_HIP(hipMemset(ix->d_pair_keys, 0xFF, (size_t)kPairCap * 8));
    CQ_HIP(hipMemset(ix->d_pair_cnts, 0, (size_t)kPairCap * 8));
    uint64_t n = 0;
    for (uint32_t i = 0; i < kPairCap; i++) {
        if (k[i] == CQ_EMPTY_KEY) continue;
        if (n < pair_cap && pair_a && pair_b && pair_cnt) {
            pair_a[n] = (uint32_t)(k[i] >> 32);
            pair_b[n] = (uint32_t)k[i];
            pair_cnt[n] = c[i];
        }
        n++;
    }
    *n_pairs = n;
    if (n > pair_cap) return fail(CQ_ERR_LIMIT, "more distinct pairs than pair_cap");
    return CQ_OK;
}

namespace {

// Grow one staging slot to hold n reads of sw words.
int slot_reserve(cq_index::Slot &sl, uint64_t n, uint32_t sw)
{
    const size_t words = (size_t)n * sw;
    if (sl.cap_words < words) {
        if (sl.h_packed) (void)hipHostFree(sl.h_packed);
        if (sl.d_packed) (void)hipFree(sl.d_packed);
        sl.h_packed = nullptr; sl.d_packed = nullptr; sl.cap_words = 0;
        CQ_HIP(hipHostMalloc((void **)&sl.h_packed, words * 4, hipHostMallocDefault));
        CQ_HIP(hipMalloc((void **)&sl.d_packed, words * 4));
        sl.cap_words = words;
    }
    if (sl.cap_reads < n) {
        if (sl.h_lens) (void)hipHostFree(sl.h_lens);
        if (sl.d_lens) (void)hipFree(sl.d_lens);
        sl.h_lens = nullptr; sl.d_lens = nullptr; sl.cap_reads = 0;
        CQ_HIP(hipHostMalloc((void **)&sl.h_lens, n, hipHostMallocDefault));
        CQ_HIP(hipMalloc((void **)&sl.d_lens, n));
        sl.cap_reads = n;
    }
    if (!sl.copied) CQ_HIP(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
    if (!sl.done) CQ_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    return CQ_OK;
}

}  // namespace

int cq_query(cq_index *ix, int mode, const uint8_t *bases, const uint64_t *offsets,
             uint64_t n_reads, uint32_t n_genomes, cq_counts *out)
{
    if (!ix || !out || !offsets) return fail(CQ_ERR_ARG, "cq_query: NULL argument");
    if (!out->cnt_u || !out->cnt_d) return fail(CQ_ERR_ARG, "cq_query: cnt_u / cnt_d must be provided");
    if (mode == CQ_MODE_P && ((!out->rcount_u && ix->img.n_leaves[0]) || (!out->rcount_d && ix->img.n_leaves[1])))
        return fail(CQ_ERR_ARG, "cq_query: rcount_u / rcount_d are mandatory in CQ_MODE_P (the ILP reads them)");
    if (ix->device < 0) return fail(CQ_ERR_NO_DEVICE, "index was loaded host-only (CQ_DEVICE_NONE); no CPU classify path exists");
    if (ix->img.max_refid > n_genomes)
        return fail(CQ_ERR_RANGE, "index holds refID " + std::to_string(ix->img.max_refid) + " > n_genomes");
    CQ_HIP(hipSetDevice(ix->device));
    if (!ix->s_copy) CQ_HIP(hipStreamCreateWithFlags(&ix->s_copy, hipStreamNonBlocking));
    if (!ix->s_comp) CQ_HIP(hipStreamCreateWithFlags(&ix->s_comp, hipStreamNonBlocking));

    const uint64_t G1 = (uint64_t)n_genomes + 1, cw = cq_counter_words(n_genomes);
    const uint64_t nl = ix->img.n_leaves[0] + ix->img.n_leaves[1];
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

    // One call = one FASTQ.  Chunks of 2 M reads alternate between two staging slots:
    //   CPU   pack(c+1) ........ pack(c+2) ........
    //   copy           H2D(c+1) ...........H2D(c+2)
    //   comp  kernel(c) ........ kernel(c+1) .......
    const uint64_t kChunk = 1ull << 21;
    int rc = CQ_OK;
    uint64_t c = 0;
    for (uint64_t c0 = 0; c0 < n_reads && rc == CQ_OK; c0 += kChunk, c++) {
        cq_index::Slot &sl = ix->slot[c & 1];
        const uint64_t n = std::min(kChunk, n_reads - c0);
        uint64_t max_len = 0;
        for (uint64_t r = c0; r < c0 + n; r++) {
            const uint64_t l = offsets[r + 1] - offsets[r];
            if (l <= 255 && l > max_len) max_len = l;
        }
        const uint32_t sw = cq_pack_stride_words((uint32_t)max_len);
        if (sl.done) CQ_HIP(hipEventSynchronize(sl.done));      // the kernel that last used this slot
        rc = slot_reserve(sl, n, sw);
        if (rc != CQ_OK) break;
        uint64_t sk = 0;
        rc = cq_pack_reads(bases, offsets + c0, n, ix->img.hash_len, sw, sl.h_packed, sl.h_lens, &sk);
        if (rc != CQ_OK) { fail(rc, "cq_pack_reads failed"); break; }
        CQ_HIP(hipMemcpyAsync(sl.d_packed, sl.h_packed, (size_t)n * sw * 4, hipMemcpyHostToDevice, ix->s_copy));
        CQ_HIP(hipMemcpyAsync(sl.d_lens, sl.h_lens, n, hipMemcpyHostToDevice, ix->s_copy));
        CQ_HIP(hipEventRecord(sl.copied, ix->s_copy));
        CQ_HIP(hipStreamWaitEvent(ix->s_comp, sl.copied, 0));
        rc = cq_query_device(ix, mode, sl.d_packed, sl.d_lens, n, sw, (uint32_t)max_len, n_genomes, ix->d_ctr, d_rc,
                             ix->s_comp);
        if (rc != CQ_OK) break;
        CQ_HIP(hipEventRecord(sl.done, ix->s_comp));
    }
    if (hipStreamSynchronize(ix->s_comp) != hipSuccess && rc == CQ_OK) rc = fail(CQ_ERR_HIP, "classify kernel failed");
    if (rc != CQ_OK) return rc;

    std::vector<uint64_t> ctr(cw, 0);
    CQ_HIP(hipMemcpy(ctr.data(), ix->d_ctr, cw * 8, hipMemcpyDeviceToHost));
    if (d_rc) {
        if (ix->img.n_leaves[0])
            CQ_HIP(hipMemcpy(out->rcount_u, d_rc, ix->img.n_leaves[0] * 4, hipMemcpyDeviceToHost));
        if (ix->img.n_leaves[1])
            CQ_HIP(hipMemcpy(out->rcount_d, d_rc + ix->img.n_leaves[0], ix->img.n_leaves[1] * 4, hipMemcpyDeviceToHost));
    }
    memcpy(out->cnt_u, ctr.data(), G1 * 8);
    memcpy(out->cnt_d, ctr.data() + G1, G1 * 8);
    out->nundet = ctr[CQ_CTR_NUNDET(n_genomes)];
    out->nconf = ctr[CQ_CTR_NCONF(n_genomes)];
    out->nskipped = ctr[CQ_CTR_NSKIP(n_genomes)];
    out->n_pairs = 0;
    if (mode == CQ_MODE_SC) {
        if (ctr[CQ_CTR_FLAGS(n_genomes)] & 1ull) return fail(CQ_ERR_LIMIT, "device pair table full");
        uint64_t np = 0;
        rc = cq_pairs_fetch(ix, out->pair_a, out->pair_b, out->pair_cnt, out->pair_cap, &np);
        out->n_pairs = np;
        if (rc != CQ_OK) return rc;
    }
    return CQ_OK;
}

}  // extern "C"
