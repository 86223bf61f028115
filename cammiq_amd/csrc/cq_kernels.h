// cq_kernels.h -- launch interface between the C ABI (cq_api.cpp) and the HIP kernels.
#ifndef CQ_KERNELS_H_
#define CQ_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cammiq_hip.h"

namespace cq {

// Device view of the index (all pointers into HBM).
struct DevIndex {
    const uint4 *slots;      // n_buckets_alloc buckets of 4 x uint4: key_lo | key_hi | val_u | val_d (cq_device.h)
    const uint4 *nodes;      // array trie, 16 B per node
    const uint2 *leaf_rids;  // global leaf id -> (refID1, refID2)
    uint32_t n_buckets;      // hash range
    uint32_t hash_len;
    uint32_t minimizer_len;  // m of this index's table (cq_device.h: 16, or 18 for large tables)
};

constexpr uint32_t kStaticSixteenths = 12;   // 3/4 by stride, the last quarter off the counter (8 / 12 / 14 / 15 sixteenths: equal on a 50 M-read launch, 12 and 8 1.6-1.9 % ahead of 14 on 10 M- and 2 M-read launches; profiles/r04_dynamic_tail_static_share.txt)
constexpr uint32_t kWorkStripes = 64, kWorkStripeWords = 32;   // the kernel's work counter: 64 words, each in a 128-byte line of its own

struct QueryArgs {
    const uint32_t *packed;  // n_reads rows of stride_words uint32 (null when `tight` is given)
    const uint8_t *tight;    // or: n_reads rows of tight_sb BYTES (cq_pack_reads_tight), widened by the kernel's staging itself;
    uint32_t tight_sb;       //     the buffer must be readable 16 bytes past its last row
    const uint8_t *lens;
    uint64_t n_reads;
    uint64_t read0;          // index of packed[0] within the call's rows (set by the launcher when it cuts a call into launches)
    uint32_t stride_words;
    uint32_t wmax;           // max windows per read in this batch: max_len - h + 1
    uint32_t pmax;           // max m-mer positions per read: max_len - m + 1   (set by the launcher)
    uint32_t pstride;        // words per read in the LDS array of m-mer hashes (set by the launcher)
    uint32_t magic_w;        // ceil(2^19 / window groups per read), ceil(2^19 / pmax): exact small divisions
    uint32_t magic_p;
    uint32_t magic_s;        // ceil(2^19 / stride_words)
    uint32_t magic_pp;       // ceil(2^19 / position groups per read (pre-pass))
    uint32_t n_genomes;
    int mode;
    uint64_t *counters;      // cq_counter_words(n_genomes)
    uint32_t *rcount;        // may be null
    uint32_t *ovf_list;      // reads whose hit list overflowed the fast path
    uint32_t *ovf_count;
    uint32_t *work_counter;  // kWorkStripes x kWorkStripeWords device words (zeroed per launch by the launcher): the tail of the sub-tiles is handed out from here, one stripe per group of waves; null: stride only
    uint32_t max_sub;        // sub-tiles one wave may take (LDS histogram; set by the launcher, 0: no bound)
    uint32_t static_16ths;   // sixteenths of a wave's share it takes by stride before it turns to the counter (set by the launcher)
    uint32_t ovf_cap;
    uint32_t use_lds_hist;
    uint64_t *pair_keys;     // SC mode pair map (power-of-two capacity)
    uint64_t *pair_cnts;
    uint32_t pair_cap;
    uint64_t *stamps;        // diagnostic builds only (CQ_STAMPS): per-phase cycle sums, else null
};

// What launch_classify chose for a launch (cq_last_launch_info): the fast kernel's instantiation.
struct LaunchInfo {
    int reads_per_subtile = 0;   // R: 8, or 4 for long reads
    int hit_slots = 0;           // CAP of the fast kernel
    int lds_hist = 0;            // per-genome counters in an LDS histogram (1) or as global atomics (0)
    int fixed_shape = 0;         // 1: the instantiation with hash length and batch shape as compile-time constants
    int fixed_h = 0, fixed_read_len = 0;
    int minimizer_len = 0;       // m of the index the launch ran against
    int blocks_per_cu = 0;       // resident workgroups per CU the persistent grid was sized for
};

// ev_start | fast kernel | ev_mid | exact slow-path kernel | ev_stop  (events may be null; info may be null)
hipError_t launch_classify(const DevIndex &ix, QueryArgs a, int n_cus, hipStream_t stream,
                           hipEvent_t ev_start, hipEvent_t ev_mid, hipEvent_t ev_stop, LaunchInfo *info = nullptr);

// dst += src, element-wise, for a counter block (uint64) and an rcount array (uint32; may be null):
// sums shards that ran on the SAME device (cq_multi rehearsal); distinct devices meet in RCCL.
hipError_t launch_accumulate(uint64_t *dst64, const uint64_t *src64, uint64_t n64, uint32_t *dst32,
                             const uint32_t *src32, uint64_t n32, hipStream_t stream);

// Tight rows (stride of sb bytes, most significant base first) -> word rows of sw words, zero-filled past the
// row: what cq_query_packed_tight runs on every chunk after its H2D copy (include/cammiq_hip.h).
hipError_t launch_widen_rows(const uint8_t *tight, uint32_t sb, uint32_t *rows, uint32_t sw, uint64_t n_reads,
                             hipStream_t stream);

// rcount (uint32 per leaf, device) -> one byte per leaf saturated at 255, WRITTEN BY THE KERNEL INTO PAGE-LOCKED HOST
// MEMORY (host_out8: n bytes), + an escape list of (leaf, count) for the entries of 255 and more (device; *esc_count counts
// them all, esc holds the first esc_cap).  host_flags[s] = epoch once segment s (kNarrowSeg entries) is on the host;
// host_flags[n_segments] = epoch and *host_esc_count = the number of escapes once all of them are.  blocks_done: a device
// word, zero between calls.  rc must be 16-byte aligned (cq_api.cpp narrow_start).
constexpr uint64_t kNarrowSeg = 1ull << 16;
hipError_t launch_narrow_rcount(const uint32_t *rc, uint64_t n, uint8_t *host_out8, uint64_t seg, uint32_t *host_flags, uint32_t epoch, uint2 *esc,
                                uint32_t *esc_count, uint32_t esc_cap, uint32_t *blocks_done, uint32_t *host_esc_count, hipStream_t stream);

// Board calibrators behind cq_calibrate (diagnostic): `grid` workgroups of 256 lanes, each lane `iters` random 16-byte
// loads from tab[0 .. n_units); mix = with a returnless atomic into atom[0 .. n_atom) per 16 loads and LDS traffic
// beside the loads.  stamps: 2 x grid words (shader cycles, 100 MHz ticks per workgroup); sink: one word.
hipError_t launch_calib_gather(bool mix, const uint4 *tab, uint64_t n_units, int iters, uint32_t *atom, uint64_t n_atom,
                               uint64_t *stamps, uint32_t *sink, int grid, hipStream_t stream);

// ... and the latency calibrator: `grid` single waves, every lane `iters` DEPENDENT random 16-byte loads (one in flight per lane).
hipError_t launch_calib_chase(const uint4 *tab, uint64_t n_units, int iters, uint64_t *stamps, uint32_t *sink, int grid, hipStream_t stream);

}  // namespace cq
#endif
