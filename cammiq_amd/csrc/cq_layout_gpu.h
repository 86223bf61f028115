// cq_layout_gpu.h -- the table of the flat image laid out on the device (cq_layout_gpu.hip).
#ifndef CQ_LAYOUT_GPU_H_
#define CQ_LAYOUT_GPU_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cq {

struct DeviceLayoutResult {
    void *d_table = nullptr;          // n_buckets_alloc x 64 bytes, hipMalloc'ed: the caller owns it
    uint64_t n_buckets_alloc = 0;     // hash range + spill tail (grown when the last homes are crowded, as on the host)
    uint64_t n_keys = 0;              // distinct keys over both tables
    uint64_t n_overflowed = 0;        // buckets whose overflow flag is set
    uint32_t max_chain = 1;           // longest bucket chain a lookup can walk
    bool limit = false;               // the table would pass 2^32 buckets (CQ_ERR_LIMIT)
};

// d_keys[n] / d_vals[n], n = nb_u + nb_d: the buckets of ht_u in file order, then those of ht_d (keys = the reference's
// map64 keys, vals = final trie codes, cq_layout.cpp prepare_image); d_leaf_rids as the kernels read it.  Lays out the
// merged open-addressing table for a hash range of n_buckets exactly as finish_image_host does (byte-identical).
// unsupported = true (and hipSuccess): a home bucket holds more keys than the device path sorts -- use the host builder.
hipError_t layout_table_on_device(const uint64_t *d_keys, const uint32_t *d_vals, uint64_t nb_u, uint64_t nb_d, uint32_t h, uint32_t m,
                                  uint32_t n_buckets, const uint2 *d_leaf_rids, int n_cus, DeviceLayoutResult &out, bool &unsupported,
                                  void *prealloc = nullptr, uint64_t prealloc_buckets = 0);
// prealloc: a hipMalloc'ed block of prealloc_buckets x 64 bytes the table may be built in (taken over, or freed when too
// small): the 80 GB hipMalloc of a configs[4]-size table takes ~2 s, which the caller can spend beside the host part.

}  // namespace cq
#endif
