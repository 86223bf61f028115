/*
 * cammiq_hip.h -- C ABI of libcammiq_hip.so, the MI355X (gfx950) engine for CAMMiQ's
 * read-classification hot path.
 *
 * The reference (mounted at /root/reference; citations relative to its src/) has no
 * plugin or FFI seam: the path is two C++ member functions that mutate FqReader state,
 *
 *     void FqReader::query64_p  (size_t file_idx)      query.hpp:113, query.cpp:458-648
 *     void FqReader::query64mt_p(size_t file_idx)      query.hpp:114, query.cpp:650-889
 *     void FqReader::query64_sc (size_t file_idx)      query.hpp:115, query.cpp:891-1080
 *
 * called from queryFastq_p (query.cpp:247-250,286-289) / queryFastq_sc (:319,356), on top
 * of Hash::loadIdx64_p (hashtrie.cpp:486-507) and Hash::find64_p (:350-369).  This header
 * is the boundary a maintainer binds instead (INTEGRATION.md shows the ~40-line patch):
 * plain pointers and sizes, no C++ or torch types, caller owns every in/out array, the
 * library owns the index and its device buffers behind an opaque handle.
 *
 * Every function returns CQ_OK (0) or a negative cq_status; the text of the last error
 * on the calling thread is available from cq_last_error().  The library never aborts and
 * has NO CPU fallback for the classify step: without a usable HIP device cq_query*
 * fail with CQ_ERR_NO_DEVICE.
 *
 * Threading (as the reference: one caller, FqReader is not re-entrant): a handle carries ONE
 * query at a time.  Its workspace -- slow-path list, kernel-timing events, SC pair map, staging
 * buffers -- is shared by all calls on it, so do not overlap two cq_query* calls (or two
 * cq_query_device launches on different streams) on the same handle; different handles are
 * independent, and consecutive cq_query_device launches on ONE stream may be queued back to back.
 */
#ifndef CAMMIQ_HIP_H_
#define CAMMIQ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CQ_ABI_VERSION 6   /* 6: cq_query_reads, cq_multi_query_reads (the reference's reads[] / rlengths[] as they are); 5: cq_calibrate, cq_rcount_fetch; 4: cq_last_launch_info; every host-fed SC-mode query starts from an empty pair map */

/*
 * Design limits of one handle (= one GPU's replica of the index).  The reference's pointer trie has none beyond its
 * host's memory (hashtrie.hpp:8-13,37-50; util.hpp:12-14 bounds genomes, not markers); here cq_index_load returns
 * CQ_ERR_LIMIT with a message, never a wrapped index:
 *     markers (leaves of ht_u + ht_d)   <= 2^31 - 2     trie codes keep two flag bits of their 32 (cq_device.h)
 *     trie nodes after path compression <  2^30
 *     table buckets                     <  2^32          (one bucket per key by default: implied by the leaf limit)
 *     key length                        <= 255          (as the reference: pleafNode::depth is a uint8_t)
 * 2.1e9 markers are ~137 GB of table + ~26 GB of refIDs and rcount on the device (of 288 GB) and ~1.7 x RefSeq-scale
 * (1-2e8 markers per 1000 bacterial genomes -> 1.5-3e9 for configs[4]'s ~15 000): an index beyond the
 * limit is served by splitting its GENOMES over two handles / GPUs -- the markers of a genome subset are a valid
 * index, and cnt_u/cnt_d/rcount of disjoint marker sets do not add up to the whole (a read's decision looks at all
 * its hits), so that split needs the two hit lists merged before the decision: not built, stated here as the limit.
 * Measured at 1.26e9 markers (15 000 genomes x 3.45 Mbp at the survey's density): DESIGN.md section 6.3.
 */

typedef enum cq_status {
    CQ_OK = 0,
    CQ_ERR_ARG = -1,       /* NULL / out-of-range argument */
    CQ_ERR_IO = -2,        /* cannot open/read a file (reference: abort(), binaryio.cpp:190-197) */
    CQ_ERR_FORMAT = -3,    /* malformed .binN/.aux (reference: assert, hashtrie.cpp:446,492) */
    CQ_ERR_HASHLEN = -4,   /* hash_len_u != hash_len_d (reference: assert, query.cpp:460) */
    CQ_ERR_RANGE = -5,     /* a refID exceeds n_genomes (reference: genomes[rid] out of bounds) */
    CQ_ERR_NO_DEVICE = -6, /* no HIP device / handle was loaded host-only */
    CQ_ERR_HIP = -7,       /* a HIP runtime call failed */
    CQ_ERR_NOMEM = -8,
    CQ_ERR_LIMIT = -9,     /* a design limit above is exceeded, or the caller's pair arrays are too small */
    CQ_ERR_COMM = -10      /* an RCCL call failed (multi-GPU entry points) */
} cq_status;

/* Which classify routine is restated. */
#define CQ_MODE_P 0  /* query64_p / query64mt_p: counts + per-leaf rcount            */
#define CQ_MODE_SC 1 /* query64_sc: no rcount, pair counts, |P|>=2&|I|=1 bumps u too */

#define CQ_TABLE_U 0 /* ht_u  (index_u.bin1) */
#define CQ_TABLE_D 1 /* ht_d  (index_d.bin2) */

#define CQ_DEVICE_NONE (-1) /* cq_index_load: decode + lay out on the host only (inspection) */

typedef struct cq_index cq_index; /* opaque: replaces FqReader::ht_u / ht_d (query.hpp:57-58) */

/* One decoded leaf = pleafNode minus rcount (hashtrie.hpp:37-47).  16 bytes. */
typedef struct cq_leaf {
    uint32_t refID1;
    uint32_t refID2;  /* 0 for a unique marker */
    uint16_t ucount1;
    uint16_t ucount2;
    uint8_t depth;    /* total key length = hash_len + trie depth (hashtrie.cpp:442,470) */
    uint8_t pad_[3];
} cq_leaf;

typedef struct cq_index_info {
    uint32_t abi_version;
    uint32_t hash_len;         /* h, shared by both tables */
    uint32_t max_refid;        /* largest refID in any leaf */
    int32_t device;            /* HIP ordinal or CQ_DEVICE_NONE */
    uint32_t doubly_flag[2];   /* header bit of each file (hashtrie.cpp:489) */
    uint64_t n_leaves[2];      /* Hash::leaf_cnt per table */
    uint64_t n_file_buckets[2];/* buckets decoded per file (== map64.size() unless duplicated) */
    uint64_t n_trie_nodes;     /* internal trie nodes, both tables */
    uint64_t n_keys;           /* distinct h-mer prefixes over both tables (merged table) */
    uint64_t n_table_buckets;  /* 64-byte buckets in the device table (incl. spill tail) */
    uint64_t n_overflowed;     /* buckets whose overflow bit is set */
    uint32_t max_chain;        /* longest bucket chain a lookup can walk */
    uint32_t reserved_;        /* 1 when the image came from the CAMMIQ_IMAGE_CACHE file "<path_u>.cqimg" (=1: handles without a device and
                                  tables the host lays out; =force: every handle) */
    uint64_t device_bytes;     /* HBM held by this handle */
    uint32_t minimizer_len;    /* m-mer length the table is addressed by: 16, or 18 from 2.5e8 keys on (CQ_MINIMIZER_LARGE_FROM, cq_device.h) */
    uint32_t reserved2_;
} cq_index_info;

/*
 * Replaces FqReader::loadIdx_p -> Hash::loadIdx64_p (query.cpp:109-123, hashtrie.cpp:486-507).
 * Reads path_u and path_u+".aux" (and the same for path_d) in the reference's on-disk format,
 * unchanged.  path_d may be NULL or "": the doubly-unique table is then empty (what a
 * --unique build leaves behind; the reference itself would abort on the missing file).
 * device >= 0 uploads the flat table + trie to that GPU; CQ_DEVICE_NONE keeps it on the host.
 */
int cq_index_load(const char *path_u, const char *path_d, int device, cq_index **out);

int cq_index_get_info(const cq_index *idx, cq_index_info *info);

/*
 * Leaves of one table in FILE DECODE ORDER -- the order Hash::map_sp is filled in
 * (hashtrie.cpp:452-453,476), so the host can rebuild map_sp[g] exactly:
 * for i in order: push i to map_sp[refID1] (and to map_sp[refID2] when refID2 != 0).
 * out must hold info.n_leaves[table] entries.
 */
int cq_index_leaves(const cq_index *idx, int table, cq_leaf *out);

/*
 * Diagnostic: host-side lookup of one h-mer (2*hash_len-bit value, the reference's map64 key)
 * in the merged table, following exactly the bucket chain the GPU follows.  code_u / code_d
 * receive the bucket root in ht_u / ht_d: 0 absent, 0x80000000|global leaf id (u leaves
 * first, then d leaves) or a trie node index.  chain = buckets read.  Not a classify path.
 */
int cq_index_probe(const cq_index *idx, uint64_t hv, uint32_t *code_u, uint32_t *code_d, uint32_t *chain);

void cq_index_free(cq_index *idx);

/* Outputs of one classify call == the FqReader state query64_* mutates. */
typedef struct cq_counts {
    uint64_t *cnt_u;     /* [n_genomes+1]  Genome::read_cnts_u (query.hpp:15), index 0 unused */
    uint64_t *cnt_d;     /* [n_genomes+1]  Genome::read_cnts_d (query.hpp:16) */
    uint32_t *rcount_u;  /* [n_leaves[U]]  pleafNode::rcount, decode order; may be NULL in SC mode */
    uint32_t *rcount_d;  /* [n_leaves[D]]  (both are written in full by every CQ_MODE_P query.  Over the link they travel
                            as one byte per leaf + an escape list and are widened into these arrays by the library's own
                            threads -- bit-exact; page-locked arrays from cq_host_alloc and plain arrays both work) */
    uint64_t nundet;     /* FqReader::nundet (query.hpp:43) */
    uint64_t nconf;      /* FqReader::nconf  (query.hpp:40) */
    uint64_t nskipped;   /* reads outside the parity domain (len < h, len > 255, non-ACGT byte):
                            not classified, not counted anywhere else.  Reference: UB. */
    /* CQ_MODE_SC only: FqReader::read_cnts_b (query.hpp:49) as triples, any order. */
    uint32_t *pair_a;    /* [pair_cap] smaller refID */
    uint32_t *pair_b;    /* [pair_cap] larger refID  */
    uint64_t *pair_cnt;  /* [pair_cap] */
    uint64_t pair_cap;
    uint64_t n_pairs;    /* out: distinct pairs (CQ_ERR_LIMIT if > pair_cap) */
} cq_counts;

/*
 * Replaces one call of query64_p / query64mt_p / query64_sc on reads that sit in host
 * memory as ASCII, exactly as FqReader::readFastq leaves them (query.cpp:371-393):
 * read r is bases[offsets[r] .. offsets[r+1]).  Packs to 2 bit, copies to the GPU,
 * classifies, copies the counters back.  Counters are OVERWRITTEN (one call == one FASTQ
 * after resetCounters, query.cpp:1820-1840).  Synchronous.
 */
int cq_query(cq_index *idx, int mode, const uint8_t *bases, const uint64_t *offsets,
             uint64_t n_reads, uint32_t n_genomes, cq_counts *out);

/*
 * The same call on the reference's own two arrays, as they are: reads[r] = one heap block of ASCII per read (no
 * terminator), rlengths[r] = its length -- FqReader::reads[f].data() and FqReader::rlengths[f].data() (query.hpp:35-36,
 * filled by readFastq, query.cpp:371-393).  Nothing is flattened or copied on the caller's side: the library's packer
 * threads read every block where it lies and write 2-bit rows into page-locked chunks while the GPU classifies the
 * previous chunk.  Same results as cq_query on the same reads; a NULL block of non-zero length is skipped (nskipped).
 */
int cq_query_reads(cq_index *idx, int mode, const uint8_t *const *reads, const uint8_t *rlengths,
                   uint64_t n_reads, uint32_t n_genomes, cq_counts *out);

/*
 * Same call for reads the host has already packed with cq_pack_reads (packed + lens, row stride
 * stride_words, longest read max_len): chunks are copied to the GPU on a copy stream while the
 * previous chunk is being classified, counters come back at the end -- H2D + kernels + D2H, the
 * bracket of the reference's "Time for query" (query.cpp:459,645-647) without the host-side
 * packing.  Buffers from cq_host_alloc (pinned) make both directions run at link speed; pageable
 * memory works too, slower.  max_len is a hint (0 = unknown): the longest length found in `lens` is what
 * sizes the work; a length above 16 * stride_words is CQ_ERR_ARG.
 */
int cq_query_packed(cq_index *idx, int mode, const uint32_t *packed, const uint8_t *lens,
                    uint64_t n_reads, uint32_t stride_words, uint32_t max_len, uint32_t n_genomes,
                    cq_counts *out);

/* Page-locked host memory for packed reads and for the rcount arrays of cq_counts. */
int cq_host_alloc(void **p, size_t bytes);
void cq_host_free(void *p);

/* ---- packed / device-resident interface (multi-GPU hosts, benchmarks, pipelines) ---- */

/* Words (uint32) per packed read for reads up to max_len bases: ceil(max_len / 16), 1..16.  Any stride in
 * that range that holds the longest read is accepted by the calls below (rows are not padded to 16 bytes). */
uint32_t cq_pack_stride_words(uint32_t max_len);

/*
 * ASCII -> 2-bit rows.  Base j of a read goes to word j/16, bits [31-2(j%16) : 30-2(j%16)]
 * (A=0 C=1 G=2 T=3, either case -- FqReader::symbolIdx, query.cpp:1860-1873).
 * lens[r] = read length, or 0 when the read is outside the parity domain (see nskipped).
 * packed: n_reads*stride_words uint32; lens: n_reads uint8.  Host only, multi-threaded.
 */
int cq_pack_reads(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                  uint32_t hash_len, uint32_t stride_words, uint32_t *packed, uint8_t *lens,
                  uint64_t *n_skipped);

/* One read (seq[0..len)) into one row of stride_words words; *len_out = len, or 0 when the read is outside the
 * parity domain.  Thread-safe, allocation-free: a FASTQ parser's threads call it on the sequence lines of a
 * memory-mapped file and never materialise the ASCII reads (cammiq_main.cpp).  hash_len = 1 leaves the
 * "shorter than h" rule to the kernel, which counts such reads in nskipped as well. */
int cq_pack_read(const uint8_t *seq, uint32_t len, uint32_t hash_len, uint32_t stride_words, uint32_t *row,
                 uint8_t *len_out);

/*
 * Tight rows: the same 2-bit string at a stride of whole BYTES -- ceil(max_len / 4), 1..64: 25 bytes for a
 * 100-bp read instead of 28 -- for reads that have to cross the host link (H2D is what bounds a host-fed
 * query: 57 GB/s against 2.2 G reads/s of kernel).  Base j of a read goes to byte j/4, bits
 * [7-2(j%4) : 6-2(j%4)]; row r starts at packed + r * stride_bytes, no alignment.  cq_query_packed_tight copies
 * the tight rows to the GPU, where the classify kernel's own staging widens them (no second copy of the rows in
 * HBM; rounds 2-3 ran a small widening kernel per chunk); lengths of a chunk whose reads all have one length are
 * not sent at all; everything else is cq_query_packed.  A length above 4 * stride_bytes is CQ_ERR_ARG.
 */
uint32_t cq_pack_stride_bytes(uint32_t max_len);
int cq_pack_reads_tight(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                        uint32_t hash_len, uint32_t stride_bytes, uint8_t *packed, uint8_t *lens,
                        uint64_t *n_skipped);
int cq_pack_read_tight(const uint8_t *seq, uint32_t len, uint32_t hash_len, uint32_t stride_bytes, uint8_t *row,
                       uint8_t *len_out);
int cq_query_packed_tight(cq_index *idx, int mode, const uint8_t *packed, const uint8_t *lens,
                          uint64_t n_reads, uint32_t stride_bytes, uint32_t max_len, uint32_t n_genomes,
                          cq_counts *out);

/* Number of uint64 in the device counter block for n_genomes:
 * [cnt_u[G+1] | cnt_d[G+1] | nundet nconf nskipped flags nslow 0 0 0].  flags != 0: increments of the
 * SC pair map were lost (map full).  Every word adds up across GPUs, flags included. */
uint64_t cq_counter_words(uint32_t n_genomes);

/*
 * Classify reads that are already resident in HBM.  Asynchronous on `stream` (a hipStream_t,
 * NULL = default stream).  ACCUMULATES into d_counters (cq_counter_words(n_genomes) uint64,
 * layout above) and d_rcount (n_leaves[U]+n_leaves[D] uint32, U first; may be NULL in SC
 * mode); the caller zeroes them, sums them across GPUs (RCCL) and copies them back.
 * d_packed/d_lens as produced by cq_pack_reads; max_len = longest read in the batch (an upper
 * bound is fine: it only sizes the lane grid; 0 = stride_words*16).  n_reads <= 2^31 per call.
 * SC-mode pair counts accumulate inside the handle; fetch with cq_pairs_fetch.
 */
int cq_query_device(cq_index *idx, int mode, const uint32_t *d_packed, const uint8_t *d_lens,
                    uint64_t n_reads, uint32_t stride_words, uint32_t max_len, uint32_t n_genomes,
                    uint64_t *d_counters, uint32_t *d_rcount, void *stream);

/* SC mode: copy out and clear the pair counters accumulated by cq_query_device.  Synchronises.
 * With pair_a == NULL only *n_pairs is written (nothing copied, nothing cleared); with arrays that
 * are too small: CQ_ERR_LIMIT, *n_pairs = the number needed, nothing cleared -- call again.
 * The host-fed entries (cq_query*, cq_multi_query*) start every SC-mode query from an EMPTY map:
 * pairs kept after a CQ_ERR_LIMIT can be fetched here only until the next query on the handle. */
int cq_pairs_fetch(cq_index *idx, uint32_t *pair_a, uint32_t *pair_b, uint64_t *pair_cnt,
                   uint64_t pair_cap, uint64_t *n_pairs);

/* SC mode, device interface: size the pair map for at least n_slots slots (rounded up to a power of
 * two; keep it under half full) and clear it.  cq_query / cq_query_packed / cq_multi_query grow the map
 * by themselves and classify again when it fills up; cq_query_device cannot (it only reports
 * flags != 0 in the counter block). */
int cq_pairs_reserve(cq_index *idx, uint64_t n_slots);

/* Duration in ms of the classify kernels of the most recent cq_query_device call on this handle
 * (HIP events on the call's stream; synchronises on the last one): the main kernel and the exact
 * slow-path kernel that re-classifies reads with more hits than the main kernel's lists hold.
 * cq_last_kernel_ms = their sum. */
int cq_last_kernel_times(cq_index *idx, float *fast_ms, float *slow_ms);
int cq_last_kernel_ms(cq_index *idx, float *ms);

/* Which instantiation of the classify kernel the most recent cq_query_device launch on this handle ran
 * (profiling records name the kernel they measured by this, not by a constant): reads per wave sub-tile
 * (8; 4 for reads >~ 235 bp), hit slots per read, per-genome counters in an LDS histogram (1) or as
 * global atomics (0), and whether the build with CAMMiQ's default hash length 26
 * (/root/reference/src/main.cpp:335-346) and the batch's read length (100 / 150) folded in as
 * compile-time constants was used (same results; tests run both). */
typedef struct cq_launch_info {
    int32_t reads_per_subtile;
    int32_t hit_slots;
    int32_t lds_hist;
    int32_t fixed_shape;
    int32_t fixed_hash_len;   /* 0 unless fixed_shape */
    int32_t fixed_read_len;   /* 0 unless fixed_shape */
    int32_t blocks_per_cu;    /* resident workgroups per CU the persistent grid was sized for */
    int32_t minimizer_len;    /* m of the index (16 / 18): part of the instantiation's name when fixed_shape */
} cq_launch_info;
int cq_last_launch_info(cq_index *idx, cq_launch_info *out);

/*
 * The device door's way back for rcount: d_rcount (n_leaves[U] + n_leaves[D] uint32 in HBM, U first, 16-byte aligned --
 * what cq_query_device accumulated into, after cq_counts_allreduce where several GPUs take part) -> the caller's two
 * host arrays, the reference's pleafNode::rcount per leaf in decode order (hashtrie.hpp:43; read by the ILP,
 * query.cpp:1161,1176-1177).  Ordered behind everything queued on `stream`; synchronous: returns with the arrays
 * filled.  The transport is the host-fed doors': one byte per leaf + an escape list written by a kernel straight into
 * page-locked memory, widened by the library's threads while it arrives (a quarter of a plain uint32 copy's time on
 * the link); plain arrays and cq_host_alloc'ed ones both work.
 */
int cq_rcount_fetch(cq_index *idx, const uint32_t *d_rcount, void *stream, uint32_t *rcount_u, uint32_t *rcount_d);

/*
 * Diagnostic, no reference analogue: what THIS board gives the classify kernel to work with, measured in about a
 * second on the handle's own table in HBM (same bytes, same allocation, nothing is modified).  The classify kernel is
 * bound by the chip's rate of random loads from a multi-GB table, and boards of one pool differ in it by more than
 * 10 %: a benchmark line that carries these numbers says whether a slow run was the board or the build.
 *   gather16_Glines_s      random 16-byte loads per second (in 10^9), four independent loads in flight per lane, best of
 *                          4 / 6 / 8 workgroups per CU -- the ceiling the probe loop's bucket reads are held against
 *   gather16_mix_Glines_s  the same loads with a returnless global atomic per 16 loads into an rcount-sized array and
 *                          LDS stores / reads beside them (what the real kernel has and a plain gather lacks)
 *   chase16_Glines_s       DEPENDENT random 16-byte loads per second (one in flight per lane: the next address is a
 *                          function of the loaded quad) from ONE wave per CU -- far below what saturates the memory
 *                          system; chase_latency_ns = lanes in flight / that rate -- what the memory system answers
 *                          a lone request in, which a latency-hiding kernel follows and a saturated gather does not show
 *   clock_MHz_*            shader clock held DURING each of the kernels: shader cycles / 100 MHz constant-clock
 *                          ticks (s_memtime / s_memrealtime), median over workgroups
 * Not part of any query; a handle without a device returns CQ_ERR_NO_DEVICE.
 */
typedef struct cq_calibration {
    double gather16_Glines_s;
    double gather16_mix_Glines_s;
    double clock_MHz_gather;
    double clock_MHz_mix;
    double table_bytes;         /* bytes of the table the loads were spread over */
    double seconds;             /* wall time the calibration took */
    int32_t gather_blocks_per_cu, mix_blocks_per_cu;   /* the occupancy that gave the best rate */
    double chase16_Glines_s;
    double chase_latency_ns;
    double clock_MHz_chase;
} cq_calibration;
int cq_calibrate(cq_index *idx, cq_calibration *out);

/* ---- multi-GPU -----------------------------------------------------------------------------
 * No reference analogue: the reference's one parallel axis is the OpenMP loop over reads
 * (query.cpp:664-665).  Here that loop is cut into P contiguous shards, one per GPU, against an
 * index replicated in every GPU's HBM; the outputs are commutative integer sums, so ONE RCCL
 * all-reduce(sum) of the counter block (uint64) and of rcount (uint32) at the end of a query gives
 * every GPU -- and the host, before it hands the counts to the ILP, query.cpp:251-258 -- exactly
 * the single-GPU result.  Shard p of P takes reads [n*p/P, n*(p+1)/P). */
int cq_shard_range(uint64_t n_reads, int rank, int n_ranks, uint64_t *lo, uint64_t *hi);

/* (1) One process, several GPUs: one host thread per device inside the library. */
typedef struct cq_multi cq_multi;

/* Decode + lay out once, upload to every listed device, create the communicators
 * (ncclCommInitAll).  A device may be listed more than once (rehearsal on a box with fewer GPUs
 * than shards): shards on one device are summed by a kernel, distinct devices by RCCL. */
int cq_multi_load(const char *path_u, const char *path_d, const int *devices, int n_dev, cq_multi **out);
int cq_multi_size(const cq_multi *m);
/* Borrowed handle of shard i (cq_index_get_info, cq_index_leaves, ...); owned by m. */
cq_index *cq_multi_index(cq_multi *m, int i);
/* cq_query / cq_query_packed over all devices of m; same arguments, same results. */
int cq_multi_query(cq_multi *m, int mode, const uint8_t *bases, const uint64_t *offsets,
                   uint64_t n_reads, uint32_t n_genomes, cq_counts *out);
int cq_multi_query_reads(cq_multi *m, int mode, const uint8_t *const *reads, const uint8_t *rlengths,
                         uint64_t n_reads, uint32_t n_genomes, cq_counts *out);
int cq_multi_query_packed(cq_multi *m, int mode, const uint32_t *packed, const uint8_t *lens,
                          uint64_t n_reads, uint32_t stride_words, uint32_t max_len, uint32_t n_genomes,
                          cq_counts *out);
int cq_multi_query_packed_tight(cq_multi *m, int mode, const uint8_t *packed, const uint8_t *lens,
                                uint64_t n_reads, uint32_t stride_bytes, uint32_t max_len, uint32_t n_genomes,
                                cq_counts *out);
void cq_multi_free(cq_multi *m);

/* (2) One process per GPU (torchrun / mpirun style): rank 0 makes an id, the launcher's own
 * channel carries its CQ_COMM_ID_BYTES bytes to the other ranks, every rank joins with its
 * handle's device (ncclCommInitRank), classifies its shard with cq_query_device and ends the
 * query with cq_counts_allreduce (asynchronous on `stream`, in place). */
#define CQ_COMM_ID_BYTES 128
typedef struct cq_comm cq_comm;
int cq_comm_unique_id(uint8_t *id);
int cq_comm_init_rank(cq_index *idx, const uint8_t *id, int rank, int n_ranks, cq_comm **out);
int cq_comm_info(const cq_comm *comm, int *rank, int *n_ranks);
int cq_counts_allreduce(cq_comm *comm, uint64_t *d_counters, uint64_t n_counter_words,
                        uint32_t *d_rcount, uint64_t n_rcount, void *stream);
void cq_comm_free(cq_comm *comm);

const char *cq_last_error(void);
int cq_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CAMMIQ_HIP_H_ */
