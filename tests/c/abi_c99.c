/* The boundary is a plain C ABI: this file is compiled as strict C99 against include/cammiq_hip.h
 * and linked with libcammiq_hip.so only.  No compute call (no GPU needed): load an index host-only,
 * read its info and leaves, probe it, pack a read, and check the error contract. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cammiq_hip.h"

int main(int argc, char **argv)
{
    cq_index *ix = NULL;
    cq_index_info info;
    cq_leaf *leaves;
    const char *pd = (argc > 2 && strcmp(argv[2], "-") != 0) ? argv[2] : NULL;
    uint32_t cu = 1, cd = 1, chain = 0;
    const uint8_t read[] = "ACGTACGTACGTACGTACGTACGTACGTACGT";
    uint64_t offs[2], skipped = 99;
    uint32_t packed[8];
    uint8_t len = 0;

    if (argc < 2) return 2;
    if (cq_abi_version() != CQ_ABI_VERSION) return 3;
    if (cq_index_load("/nonexistent/index_u.bin1", NULL, CQ_DEVICE_NONE, &ix) != CQ_ERR_IO || ix != NULL) return 4;
    if (strlen(cq_last_error()) == 0) return 5;
    if (cq_index_load(argv[1], pd, CQ_DEVICE_NONE, &ix) != CQ_OK) { fprintf(stderr, "%s\n", cq_last_error()); return 6; }
    if (cq_index_get_info(ix, &info) != CQ_OK) return 7;
    leaves = (cq_leaf *)malloc((size_t)(info.n_leaves[0] ? info.n_leaves[0] : 1) * sizeof(cq_leaf));
    if (cq_index_leaves(ix, 0, leaves) != CQ_OK) return 8;
    if (info.n_leaves[0] && leaves[0].refID1 == 0) return 9;
    if (cq_index_probe(ix, 0xFFFFFFFFFFFFFFFFull, &cu, &cd, &chain) != CQ_OK || cu != 0 || cd != 0) return 10;
    /* no CPU classify path: a host-only handle must refuse to query */
    {
        uint64_t c1[8] = {0}, c2[8] = {0};
        cq_counts c;
        memset(&c, 0, sizeof c);
        c.cnt_u = c1; c.cnt_d = c2;
        offs[0] = 0; offs[1] = 32;
        if (cq_query(ix, CQ_MODE_SC, read, offs, 1, 4, &c) != CQ_ERR_NO_DEVICE) return 11;
    }
    offs[0] = 0; offs[1] = 32;
    if (cq_pack_stride_words(32) != 2 || cq_pack_stride_words(100) != 7) return 12;
    if (cq_pack_reads(read, offs, 1, info.hash_len, 4, packed, &len, &skipped) != CQ_OK) return 13;
    if (len != 32 || skipped != 0 || packed[0] != 0x1B1B1B1Bu) return 14;   /* ACGT = 00 01 10 11, MSB first */
    {   /* round-2 entry points that need no GPU: single-row packer, sharding rule, packed query on a host-only handle */
        uint32_t row[2] = {0, 0};
        uint8_t l2 = 0;
        uint64_t lo = 9, hi = 9;
        uint64_t c1[8] = {0}, c2[8] = {0};
        cq_counts c;
        if (cq_pack_read(read, 32, 1, 2, row, &l2) != CQ_OK || l2 != 32 || row[0] != 0x1B1B1B1Bu || row[1] != 0x1B1B1B1Bu) return 15;
        if (cq_shard_range(10, 1, 3, &lo, &hi) != CQ_OK || lo != 3 || hi != 6) return 16;
        if (cq_shard_range(10, 3, 3, &lo, &hi) != CQ_ERR_ARG) return 17;
        memset(&c, 0, sizeof c);
        c.cnt_u = c1; c.cnt_d = c2;
        if (cq_query_packed(ix, CQ_MODE_SC, packed, &len, 1, 4, 32, 4, &c) != CQ_ERR_NO_DEVICE) return 18;
        if (cq_multi_size(NULL) != 0 || cq_multi_index(NULL, 0) != NULL) return 19;
        {   /* tight rows (ABI 3): 32 bases = 8 bytes, most significant base first */
            uint8_t trow[8] = {0}, tl = 0;
            int k;
            if (cq_pack_stride_bytes(100) != 25 || cq_pack_stride_bytes(32) != 8) return 20;
            if (cq_pack_read_tight(read, 32, 1, 8, trow, &tl) != CQ_OK || tl != 32) return 21;
            for (k = 0; k < 8; k++) if (trow[k] != 0x1B) return 22;
            if (cq_pack_reads_tight(read, offs, 1, info.hash_len, 8, trow, &tl, &skipped) != CQ_OK || tl != 32 || skipped != 0) return 23;
            if (cq_query_packed_tight(ix, CQ_MODE_SC, trow, &tl, 1, 8, 32, 4, &c) != CQ_ERR_NO_DEVICE) return 24;
            if (cq_multi_query_packed_tight(NULL, CQ_MODE_SC, trow, &tl, 1, 8, 32, 4, &c) != CQ_ERR_ARG) return 25;
        }
    }
    printf("ok hash_len %u leaves %llu+%llu keys %llu\n", info.hash_len, (unsigned long long)info.n_leaves[0],
           (unsigned long long)info.n_leaves[1], (unsigned long long)info.n_keys);
    free(leaves);
    cq_index_free(ix);
    return 0;
}
