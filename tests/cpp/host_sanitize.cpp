// host_sanitize.cpp -- ASAN/UBSAN driver for the host-side code of libcammiq_hip.so that needs no GPU:
// decoder, layout builder (incl. path compression and the host mirror of the device lookup) and packer.
// Built and run by tests/test_sanitize.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all ...
// usage: host_sanitize index_u.bin1 [index_d.bin2|-] reads.txt
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../cammiq_amd/csrc/cq_index.hpp"
#include "../../include/cammiq_glue.hpp"

struct MetaGenome { uint64_t read_cnts_u = 0, read_cnts_d = 0; uint32_t glength = 0, nus = 0, nds = 0; };

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    cq::DecodedTable u, d;
    std::string err;
    int rc = cq::decode_table(argv[1], u, err);
    if (rc != CQ_OK) { printf("decode_u %d %s\n", rc, err.c_str()); return 0; }   // a clean error is a pass
    if (strcmp(argv[2], "-") != 0) {
        rc = cq::decode_table(argv[2], d, err);
        if (rc != CQ_OK) { printf("decode_d %d %s\n", rc, err.c_str()); return 0; }
    } else cq::make_empty_table(u.hash_len, d);
    // the layout's choice of the minimizer length (cq_device.h): 16-mers below 2.5e8 keys, 18-mers from there on, never longer than h
    if (cq_choose_minimizer_len(26, 249999999ull) != 16 || cq_choose_minimizer_len(26, 250000000ull) != 18 ||
        cq_choose_minimizer_len(12, 1ull << 40) != 12 || cq_choose_minimizer_len(17, 1ull << 40) != 17) { printf("MINIMIZER CHOICE\n"); return 1; }
    cq::FlatImage img;
    uint64_t found = 0;
    // the table addressed by 18-mer minimizers (large tables; 64-bit m-mers), then by the automatic choice
    for (uint32_t mlen : {18u, 0u}) {
        rc = cq::build_image(u, d, 0.0, mlen ? std::min(mlen, u.hash_len) : 0u, img, err);
        if (rc != CQ_OK) { printf("layout %d %s\n", rc, err.c_str()); return 0; }
        // every bucket key must be found again, with a sane chain length
        for (const cq::DecodedTable *t : {&u, &d})
            for (uint64_t k : t->bucket_key) {
                uint32_t vu, vd, chain;
                cq::image_lookup(img, k, vu, vd, &chain);
                if ((vu | vd) == 0 || chain > img.max_chain) { printf("LOOKUP FAILED\n"); return 1; }
                found++;
            }
    }
    // pack the reads (one per line; anything goes)
    std::ifstream in(argv[3], std::ios::binary);
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offs(1, 0);
    std::string line;
    while (std::getline(in, line)) { bases.insert(bases.end(), line.begin(), line.end()); offs.push_back(bases.size()); }
    const uint64_t n = offs.size() - 1;
    const uint32_t sw = cq_pack_stride_words(255);
    std::vector<uint32_t> packed(n * sw + 1);
    std::vector<uint8_t> lens(n + 1);
    uint64_t sk = 0;
    rc = cq_pack_reads(bases.data(), offs.data(), n, img.hash_len, sw, packed.data(), lens.data(), &sk);
    // the same reads once more at the tight stride of each read, one row at a time (cq_pack_read: what the CLI's FASTQ
    // loader calls), rows allocated exactly so that a write past a row's end is caught
    for (uint64_t r = 0; r < n; r++) {
        const uint64_t len = offs[r + 1] - offs[r];
        const uint32_t w = cq_pack_stride_words(len > 255 ? 255 : (uint32_t)len);
        std::vector<uint32_t> row(w);
        uint8_t lo = 7;
        if (cq_pack_read(bases.data() + offs[r], (uint32_t)(len > 0xFFFFFFFFull ? 0 : len), 1, w, row.data(), &lo) != CQ_OK) { printf("PACK_READ FAILED\n"); return 1; }
        if (lo != lens[r] && !(lens[r] == 0 && len < img.hash_len)) { printf("PACK_READ LEN %u vs %u\n", lo, lens[r]); return 1; }
    }
    // and as tight rows (cq_pack_read_tight / cq_pack_reads_tight), again with exact allocations
    for (uint64_t r = 0; r < n; r++) {
        const uint64_t len = offs[r + 1] - offs[r];
        const uint32_t sb = cq_pack_stride_bytes(len > 255 ? 255 : (uint32_t)len);
        std::vector<uint8_t> row(sb);
        uint8_t lo = 7;
        if (cq_pack_read_tight(bases.data() + offs[r], (uint32_t)(len > 0xFFFFFFFFull ? 0 : len), 1, sb, row.data(), &lo) != CQ_OK) { printf("PACK_READ_TIGHT FAILED\n"); return 1; }
        if (lo != lens[r] && !(lens[r] == 0 && len < img.hash_len)) { printf("PACK_READ_TIGHT LEN %u vs %u\n", lo, lens[r]); return 1; }
    }
    {
        const uint32_t sb = cq_pack_stride_bytes(100);
        std::vector<uint8_t> tight(n * sb + 1), tl(n + 1);
        uint64_t tsk = 0;
        if (cq_pack_reads_tight(bases.data(), offs.data(), n, img.hash_len, sb, tight.data(), tl.data(), &tsk) != CQ_OK) { printf("PACK_READS_TIGHT FAILED\n"); return 1; }
    }
    // the reads as the reference holds them (cq_query_reads' source): every read a heap block of EXACTLY its length, so that a
    // load past a block's end is caught; rows of exactly n * sb bytes; must give the tight rows of the flattened source
    {
        std::vector<uint8_t *> blocks;
        std::vector<uint8_t> bl;
        std::vector<uint64_t> which;
        for (uint64_t r = 0; r < n; r++) {
            const uint64_t len = offs[r + 1] - offs[r];
            if (len > 255) continue;
            uint8_t *b = len ? new uint8_t[len] : nullptr;
            if (len) memcpy(b, bases.data() + offs[r], len);
            blocks.push_back(b); bl.push_back((uint8_t)len); which.push_back(r);
        }
        const uint64_t m = blocks.size();
        const uint32_t sb = cq_pack_stride_bytes(255);
        std::vector<uint8_t> rows(m * sb), rl(m), ref(n * sb + 1), reflen(n + 1);
        cq::ReadSource src;
        src.ptrs = blocks.data();
        src.lens = bl.data();
        uint64_t psk = 0, rsk = 0;
        uint32_t mn = 0, mx = 0;
        cq::pack_tight_slice(src, 0, m, img.hash_len, sb, rows.data(), rl.data(), &psk, &mn, &mx);
        if (cq_pack_reads_tight(bases.data(), offs.data(), n, img.hash_len, sb, ref.data(), reflen.data(), &rsk) != CQ_OK) { printf("PACK_READS_TIGHT FAILED\n"); return 1; }
        for (uint64_t i = 0; i < m; i++)
            if (rl[i] != reflen[which[i]] || memcmp(rows.data() + i * sb, ref.data() + which[i] * sb, sb)) { printf("POINTER SOURCE DIFFERS at read %llu\n", (unsigned long long)which[i]); return 1; }
        for (uint8_t *b : blocks) delete[] b;
    }
    // meta files next to index_u (glue header): whatever is there -- present, absent or damaged -- must parse cleanly
    {
        std::vector<MetaGenome> genomes(std::min<uint64_t>((uint64_t)img.max_refid + 2, 100000));   // a damaged index may carry any refID
        const char *msg = cq_glue::load_genome_meta(cq_glue::index_dir(argv[1]), genomes, true);
        printf("meta %s", msg ? msg : "loaded\n");
    }
    printf("ok leaves %zu+%zu keys %llu nodes %zu found %llu reads %llu skipped %llu rc %d\n", u.leaves.size(), d.leaves.size(),
           (unsigned long long)img.n_keys, img.nodes.size(), (unsigned long long)found, (unsigned long long)n,
           (unsigned long long)sk, rc);
    return 0;
}
