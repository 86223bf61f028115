// ilp_handoff.cpp -- compiles the reference-side binding (include/cammiq_glue.hpp, the code
// INTEGRATION.md asks a maintainer to add to FqReader) against a small OWN mock of the three
// reference types it touches, and prints the state the unmodified ILP would read afterwards.
// The mock repeats only member NAMES and types of /root/reference/src/query.hpp:13-25 (Genome),
// hashtrie.hpp:37-47 (pleafNode), :50,55,60 (HashMapSP, Hash::map_sp, leaf_cnt); no reference
// header or source is included.  tests/test_handoff.py checks the printed state against the oracle.
//
//   ilp_handoff <index_u> <index_d|-> <n_genomes> <device|-1> [reads.txt [P|SC]]
//
// device -1: host-only handle (CQ_DEVICE_NONE) -- map_sp + meta files only, runs without a GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/cammiq_glue.hpp"

// ---- mock of the reference's host-side types (names as in the reference) ----
struct Genome {
    uint64_t read_cnts_u = 0, read_cnts_d = 0;
    uint32_t glength = 0, nus = 0, nds = 0, taxID = 0;
    std::string name;
};
struct pleafNode {
    uint32_t refID1 = 0, refID2 = 0;
    uint8_t depth = 0;
    uint16_t ucount1 = 0, ucount2 = 0;
    uint32_t rcount = 0;
};
typedef std::unordered_map<uint32_t, std::vector<pleafNode *>> HashMapSP;
struct Hash {
    HashMapSP map_sp;
    uint64_t leaf_cnt = 0;
};
struct FqReader {
    std::vector<Genome *> genomes;
    Hash *ht_u = new Hash(), *ht_d = new Hash();
    size_t nundet = 0, nconf = 0;
    std::map<std::pair<uint32_t, uint32_t>, uint64_t> read_cnts_b;
    std::string IDXFILEU, IDXFILED, IDXDIR = "./";
    uint32_t hash_len_u = 0, hash_len_d = 0;
    // ---- what INTEGRATION.md adds ----
    cq_index *gpu_idx = NULL;
    std::vector<pleafNode> gpu_leaves[2];
    int loadIdx_gpu(int device);
    int query64_gpu(size_t file_idx, int mode);
    // ---- the reads as FqReader::readFastq leaves them (query.hpp:35-36, query.cpp:371-393): one heap block per read, no
    //      terminator, and its length in a byte
    std::vector<uint8_t *> reads[1];
    std::vector<uint8_t> rlengths[1];
};

// FqReader::loadIdx_gpu -- replaces loadIdx_p (query.cpp:109-123)
int FqReader::loadIdx_gpu(int device)
{
    int rc = cq_index_load(IDXFILEU.c_str(), IDXFILED.empty() ? NULL : IDXFILED.c_str(), device, &gpu_idx);
    if (rc != CQ_OK) { fprintf(stderr, "%s\n", cq_last_error()); return rc; }
    cq_index_info info;
    cq_index_get_info(gpu_idx, &info);
    hash_len_u = hash_len_d = info.hash_len;
    rc = cq_glue::rebuild_map_sp(gpu_idx, CQ_TABLE_U, *ht_u, gpu_leaves[0]);
    if (rc == CQ_OK) rc = cq_glue::rebuild_map_sp(gpu_idx, CQ_TABLE_D, *ht_d, gpu_leaves[1]);
    return rc;
}

// FqReader::query64_gpu -- replaces query64_p / query64mt_p / query64_sc
int FqReader::query64_gpu(size_t file_idx, int mode)
{
    const size_t G = genomes.size() - 1;
    std::vector<uint64_t> cu(G + 1), cd(G + 1), pc(1 << 12);
    std::vector<uint32_t> ru(gpu_leaves[0].size()), rd(gpu_leaves[1].size()), pa(1 << 12), pb(1 << 12);
    cq_counts c;
    memset(&c, 0, sizeof c);
    c.cnt_u = cu.data(); c.cnt_d = cd.data();
    c.rcount_u = ru.empty() ? NULL : ru.data(); c.rcount_d = rd.empty() ? NULL : rd.data();
    c.pair_a = pa.data(); c.pair_b = pb.data(); c.pair_cnt = pc.data(); c.pair_cap = pc.size();
    // the two arrays as they are: nothing is flattened or copied on this side
    int rc = cq_query_reads(gpu_idx, mode, reads[file_idx].data(), rlengths[file_idx].data(), reads[file_idx].size(), (uint32_t)G, &c);
    if (rc == CQ_ERR_LIMIT && c.n_pairs > pc.size()) {   // more pairs than the arrays hold: the library kept them
        pa.resize(c.n_pairs); pb.resize(c.n_pairs); pc.resize(c.n_pairs);
        c.pair_a = pa.data(); c.pair_b = pb.data(); c.pair_cnt = pc.data(); c.pair_cap = pc.size();
        rc = cq_pairs_fetch(gpu_idx, c.pair_a, c.pair_b, c.pair_cnt, c.pair_cap, &c.n_pairs);
    }
    if (rc != CQ_OK) { fprintf(stderr, "%s\n", cq_last_error()); return rc; }
    cq_glue::add_counts(c, genomes, gpu_leaves[0], gpu_leaves[1], nundet, nconf, read_cnts_b);
    return CQ_OK;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: ilp_handoff index_u index_d|- n_genomes device|-1 [reads.txt [P|SC]]\n"); return 2; }
    FqReader fq;
    fq.IDXFILEU = argv[1];
    fq.IDXFILED = strcmp(argv[2], "-") ? argv[2] : "";
    fq.IDXDIR = cq_glue::index_dir(fq.IDXFILEU);
    const size_t G = (size_t)atoi(argv[3]);
    const int device = atoi(argv[4]);
    fq.genomes.push_back(NULL);                       // loadSmap: genomes[0] = NULL, ids are 1-based (query.cpp:126)
    for (size_t g = 1; g <= G; g++) fq.genomes.push_back(new Genome());
    if (fq.loadIdx_gpu(device) != CQ_OK) return 1;
    if (const char *msg = cq_glue::load_genome_meta(fq.IDXDIR, fq.genomes, fq.IDXFILED.empty())) { fputs(msg, stderr); return 3; }
    if (argc > 5) {
        std::ifstream in(argv[5]);
        std::string line;
        while (std::getline(in, line)) {   // readFastq's own bookkeeping: a block of exactly the read's length per read
            if (line.empty() || line.size() > 255) continue;
            uint8_t *blk = new uint8_t[line.size()];
            memcpy(blk, line.data(), line.size());
            fq.reads[0].push_back(blk);
            fq.rlengths[0].push_back((uint8_t)line.size());
        }
        const int mode = (argc > 6 && !strcmp(argv[6], "SC")) ? CQ_MODE_SC : CQ_MODE_P;
        if (fq.query64_gpu(0, mode) != CQ_OK) return 1;
        for (uint8_t *blk : fq.reads[0]) delete[] blk;
    }
    // ---- the state runILP_* reads (query.cpp:1100-1226), in the order it iterates it
    printf("nundet %zu nconf %zu\n", fq.nundet, fq.nconf);
    Hash *ht[2] = {fq.ht_u, fq.ht_d};
    for (size_t i = 1; i <= G; i++) {
        const Genome &g = *fq.genomes[i];
        printf("G %zu %lu %lu %u %u %u %zu %zu\n", i, (unsigned long)g.read_cnts_u, (unsigned long)g.read_cnts_d, g.glength, g.nus,
               g.nds, fq.ht_u->map_sp[(uint32_t)i].size(), fq.ht_d->map_sp[(uint32_t)i].size());
        for (int t = 0; t < 2; t++)
            for (const pleafNode *pn : ht[t]->map_sp[(uint32_t)i])
                printf("%c %zu %u %u %u %u %u %u\n", t ? 'd' : 'u', (size_t)(pn - fq.gpu_leaves[t].data()), pn->refID1, pn->refID2,
                       (unsigned)pn->depth, (unsigned)pn->ucount1, (unsigned)pn->ucount2, pn->rcount);
    }
    for (const auto &kv : fq.read_cnts_b) printf("P %u %u %lu\n", kv.first.first, kv.first.second, (unsigned long)kv.second);
    printf("leaf_cnt %lu %lu\n", (unsigned long)fq.ht_u->leaf_cnt, (unsigned long)fq.ht_d->leaf_cnt);
    cq_index_free(fq.gpu_idx);
    return 0;
}
