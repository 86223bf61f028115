// cammiq_glue.hpp -- C++ glue between the C ABI (cammiq_hip.h) and the host-side state the
// reference's unmodified ILP reads.  Header-only templates, so that ONE piece of code compiles
//   * inside the reference (FqReader::loadIdx_gpu / query64_gpu, against the
//     reference's own Genome / pleafNode / Hash, /root/reference/src/query.hpp:13-25,
//     hashtrie.hpp:37-60),
//   * inside this repo's `cammiq` shell (cammiq_amd/csrc/cammiq_main.cpp), and
//   * against the small mock of those three types in tests/cpp/ilp_handoff.cpp, where it is
//     type-checked and unit-tested.
// What runILP_* consumes (query.cpp:1083-1783): per genome read_cnts_u, read_cnts_d, glength,
// nus, nds (query.hpp:15-19); per genome g the leaves of ht_u->map_sp[g] / ht_d->map_sp[g] in
// DECODE ORDER with refID1, refID2, depth, ucount1, ucount2, rcount (hashtrie.hpp:37-47;
// filled at hashtrie.cpp:452-453,476); reads[file].size() and tlengths[file].
//
// Type requirements (duck-typed, the reference's names):
//   Genome : read_cnts_u, read_cnts_d (uint64_t); glength, nus, nds (uint32_t)
//   PLeaf  : refID1, refID2, rcount (uint32_t); depth (uint8_t); ucount1, ucount2 (uint16_t)
//   Hash   : map_sp  -- operator[](uint32_t) -> std::vector<PLeaf*>& ;  leaf_cnt
#ifndef CAMMIQ_GLUE_HPP_
#define CAMMIQ_GLUE_HPP_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "cammiq_hip.h"

namespace cq_glue {

template <class T> inline T &deref(T *p) { return *p; }
template <class T> inline T &deref(T &r) { return r; }

// FqReader::IDXDIR (query.cpp:42-44): directory of index_u including the trailing '/', "./" if none.
inline std::string index_dir(const std::string &index_u)
{
    const size_t k = index_u.find_last_of('/');
    return k == std::string::npos ? std::string("./") : index_u.substr(0, k + 1);
}

// One "id <ws> value" file of FqReader::loadGenomeLength (query.cpp:158-205).  Returns false when
// the file cannot be opened.  Lines whose id is not in [1, genomes.size()) are ignored (the
// reference indexes genomes[] out of bounds there); line order is irrelevant (it is the build's
// unordered_map order).
template <class GenomeVec, class Setter>
bool load_id_value_file(const std::string &fn, GenomeVec &genomes, Setter set)
{
    std::ifstream in(fn.c_str());
    if (!in.is_open()) return false;
    std::string line, id, val;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        id.clear(); val.clear();
        ls >> id >> val;
        if (id.empty() || val.empty()) continue;
        char *e1 = nullptr, *e2 = nullptr;
        const unsigned long i = strtoul(id.c_str(), &e1, 10), v = strtoul(val.c_str(), &e2, 10);
        if (*e1 || *e2 || i == 0 || i >= genomes.size()) continue;
        set(deref(genomes[i]), (uint32_t)v);
    }
    return true;
}

// FqReader::loadGenomeLength (query.cpp:158-205): genome_lengths.out -> glength,
// unique_lmer_count_u.out -> nus, unique_lmer_count_d.out -> nds, all next to index_u.
// Returns NULL, or the reference's own message for the first file that cannot be opened (the
// reference then abort()s; the caller decides).  d_optional: a --unique build writes no
// unique_lmer_count_d.out (build.cpp:671-698) -- with it set, that one file may be absent (nds stay 0).
template <class GenomeVec>
const char *load_genome_meta(const std::string &idxdir, GenomeVec &genomes, bool d_optional = false)
{
    typedef decltype(deref(genomes[0])) G;
    if (!load_id_value_file(idxdir + "genome_lengths.out", genomes, [](G g, uint32_t v) { g.glength = v; }))
        return "Can not open genome length file.\n";
    if (!load_id_value_file(idxdir + "unique_lmer_count_u.out", genomes, [](G g, uint32_t v) { g.nus = v; }))
        return "Can not open unique count file.\n";
    if (!load_id_value_file(idxdir + "unique_lmer_count_d.out", genomes, [](G g, uint32_t v) { g.nds = v; }) && !d_optional)
        return "Can not open doubly-unique count file.\n";
    return nullptr;
}

// Rebuild one table's pleafNode records and Hash::map_sp exactly as decodeTrie_p leaves them
// (hashtrie.cpp:441-453, 469-476): leaves in file decode order; every leaf is pushed to
// map_sp[refID1] and, when it is doubly-unique, to map_sp[refID2] as well.  `store` owns the
// records (map_sp holds pointers into it, so it must not be resized afterwards).
template <class Hash, class PLeaf>
int rebuild_map_sp(const cq_index *idx, int table, Hash &ht, std::vector<PLeaf> &store)
{
    cq_index_info info;
    int rc = cq_index_get_info(idx, &info);
    if (rc != CQ_OK) return rc;
    std::vector<cq_leaf> lv(info.n_leaves[table]);
    if (!lv.empty()) {
        rc = cq_index_leaves(idx, table, lv.data());
        if (rc != CQ_OK) return rc;
    }
    store.resize(lv.size());
    for (size_t i = 0; i < lv.size(); i++) {
        PLeaf &p = store[i];
        p.refID1 = lv[i].refID1;
        p.refID2 = lv[i].refID2;
        p.depth = lv[i].depth;
        p.ucount1 = lv[i].ucount1;
        p.ucount2 = lv[i].ucount2;
        p.rcount = 0;
    }
    for (size_t i = 0; i < store.size(); i++) {
        ht.map_sp[store[i].refID1].push_back(&store[i]);
        if (store[i].refID2) ht.map_sp[store[i].refID2].push_back(&store[i]);
    }
    ht.leaf_cnt = store.size();
    return CQ_OK;
}

// Add the outputs of one cq_query* call into the state query64_p / query64_sc mutate
// (query.cpp:458-648, 891-1080): counters ACCUMULATE until resetCounters (query.cpp:1820-1840),
// so this adds.  PairMap = std::map<std::pair<uint32_t,uint32_t>, uint64_t> (FqReader::read_cnts_b).
template <class GenomeVec, class PLeaf, class PairMap>
void add_counts(const cq_counts &c, GenomeVec &genomes, std::vector<PLeaf> &leaves_u, std::vector<PLeaf> &leaves_d,
                size_t &nundet, size_t &nconf, PairMap &read_cnts_b)
{
    for (size_t g = 1; g < genomes.size(); g++) {
        deref(genomes[g]).read_cnts_u += c.cnt_u[g];
        deref(genomes[g]).read_cnts_d += c.cnt_d[g];
    }
    if (c.rcount_u) for (size_t i = 0; i < leaves_u.size(); i++) leaves_u[i].rcount += c.rcount_u[i];
    if (c.rcount_d) for (size_t i = 0; i < leaves_d.size(); i++) leaves_d[i].rcount += c.rcount_d[i];
    nundet += c.nundet;
    nconf += c.nconf;
    for (uint64_t i = 0; i < c.n_pairs; i++) read_cnts_b[std::make_pair(c.pair_a[i], c.pair_b[i])] += c.pair_cnt[i];
}

}  // namespace cq_glue
#endif
