// cammiq_main.cpp -- `cammiq --query ...` command-line shell on top of libcammiq_hip.so.
//
// Keeps the query-side contract of the reference CLI (/root/reference/src/main.cpp:74-446,
// README "Query" section): same flags, same stderr lines, same --read_cnts TSV
// (FqReader::outputUniqueCnts, /root/reference/src/query.cpp:1786-1818), while the classify
// step runs on the GPU through the C ABI.  Own code: a thin restatement of the driver
// (FqReader::loadSmap query.cpp:125-156, readFastq :371-425, queryFastq_p/_sc :231-369),
// not a port of it.  Out of scope here, exactly as in BASELINE.json's north star:
//   --build        index construction stays with the reference's own binary;
//   the ILP        runILP_* needs CPLEX/Gurobi; with --dump_counts FILE this shell writes
//                  what the unmodified ILP consumes besides the index itself (per genome: counts,
//                  glength, nus, nds; per leaf with rcount > 0: its record and rcount; reads and
//                  bases of the query) so a host with a solver can pick it up
//                  (include/cammiq_glue.hpp is the in-process version of the same hand-off).
// Extensions (not in the reference): --device N, --gpus N / --devices a,b,.. (shards the reads over
// several GPUs, RCCL all-reduce of the counts), --dump_counts FILE, --image_cache, missing .bin2 allowed.
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <future>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cammiq_hip.h"
#include "../../include/cammiq_glue.hpp"

namespace {

// Genome (query.hpp:13-25): the per-genome state the ILP reads
struct Genome {
    uint32_t taxID = 0;
    std::string name;
    uint64_t read_cnts_u = 0, read_cnts_d = 0;
    uint32_t glength = 0, nus = 0, nds = 0;
    Genome() {}
    Genome(uint32_t t, const std::string &n) : taxID(t), name(n) {}
};

bool valid_file(const char *p) { struct stat st; return stat(p, &st) == 0; }

std::string ext_of(const std::string &f) { return f.substr(f.find_last_of('.') + 1); }

std::string base_name(const std::string &p)
{
    size_t k = p.find_last_of('/');
    return k == std::string::npos ? p : p.substr(k + 1);
}

[[noreturn]] void die(const char *fmt, const char *arg = nullptr)
{
    if (arg) fprintf(stderr, fmt, arg); else fputs(fmt, stderr);
    exit(EXIT_FAILURE);
}

// FqReader::loadSmap (query.cpp:125-156): genomes[] is filled in map-file LINE ORDER and
// indexed by refID; a repeated taxID appends the name to an existing entry.
std::vector<Genome> load_map(const std::string &fn)
{
    std::vector<Genome> g(1);
    std::ifstream in(fn);
    if (!in.is_open()) die("Can not open map file %s.\n", fn.c_str());
    std::string line;
    std::set<uint32_t> taxids;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string file, id, taxid, name;
        std::getline(ls, file, '\t'); std::getline(ls, id, '\t'); std::getline(ls, taxid, '\t'); std::getline(ls, name, '\t');
        if (id.empty() || taxid.empty()) continue;
        uint32_t t = (uint32_t)atoi(taxid.c_str());
        if (taxids.count(t)) {
            size_t i = (size_t)atoi(id.c_str());
            if (i < g.size()) g[i].name += "/" + name;
        } else {
            g.push_back(Genome(t, name));
            taxids.insert(t);
        }
    }
    fprintf(stderr, "Loaded genome map file.\n");
    return g;
}

// FqReader::readFastq (query.cpp:371-425): second line of every four; reads shorter than
// min_l are dropped; every 'N' of a read is replaced by ONE random base per read.  The
// reference seeds rand() from the clock, so its output on reads with N is not reproducible;
// this shell derives the base from the read's index instead (documented deviation).
// Unlike the reference (one getline + one new[] per read, single thread) the file is
// memory-mapped and split at line boundaries across threads.  Two passes: (1) every chunk counts
// its lines and, for each of the four possible positions of its first line in a FASTQ record, the
// sequence lines, their bytes and their longest length; a prefix sum over the chunks then fixes
// everything; (2) every chunk packs its sequence lines straight into tight 2-bit rows (cq_pack_read_tight:
// 25 bytes per 100-bp read, what crosses the host link), which is what cq_query_packed_tight takes -- the
// ASCII reads are never copied (--fastq_stats keeps an
// ASCII pass 2 for its digest).  The output buffers are not initialised first (their pages are
// first touched by the copying threads).
struct Reads {
    std::unique_ptr<uint8_t[]> bases;   // ASCII (--fastq_stats only)
    std::unique_ptr<uint64_t[]> offs;   // n_reads + 1
    std::unique_ptr<uint8_t[]> packed;  // tight 2-bit rows, stride sb bytes (queries: the ASCII reads are never materialised)
    std::unique_ptr<uint8_t[]> lens;
    uint32_t sb = 1, max_len = 0;
    size_t n_reads = 0, n_bases = 0;
};

// pack = true: pass 2 writes 2-bit rows (cq_pack_read_tight) straight from the mapped file instead of an ASCII copy.
void read_fastq(const std::string &fn, size_t min_l, Reads &out, bool pack = false)
{
    const bool timing = getenv("CAMMIQ_LOAD_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    out = Reads();
    out.offs.reset(new uint64_t[1]);
    out.offs[0] = 0;
    int fd = open(fn.c_str(), O_RDONLY);
    if (fd < 0) die("Failed to find input file %s.\n", fn.c_str());
    struct stat st;
    fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    if (n == 0) { close(fd); fprintf(stderr, "Loaded query file %s.\n", fn.c_str()); return; }
    const char *p = (const char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) die("Failed to map input file %s.\n", fn.c_str());
    madvise((void *)p, n, MADV_SEQUENTIAL);
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned T = (n < (1u << 22)) ? 1u : std::max(1u, std::min(hw ? hw : 1u, 32u));
    // chunk c covers [cut[c], cut[c+1]); every cut sits right after a newline
    std::vector<size_t> cut(T + 1, n);
    cut[0] = 0;
    for (unsigned c = 1; c < T; c++) {
        size_t q = n / T * c;
        const char *nl = (const char *)memchr(p + q, '\n', n - q);
        cut[c] = nl ? (size_t)(nl - p) + 1 : n;
    }
    auto for_chunks = [&](auto &&fn_) {
        std::vector<std::thread> th;
        for (unsigned c = 0; c < T; c++) th.emplace_back(fn_, c);
        for (auto &x : th) x.join();
    };
    // pass 1: lines per chunk; kept reads / bytes for each residue (line number within the chunk) mod 4
    struct Count { uint64_t lines = 0, reads[4] = {0, 0, 0, 0}, bytes[4] = {0, 0, 0, 0}; uint32_t maxlen[4] = {0, 0, 0, 0}; };
    std::vector<Count> cnt(T);
    for_chunks([&](unsigned c) {
        Count k;
        for (const char *s = p + cut[c], *e = p + cut[c + 1]; s < e;) {
            const char *nl = (const char *)memchr(s, '\n', (size_t)(e - s));
            size_t len = nl ? (size_t)(nl - s) : (size_t)(e - s);
            if (len && s[len - 1] == '\r') len--;
            if (len >= min_l) {
                k.reads[k.lines & 3u]++; k.bytes[k.lines & 3u] += len;
                if (len <= 255 && len > k.maxlen[k.lines & 3u]) k.maxlen[k.lines & 3u] = (uint32_t)len;
            }
            k.lines++;
            if (!nl) break;
            s = nl + 1;
        }
        cnt[c] = k;
    });
    // a chunk whose first line is line L of the file holds its sequence lines at residue (1 - L) mod 4
    std::vector<uint64_t> line0(T + 1, 0), r0(T + 1, 0), b0(T + 1, 0);
    for (unsigned c = 0; c < T; c++) {
        const unsigned res = (unsigned)((1u - (unsigned)(line0[c] & 3u)) & 3u);
        line0[c + 1] = line0[c] + cnt[c].lines;
        r0[c + 1] = r0[c] + cnt[c].reads[res];
        b0[c + 1] = b0[c] + cnt[c].bytes[res];
        out.max_len = std::max(out.max_len, cnt[c].maxlen[res]);
    }
    out.n_reads = r0[T];
    out.n_bases = b0[T];
    out.sb = cq_pack_stride_bytes(out.max_len);
    const uint32_t sb = out.sb;
    if (pack) {
        out.packed.reset(new uint8_t[out.n_reads * sb + 1]);
        out.lens.reset(new uint8_t[out.n_reads + 1]);
    } else {
        out.bases.reset(new uint8_t[out.n_bases ? out.n_bases : 1]);
        out.offs.reset(new uint64_t[out.n_reads + 1]);
        out.offs[out.n_reads] = out.n_bases;
    }
    // pass 2: copy
    const char alphabet[4] = {'A', 'C', 'G', 'T'};
    uint8_t *const bases = out.bases.get();
    uint64_t *const offs = out.offs.get();
    uint8_t *const packed = out.packed.get();
    uint8_t *const lens = out.lens.get();
    for_chunks([&](unsigned c) {
        uint64_t r = r0[c], b = b0[c], li = line0[c];
        uint8_t tmp[256];
        for (const char *s = p + cut[c], *e = p + cut[c + 1]; s < e; li++) {
            const char *nl = (const char *)memchr(s, '\n', (size_t)(e - s));
            size_t len = nl ? (size_t)(nl - s) : (size_t)(e - s);
            if ((li & 3u) == 1u) {
                if (len && s[len - 1] == '\r') len--;
                if (len >= min_l) {
                    const bool has_n = memchr(s, 'N', len) != nullptr;
                    uint64_t z = ((li >> 2) + 1) * 0x9E3779B97F4A7C15ull;
                    z ^= z >> 29;
                    const uint8_t sub = (uint8_t)alphabet[(z >> 7) & 3];
                    if (pack) {
                        const uint8_t *src = (const uint8_t *)s;
                        if (has_n && len <= 255) {
                            for (size_t i = 0; i < len; i++) tmp[i] = s[i] == 'N' ? sub : (uint8_t)s[i];
                            src = tmp;
                        }
                        // reads longer than 255 bases are outside the parity domain (the reference keeps lengths in a uint8_t)
                        if (len > 255) { memset(packed + r * sb, 0, sb); lens[r] = 0; }
                        else cq_pack_read_tight(src, (uint32_t)len, 1, sb, packed + r * sb, lens + r);
                        r++;
                    } else {
                        offs[r++] = b;
                        uint8_t *dst = bases + b;
                        memcpy(dst, s, len);
                        if (has_n) for (size_t i = 0; i < len; i++) if (dst[i] == 'N') dst[i] = sub;
                    }
                    b += len;
                }
            }
            if (!nl) break;
            s = nl + 1;
        }
    });
    munmap((void *)p, n);
    close(fd);
    fprintf(stderr, "Loaded query file %s.\n", fn.c_str());
    if (timing)
        fprintf(stderr, "[read_fastq] %.1f ms for %.2f GB, %zu reads, %u threads\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), n / 1e9,
                out.n_reads, T);
}

void list_fastq(const std::string &dir, std::vector<std::string> &out)
{
    DIR *d = opendir(dir.c_str());
    if (!d) die("Input directory not exists.\n");
    while (dirent *e = readdir(d)) {
        std::string f = e->d_name;
        if (f.size() >= 7) {
            std::string x = ext_of(f);
            if (x == "fq" || x == "fastq") out.push_back(dir + f);
        }
    }
    closedir(d);
}

}  // namespace

int main(int argc, char **argv)
{
    int mode = -1, id_mode = 0, t = 1, h = -1, h1 = -1, h2 = -1, device = 0, gpus = 1;
    if (const char *e = getenv("CAMMIQ_GPUS")) gpus = atoi(e);
    size_t min_rl = 0;
    std::vector<int> dev_list;
    std::string fi1, fi2, fm, output = "CAMMiQ_output.txt", fq_dir, dump;
    std::vector<std::string> fq_names;
    float erate = 0.01f;
    auto need = [&](int &i, const char *msg) { if (++i >= argc) die(msg); return argv[i]; };

    for (int i = 1; i < argc; i++) {
        std::string v(argv[i]);
        if (v == "--build") { mode = 0; continue; }
        if (v == "--query") { mode = 1; continue; }
        if (v == "--read_cnts") {
            if (mode <= 0) die("Option --read_cnts is only valid in mode QUERY.\n");
            id_mode = 1; continue;
        }
        if (v == "--doubly_unique") {
            if (mode == 0) continue;
            if (id_mode == 0) die("Option --doubly_unique is only valid in --read_cnts queries.\n");
            id_mode = 2; continue;
        }
        if (v == "--unique" || v == "--both") {
            if (mode > 0) die("Option --unique is only valid in mode BUILD.\n");
            continue;
        }
        if (v == "--enable_ilp_display") { if (mode <= 0) die("Option --enable_ilp_display is only valid in mode QUERY.\n"); continue; }
        if (v == "--read_length_filter") {
            if (mode <= 0) die("Option --read_length_filter is only valid in mode QUERY.\n");
            min_rl = (size_t)atoi(need(i, "Please specify a parameter value for --read_length_filter.\n")); continue;
        }
        if (v == "--read_cnt_thres" || v == "--easy_to_identify_thres" || v == "--unique_read_cnt_thres" ||
            v == "--doubly_unique_read_cnt_thres" || v == "--ilp_alpha" || v == "--ilp_epsilon" || v == "--ilp_max_cov" ||
            v == "--ilp_resolution") {   // fine parameters feed the ILP only (main.cpp:141-222)
            need(i, "Please specify a parameter value.\n"); continue;
        }
        if (v == "--fastq_stats") {   // diagnostic: parse one FASTQ like a query would, print a digest, exit
            const char *f = need(i, "Please specify a fastq file.\n");
            Reads rd;
            read_fastq(f, min_rl, rd);
            uint64_t hsh = 1469598103934665603ull;
            for (size_t k = 0; k < rd.n_bases; k++) hsh = (hsh ^ rd.bases[k]) * 1099511628211ull;
            for (size_t k = 0; k <= rd.n_reads; k++) hsh = (hsh ^ rd.offs[k]) * 1099511628211ull;
            printf("reads %zu bases %zu fnv %016llx\n", rd.n_reads, rd.n_bases, (unsigned long long)hsh);
            return 0;
        }
        if (v == "--device") { device = atoi(need(i, "Please specify the GPU ordinal.\n")); continue; }
        if (v == "--gpus") { gpus = atoi(need(i, "Please specify the number of GPUs.\n")); continue; }
        if (v == "--devices") {   // explicit ordinals, comma separated; always takes the multi-GPU path
            std::istringstream ls(need(i, "Please specify the GPU ordinals.\n"));
            std::string tok;
            while (std::getline(ls, tok, ',')) if (!tok.empty()) dev_list.push_back(atoi(tok.c_str()));
            continue;
        }
        if (v == "--image_cache") { setenv("CAMMIQ_IMAGE_CACHE", "1", 0); continue; }   // see cq_cache.cpp (an existing value, e.g. "force", stays)
        if (v == "--dump_counts") { dump = need(i, "Please specify the counts file name.\n"); continue; }
        if (v == "-h") {
            need(i, "Please specify the hash length.\n");
            if (i + 1 < argc && argv[i + 1][0] != '-') { h1 = atoi(argv[i++]); h2 = atoi(argv[i]); }
            else h = atoi(argv[i]);
            for (int x : {h, h1, h2})
                if (x != -1 && (x <= 4 || x >= 32)) die("The hash length should be in range [5, 31].\n");
            continue;
        }
        if (v == "-i") {
            if (++i >= argc) die("Please specify index file names.\n");
            while (i < argc && argv[i][0] != '-') {
                std::string f = argv[i++], x = ext_of(f);
                if (x == "idx1" || x == "bin1") fi1 = f;
                if (x == "idx2" || x == "bin2") fi2 = f;
            }
            i--; continue;
        }
        if (v == "-o") { if (mode <= 0) die("Parameter o is only valid in mode QUERY.\n"); output = need(i, "Please specify the output file name.\n"); continue; }
        if (v == "-e") { erate = (float)atof(need(i, "Please specify the error rate.\n")); continue; }
        if (v == "-t") { t = atoi(need(i, "Please specify the worker threads number.\n")); continue; }
        if (v == "-f") {
            if (++i >= argc) die("Please specify file names.\n");
            while (i < argc && argv[i][0] != '-') {
                std::string f = argv[i++], x = ext_of(f);
                if (x == "out" || x == "map") fm = f;
            }
            i--; continue;
        }
        if (v == "-q") {
            if (++i >= argc) die("Please specify query file names.\n");
            while (i < argc && argv[i][0] != '-') {
                std::string f = argv[i++], x = ext_of(f);
                if (x == "fq" || x == "fastq") {
                    if (!valid_file(f.c_str())) die("Failed to find input file %s.\n", f.c_str());
                    fq_names.push_back(f);
                }
            }
            i--; continue;
        }
        if (v == "-Q") {
            fq_dir = need(i, "Please specify the directory containing fastq files.\n");
            if (!valid_file(fq_dir.c_str())) die("Failed to find input directory %s.\n", fq_dir.c_str());
            continue;
        }
        if (v == "-k" || v == "-L" || v == "-Lmax" || v == "-D") { need(i, "Please specify a value.\n"); continue; }   // build-side
        fprintf(stderr, "Failed to recognize option: %s. \n", v.c_str());
        return EXIT_FAILURE;
    }
    (void)erate; (void)t;
    if (mode == 0) die("cammiq (MI355X query engine): index construction is out of scope; build the index with the "
                       "reference CAMMiQ and query it here.\n");
    if (mode != 1) die("Please specify --query.\n");
    if (fi1.empty()) die("Please specify index file names.\n");
    if (fm.empty()) die("Please specify file names.\n");
    if (fq_names.empty()) {
        if (fq_dir.empty()) die("Please specify at least one query file or directory.\n");
        list_fastq(fq_dir, fq_names);
    }

    // FASTQ parsing does not need the index: file f+1 is parsed on a second thread while the index loads
    // (f = 0) or while file f is being classified.  (The reference reads and queries strictly in turn,
    // query.cpp:371-425; the stderr lines keep its order.)
    auto parse_async = [&](size_t f) {
        return std::async(std::launch::async, [&, f] {
            Reads p;
            read_fastq(fq_names[f], min_rl, p, true);   // straight to 2-bit rows
            return p;
        });
    };
    std::future<Reads> next_fq = parse_async(0);

    auto t0 = std::chrono::high_resolution_clock::now();
    // one GPU: a plain handle; several: the library shards the reads over devices device .. device+gpus-1
    // and all-reduces the counts over RCCL (cq_multi_*); the results are the same, bit for bit
    cq_index *ix = nullptr;
    cq_multi *mx = nullptr;
    if (!dev_list.empty()) gpus = (int)dev_list.size();
    if (gpus < 1 || gpus > 64) die("The number of GPUs should be in range [1, 64].\n");
    if (gpus == 1 && dev_list.empty()) {
        if (cq_index_load(fi1.c_str(), fi2.empty() ? nullptr : fi2.c_str(), device, &ix) != CQ_OK) {
            fprintf(stderr, "%s\n", cq_last_error());
            return EXIT_FAILURE;
        }
    } else {
        std::vector<int> devs = dev_list;
        if (devs.empty()) for (int d = 0; d < gpus; d++) devs.push_back(device + d);
        if (cq_multi_load(fi1.c_str(), fi2.empty() ? nullptr : fi2.c_str(), devs.data(), gpus, &mx) != CQ_OK) {
            fprintf(stderr, "%s\n", cq_last_error());
            return EXIT_FAILURE;
        }
        ix = cq_multi_index(mx, 0);
    }
    cq_index_info info;
    cq_index_get_info(ix, &info);
    fprintf(stderr, "Index: %s\nHash Length: %u\n", fi1.c_str(), info.hash_len);
    if (!fi2.empty()) fprintf(stderr, "Index: %s\nHash Length: %u\n", fi2.c_str(), info.hash_len);
    for (int x : {h, h1, h2})
        if (x != -1 && (uint32_t)x != info.hash_len) die("Hash length given with -h differs from the one encoded in the index.\n");
    fprintf(stderr, "Loaded index files into memory.\n");
    fprintf(stderr, "Time for loading index: %lu ms.\n",
            (unsigned long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count());

    std::vector<Genome> genomes = load_map(fm);
    const uint32_t G = (uint32_t)genomes.size() - 1;
    std::vector<cq_leaf> leaves[2];
    for (int tb = 0; tb < 2; tb++) {
        leaves[tb].resize(info.n_leaves[tb]);
        cq_index_leaves(ix, tb, leaves[tb].data());
    }
    // FqReader::loadGenomeLength (query.cpp:158-205): glength / nus / nds from the three text files next
    // to index_u.  The reference calls it in every quantification query and in --read_cnts queries given
    // with -q (query.cpp:245,283,347) but not in --read_cnts queries given with -Q (:303-331), and aborts
    // when a file is missing.  Same here (exit instead of abort); only a --unique index (no .bin2) may
    // lack unique_lmer_count_d.out, which such a build never writes (build.cpp:671-698).
    if (id_mode == 0 || fq_dir.empty()) {
        if (const char *msg = cq_glue::load_genome_meta(cq_glue::index_dir(fi1), genomes, fi2.empty())) die(msg);
        fprintf(stderr, "Loaded genome length file.\n");
    }

    std::vector<uint64_t> cu(G + 1), cd(G + 1), pc(1 << 16);
    std::vector<uint32_t> ru(info.n_leaves[0]), rd(info.n_leaves[1]), pa(1 << 16), pb(1 << 16);
    for (size_t f = 0; f < fq_names.size(); f++) {
        Reads fq = next_fq.get();
        if (f + 1 < fq_names.size()) next_fq = parse_async(f + 1);
        const std::string cur = base_name(fq_names[f]);
        if (id_mode && t > 1) fprintf(stderr, "Single cell queries only support one thread.\n");
        fprintf(stderr, "Querying %s.\n", cur.c_str());
        auto q0 = std::chrono::high_resolution_clock::now();
        cq_counts c;
        const int qmode = id_mode ? CQ_MODE_SC : CQ_MODE_P;
        int rc;
        for (;;) {
            memset(&c, 0, sizeof c);
            c.cnt_u = cu.data(); c.cnt_d = cd.data();
            c.rcount_u = ru.empty() ? nullptr : ru.data(); c.rcount_d = rd.empty() ? nullptr : rd.data();
            c.pair_a = pa.data(); c.pair_b = pb.data(); c.pair_cnt = pc.data(); c.pair_cap = pc.size();
            rc = mx ? cq_multi_query_packed_tight(mx, qmode, fq.packed.get(), fq.lens.get(), fq.n_reads, fq.sb, fq.max_len, G, &c)
                    : cq_query_packed_tight(ix, qmode, fq.packed.get(), fq.lens.get(), fq.n_reads, fq.sb, fq.max_len, G, &c);
            if (rc == CQ_ERR_LIMIT && c.n_pairs > pc.size()) {   // read_cnts_b has more entries than the arrays: grow, ask again
                pa.resize(c.n_pairs); pb.resize(c.n_pairs); pc.resize(c.n_pairs);
                if (!mx) {   // the single-GPU library kept the pairs: fetch them, no second classify
                    uint64_t np = 0;
                    rc = cq_pairs_fetch(ix, pa.data(), pb.data(), pc.data(), pc.size(), &np);
                    c.pair_a = pa.data(); c.pair_b = pb.data(); c.pair_cnt = pc.data(); c.n_pairs = np;
                    break;
                }
                continue;
            }
            break;
        }
        if (rc != CQ_OK) { fprintf(stderr, "%s\n", cq_last_error()); return EXIT_FAILURE; }
        fprintf(stderr, "Processed %lu reads.\r", (unsigned long)fq.n_reads);
        fprintf(stderr, "\nNumber of unlabeled reads: %lu.\n", (unsigned long)c.nundet);
        fprintf(stderr, "Number of reads with conflict labels: %lu.\n", (unsigned long)c.nconf);
        if (c.nskipped) fprintf(stderr, "Number of reads outside the supported domain (skipped): %lu.\n", (unsigned long)c.nskipped);
        fprintf(stderr, "Completed query %s.\n", cur.c_str());
        fprintf(stderr, "Time for query: %lu ms.\n",
                (unsigned long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - q0).count());

        if (id_mode == 1) {   // outputUniqueCnts, query.cpp:1786-1818
            FILE *fo = fopen(output.c_str(), f == 0 ? "w" : "a");
            if (!fo) die("Can not open output file %s.\n", output.c_str());
            if (f == 0) {
                fprintf(fo, "QUERY/TAXID\t");
                for (uint32_t i = 1; i <= G; i++) fprintf(fo, i < G ? "%u\t" : "%u\n", genomes[i].taxID);
            }
            fprintf(fo, "%s\t", cur.c_str());
            for (uint32_t i = 1; i <= G; i++) fprintf(fo, i < G ? "%lu\t" : "%lu\n", (unsigned long)cu[i]);
            fclose(fo);
        }
        if (!dump.empty()) {
            // What runILP_* reads from FqReader state besides the index (query.cpp:1083-1226): reads[f].size()
            // and tlengths[f]; per genome read_cnts_u/_d, glength, nus, nds; per leaf its record and rcount --
            // leaves with rcount 0 are not listed, the consumer has them from the index (cq_index_leaves,
            // decode order = leaf index; map_sp[g] = the leaves naming g, in that order).
            FILE *fo = fopen(dump.c_str(), f == 0 ? "w" : "a");
            if (!fo) die("Can not open output file %s.\n", dump.c_str());
            if (f == 0) fprintf(fo, "#cammiq_counts\t2\n");
            fprintf(fo, "#query\t%s\tn_reads\t%lu\ttotal_bases\t%lu\tnundet\t%lu\tnconf\t%lu\n", cur.c_str(),
                    (unsigned long)fq.n_reads, (unsigned long)fq.n_bases, (unsigned long)c.nundet, (unsigned long)c.nconf);
            for (uint32_t g = 1; g <= G; g++)
                fprintf(fo, "G\t%u\t%u\t%lu\t%lu\t%u\t%u\t%u\n", g, genomes[g].taxID, (unsigned long)cu[g], (unsigned long)cd[g],
                        genomes[g].glength, genomes[g].nus, genomes[g].nds);
            if (!id_mode)
                for (int tb = 0; tb < 2; tb++)
                    for (size_t i = 0; i < leaves[tb].size(); i++) {
                        const uint32_t r = tb ? rd[i] : ru[i];
                        const cq_leaf &lf = leaves[tb][i];
                        if (r) fprintf(fo, "L\t%c\t%zu\t%u\t%u\t%u\t%u\t%u\t%u\n", tb ? 'd' : 'u', i, lf.refID1, lf.refID2,
                                       (unsigned)lf.depth, (unsigned)lf.ucount1, (unsigned)lf.ucount2, r);
                    }
            for (uint64_t i = 0; i < c.n_pairs; i++)
                fprintf(fo, "P\t%u\t%u\t%lu\n", pa[i], pb[i], (unsigned long)pc[i]);
            fclose(fo);
        }
    }
    if (mx) cq_multi_free(mx);
    else cq_index_free(ix);
    return 0;
}
