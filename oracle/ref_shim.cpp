// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" window onto the *real* reference (muellan/metacache), compiled together with
// the reference's own objects into oracle/_ref/libmcref_{u32,u16}.so by oracle/Makefile.
// It #includes the reference headers from where they lie (-I/root/reference/src); nothing of the
// reference is copied into this repository.  It exists so that tests, smoke() and bench.py's
// cpu_baseline leg can ask the reference itself for:
//   * window sketches of arbitrary strings      (hash_dna.hpp:208-255)
//   * database::query_host on a loaded database (database.hpp:399-407 -> host_hashmap.hpp:695-723)
//   * the target lineage table                  (taxonomy.hpp:919-1030)
// The product never links or loads this file.

#include "database.hpp"
#include "candidate_generation.hpp"
#include "options.hpp"
#include "query_handler.hpp"

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace mc;

namespace {

sequence make_seq(const char* s, uint64_t n) {
    sequence q;
    if (n > 0) {
        q.resize(n);
        std::memcpy(q.data(), s, n);
    }
    return q;
}

struct ref_db {
    database db;
    std::vector<const taxon*> taxa;   // index -> taxon (non-target taxa first, then targets)
    std::string error;
};

struct ref_handler {
    query_handler<location> qh;
};

// what make_candidate_generation_rules needs from a query (candidate_structs.hpp:134-151)
struct fake_query { sequence seq1; sequence seq2; };

}  // namespace

extern "C" {

// sizeof(target_id) this library was compiled for (2 or 4)
int ref_target_id_bytes() { return int(sizeof(target_id)); }
int ref_location_bytes() { return int(sizeof(location)); }

// Sketch all windows of one sequence.  feats: [maxWindows * s] (unused tail of a window = ~0),
// counts[w] = number of valid features of window w.  Returns the number of windows for which the
// reference called 'consume' (windows shorter than k do not count), or -1 if maxWindows is too small.
int64_t ref_sketch(const char* seq, uint64_t len, uint32_t k, uint32_t s, uint32_t w, uint32_t stride,
                   uint32_t* feats, uint32_t* counts, uint64_t maxWindows)
{
    sketching_opt opt;
    opt.kmerlen = numk_t(k); opt.sketchlen = s; opt.winlen = w; opt.winstride = stride;
    sketcher sk;
    sequence q = make_seq(seq, len);
    int64_t nwin = 0;
    bool overflow = false;
    sk.for_each_sketch(q.begin(), q.end(), opt, [&](const auto& sketch) {
        if (uint64_t(nwin) >= maxWindows) { overflow = true; return; }
        uint32_t c = 0;
        for (auto f : sketch) { feats[nwin * s + c] = f; ++c; }
        counts[nwin] = c;
        for (; c < s; ++c) feats[nwin * s + c] = ~uint32_t(0);
        ++nwin;
    });
    return overflow ? -1 : nwin;
}

void* ref_db_open(const char* name)
{
    auto h = std::make_unique<ref_db>();
    try {
        h->db.read(name, -1, 1, database::scope::everything, info_level::silent);
    } catch (std::exception& e) {
        fprintf(stderr, "ref_db_open: %s\n", e.what());
        return nullptr;
    }
    return h.release();
}

void ref_db_close(void* h) { delete static_cast<ref_db*>(h); }

// info[0..7] = k, s, w, stride, maxLocationsPerFeature, targetCount, partCount, locationCount
void ref_db_info(void* hv, uint64_t* info)
{
    auto& db = static_cast<ref_db*>(hv)->db;
    info[0] = db.target_sketching().kmerlen;
    info[1] = db.target_sketching().sketchlen;
    info[2] = db.target_sketching().winlen;
    info[3] = db.target_sketching().winstride;
    info[4] = db.max_locations_per_feature();
    info[5] = db.target_count();
    info[6] = db.part_count();
    info[7] = db.location_count();
}

void ref_db_max_locations_per_feature(void* hv, uint64_t n) {
    static_cast<ref_db*>(hv)->db.max_locations_per_feature(database::bucket_size_type(n));
}
uint64_t ref_db_remove_features_with_more_locations_than(void* hv, uint64_t n) {
    return static_cast<ref_db*>(hv)->db.remove_features_with_more_locations_than(database::bucket_size_type(n));
}

// lineage table as taxon ids: out[tgt*21 + rank] = taxon id (0 = null)  (taxonomy.hpp:368, 919-1030)
void ref_db_lineages(void* hv, int64_t* out)
{
    auto& db = static_cast<ref_db*>(hv)->db;
    const auto& lins = db.taxa().target_lineages();
    for (size_t t = 0; t < lins.size(); ++t)
        for (int r = 0; r < taxonomy::num_ranks; ++r)
            out[t * taxonomy::num_ranks + r] = lins[t][r] ? lins[t][r]->id() : 0;
}

// name of target t (what -tophits / -allhits print); returns length, copies up to cap bytes
int64_t ref_db_target_name(void* hv, uint64_t tgt, char* buf, uint64_t cap)
{
    auto& db = static_cast<ref_db*>(hv)->db;
    const taxon* t = db.taxa().cached_taxon_of_target(target_id(tgt));
    if (!t) return -1;
    const auto& n = t->name();
    std::memcpy(buf, n.data(), std::min<uint64_t>(cap, n.size()));
    return int64_t(n.size());
}

void* ref_handler_new() { return new ref_handler; }
void ref_handler_free(void* q) { delete static_cast<ref_handler*>(q); }

// One query through database::query_host with the caller logic of database_query.hpp:126-142.
//   sketch opts: k from the DB; s,w,stride as given (0 = take from DB) like querying.cpp:232-243
// Outputs (pointers valid until the next call on the same handler):
//   *allhits  -> n_all  x {u32 win, u32 tgt}   (converted from the reference's packed location)
//   *tophits  -> n_top  x {i64 taxid, u32 tgt, u32 hits, u32 beg, u32 end}
struct ref_hit { uint32_t win; uint32_t tgt; };
struct ref_cand { int64_t taxid; uint32_t tgt; uint32_t hits; uint32_t beg; uint32_t end; };

struct ref_result_buf { std::vector<ref_hit> hits; std::vector<ref_cand> cands; };
static thread_local ref_result_buf tl_buf;

int ref_query(void* hv, void* qv,
              const char* s1, uint64_t l1, const char* s2, uint64_t l2,
              uint32_t sketchlen, uint32_t winlen, uint32_t winstride,
              uint64_t maxCand, int lowestRank, uint64_t insertSizeMax,
              const ref_hit** allhits, uint64_t* nAll,
              const ref_cand** tophits, uint64_t* nTop)
{
    auto& db = static_cast<ref_db*>(hv)->db;
    auto& qh = static_cast<ref_handler*>(qv)->qh;

    fake_query q{make_seq(s1, l1), make_seq(s2, l2)};

    sketching_opt sk = db.target_sketching();
    if (sketchlen) sk.sketchlen = sketchlen;
    if (winlen)    sk.winlen = winlen;
    if (winstride) sk.winstride = winstride;

    classification_options copt;
    copt.lowestRank = taxon_rank(lowestRank);
    copt.insertSizeMax = insertSizeMax;
    copt.maxNumCandidatesPerQuery = maxCand ? maxCand : std::numeric_limits<std::size_t>::max();

    auto rules = make_candidate_generation_rules(q, copt, db.target_sketching().winstride);

    db.query_host(q.seq1, q.seq2, qh, sk, rules);

    auto& buf = tl_buf;
    buf.hits.clear(); buf.cands.clear();
    for (const auto& l : qh.allhits()) buf.hits.push_back({uint32_t(l.win), uint32_t(l.tgt)});
    for (const auto& c : qh.tophits())
        buf.cands.push_back({c.tax ? int64_t(c.tax->id()) : 0, uint32_t(c.tgt), c.hits, c.pos.beg, c.pos.end});
    *allhits = buf.hits.data();  *nAll = buf.hits.size();
    *tophits = buf.cands.data(); *nTop = buf.cands.size();
    return 0;
}

// Throughput probe for bench.py's cpu_baseline leg: runs 'n' single-end reads (concatenated in
// 'seqs', read i = [offs[i], offs[i+1])) through query_host on 'threads' host threads, each with
// its own query_handler (thread model of database_query.hpp:204-205), and returns elapsed seconds.
// If cands != nullptr: cands[i*maxCand + j] = top candidate j of read i (hits == 0 -> unused).
double ref_query_many(void* hv, const char* seqs, const uint64_t* offs, uint64_t n,
                      uint64_t maxCand, int lowestRank, uint64_t insertSizeMax, int threads,
                      ref_cand* cands);

}  // extern "C"

#include <chrono>
#include <thread>

extern "C" double ref_query_many(void* hv, const char* seqs, const uint64_t* offs, uint64_t n,
                                 uint64_t maxCand, int lowestRank, uint64_t insertSizeMax, int threads,
                                 ref_cand* cands)
{
    auto& db = static_cast<ref_db*>(hv)->db;
    if (threads < 1) threads = 1;
    classification_options copt;
    copt.lowestRank = taxon_rank(lowestRank);
    copt.insertSizeMax = insertSizeMax;
    copt.maxNumCandidatesPerQuery = maxCand ? maxCand : std::numeric_limits<std::size_t>::max();
    const sketching_opt sk = db.target_sketching();

    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        pool.emplace_back([&, t] {
            query_handler<location> qh;
            const uint64_t lo = n * uint64_t(t) / threads, hi = n * uint64_t(t + 1) / threads;
            fake_query q;
            for (uint64_t i = lo; i < hi; ++i) {
                const uint64_t len = offs[i + 1] - offs[i];
                q.seq1.resize(len);
                std::memcpy(q.seq1.data(), seqs + offs[i], len);
                auto rules = make_candidate_generation_rules(q, copt, sk.winstride);
                db.query_host(q.seq1, q.seq2, qh, sk, rules);
                if (cands) {
                    uint64_t j = 0;
                    for (const auto& c : qh.tophits()) {
                        if (j >= maxCand) break;
                        cands[i * maxCand + j] = {c.tax ? int64_t(c.tax->id()) : 0, uint32_t(c.tgt), c.hits, c.pos.beg, c.pos.end};
                        ++j;
                    }
                    for (; j < maxCand; ++j) cands[i * maxCand + j] = {0, 0, 0, 0, 0};
                }
            }
        });
    }
    for (auto& th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}
