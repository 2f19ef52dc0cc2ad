// mcq -- the reference's command lines on MI355X, host C++14 above the C ABI (include/metacache_amd.h):
//   mcq query <database> [<reads.fa|fq[.gz]|directory>...] [options]     mode_query.cpp, querying.cpp (interactive without files)
//   mcq build <database> <sequence files|directories>... [options]        mode_build.cpp, building.cpp        (mcq_build.h)
//   mcq modify <database> <sequence files|directories>... [options]       mode_build.cpp:74-88: adds to an existing database (mcq_build.h)
//   mcq build+query -targets <files>... [-query <files>...] [options]     mode_build_query.cpp                (mcq_build.h)
//   mcq merge <result files>... -taxonomy <dir> [options]                 mode_merge.cpp
//   mcq info [<database> [targets [name...] | lineages | rank <r>]]       mode_info.cpp (metadata topics)
// Query mode mirrors
//   option handling      options.cpp:860-1430 (query subset), querying.cpp:225-269 (adapt_options_to_database)
//   read ingest          sequence_io.cpp:160-228, :293-322 (FASTA / FASTQ, -pairfiles / -pairseq), database_query.hpp:258-284;
//                        files are memory-mapped (gzip: inflated into memory), record starts are indexed by all
//                        threads in parallel (query ids = record order, as reader.index()), and every worker thread drives one
//                        batch slot: parse -> mc_batch_add -> submit -> wait -> classify + format; output is written in file order
//   classification       classification.cpp:146-189 (classify: ranked-LCA vote over the top candidates), :104-137 (ground truth),
//                        :272-295 (evaluation), :304-374 (abundance estimation)
//   output lines         classification.cpp:470-526 (show_query_mapping), printing.cpp:47-620 (parameters, candidates, matches,
//                        hits per reference sequence, abundance tables, summary)
// Options: -out -split-out -lowest -highest -hitmin -hitdiff -maxcand -tophits -allhits -locations -queryids -mapped-only -no-map
//   -taxids -taxids-only -omit-ranks -separate-cols -separator -comment -lineage -pairfiles -pairseq -insertsize -min-readlen
//   -max-readlen -query-limit -sketchlen -winlen -winstride -max-locations-per-feature -remove-overpopulated-features -max-load-fac
//   -no-query-params -no-summary -no-err -threads -batch-size -abundances [file] -abundance-per <rank> -hits-per-ref [file]
//   -ground-truth -precision -taxon-coverage.  Not offered: -cov-percentile, -align.
#include "mcq_build.h"

#include <fcntl.h>
#include <unistd.h>

namespace {

using namespace mcq;

struct Options {
    std::string db, outfile;
    std::vector<std::string> infiles;
    int lowest = 0, highest = 19;                    // sequence .. domain (options.hpp:246-260)
    int hitsMin = 0; float hitsDiff = 1.0f;
    float covPercentile = 0.0f;          // -cov-percentile (options.cpp:918-930, :1313)
    uint64_t maxCand = 2, insertMax = 0;
    enum Pairing { unpaired, files, sequences } pairing = unpaired;
    bool tophits = false, allhits = false, locations = false, queryIds = false, lineage = false, separateCols = false;
    bool showName = true, showRank = true, showId = false;
    enum MapView { mv_none, mv_mapped, mv_all } mapView = mv_all;
    bool collapseUnclassified = true;
    std::string comment = "# ", none = "--", column = "\t|\t", taxSep = ",", rankSuffix = ":", idPrefix = "(", idSuffix = ")";
    bool showQueryParams = true, showSummary = true, showErrors = true, splitOut = false;
    uint32_t sketchlen = 0, winlen = 0, winstride = 0, batchSize = 1u << 16;
    uint32_t residentParts = 0;          // -resident-parts n (this program's own): a partitioned database n parts at a time, the next group loading behind
                                         // the queries (mc_partset_*; the reference's workflow: one query run per part + merge, docs/partitioning.md:116-153)
    std::vector<int32_t> gpus;           // -gpus a,b,...: the resident parts dealt out over these GPUs, per-part candidates gathered over RCCL
    bool shardKeys = false;              // -shard keys: ONE database key-sharded over the GPUs of -gpus (mc_keyset_*: every GPU holds the features it owns, the partial
                                         // location lists travel over RCCL to the GPU that owns the read); -shard parts = the default of -gpus (parts over GPUs)
    bool shardTargets = false;           // -shard targets: ONE database file cut into contiguous target ranges at load (mc_config.target_shard_*), a range per GPU of -gpus
    uint32_t targetShards = 0;           // -target-shards n: number of ranges (default: one per GPU of -gpus); with -resident-parts m: m ranges in HBM at a time
    uint32_t keyShards = 0;              // -key-shards n: number of key shards (default: one per GPU of -gpus; more than one per GPU on a single device)
    uint32_t replication = 1;            // -replicate: copies of the table on GPUs 0 .. n-1, the workers are dealt out over them (options.cpp:1155-1163)
    uint32_t refBatchSize = 0;           // -batch-size as given (the reference's batches matter for -cov-percentile)
    int maxLocs = -1, threads = 0;
    bool removeOverpopulated = false; float maxLoadFac = 0;
    uint64_t minReadLen = 0, maxReadLen = std::numeric_limits<uint64_t>::max();
    int64_t queryLimit = std::numeric_limits<int64_t>::max();
    bool hitsPerRef = false, abundances = false;
    bool showGroundTruth = false, determineGroundTruth = false, precision = false, taxonCoverage = false;
    int abundancePer = kNumRanks;                    // none
    std::string targetsFile, abundanceFile;
};

std::string sanitize_special_chars(const std::string& s)   // cmdline_utility: "\t" etc. typed literally
{
    std::string r;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '\\' && i + 1 < s.size()) {
            const char c = s[i + 1];
            if (c == 't') { r += '\t'; ++i; continue; }
            if (c == 'n') { r += '\n'; ++i; continue; }
        }
        r += s[i];
    }
    return r;
}

// Command line options of `metacache query` (options.cpp:860-1295); 'o' carries the defaults (interactive mode: the options of
// the initial command line).  args = everything after the database name.
Options parse(const std::vector<std::string>& args, Options o)
{
    o.infiles.clear();
    auto need = [&](size_t& i) -> std::string { if (i + 1 >= args.size()) throw std::runtime_error("value missing after '" + args[i] + "'"); return args[++i]; };
    for (size_t i = 0; i < args.size(); ++i) {
        const std::string& a = args[i];
        if (a.empty()) continue;
        if (a[0] != '-') { o.infiles.push_back(a); continue; }
        if (a == "-out") o.outfile = need(i);
        else if (a == "-split-out" || a == "-splitout") { o.splitOut = true; o.outfile = need(i); }
        else if (a == "-lowest") { int r = rank_from_name(need(i)); if (r < 0) throw std::runtime_error("unknown rank"); o.lowest = r; }
        else if (a == "-highest") { int r = rank_from_name(need(i)); if (r < 0) throw std::runtime_error("unknown rank"); o.highest = r; }
        else if (a == "-cov-percentile") { o.covPercentile = std::stof(need(i)); if (o.covPercentile > 1) o.covPercentile *= 0.01f; }
        else if (a == "-hitmin" || a == "-hit-min" || a == "-hits-min" || a == "-hitsmin") o.hitsMin = std::stoi(need(i));
        else if (a == "-hitdiff" || a == "-hit-diff" || a == "-hitsdiff" || a == "-hits-diff") o.hitsDiff = std::stof(need(i));
        else if (a == "-maxcand" || a == "-max-cand") o.maxCand = std::stoull(need(i));
        else if (a == "-tophits" || a == "-top-hits") o.tophits = true;
        else if (a == "-allhits" || a == "-all-hits") o.allhits = true;
        else if (a == "-locations") { o.locations = true; o.tophits = true; }
        else if (a == "-queryids" || a == "-query-ids") o.queryIds = true;
        else if (a == "-ground-truth") { o.determineGroundTruth = true; o.showGroundTruth = true; }
        else if (a == "-precision") { o.precision = true; o.determineGroundTruth = true; }
        else if (a == "-taxon-coverage") { o.taxonCoverage = true; o.precision = true; o.determineGroundTruth = true; }
        // an optional file name follows (clipp opt_value: the next word unless it is an option)
        else if (a == "-abundances" || a == "-abundance") { o.abundances = true; if (i + 1 < args.size() && args[i + 1][0] != '-') o.abundanceFile = args[++i]; }
        else if (a == "-abundance-per") { int r = rank_from_name(need(i)); if (r < 0) throw std::runtime_error("unknown rank"); if (r < kNumRanks - 1) o.abundancePer = r; }
        else if (a == "-hits-per-ref" || a == "-hits-per-seq" || a == "-hits-per-tgt" || a == "-hits-per-target") {
            o.hitsPerRef = true; if (i + 1 < args.size() && args[i + 1][0] != '-') o.targetsFile = args[++i]; }
        else if (a == "-mapped-only" || a == "-mappedonly") o.mapView = Options::mv_mapped;
        else if (a == "-no-map" || a == "-nomap") o.mapView = Options::mv_none;
        else if (a == "-taxids" || a == "-taxid") o.showId = true;
        else if (a == "-taxids-only" || a == "-taxidsonly") { o.showId = true; o.showName = false; }
        else if (a == "-omit-ranks" || a == "-omitranks") o.showRank = false;
        else if (a == "-separate-cols" || a == "-separatecols") o.separateCols = true;
        else if (a == "-separator") o.column = sanitize_special_chars(need(i));
        else if (a == "-comment") o.comment = need(i);
        else if (a == "-lineage" || a == "-lineages") o.lineage = true;
        else if (a == "-pairfiles" || a == "-pair-files" || a == "-paired-files") o.pairing = Options::files;
        else if (a == "-pairseq" || a == "-pair-seq" || a == "-paired-seq") o.pairing = Options::sequences;
        else if (a == "-insertsize" || a == "-insert-size") o.insertMax = std::stoull(need(i));
        else if (a == "-min-readlen") o.minReadLen = std::stoull(need(i));
        else if (a == "-max-readlen") o.maxReadLen = std::stoull(need(i));
        else if (a == "-query-limit") o.queryLimit = std::stoll(need(i));
        else if (a == "-sketchlen") o.sketchlen = (uint32_t)std::stoul(need(i));
        else if (a == "-winlen") o.winlen = (uint32_t)std::stoul(need(i));
        else if (a == "-winstride") o.winstride = (uint32_t)std::stoul(need(i));
        else if (a == "-max-locations-per-feature") o.maxLocs = std::stoi(need(i));
        else if (a == "-remove-overpopulated-features") o.removeOverpopulated = true;
        else if (a == "-max-load-fac" || a == "-max-load-factor") o.maxLoadFac = std::stof(need(i));
        else if (a == "-no-query-params" || a == "-no-queryparams") o.showQueryParams = false;
        else if (a == "-no-summary" || a == "-nosummary") o.showSummary = false;
        else if (a == "-no-err" || a == "-no-errors") o.showErrors = false;
        else if (a == "-threads") o.threads = std::stoi(need(i));
        else if (a == "-replicate") o.replication = (uint32_t)std::max(1, std::stoi(need(i)));
        else if (a == "-resident-parts") o.residentParts = (uint32_t)std::max(1, std::stoi(need(i)));
        else if (a == "-shard") {
            const std::string v = need(i);
            o.shardKeys = v == "keys"; o.shardTargets = v == "targets";
            if (v != "keys" && v != "parts" && v != "targets") throw std::runtime_error("-shard parts|keys|targets");
        }
        else if (a == "-target-shards") o.targetShards = (uint32_t)std::max(1, std::stoi(need(i)));
        else if (a == "-key-shards") o.keyShards = (uint32_t)std::max(1, std::stoi(need(i)));
        else if (a == "-gpus") {
            std::stringstream ss(need(i));
            for (std::string t; std::getline(ss, t, ',');) if (!t.empty()) o.gpus.push_back((int32_t)std::stoi(t));
        }
        else if (a == "-batch-size" || a == "-batchsize") { o.batchSize = (uint32_t)std::stoul(need(i)); o.refBatchSize = o.batchSize; }
        else throw std::runtime_error("unknown option '" + a + "'");
    }
    // process_query_options (options.cpp:1297-1366)
    {   // replace_directories_with_contained_files (options.cpp:146-165)
        std::vector<std::string> expanded;
        for (const auto& name : o.infiles) {
            auto sub = files_in_directory(name);
            if (sub.empty()) expanded.push_back(name); else expanded.insert(expanded.end(), sub.begin(), sub.end());
        }
        o.infiles.swap(expanded);
    }
    if (o.pairing == Options::files) { if (o.infiles.size() > 1) std::sort(o.infiles.begin(), o.infiles.end()); else o.pairing = Options::unpaired; }
    if (o.hitsDiff > 1) o.hitsDiff *= 0.01;       // double factor, as options.cpp:1312
    if (o.lowest > o.highest) o.lowest = o.highest;
    if (o.batchSize < 1) o.batchSize = 1;
    if (o.queryLimit < 0) o.queryLimit = 0;
    if (o.targetsFile == o.outfile) o.targetsFile.clear();
    if (o.abundanceFile == o.outfile) o.abundanceFile.clear();
    if (o.hitsPerRef) o.queryIds = true;
    if (o.separateCols) { o.collapseUnclassified = false; o.taxSep = o.column; o.rankSuffix = o.column; o.idPrefix = o.column; o.idSuffix = ""; }
    if (o.mapView == Options::mv_none && o.tophits) o.mapView = Options::mv_mapped;
    else if (o.allhits) o.mapView = Options::mv_all;
    return o;
}

// host threads worth running: the cgroup's CPU quota where there is one (a box may show 256 CPUs and grant 16), else the hardware threads
static unsigned granted_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0}; long long period = 0;
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const long long quota = std::atoll(q);
            if (quota > 0) n = std::max(1u, std::min(n, (unsigned)((quota + period / 2) / period)));
        }
        std::fclose(f);
    }
    return n;
}

// ---- output (printing.cpp:160-365) ----------------------------------------------------------------------------------
// The mapping lines of a batch are put together by its worker thread in a plain byte buffer: the inserters the line formatters use,
// without a stream's locale and sentry work per field (10^7 lines of `-tophits -queryids` took 5.7 s of CPU through std::ostringstream --
// 0.36 s of a 0.52 s query phase on the 16 granted cores -- against 0.6 s of kernels).  Same bytes as operator<< of std::ostream writes
// for these types (integers in decimal, strings and characters as they are).
struct FastOut {
    std::string s;
    FastOut& operator<<(char c) { s.push_back(c); return *this; }
    FastOut& operator<<(const char* p) { s.append(p); return *this; }
    FastOut& operator<<(const std::string& t) { s.append(t); return *this; }
    FastOut& put_u64(uint64_t v)
    {
        char b[24]; int i = 24;
        do { b[--i] = (char)('0' + v % 10); v /= 10; } while (v);
        s.append(b + i, (size_t)(24 - i));
        return *this;
    }
    FastOut& put_i64(int64_t v) { if (v < 0) { s.push_back('-'); return put_u64(0 - (uint64_t)v); } return put_u64((uint64_t)v); }
    template <class T, typename std::enable_if<std::is_integral<T>::value && std::is_unsigned<T>::value && !std::is_same<T, char>::value && !std::is_same<T, bool>::value, int>::type = 0>
    FastOut& operator<<(T v) { return put_u64((uint64_t)v); }
    template <class T, typename std::enable_if<std::is_integral<T>::value && std::is_signed<T>::value && !std::is_same<T, char>::value, int>::type = 0>
    FastOut& operator<<(T v) { return put_i64((int64_t)v); }
    FastOut& write(const char* p, std::streamsize n) { s.append(p, (size_t)n); return *this; }
    void str(const std::string& t) { s = t; }                  // (reset, as std::ostringstream::str(std::string()))
    const std::string& str() const { return s; }
};

template <class OS>
void print_taxon(OS& os, const Options& o, const std::string& name, int64_t id, int rank)
{
    if (o.showRank) { if (rank == kNumRanks) os << o.none; else os << kRankNames[rank]; os << o.rankSuffix; }
    if (o.showName) { os << name; if (o.showId) os << o.idPrefix << id << o.idSuffix; }
    else if (o.showId) os << id;
}

template <class OS>
void show_lineage(OS& os, const Options& o, const Taxonomy& tx, const Lineage& lin, int lowest, int highest)
{
    if (lowest == kNumRanks) return;
    if (highest == kNumRanks) highest = kNumRanks - 1;
    for (int r = lowest; r <= highest; ++r) {
        const Taxon* t = tx.taxon(lin[r]);
        if (t) print_taxon(os, o, t->name, t->id, t->rank); else print_taxon(os, o, o.none, 0, r);
        if (r < highest) os << o.taxSep;
    }
}

template <class OS>
void show_taxon(OS& os, const Options& o, const Taxonomy& tx, uint32_t best /* idx+1 */, bool bestIsTarget, uint32_t bestTgt)
{
    const Taxon* t = tx.taxon(best);
    if (!t || t->rank > o.highest) {
        if (o.collapseUnclassified) {
            if (o.showId && !o.showName && !o.showRank) os << 0; else os << o.none;
        } else {
            const int rmax = o.lineage ? o.highest : o.lowest;
            for (int r = o.lowest; r <= rmax; ++r) { print_taxon(os, o, o.none, 0, kNumRanks); if (r < rmax) os << o.taxSep; }
        }
    } else {
        const int rmin = o.lowest < t->rank ? t->rank : o.lowest;
        const int rmax = o.lineage ? o.highest : rmin;
        const Lineage lin = bestIsTarget ? tx.target_ranks(bestTgt) : tx.ranks_of(best);
        show_lineage(os, o, tx, lin, rmin, rmax);
    }
}

struct Cand { uint32_t tgt, hits, beg, end; uint32_t tax; /* idx+1 */ };

template <class OS>
void show_candidates(OS& os, const Options& o, const Taxonomy& tx, const std::vector<Cand>& c)
{
    for (size_t i = 0; i < c.size() && c[i].hits > 0; ++i) {
        if (i > 0) os << ',';
        const Taxon* t = tx.taxon(c[i].tax);
        if (o.lowest == 0) { if (t) os << t->name << ':' << c[i].hits; }
        else {
            const Taxon* a = t;
            if (t && t->rank < o.lowest) a = tx.taxon(tx.target_ranks(c[i].tgt)[o.lowest]);
            if (a) os << a->id; else os << t->name;
            os << ':' << c[i].hits;
        }
    }
}

template <class OS>
void show_matches(OS& os, const Options& o, const Taxonomy& tx, const mc_location* hits, uint64_t n)
{
    if (n == 0) return;
    auto emit = [&](const mc_location& cur, int count) {
        const Lineage lin = tx.target_ranks(cur.tgt);
        if (o.lowest == 0) { const Taxon* t = tx.taxon(lin[0]); if (t) os << t->name << '/' << int(cur.win) << ':' << count << ','; }
        else { const Taxon* t = tx.taxon(lin[o.lowest]); if (!t) t = tx.taxon(lin[0]); os << t->name << ':' << count << ','; }
    };
    uint64_t cur = 0; int count = 1;
    for (uint64_t i = 1; i < n; ++i) {
        if (hits[cur].tgt == hits[i].tgt && hits[cur].win == hits[i].win) ++count;
        else { emit(hits[cur], count); cur = i; count = 1; }
    }
    emit(hits[cur], count);
}

// ---- ground truth from the query header (classification.cpp:104-137; sequence_io.cpp:479-673) ---------------------------
uint32_t ground_truth(const Taxonomy& tx, const std::string& header)
{
    if (header.empty()) return 0;
    std::smatch m;
    std::regex_search(header, m, accession_regex());
    uint32_t t = tx.with_name(m[4].length() ? m[2].str() : std::string());      // accession.version
    if (t) return tx.next_ranked_ancestor(t);
    t = tx.with_similar_name(m[3].str());                                       // accession, any version
    if (t) return tx.next_ranked_ancestor(t);
    t = tx.with_id(taxon_id_in_header(header));
    if (t) return tx.next_ranked_ancestor(t);
    t = tx.with_name(header);
    if (t) return tx.next_ranked_ancestor(t);
    t = tx.with_name(leading_word(header));
    if (t) return tx.next_ranked_ancestor(t);
    t = tx.with_name(filename_without_extension(header));
    if (t) return tx.next_ranked_ancestor(t);
    return 0;
}

// classification.cpp:146-189
uint32_t classify(const Options& o, const Taxonomy& tx, const std::vector<Cand>& cand, bool& isTarget, uint32_t& tgt)
{
    isTarget = false; tgt = 0;
    if (cand.empty() || cand[0].hits == 0 || !cand[0].tax) return 0;
    if (cand[0].hits < (uint32_t)o.hitsMin) return 0;
    uint32_t lca = cand[0].tax;
    const float threshold = cand[0].hits > (uint32_t)o.hitsMin ? (cand[0].hits - o.hitsMin) * o.hitsDiff : 0;
    auto ranks = [&](const Cand& c) { return c.tgt < tx.numTargets ? tx.target_ranks(c.tgt) : tx.ranks_of(c.tax); };   // classification.cpp:166-176
    const Lineage top = ranks(cand[0]);
    for (size_t i = 1; i < cand.size() && cand[i].hits > 0; ++i) {
        if (cand[i].hits > threshold) {
            const Lineage cr = ranks(cand[i]);
            const int from = tx.taxon(lca)->rank;
            uint32_t l = 0;
            for (int r = from; r <= kNumRanks - 1; ++r) if (top[r] && top[r] == cr[r]) { l = top[r]; break; }   // ranked_lca
            lca = l;
            if (!lca || tx.taxon(lca)->rank > o.highest) return 0;
        } else break;
    }
    if (!lca || tx.taxon(lca)->rank > o.highest) return 0;
    if (lca == cand[0].tax && tx.taxon(lca)->rank == 0) { isTarget = true; tgt = cand[0].tgt; }
    return lca;
}

// ---- database session: the loaded context is kept between jobs (interactive mode) as long as its load-time settings fit ----
// Rows 9-10 on the host for ONE read's sorted location list: the reference's loop (for_all_contiguous_window_ranges,
// candidate_generation.hpp:47-108) and its sorted insert without a limit (:172-231, same std:: calls so that equal hit counts fall
// the same way).  Used when -maxcand 0 meets a read with more candidates than the device list holds.
static void host_candidates(const mc_location* h, uint64_t n, uint32_t maxWin, const Taxonomy& tx, int lowest, std::vector<Cand>& top)
{
    top.clear();
    auto greater = [](const Cand& a, const Cand& b) { return a.hits > b.hits; };
    auto insert = [&](Cand c) {
        c.tax = 0;
        if (c.tgt < tx.numTargets) {
            const uint32_t* lin = tx.targetLineages + (size_t)c.tgt * kNumRanks;
            if (lowest > 0) { for (int rk = lowest; rk < kNumRanks; ++rk) if (lin[rk]) { c.tax = lin[rk]; break; } }
            else c.tax = lin[0];
        }
        if (!c.tax) return;
        if (lowest <= 0) { top.insert(std::upper_bound(top.begin(), top.end(), c, greater), c); return; }
        auto i = std::find_if(top.begin(), top.end(), [&](const Cand& x) { return x.tax == c.tax; });
        if (i != top.end()) { if (c.hits > i->hits) { *i = c; std::sort(top.begin(), i + 1, greater); } }
        else top.insert(std::upper_bound(top.begin(), top.end(), c, greater), c);
    };
    if (n == 0) return;
    uint64_t fst = 0;
    uint32_t hits = 1;
    Cand best{h[0].tgt, 1, h[0].win, h[0].win, 0};
    for (uint64_t lst = 1; lst < n; ++lst) {
        if (h[lst].tgt == best.tgt) {
            ++hits;
            while (fst != lst && (uint32_t)(h[lst].win - h[fst].win) >= maxWin) { --hits; ++fst; }
            if (hits > best.hits) { best.hits = hits; best.beg = h[fst].win; best.end = h[lst].win; }
        } else {
            insert(best);
            fst = lst; hits = 1;
            best = Cand{h[lst].tgt, 1, h[lst].win, h[lst].win, 0};
        }
    }
    insert(best);
}

struct Session {
    mc_ctx* ctx = nullptr;
    mc_config cfg{};
    std::string db;
    Taxonomy tx;
    uint32_t dbStride = 0, dbSketch = 0, dbWinlen = 0;
    unsigned threads = 1, workers = 1;
    std::vector<mc_ctx*> replicas;                   // -replicate n: the same table on the GPUs 1 .. n-1 (ctx is the one on GPU 0)
    uint32_t replication = 1;
    mc_partset* partset = nullptr;                   // -resident-parts / -gpus: the parts as contexts of their own (ctx then holds the metadata only)
    mc_keyset* keyset = nullptr;                     // -shard keys: key shards as contexts of their own (ctx holds the metadata only)
    mc_ctx* replica(unsigned i) const { return i == 0 ? ctx : replicas[i - 1]; }
    void close_replicas() { for (mc_ctx* r : replicas) mc_destroy(r); replicas.clear(); }
    ~Session() { close_replicas(); if (partset) mc_partset_close(partset); if (keyset) mc_keyset_close(keyset); if (ctx) mc_destroy(ctx); }

    BuiltDatabase* built = nullptr;                  // build+query: the table comes from the builder's device arrays, not from files
    std::vector<uint32_t> builtLineages;

    static uint32_t unlimitedCap()
    {
        const char* e = std::getenv("MCQ_UNLIMITED_CAP");                       // tests: a small cap makes the host path run on toy data
        return e ? (uint32_t)std::max(1, std::atoi(e)) : 256u;
    }
    void open(const Options& o)
    {
        mc_config c; mc_config_default(&c);
        c.kmerlen = 0; c.sketchlen = o.sketchlen; c.winlen = o.winlen; c.winstride = o.winstride;
        // -maxcand 0 = no limit (options.cpp:1315-1317): the device lists hold unlimitedCap() candidates; a read that fills its list gets
        // its candidates from the HOST instead -- rows 9-10 over the read's sorted location list, which is copied back for that
        c.max_candidates = o.maxCand < 1 ? unlimitedCap() : (uint32_t)std::min<uint64_t>(o.maxCand, 4096);
        c.copy_allhits = (o.allhits || o.maxCand < 1) ? 1 : 0;
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        threads = o.threads > 0 ? (unsigned)o.threads : hw;                     // options.hpp: numThreads defaults to all hardware threads
        workers = std::min(threads, 64u);                                      // one batch slot (pinned staging) per worker
        if (o.threads <= 0) workers = std::min(workers, std::max(8u, 2u * granted_cpus()));   // (a container's CPU quota: more runnable threads than twice that only get the group throttled)
        const uint32_t nrep = built ? 1u : std::max(1u, o.replication);
        workers = std::max(workers, nrep);
        c.num_slots = (workers + nrep - 1) / nrep;                              // worker w: replica w % n, slot w / n
        c.slot_max_queries = o.batchSize;
        c.slot_max_chars = std::max<uint32_t>(1u << 24, o.batchSize * 320u);
        c.max_load_factor = o.maxLoadFac;
        if (built) {
            if (built->opt.maxLoadFac > 0.4f && built->opt.maxLoadFac < 0.99f) c.max_load_factor = built->opt.maxLoadFac;
        } else if (o.removeOverpopulated) {                                     // read_database, mode_query.cpp:69-92
            int maxlpf = o.maxLocs - 1;
            if (maxlpf < 0 || maxlpf >= 254) maxlpf = 253;
            c.remove_overpopulated = (uint32_t)maxlpf;                             // clamped to the DB's cap - 1 by mc_open_database
            c.max_locations_per_feature = o.maxLocs < 0 ? 0 : (uint32_t)std::max(1, o.maxLocs);
        } else if (o.maxLocs > 1) c.max_locations_per_feature = (uint32_t)o.maxLocs;
        const bool wantKeys = !built && o.shardKeys;
        const bool wantTargets = !built && o.shardTargets;
        const bool wantSet = !built && (o.residentParts > 0 || !o.gpus.empty() || wantKeys || wantTargets);
        if (ctx && db == o.db && std::memcmp(&c, &cfg, sizeof(c)) == 0 && replication == nrep && !wantSet && !partset && !keyset) return;  // same table, same slots: keep it
        close_replicas();
        if (partset) { mc_partset_close(partset); partset = nullptr; }
        if (keyset) { mc_keyset_close(keyset); keyset = nullptr; }
        if (ctx) { mc_destroy(ctx); ctx = nullptr; }
        replication = nrep;
        tx = Taxonomy{};
        if (wantSet) {
            if (o.allhits || o.maxCand < 1 || o.maxCand > 4 || o.covPercentile > 0)
                throw std::runtime_error("-resident-parts / -gpus / -shard: top candidates only (-maxcand 1..4, no -allhits, no -cov-percentile)");
            c.num_slots = 1;
            c.slot_max_queries = std::max<uint32_t>(o.batchSize, 1u << 16);
            c.slot_max_chars = std::max<uint32_t>(1u << 24, c.slot_max_queries * 320u);
            // (-shard targets: the part set's "parts" are the target ranges of the one file, mc_config.target_shard_count)
            if (wantTargets) c.target_shard_count = o.targetShards ? o.targetShards : (uint32_t)std::max<size_t>(o.gpus.size(), 1);
            if (wantKeys) {
                if (mc_keyset_open(o.db.c_str(), &c, o.keyShards, o.gpus.empty() ? nullptr : o.gpus.data(), (uint32_t)o.gpus.size(), &keyset) != MC_OK)
                    throw std::runtime_error(mc_keyset_last_error(nullptr));
            } else if (mc_partset_open(o.db.c_str(), &c, o.residentParts, o.gpus.empty() ? nullptr : o.gpus.data(), (uint32_t)o.gpus.size(), &partset) != MC_OK)
                throw std::runtime_error(mc_partset_last_error(nullptr));
            if (mc_open_metadata(o.db.c_str(), &ctx) != MC_OK) throw std::runtime_error(mc_last_error(nullptr));
            uint64_t nt = 0; mc_db_num_taxa(ctx, &nt); tx.taxa.resize(nt);
            for (uint64_t i = 0; i < nt; ++i) {
                uint32_t rk; const char* nm;
                mc_db_taxon(ctx, i, &tx.taxa[i].id, &tx.taxa[i].parent, &rk, &nm);
                tx.taxa[i].rank = int(rk); tx.taxa[i].name = nm;
                mc_db_taxon_source(ctx, i, nullptr, nullptr, &tx.taxa[i].windows);
            }
        } else if (built) {
            // add_to_database_and_query (mode_build_query.cpp:41-77): a query context over the builder's arrays
            for (mc_builder* b : built->bs) if (mc_build_set_query_config(b, &c) != MC_OK) throw std::runtime_error(mc_build_last_error(b));
            if (mc_build_finish_shards(built->bs.data(), (uint32_t)built->bs.size(), &ctx) != MC_OK)
                throw std::runtime_error(mc_build_last_error(built->bs[0]));
            tx.taxa = built->nonTarget;
            tx.taxa.insert(tx.taxa.end(), built->targets.begin(), built->targets.end());
        } else {
            if (mc_open_database(o.db.c_str(), &c, &ctx) != MC_OK) throw std::runtime_error(mc_last_error(nullptr));
            for (uint32_t r = 1; r < nrep; ++r) {                                // database::read with replication (database.cpp:207-215)
                mc_config cr = c; cr.device = (int32_t)r;
                mc_ctx* rc = nullptr;
                if (mc_open_database(o.db.c_str(), &cr, &rc) != MC_OK)
                    throw std::runtime_error(std::string("-replicate ") + std::to_string(nrep) + ": GPU " + std::to_string(r) + ": " + mc_last_error(nullptr));
                replicas.push_back(rc);
            }
            uint64_t nt = 0; mc_db_num_taxa(ctx, &nt); tx.taxa.resize(nt);
            for (uint64_t i = 0; i < nt; ++i) {
                uint32_t rk; const char* nm;
                mc_db_taxon(ctx, i, &tx.taxa[i].id, &tx.taxa[i].parent, &rk, &nm);
                tx.taxa[i].rank = int(rk); tx.taxa[i].name = nm;
                mc_db_taxon_source(ctx, i, nullptr, nullptr, &tx.taxa[i].windows);
            }
        }
        cfg = c; db = o.db;
        uint64_t info[8]; mc_db_info(ctx, info);
        dbSketch = (uint32_t)info[1]; dbWinlen = (uint32_t)info[2]; dbStride = (uint32_t)info[3];
        for (size_t i = 0; i < tx.taxa.size(); ++i) {
            tx.byId.emplace(tx.taxa[i].id, (uint32_t)i);
            if (tx.taxa[i].rank == 0 && tx.taxa[i].id < 0) tx.targetByName.emplace(tx.taxa[i].name, (uint32_t)i + 1);
        }
        if (built) {                                                            // ranked lineages of the targets (taxonomy.hpp:576-597)
            const size_t nt = built->targets.size(), base = built->nonTarget.size();
            builtLineages.assign(nt * kNumRanks, 0);
            for (size_t t = 0; t < nt; ++t) {
                const Lineage l = tx.ranks_of((uint32_t)(base + t) + 1);
                std::copy(l.begin(), l.end(), builtLineages.begin() + t * kNumRanks);
            }
            if (mc_set_lineages(ctx, builtLineages.data(), nt) != MC_OK) throw std::runtime_error(mc_last_error(ctx));
        }
        mc_db_lineages(ctx, &tx.targetLineages, &tx.numTargets);
        tx.build_covered();
    }
};

// process_input_files (querying.cpp:40-128): parameters, table layout, mappings of all given files, summary -> one output
// merge mode (mode_merge.cpp): the candidates come from result files instead of the database
struct MergedInput { std::vector<std::string> files, headers; std::vector<std::vector<Cand>> cands; };

void run_job(Session& S, Options o, const std::vector<std::string>& infiles, const std::string& outfile, const std::string& targetsFile,
             const std::string& abundanceFile, const MergedInput* merged = nullptr)
{
        if (!merged) S.open(o);
        mc_ctx* ctx = S.ctx;
        const Taxonomy& tx = S.tx;
        const mc_config& cfg = S.cfg;
        const uint32_t dbStride = S.dbStride, dbSketch = S.dbSketch;
        const unsigned threads = S.threads, workers = S.workers;
        const bool unlimited = o.maxCand < 1;
        if (o.hitsMin < 1) o.hitsMin = dbSketch >= 6 ? int(dbSketch / 3.0) : (dbSketch >= 4 ? 2 : 1);        // querying.cpp:257-268
        o.infiles = infiles;

        std::ofstream fout;
        if (!outfile.empty()) { fout.open(outfile); if (!fout.good()) throw std::runtime_error("Could not write to file " + outfile); }
        std::ostream& os = outfile.empty() ? std::cout : fout;
        std::ofstream ftargets, ftaxa;                                          // process_input_files, querying.cpp:84-112
        if (!targetsFile.empty()) { ftargets.open(targetsFile); if (!ftargets.good()) throw std::runtime_error("Could not write to file " + targetsFile); }
        if (!abundanceFile.empty()) { ftaxa.open(abundanceFile); if (!ftaxa.good()) throw std::runtime_error("Could not write to file " + abundanceFile); }
        std::ostream& perTargetOut = targetsFile.empty() ? os : ftargets;
        std::ostream& perTaxonOut = abundanceFile.empty() ? os : ftaxa;
        const bool taxCountsWanted = o.abundances || o.abundancePer != kNumRanks;
        struct Cover { uint32_t tgt; uint64_t qid; uint32_t beg, end, hits; };
        std::vector<Cover> covers;                                              // matches_per_target, all workers
        std::map<uint32_t, double> bestCounts;                                  // taxon (index + 1) -> queries classified as it

        if (o.showQueryParams) {                                                 // printing.cpp:47-131
            if (o.mapView != Options::mv_none) {
                os << o.comment << "Reporting per-read mappings (non-mapping lines start with '" << o.comment << "').\n";
                os << o.comment << (o.lineage ? "The complete lineage will be reported starting with the lowest match.\n" : "Only the lowest matching rank will be reported.\n");
            } else os << o.comment << "Per-Read mappings will not be shown.\n";
            if (o.minReadLen > 0) os << o.comment << "Only reads with a minimum length of " << o.minReadLen << " bp will be mapped.\n";
            if (o.maxReadLen < std::numeric_limits<uint64_t>::max()) os << o.comment << "Only reads with a maximum length of " << o.maxReadLen << " bp will be mapped.\n";
            os << o.comment << "Classification will be constrained to ranks from '" << kRankNames[o.lowest] << "' to '" << kRankNames[o.highest] << "'.\n";
            os << o.comment << "Classification hit threshold is " << o.hitsMin << " per query\n";
            os << o.comment << "At maximum " << (unlimited ? std::numeric_limits<size_t>::max() : o.maxCand) << " classification candidates will be considered per query.\n";
            if (o.pairing == Options::files) os << o.comment << "File based paired-end mode:\n" << o.comment << "  Reads from two consecutive files will be interleaved.\n" << o.comment << "  Max insert size considered " << o.insertMax << ".\n";
            else if (o.pairing == Options::sequences) os << o.comment << "Per file paired-end mode:\n" << o.comment << "  Reads from two consecutive sequences in each file will be paired up.\n" << o.comment << "  Max insert size considered " << o.insertMax << ".\n";
            if (o.hitsPerRef) os << o.comment << "A list of hits per reference sequence will be generated after the read mapping.\n";
            if (o.abundances) os << o.comment << "A list of absolute and relative abundances per taxon will be generated after the read mapping.\n";
            if (o.abundancePer != kNumRanks) os << o.comment << "A list of absolute and relative abundances for each '" << kRankNames[o.abundancePer] << "' will be generated after the read mapping.\n";
            os << o.comment << "Using " << threads << " threads\n";
        }
        if (merged) {                                                            // merge_result_files, mode_merge.cpp:266-269
            os << o.comment << "Merging " << merged->files.size() << " files:\n";
            for (const auto& f : merged->files) os << o.comment << f << '\n';
        }
        if (o.mapView != Options::mv_none) {                                     // show_query_mapping_header, classification.cpp:432-460
            os << o.comment << "TABLE_LAYOUT: ";
            if (o.queryIds) os << "query_id" << o.column;
            os << "query_header" << o.column;
            const int rmax = o.lineage ? o.highest : o.lowest;
            auto taxon_header = [&](const std::string& prefix) {              // show_taxon_header, printing.cpp:133-172
                auto hdr = [&](int r, bool named) {
                    if (o.showRank) os << prefix << (named ? kRankNames[r] : "rank") << o.rankSuffix;
                    if (o.showName) { os << prefix << "taxname"; if (o.showId) os << o.idPrefix << prefix << "taxid" << o.idSuffix; }
                    else if (o.showId) os << prefix << "taxid";
                };
                if (o.lowest == rmax) hdr(o.lowest, false);
                else for (int r = o.lowest; r <= rmax; ++r) { hdr(r, true); if (r < rmax) os << o.taxSep; }
            };
            if (o.showGroundTruth) { taxon_header("truth_"); os << o.column; }
            if (o.allhits) os << "all_hits" << o.column;
            if (o.tophits) os << "top_hits" << o.column;
            if (o.locations) os << "candidate_locations" << o.column;
            taxon_header("");
            os << '\n';
        }

        const auto t0 = std::chrono::steady_clock::now();
        uint64_t assigned[kNumRanks + 1] = {};                                   // classification_statistics (classification_statistics.hpp)
        uint64_t known[kNumRanks + 1] = {}, correct[kNumRanks + 1] = {}, wrong[kNumRanks + 1] = {}, covFalsePos[kNumRanks + 1] = {};
        uint64_t covTotalDomain = 0;

        // ---- all inputs are indexed first: batches = runs of consecutive queries, ids continue across files -------------
        // Batches are produced while the workers already run: plain files are indexed chunk by chunk (SeqFile::stream_*), and a batch
        // goes out as soon as its records are known -- the first one after a few megabytes instead of after the whole file.
        struct Batch { size_t f1, f2; size_t qBeg, qEnd; uint64_t idBase; std::string prefix; const std::vector<uint64_t>* sel; bool halfLast; };
        std::vector<std::unique_ptr<SeqFile>> files;
        files.reserve(o.infiles.size() + 2);                                   // workers read files[...] while the producer appends
        std::vector<std::unique_ptr<std::vector<uint64_t>>> selections;
        std::deque<Batch> batches;                                             // grows at the back only: references stay valid
        std::mutex batchMtx;
        std::condition_variable batchCv;
        bool producing = true;
        std::string producerError;
        auto push_batch = [&](Batch&& B) {
            { std::lock_guard<std::mutex> l(batchMtx); batches.push_back(std::move(B)); }
            batchCv.notify_all();
        };
        const bool lengthFilter = o.minReadLen > 0 || o.maxReadLen < std::numeric_limits<uint64_t>::max();
        double tIndexed = 0;
        auto produce = [&]() {
          try {
            uint64_t idOffset = 0;
            const size_t stride = o.pairing == Options::files ? 2 : 1;
            for (size_t fi = 0; fi < o.infiles.size(); fi += stride) {
                if (o.pairing == Options::files && fi + 1 >= o.infiles.size()) break;
                std::string prefix = o.comment + o.infiles[fi];                  // appendToOutput, classification.cpp:825-827
                if (o.pairing == Options::files) prefix += " + " + o.infiles[fi + 1];
                prefix += '\n';
                size_t nq = 0, f1 = files.size(), f2 = files.size();
                bool streamed = false;
                try {
                    files.emplace_back(new SeqFile(o.infiles[fi]));
                    SeqFile& F = *files.back();
                    if (F.can_stream() && o.pairing != Options::files && !lengthFilter && o.queryLimit >= (int64_t)1 << 62 && !std::getenv("MCQ_NO_STREAM")) {
                        // streaming: batches of this file go out while its later chunks are still being indexed
                        streamed = true;
                        const size_t per = o.pairing == Options::sequences ? 2 : 1;
                        F.stream_begin(std::max(1u, std::min(workers, 16u)));
                        size_t q = 0;
                        bool first = true;
                        for (;;) {
                            const size_t readable = F.stream_wait((q + o.batchSize) * per);
                            const bool done = F.stream_done();
                            const size_t qAvail = done ? (F.records() + per - 1) / per : readable / per;
                            while (q + o.batchSize <= qAvail || (done && q < qAvail)) {
                                const size_t qe = std::min(qAvail, q + o.batchSize);
                                const bool half = done && per == 2 && qe == qAvail && (F.records() & 1u);
                                push_batch(Batch{f1, f2, q, qe, idOffset, first ? prefix : std::string(), nullptr, half});
                                first = false; q = qe;
                            }
                            if (done) break;
                        }
                        F.stream_end();
                        if (first) push_batch(Batch{f1, f2, 0, 0, idOffset, prefix, nullptr, false});     // no records: the file's comment line only
                        nq = (F.records() + per - 1) / per;
                        size_t nread = nq;
                        if (per == 2 && nq > 0 && (F.records() & 1u)) --nread;                           // (the half pair did not count)
                        idOffset += nread;
                        continue;
                    }
                    files.back()->index(workers);
                    nq = files.back()->records();
                    if (o.pairing == Options::sequences) nq = (nq + 1) / 2;
                    if (o.pairing == Options::files) {
                        f2 = files.size();
                        files.emplace_back(new SeqFile(o.infiles[fi + 1]));
                        files.back()->index(workers);
                        nq = std::min(nq, files.back()->records());
                    }
                } catch (std::exception& e) { if (o.showErrors) std::cerr << "FAIL: " << e.what() << '\n'; nq = 0; }   // database_query.hpp:397-399
                // query_batched (database_query.hpp:258-284): at most -query-limit queries per file; with a read length filter a
                // query that fails it is replaced by the next record -- except when the file ends there
                const std::vector<uint64_t>* sel = nullptr;
                size_t nsel = std::min<uint64_t>(nq, (uint64_t)o.queryLimit), nread = nsel;
                if (lengthFilter && nq > 0) {
                    std::vector<uint32_t> len(nq);
                    {
                        std::vector<std::thread> pool;
                        const size_t per = (nq + workers - 1) / workers;
                        for (unsigned t = 0; t < workers; ++t)
                            pool.emplace_back([&, t] {
                                View h, sq; std::string scratch;
                                for (size_t q = t * per; q < std::min(nq, (t + 1) * per); ++q) {
                                    files[f1]->record(o.pairing == Options::sequences ? 2 * q : q, h, sq, scratch);
                                    len[q] = (uint32_t)std::min<size_t>(sq.n, 0xFFFFFFFFu);
                                }
                            });
                        for (auto& th : pool) th.join();
                    }
                    auto fails = [&](size_t q) { return len[q] < o.minReadLen || len[q] > o.maxReadLen; };
                    selections.emplace_back(new std::vector<uint64_t>());
                    auto& kept = *selections.back();
                    size_t q = 0, discarded = 0;
                    for (int64_t limit = o.queryLimit; q < nq && limit >= 1; --limit) {
                        size_t cur = q++;
                        while (fails(cur)) { ++discarded; if (q >= nq) break; cur = q++; }
                        kept.push_back(cur);
                    }
                    nread = q; nsel = kept.size(); sel = &kept;
                }
                const bool oddFile = o.pairing == Options::sequences && nq > 0 && (files[f1]->records() & 1u);
                for (size_t q = 0; q == 0 || q < nsel; q += o.batchSize) {
                    const size_t qe = std::min<size_t>(nsel, q + o.batchSize);
                    // the file's last query is a half pair: when it is among the selected ones, it is the last of them
                    const bool half = oddFile && qe == nsel && nsel > 0 && (sel ? (*sel)[nsel - 1] : nsel - 1) == nq - 1;
                    push_batch(Batch{f1, f2, q, qe, idOffset, q == 0 ? prefix : std::string(), sel, half});
                    if (nsel == 0) break;
                }
                if (oddFile && nread == nq) --nread;                             // (the half pair did not count)
                idOffset += nread;                                               // reader.index(): records (pairs) consumed
            }
          } catch (std::exception& e) { std::lock_guard<std::mutex> l(batchMtx); producerError = e.what(); }
          tIndexed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          { std::lock_guard<std::mutex> l(batchMtx); producing = false; }
          batchCv.notify_all();
        };

        const bool profile = std::getenv("MCQ_PROFILE") != nullptr;              // phase times on stderr (development aid)
        std::atomic<uint64_t> nsParse{0}, nsSubmit{0}, nsWait{0}, nsClassify{0};
        auto now_ns = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        // one query: classification, statistics, mapping line (classify_and_evaluate, classification.cpp:470-559)
        struct Acc {
            uint64_t mine[kNumRanks + 1] = {}, known[kNumRanks + 1] = {}, correct[kNumRanks + 1] = {}, wrong[kNumRanks + 1] = {}, falsePos[kNumRanks + 1] = {}, covDomain = 0;
            std::map<uint32_t, double> counts;
            std::vector<Cover> covers;
            // the text show_taxon writes for a classification (taxon, or target for sequence-level results): the same few thousand over and
            // over -- its lineage walk (hash lookups up the taxonomy, taxonomy.hpp:576-597) is done once per worker and taxon
            std::unordered_map<uint64_t, std::string> taxText;
        };
        // -cov-percentile (map_queries_to_targets_default, classification.cpp:747-838): nothing is classified while the reads are
        // queried; every read's candidates are kept, the targets are filtered by their coverage afterwards, and the reads are
        // classified from the candidates that are left
        const bool covMode = o.covPercentile > 0 && !merged;
        struct Deferred { uint64_t id; View header; std::vector<Cand> cands; };
        std::deque<std::vector<Deferred>> deferred;                              // [batch]; grown (under batchMtx) when a worker takes a batch
        auto emit = [&](Acc& A, auto& out, uint64_t id, View header, const std::vector<Cand>& cands, const mc_location* hits, uint64_t nhits) {
            bool isTarget; uint32_t tgt;
            const uint32_t best = classify(o, tx, cands, isTarget, tgt);
            const int bestRank = best ? tx.taxon(best)->rank : kNumRanks;
            ++A.mine[bestRank];
            uint32_t truth = 0;
            if (o.determineGroundTruth) truth = ground_truth(tx, std::string(header.p, header.n));
            if (o.precision) {                                       // evaluate_classification, classification.cpp:272-295
                const int knownRank = truth ? tx.taxon(truth)->rank : kNumRanks;
                int correctRank = kNumRanks;                         // rank of the ranked LCA of mapping and truth
                if (best && truth) {
                    const Lineage la = tx.ranks_of(best), lb = tx.ranks_of(truth);
                    for (int r = 0; r < kNumRanks; ++r) if (la[r] && la[r] == lb[r]) { correctRank = tx.taxon(la[r])->rank; break; }
                }
                // assign_known_correct (classification_statistics.hpp:86-106)
                if (correctRank < bestRank) correctRank = bestRank;
                if (correctRank < knownRank) correctRank = knownRank;
                ++A.known[knownRank];
                if (knownRank != kNumRanks) {
                    ++A.correct[correctRank];
                    if (correctRank > knownRank && correctRank > bestRank) ++A.wrong[correctRank - 1];
                }
                if (o.taxonCoverage && truth) {                       // update_coverage_statistics, classification.cpp:242-265
                    for (uint32_t t : tx.ranks_of(truth)) {
                        if (!t) continue;
                        const int r = tx.taxon(t)->rank;
                        const bool classifiedOnRank = best && r >= bestRank;
                        if (!tx.covers(t) && classifiedOnRank) ++A.falsePos[r];
                        if (r == 19) ++A.covDomain;
                    }
                }
            }
            if (taxCountsWanted && best) ++A.counts[best];           // classify_and_evaluate, classification.cpp:552-554
            if (o.hitsPerRef && !covMode)                            // matches_per_target::insert (matches_per_target.hpp:100-110)
                for (const Cand& c : cands) if (c.tax && c.hits >= (uint32_t)o.hitsMin) A.covers.push_back(Cover{c.tgt, id, c.beg, c.end, c.hits});
            if (o.mapView == Options::mv_none || (o.mapView == Options::mv_mapped && !best)) return;
            if (o.queryIds) out << id << o.column;
            const void* sp = memchr(header.p, ' ', header.n);
            out.write(header.p, sp ? (const char*)sp - header.p : (std::streamsize)header.n);
            out << o.column;
            if (o.showGroundTruth) { show_taxon(out, o, tx, truth, false, 0); out << o.column; }
            if (o.allhits) { if (hits) show_matches(out, o, tx, hits, nhits); out << o.column; }
            if (o.tophits) { show_candidates(out, o, tx, cands); out << o.column; }
            if (o.locations) {                                       // show_candidate_ranges, printing.cpp:370-380
                for (const Cand& c : cands) out << '[' << (uint64_t)dbStride * c.beg << ',' << (uint64_t)dbStride * c.end + S.dbWinlen << "] ";
                out << o.column;
            }
            {
                const uint64_t key = isTarget ? ((1ull << 32) | tgt) : (uint64_t)best;
                auto it = A.taxText.find(key);
                if (it == A.taxText.end()) {
                    FastOut t;
                    show_taxon(t, o, tx, best, isTarget, tgt);
                    it = A.taxText.emplace(key, std::move(t.s)).first;
                }
                out << it->second;
            }
            out << '\n';
        };
        auto collect = [&](const Acc& A) {
            for (int r = 0; r <= kNumRanks; ++r) {
                assigned[r] += A.mine[r]; known[r] += A.known[r]; correct[r] += A.correct[r]; wrong[r] += A.wrong[r]; covFalsePos[r] += A.falsePos[r];
            }
            covTotalDomain += A.covDomain;
            for (const auto& kv : A.counts) bestCounts[kv.first] += kv.second;
            covers.insert(covers.end(), A.covers.begin(), A.covers.end());
        };
        // ---- workers: one batch slot each; output delivered in batch order -------------------------------------------------
        size_t nextBatch = 0;                                                   // under batchMtx
        std::mutex outMtx, errMtx;
        std::map<size_t, std::string> finished;
        size_t nextToWrite = 0;
        std::string firstError;
        std::atomic<bool> failed{false};
        // Output file: the batches' texts go out in batch order, but not one after the other -- under the lock a finished batch only gets its
        // place in the file (the sizes of the batches before it are known then); the bytes are written by the worker that formatted them,
        // side by side with the others' (pwrite).  10^7 lines of `-tophits -queryids` are 1.1 GB: one thread copying them into the page cache
        // was a quarter of a second of a 0.3 s query phase.  (No file: std::cout, in order, as before.)
        int outFd = -1;
        uint64_t outOff = 0;
        std::vector<std::string> bufPool;                                       // written batches' buffers, for the next batches (a fresh 7 MB buffer per batch is 1 700 page faults)
        auto take_buffer = [&](std::string& into) {
            std::lock_guard<std::mutex> lock(outMtx);
            if (!bufPool.empty()) { into = std::move(bufPool.back()); bufPool.pop_back(); into.clear(); }
        };
        if (!outfile.empty()) {
            fout.flush();
            const std::streamoff at = fout.tellp();
            if (at >= 0) { outFd = ::open(outfile.c_str(), O_WRONLY); outOff = (uint64_t)at; }
        }
        auto deliver = [&](size_t b, std::string&& text) {
            std::vector<std::pair<uint64_t, std::string>> mine;
            {
                std::lock_guard<std::mutex> lock(outMtx);
                finished.emplace(b, std::move(text));
                for (auto it = finished.begin(); it != finished.end() && it->first == nextToWrite; it = finished.erase(it), ++nextToWrite) {
                    if (outFd >= 0) { const uint64_t n = it->second.size(); mine.emplace_back(outOff, std::move(it->second)); outOff += n; }
                    else os.write(it->second.data(), (std::streamsize)it->second.size());
                }
            }
            for (auto& m : mine) {
                const char* p = m.second.data();
                struct Recycle { std::string& s; std::mutex& mu; std::vector<std::string>& pool; ~Recycle() { s.clear(); std::lock_guard<std::mutex> l(mu); if (pool.size() < 64) pool.push_back(std::move(s)); } } recycle{m.second, outMtx, bufPool};
                uint64_t left = m.second.size(), at = m.first;
                while (left) {
                    const ssize_t w = ::pwrite(outFd, p, (size_t)std::min<uint64_t>(left, 1ull << 30), (off_t)at);
                    if (w <= 0) { std::lock_guard<std::mutex> l(errMtx); if (!failed.exchange(true)) firstError = "Could not write to file " + outfile; break; }
                    p += w; left -= (uint64_t)w; at += (uint64_t)w;
                }
            }
        };
        auto work = [&](unsigned worker) {
            const unsigned slot = worker / S.replication;
            mc_ctx* const ctx = S.replica(worker % S.replication);
            struct Meta { uint64_t id; View header; bool empty; uint64_t len; };
            std::vector<Meta> metas;
            std::vector<Cand> cands;
            std::string scratch1, scratch2;
            FastOut out;
            Acc A;
            auto fail = [&](const std::string& m) { std::lock_guard<std::mutex> l(errMtx); if (!failed.exchange(true)) firstError = m; };
            for (;;) {
                size_t b;
                const Batch* Bp = nullptr;
                std::vector<Deferred>* Dp = nullptr;                             // taken under the lock: operator[] walks the deque's map, which emplace_back may reallocate
                {
                    std::unique_lock<std::mutex> l(batchMtx);
                    batchCv.wait(l, [&] { return failed || nextBatch < batches.size() || !producing; });
                    if (failed || nextBatch >= batches.size()) break;           // (no batch left and the producer is through)
                    b = nextBatch++;
                    Bp = &batches[b];
                    if (covMode) { while (deferred.size() <= b) deferred.emplace_back(); Dp = &deferred[b]; }
                }
                const Batch& B = *Bp;
                out.s.clear();
                if (out.s.capacity() < 64) take_buffer(out.s);                     // (a moved-from string keeps its 15-character SSO capacity, never 0)
                out << B.prefix;
                size_t q = B.qBeg;
                while (q < B.qEnd && !failed) {
                    metas.clear();
                    const uint64_t tp0 = now_ns();
                    for (; q < B.qEnd; ++q) {
                        View h1, s1, h2, s2;
                        const size_t qi = B.sel ? (size_t)(*B.sel)[q] : q;          // query index inside the file (pair)
                        if (o.pairing == Options::sequences) {
                            files[B.f1]->record(2 * qi, h1, s1, scratch1);
                            if (!(B.halfLast && q + 1 == B.qEnd)) files[B.f1]->record(2 * qi + 1, h2, s2, scratch2);
                        } else {
                            files[B.f1]->record(qi, h1, s1, scratch1);
                            if (o.pairing == Options::files) files[B.f2]->record(qi, h2, s2, scratch2);
                        }
                        if (std::max(s1.n, s2.n) >= 0xFFFFFFF0ull) { fail("sequence too long"); break; }
                        const uint32_t maxWin = (uint32_t)(2 + std::max<uint64_t>(s1.n + s2.n, o.insertMax) / dbStride);   // candidate_structs.hpp:143-145
                        const bool tooBig = s1.n + s2.n + 8 > cfg.slot_max_chars;
                        const int rc = tooBig ? MC_BATCH_FULL : mc_batch_add(ctx, slot, s1.p, (uint32_t)s1.n, s2.p, (uint32_t)s2.n, maxWin);
                        if (rc == MC_BATCH_FULL) {
                            if (!tooBig && !metas.empty()) break;                 // submit what is there, then retry this query
                            std::cerr << "query batch is too small for a single read!\n";                                    // database_query.hpp:103
                            continue;
                        }
                        if (rc < 0) { fail(mc_last_error(ctx)); break; }
                        // query id = the reader's index after the read (database_query.hpp:264); a last pair without its second
                        // sequence does not advance the index (sequence_io.cpp:312-318), so it shares the id of the pair before it
                        const bool halfPair = B.halfLast && q + 1 == B.qEnd;
                        metas.push_back(Meta{B.idBase + qi + (halfPair ? 0 : 1), h1, h1.empty() || s1.empty(), (uint64_t)s1.n + s2.n});
                    }
                    if (failed) break;
                    mc_results r;
                    const uint64_t tp1 = now_ns();
                    if (mc_batch_submit(ctx, slot, o.lowest) != MC_OK) { fail(mc_last_error(ctx)); break; }
                    const uint64_t tp2 = now_ns();
                    if (mc_batch_wait(ctx, slot, &r) != MC_OK) { fail(mc_last_error(ctx)); break; }
                    const uint64_t tp3 = now_ns();
                    for (uint32_t i = 0; i < r.num_queries; ++i) {
                        const Meta& m = metas[i];
                        if (m.empty) continue;                                    // processQuery, classification.cpp:780
                        cands.clear();
                        for (uint32_t j = 0; j < r.max_candidates; ++j) {
                            const mc_candidate& c = r.cands[(size_t)i * r.max_candidates + j];
                            if (c.hits == 0) break;
                            Cand x{c.tgt, c.hits, c.beg, c.end, 0};
                            if (c.tgt < tx.numTargets) {
                                const uint32_t* lin = tx.targetLineages + (size_t)c.tgt * kNumRanks;
                                if (o.lowest > 0) { for (int rk = o.lowest; rk < kNumRanks; ++rk) if (lin[rk]) { x.tax = lin[rk]; break; } }   // lowest_ranked_ancestor
                                else x.tax = lin[0];
                            }
                            cands.push_back(x);
                        }
                        if (o.maxCand < 1 && cands.size() == r.max_candidates) {   // list full: there may be more candidates than it holds
                            const uint32_t mw = (uint32_t)(2 + std::max<uint64_t>(m.len, o.insertMax) / dbStride);
                            host_candidates(r.hits + r.hit_offsets[i], r.hit_offsets[i + 1] - r.hit_offsets[i], mw, tx, o.lowest, cands);
                        }
                        if (covMode) Dp->push_back(Deferred{m.id, m.header, cands});
                        else emit(A, out, m.id, m.header, cands, o.allhits ? r.hits + r.hit_offsets[i] : nullptr, o.allhits ? r.hit_offsets[i + 1] - r.hit_offsets[i] : 0);
                    }
                    mc_batch_clear(ctx, slot);
                    if (profile) { nsParse += tp1 - tp0; nsSubmit += tp2 - tp1; nsWait += tp3 - tp2; nsClassify += now_ns() - tp3; }
                }
                deliver(b, std::move(out.s));
            }
            std::lock_guard<std::mutex> l(errMtx);
            collect(A);
        };
        // -shard keys / -resident-parts / -gpus: the batches STREAM -- worker threads cut the records out of the files side by side, one at a
        // time hands its batch to mc_keyset_classify / mc_partset_classify_resident (the sets take one call at a time), then formats
        // its lines.  A partitioned database is gone through part group by part group (mc_partset_select_group: the next group loads
        // behind this one's queries); between the groups a batch keeps nothing but its reads' candidate lists (`carried`), the last
        // group's pass prints.  (Round 3 collected all reads on one thread first: 2.5 s per 10^7 reads.)

        std::deque<std::vector<mc_candidate>> carried;                          // [batch]; grown under batchMtx when a worker takes a batch
        bool setHasPrior = false, setLastPass = true;                           // the part group pass the workers are in
        auto work_keyset = [&](unsigned) {
            struct Meta { uint64_t id; View header; bool empty; };
            std::vector<Meta> metas;
            std::string seq1, seq2, scratch1, scratch2;
            std::vector<uint64_t> off1, off2;
            std::vector<mc_candidate> all;
            std::vector<Cand> cands;
            FastOut out;
            Acc A;
            const bool paired = o.pairing != Options::unpaired;
            const uint32_t K = cfg.max_candidates;
            auto fail = [&](const std::string& m) { std::lock_guard<std::mutex> l(errMtx); if (!failed.exchange(true)) firstError = m; };
            for (;;) {
                size_t b;
                const Batch* Bp = nullptr;
                std::vector<mc_candidate>* slot = nullptr;
                {
                    std::unique_lock<std::mutex> l(batchMtx);
                    batchCv.wait(l, [&] { return failed || nextBatch < batches.size() || !producing; });
                    if (failed || nextBatch >= batches.size()) break;
                    b = nextBatch++;
                    Bp = &batches[b];
                    if (S.partset) { if (carried.size() <= b) carried.resize(b + 1); slot = &carried[b]; }   // (a deque: the element stays where it is)
                }
                const Batch& B = *Bp;
                metas.clear(); seq1.clear(); seq2.clear(); off1.assign(1, 0); off2.assign(1, 0);
                for (size_t q = B.qBeg; q < B.qEnd; ++q) {
                    View h1, s1, h2, s2;
                    const size_t qi = B.sel ? (size_t)(*B.sel)[q] : q;
                    if (o.pairing == Options::sequences) {
                        files[B.f1]->record(2 * qi, h1, s1, scratch1);
                        if (!(B.halfLast && q + 1 == B.qEnd)) files[B.f1]->record(2 * qi + 1, h2, s2, scratch2);
                    } else {
                        files[B.f1]->record(qi, h1, s1, scratch1);
                        if (o.pairing == Options::files) files[B.f2]->record(qi, h2, s2, scratch2);
                    }
                    const bool halfPair = B.halfLast && q + 1 == B.qEnd;
                    if (s1.n + s2.n + 8 > cfg.slot_max_chars) { std::cerr << "query batch is too small for a single read!\n"; continue; }
                    metas.push_back(Meta{B.idBase + qi + (halfPair ? 0 : 1), h1, h1.empty() || s1.empty()});
                    seq1.append(s1.p, s1.n); off1.push_back(seq1.size());
                    if (paired) { seq2.append(s2.p, s2.n); off2.push_back(seq2.size()); }
                }
                const size_t n = metas.size();
                if (slot && setHasPrior) all.swap(*slot);              // what the earlier part groups left for this batch's reads
                else all.assign(n * K, mc_candidate{});
                if (all.size() != n * K) { fail("internal: a batch changed between two part groups"); break; }
                seq1.push_back('\0'); seq2.push_back('\0');
                if (n && S.keyset) {
                    // (the key set shares its two lanes among the callers, as the part set does: this worker's batch runs beside another worker's)
                    if (mc_keyset_classify(S.keyset, seq1.data(), off1.data(), paired ? seq2.data() : nullptr, paired ? off2.data() : nullptr, n, o.lowest,
                                           o.insertMax, all.data()) != MC_OK) { fail(mc_keyset_last_error(S.keyset)); break; }
                } else if (n) {
                    // (the part set shares its two device lanes among the callers: this worker's batch runs beside another worker's,
                    // its upload under the other's kernels)
                    if (mc_partset_classify_resident(S.partset, seq1.data(), off1.data(), paired ? seq2.data() : nullptr, paired ? off2.data() : nullptr, n,
                                                     o.lowest, o.insertMax, setHasPrior ? 1 : 0, all.data()) != MC_OK) { fail(mc_partset_last_error(S.partset)); break; }
                }
                if (!setLastPass) { slot->swap(all); continue; }            // more part groups to come: nothing is printed yet
                out.s.clear();
                if (out.s.capacity() < 64) take_buffer(out.s);                     // (a moved-from string keeps its 15-character SSO capacity, never 0)
                out << B.prefix;
                for (size_t i = 0; i < n; ++i) {
                    const Meta& m = metas[i];
                    if (m.empty) continue;
                    cands.clear();
                    for (uint32_t j = 0; j < K; ++j) {
                        const mc_candidate& c = all[i * K + j];
                        if (c.hits == 0) break;
                        Cand x{c.tgt, c.hits, c.beg, c.end, 0};
                        if (c.tgt < tx.numTargets) {
                            const uint32_t* lin = tx.targetLineages + (size_t)c.tgt * kNumRanks;
                            if (o.lowest > 0) { for (int rk = o.lowest; rk < kNumRanks; ++rk) if (lin[rk]) { x.tax = lin[rk]; break; } }
                            else x.tax = lin[0];
                        }
                        cands.push_back(x);
                    }
                    emit(A, out, m.id, m.header, cands, nullptr, 0);
                }
                deliver(b, std::move(out.s));
            }
            std::lock_guard<std::mutex> l(errMtx);
            collect(A);
        };
        if (S.keyset) {
            std::thread producer(produce);
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work_keyset, w);
            work_keyset(0);
            for (auto& t : pool) t.join();
            { std::lock_guard<std::mutex> l(batchMtx); }
            batchCv.notify_all();
            producer.join();
        } else if (S.partset) {
            uint64_t pinfo[6] = {0, 0, 1, 0, 0, 0};
            mc_partset_info(S.partset, pinfo);
            const uint32_t groups = (uint32_t)std::max<uint64_t>(pinfo[2], 1);
            std::thread producer(produce);                                       // (the first pass streams behind the files' indexing)
            for (uint32_t g = 0; g < groups && !failed; ++g) {
                if (mc_partset_select_group(S.partset, g) != MC_OK) { std::lock_guard<std::mutex> l(errMtx); if (!failed.exchange(true)) firstError = mc_partset_last_error(S.partset); break; }
                { std::lock_guard<std::mutex> l(batchMtx); nextBatch = 0; }
                setHasPrior = g > 0; setLastPass = g + 1 == groups;
                std::vector<std::thread> pool;
                for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work_keyset, w);
                work_keyset(0);
                for (auto& t : pool) t.join();
            }
            { std::lock_guard<std::mutex> l(batchMtx); }
            batchCv.notify_all();
            producer.join();
        } else {
            std::thread producer(produce);
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work, w);
            work(0);
            for (auto& t : pool) t.join();
            { std::lock_guard<std::mutex> l(batchMtx); }
            batchCv.notify_all();
            producer.join();
        }
        if (outFd >= 0) { ::close(outFd); fout.seekp((std::streamoff)outOff); }   // (the stream goes on behind the batches' lines)
        if (!producerError.empty()) throw std::runtime_error(producerError);
        if (failed) throw std::runtime_error(firstError);
        if (covMode) {
            // The reference keeps (target -> candidates) in std::unordered_map objects: one per batch of -batch-size reads (4096 unless
            // given, options.hpp:232; batches do not span input files), merged into a global one batch after batch
            // (matches_per_target.hpp:100-125).  filter_targets_by_coverage (classification.cpp:591-634) then walks the global map
            // in ITS iteration order, sums float coverages in that order, std::sorts them and erases targets from the low end: the order
            // of equal coverages and the rounding of the sums come from the containers.  Same containers, same insertion sequence here
            // (the reference's -threads 1 order), so the same targets go.
            const size_t refBatch = o.refBatchSize ? o.refBatchSize : 4096;
            std::unordered_map<uint32_t, std::vector<Cover>> tgtMatches;
            {
                std::unordered_map<uint32_t, std::vector<Cover>> batchMap;
                size_t inBatch = 0, curFile = (size_t)-1;
                auto flush = [&]() {
                    for (auto& m : batchMap) { auto& t = tgtMatches[m.first]; t.insert(t.end(), m.second.begin(), m.second.end()); }
                    batchMap = std::unordered_map<uint32_t, std::vector<Cover>>();
                    inBatch = 0;
                };
                for (size_t b = 0; b < batches.size() && b < deferred.size(); ++b) {
                    if (batches[b].f1 != curFile || batches[b].qBeg == 0) { if (inBatch) flush(); curFile = batches[b].f1; }
                    for (const Deferred& d : deferred[b]) {
                        for (const Cand& c : d.cands) if (c.tax && c.hits >= (uint32_t)o.hitsMin) batchMap[c.tgt].push_back(Cover{c.tgt, d.id, c.beg, c.end, c.hits});
                        if (++inBatch == refBatch) flush();
                    }
                }
                if (inBatch) flush();
            }
            {
                using CovP = std::pair<uint32_t, float>;
                std::vector<CovP> cov;
                cov.reserve(tgtMatches.size());
                float sum = 0;
                for (const auto& m : tgtMatches) {
                    const Lineage lin = tx.target_ranks(m.first);
                    const uint32_t targetSize = tx.taxon(lin[0]) ? (uint32_t)tx.taxon(lin[0])->windows : 0u;
                    std::unordered_set<uint32_t> hitWindows;
                    for (const Cover& c : m.second) for (uint32_t w = c.beg; w <= c.end; ++w) hitWindows.emplace(w);
                    const float covP = float(hitWindows.size()) / targetSize;
                    sum += covP;
                    cov.emplace_back(m.first, covP);
                }
                std::sort(cov.begin(), cov.end(), [](CovP& a, CovP& b) { return a.second < b.second; });
                float part = 0;
                for (auto it = cov.begin(); it != cov.end(); ++it) {
                    part += it->second;
                    if (part > o.covPercentile * sum) break;
                    tgtMatches.erase(it->first);
                }
            }
            // redo_classification_batched: candidates of erased targets go, then classification and output as usual
            Acc A;
            std::ostringstream out;
            std::vector<Cand> left;
            for (size_t b = 0; b < batches.size() && b < deferred.size(); ++b) {
                for (const Deferred& d : deferred[b]) {
                    left.clear();
                    for (const Cand& c : d.cands) if (tgtMatches.find(c.tgt) != tgtMatches.end()) left.push_back(c);
                    emit(A, out, d.id, d.header, left, nullptr, 0);
                }
                os << out.str();
                out.str(std::string());
            }
            collect(A);
            for (const auto& m : tgtMatches) covers.insert(covers.end(), m.second.begin(), m.second.end());
        }
        if (merged) {                                                            // map_candidates_to_targets, classification.cpp:891-911
            Acc A;
            std::ostringstream out;
            for (size_t i = 0; i < merged->headers.size(); ++i) {
                const std::string& h = merged->headers[i];
                emit(A, out, i + 1, View{h.data(), h.size()}, merged->cands[i], nullptr, 0);
            }
            os << out.str();
            collect(A);
        }
        if (profile)
            std::cerr << "mcq profile: index " << tIndexed * 1e3 << " ms, total " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3
                      << " ms; summed over " << workers << " workers: parse+add " << nsParse / 1e6 << " ms, submit " << nsSubmit / 1e6 << " ms, wait "
                      << nsWait / 1e6 << " ms, classify+format " << nsClassify / 1e6 << " ms; batches " << batches.size() << "\n";

        uint64_t nAssigned = 0;
        for (int r = 0; r < kNumRanks; ++r) nAssigned += assigned[r];
        const uint64_t nUnassigned = assigned[kNumRanks], nTotal = nAssigned + nUnassigned;

        if (o.hitsPerRef) {                                                      // show_matches_per_targets, printing.cpp:385-420
            std::sort(covers.begin(), covers.end(), [](const Cover& a, const Cover& b) {   // per target: by window range, then query id
                if (a.tgt != b.tgt) return a.tgt < b.tgt;
                if (a.beg != b.beg) return a.beg < b.beg;
                if (a.end != b.end) return a.end < b.end;
                return a.qid < b.qid;
            });
            perTargetOut << o.comment << "--- list of hits for each reference sequence ---\n"
                         << o.comment << "window start position within sequence = window_index * window_stride(=" << dbStride << ")\n";
            perTargetOut << o.comment << "TABLE_LAYOUT: " << " sequence " << o.column << " windows_in_sequence " << o.column
                         << "queryid/first_window_index+additional_windows:hits,queryid/...\n";
            for (size_t i = 0; i < covers.size();) {
                const uint32_t tgt = covers[i].tgt;
                const Lineage lin = tx.target_ranks(tgt);
                show_lineage(perTargetOut, o, tx, lin, 0, o.lineage ? o.highest : 0);
                perTargetOut << o.column << (tx.taxon(lin[0]) ? tx.taxon(lin[0])->windows : 0) << o.column;
                for (bool first = true; i < covers.size() && covers[i].tgt == tgt; ++i, first = false) {
                    if (!first) perTargetOut << ',';
                    perTargetOut << covers[i].qid << '/' << covers[i].beg << '+' << (covers[i].end - covers[i].beg) << ':' << covers[i].hits;
                }
                perTargetOut << '\n';
            }
        }

        if (taxCountsWanted) {
            // taxon_count_map (classification.hpp:48-56): higher ranks first, then ascending taxon id
            auto higher = [&tx](uint32_t a, uint32_t b) {
                const Taxon* x = tx.taxon(a); const Taxon* y = tx.taxon(b);
                if (x->rank != y->rank) return x->rank > y->rank;
                return x->id < y->id;
            };
            std::map<uint32_t, double, decltype(higher)> counts(higher);
            for (const auto& kv : bestCounts) counts[kv.first] = kv.second;
            auto table = [&]() {                                                 // show_abundance_table, printing.cpp:424-468
                perTaxonOut << o.comment << "rank" << o.rankSuffix << "name" << o.column << "taxid" << o.column << "number of reads" << o.column
                            << "abundance\n";
                double ipart = 0.0;
                for (const auto& tc : counts) {
                    const Taxon* t = tx.taxon(tc.first);
                    perTaxonOut << (t->rank == kNumRanks ? "none" : kRankNames[t->rank]) << o.rankSuffix << t->name << o.column;
                    perTaxonOut << (t->rank == 0 ? t->parent : t->id) << o.column;
                    if (std::modf(tc.second, &ipart) == 0.0) perTaxonOut << ipart;
                    else perTaxonOut << std::setprecision(15) << tc.second << std::setprecision(6);
                    perTaxonOut << o.column << (tc.second / double(nTotal) * 100) << "%\n";
                }
                perTaxonOut << "unclassified" << o.column << "--" << o.column << '0' << o.column << nUnassigned << o.column
                            << (nTotal > 0 ? nUnassigned / double(nTotal) : 0.0) * 100 << "%\n";
            };
            if (o.abundances) { perTaxonOut << o.comment << "query summary: number of queries mapped per taxon\n"; table(); }
            if (o.abundancePer != kNumRanks) {
                // estimate_abundance (classification.cpp:304-374)
                const int rank = o.abundancePer;
                auto first_ancestor = [&](uint32_t t, int from, auto&& accept) -> uint32_t {
                    const Lineage lin = tx.ranks_of(t);
                    for (int r = from; r < kNumRanks; ++r) if (lin[r] && accept(lin[r])) return lin[r];
                    return 0;
                };
                if (rank != 0) {
                    // counts below the estimation rank move up to the closest ancestor on or above it (first position that is not
                    // "higher" than an imaginary taxon of rank-1 with id 0)
                    auto it = counts.begin();
                    while (it != counts.end()) {
                        const Taxon* t = tx.taxon(it->first);
                        if (t->rank > rank - 1 || (t->rank == rank - 1 && t->id < 0)) ++it; else break;
                    }
                    while (it != counts.end()) {
                        const uint32_t anc = first_ancestor(it->first, rank, [](uint32_t) { return true; });
                        if (anc) { counts[anc] += it->second; it = counts.erase(it); } else ++it;
                    }
                }
                std::unordered_map<uint32_t, std::vector<uint32_t>> children;
                std::unordered_map<uint32_t, uint64_t> weight;
                for (const auto& tc : counts) weight[tc.first] = 0;
                for (auto it = counts.rbegin(); it != counts.rend(); ++it) {   // leaves to root: own count + weight go to the closest counted ancestor
                    const uint32_t parent = first_ancestor(it->first, tx.taxon(it->first)->rank + 1, [&](uint32_t a) { return weight.count(a) > 0; });
                    if (parent) {
                        weight[parent] = (uint64_t)(weight[parent] + (weight[it->first] + it->second));
                        children[parent].push_back(it->first);
                    }
                }
                for (auto it = counts.begin(); it != counts.end();) {            // root to leaves: a parent's count is shared out proportionally
                    auto ch = children.find(it->first);
                    if (ch != children.end()) {
                        const uint64_t sumChildren = weight[it->first];
                        for (uint32_t c : ch->second) counts[c] += it->second * (counts[c] + weight[c]) / sumChildren;
                        it = counts.erase(it);
                    } else ++it;
                }
                perTaxonOut << o.comment << "estimated abundance (number of queries) per " << kRankNames[rank] << "\n";
                table();
            }
        }
        if (o.showSummary) {                                                     // show_summary, printing.cpp:601-620
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            uint64_t nassigned = 0;
            for (int r = 0; r < kNumRanks; ++r) nassigned += assigned[r];
            const uint64_t unassigned = assigned[kNumRanks], total = nassigned + unassigned;
            const uint64_t nq = o.pairing == Options::unpaired ? total : 2 * total;
            os << o.comment << "queries: " << nq << '\n' << o.comment << "time:    " << (long long)(s * 1000) << " ms\n"
               << o.comment << "speed:   " << nq / (s / 60.0) << " queries/min\n";
            if (total > 0) {                                                     // show_taxon_statistics, printing.cpp:502-535
                if (nassigned < 1) os << "None of the input sequences could be classified.\n";
                else {
                    if (unassigned > 0) os << o.comment << "unclassified: " << (100 * (unassigned / double(total))) << "% (" << unassigned << ")\n";
                    os << o.comment << "classified:\n";
                    const int shown[] = {0, 3, 4, 6, 10, 12, 14, 16, 18, 19, 20};
                    auto upto = [](const uint64_t* a, int r) { uint64_t s = 0; for (int x = 0; x <= r; ++x) s += a[x]; return s; };
                    auto padded = [](int r) { std::string rn = kRankNames[r]; rn.resize(11, ' '); return rn; };
                    for (int r : shown)
                        if (upto(assigned, r) > 0) os << o.comment << "  " << padded(r) << (100 * (upto(assigned, r) / double(total))) << "% (" << upto(assigned, r) << ")\n";
                    if (upto(known, kNumRanks - 1) > 0) {                         // ground truth block, printing.cpp:537-592
                        if (known[kNumRanks] > 0) os << o.comment << "ground truth unknown: " << (100 * (known[kNumRanks] / double(total))) << "% (" << known[kNumRanks] << ")\n";
                        os << o.comment << "ground truth known:\n";
                        for (int r : shown) if (upto(assigned, r) > 0) os << o.comment << "  " << padded(r) << (100 * (upto(known, r) / double(total))) << "% (" << upto(known, r) << ")\n";
                        os << o.comment << "correctly classified:\n";
                        for (int r : shown) if (upto(assigned, r) > 0) os << o.comment << "  " << padded(r) << upto(correct, r) << '\n';
                        auto wrongFrom = [&](int r) { uint64_t s = 0; for (int x = r; x < kNumRanks; ++x) s += wrong[x]; return s; };
                        os << o.comment << "precision (correctly classified / classified) if ground truth known:\n";
                        for (int r : shown) if (upto(assigned, r) > 0) {
                            const double tot = double(upto(correct, r)) + double(wrongFrom(r));
                            os << o.comment << "  " << padded(r) << (100 * (tot > 0 ? upto(correct, r) / tot : 0.0)) << "%\n";
                        }
                        os << o.comment << "sensitivity (correctly classified / all) if ground truth known:\n";
                        for (int r : shown) if (upto(assigned, r) > 0)
                            os << o.comment << "  " << padded(r) << (100 * (upto(known, r) > 0 ? upto(correct, r) / double(upto(known, r)) : 0.0)) << "%\n";
                        if (covTotalDomain > 0) {
                            os << o.comment << "false positives (hit on taxa not covered in DB):\n";
                            for (int r : shown) if (upto(assigned, r) > 0) os << o.comment << "  " << padded(r) << covFalsePos[r] << "\n";
                        }
                    }
                }
            } else std::cerr << o.comment << "No valid query sequences found.\n";
        }
}

// ---- merge mode (mode_merge.cpp): result files of queries against single database parts -> one classification ------------------
// get_results_file_properties (mode_merge.cpp:79-150): no sequence-level results, the column of the top hits, number of result lines
struct ResultsSource { std::string filename; size_t firstLine = 0, numQueries = 0; int tophitsColumn = 0; std::vector<std::string> lines; };

ResultsSource results_file_properties(const std::string& filename)
{
    ResultsSource res;
    res.filename = filename;
    std::ifstream is(filename);
    if (!is.good()) throw std::runtime_error("could not open file " + filename);
    for (std::string l; std::getline(is, l);) res.lines.push_back(std::move(l));
    size_t i = 0;
    for (;; ++i) {
        if (i >= res.lines.size() || res.lines[i].empty() || res.lines[i][0] != '#') throw std::runtime_error("classificaion ranks not found in file " + filename);
        if (res.lines[i].compare(0, 16, "# Classification") == 0) {
            if (res.lines[i].find("sequence") != std::string::npos) throw std::runtime_error("cannot merge results on sequence level");
            break;
        }
    }
    for (++i;; ++i) {
        if (i >= res.lines.size() || res.lines[i].empty() || res.lines[i][0] != '#') throw std::runtime_error("TABLE_LAYOUT not found in file " + filename);
        if (res.lines[i].compare(0, 15, "# TABLE_LAYOUT:") == 0) {
            std::istringstream ls(res.lines[i].substr(15));
            std::string column;
            ls >> column;
            if (column != "query_id") throw std::runtime_error("no query_id in file " + filename);
            int col = 0;
            while (ls.good()) {
                ls.ignore(std::numeric_limits<std::streamsize>::max(), '|');
                ls >> column;
                ++col;
                if (column == "top_hits") { res.tophitsColumn = col; break; }
            }
            break;
        }
    }
    if (res.tophitsColumn < 1) throw std::runtime_error("no top_hits in file " + filename);
    for (++i; i < res.lines.size() && !res.lines[i].empty() && res.lines[i][0] == '#'; ++i) {}
    res.firstLine = i;
    for (; i < res.lines.size(); ++i) if (res.lines[i].empty() || res.lines[i][0] != '#') ++res.numQueries;
    return res;
}

// best_distinct_matches_in_contiguous_window_ranges::insert for a candidate that names its taxon (candidate_generation.hpp:172-231)
void insert_taxon_candidate(std::vector<Cand>& top, Cand c, int mergeBelow, size_t maxCand)
{
    if (top.size() == maxCand && top.back().hits >= c.hits) return;
    if (!c.tax) return;
    auto upper = [&] { size_t j = 0; while (j < top.size() && top[j].hits >= c.hits) ++j; return j; };   // upper_bound, hits descending
    auto place = [&] {
        const size_t j = upper();
        if (j != top.size() || top.size() < maxCand) { top.insert(top.begin() + j, c); if (top.size() > maxCand) top.resize(maxCand); }
    };
    if (mergeBelow == 0) { place(); return; }
    size_t i = 0;
    while (i < top.size() && top[i].tax != c.tax) ++i;
    if (i == top.size()) { place(); return; }
    if (c.hits > top[i].hits) {
        top[i] = c;
        for (size_t j = i; j > 0 && top[j].hits > top[j - 1].hits; --j) std::swap(top[j], top[j - 1]);   // std::sort on <= 16 elements: insertion sort
    }
}

// read_results (mode_merge.cpp:158-244)
void read_results(const ResultsSource& res, const Taxonomy& tx, int mergeBelow, size_t maxCand, MergedInput& M, bool info)
{
    // "preallocate" (mode_merge.cpp:172-174) is a plain resize to this file's number of result lines: when ids have gaps (skipped
    // reads) the lists of the highest ids collected from earlier files are cut off here and start again -- kept as the reference has it
    M.cands.resize(res.numQueries); M.headers.resize(res.numQueries);
    for (size_t li = res.firstLine; li < res.lines.size(); ++li) {
        const std::string& l = res.lines[li];
        if (l.empty() || l[0] == '#') continue;
        const char* p = l.c_str();
        char* end = nullptr;
        size_t queryId = (size_t)std::strtoull(p, &end, 10);
        p = end;
        if (queryId > 0) --queryId;
        if (queryId + 1 > M.cands.size()) { M.cands.resize(queryId + 1); M.headers.resize(queryId + 1); }
        auto forward = [&](char c) { const char* q = std::strchr(p, c); p = q ? q + 1 : l.c_str() + l.size(); };
        forward('|');
        if (M.headers[queryId].empty()) {
            const char* a = p;
            while (*a && std::isspace((unsigned char)*a)) ++a;
            const char* b = a;
            while (*b && !std::isspace((unsigned char)*b)) ++b;
            if (b > a) M.headers[queryId].assign(a, b);
        }
        for (int i = 1; i < res.tophitsColumn; ++i) forward('|');
        forward('\t');
        while (*p && *p != '\t') {
            int64_t taxid = std::strtoll(p, &end, 10);
            if (end == p) { taxid = 0; if (info) std::cerr << "Query " << queryId + 1 << ": Could not read taxid.\n"; }
            p = end;
            forward(':');
            const uint32_t hits = (uint32_t)std::strtoul(p, &end, 10);
            p = end;
            const uint32_t t = tx.with_id(taxid);
            if (t) insert_taxon_candidate(M.cands[queryId], Cand{0xFFFFFFFFu, hits, 0, 0, t}, mergeBelow, maxCand);
            else if (info) std::cerr << "Query " << queryId + 1 << ": taxid " << taxid << " not found. Skipping hit.\n";
            if (!*p) break;
            if (*p == '\t') break;                                              // end of the top hits
            ++p;                                                                // ',' between them
        }
    }
}

int merge_main(const std::vector<std::string>& args)
{
    // get_merge_options (options.cpp:1776-1887)
    std::vector<std::string> qargs;
    std::vector<std::string> none;
    std::string taxPath;
    int info = 1;
    for (size_t i = 0; i < args.size(); ++i) {
        if (args[i] == "-taxonomy") { if (i + 1 >= args.size()) throw std::runtime_error("Taxonomy path missing after '-taxonomy'"); taxPath = args[++i]; }
        else if (args[i] == "-silent") info = 0;
        else if (args[i] == "-verbose") info = 2;
        else qargs.push_back(args[i]);
    }
    if (taxPath.empty()) throw std::runtime_error("Taxonomy path missing. Use '-taxonomy <path>'");
    Options o;
    o = parse(qargs, o);
    if (o.infiles.empty()) throw std::runtime_error("No query output filenames provided");
    std::sort(o.infiles.begin(), o.infiles.end());
    if (o.hitsMin == 0) o.hitsMin = 5;
    if (o.lowest < 4) o.lowest = 4;                                             // species
    if (o.infiles.size() < 2) throw std::runtime_error("At least two files are needed for merging");
    // main_mode_merge (mode_merge.cpp:401-432): a database without targets, taxa from the dumps
    BuildOptions bo;
    bo.taxPath = taxPath;
    if (bo.taxPath.back() != '/') bo.taxPath += '/';
    bo.info = info == 0 ? BuildOptions::silent : BuildOptions::moderate;
    Session S;
    {
        std::ostringstream quiet;                                               // the dump reader reports on stdout
        std::streambuf* old = std::cout.rdbuf(quiet.rdbuf());
        TaxTree t = read_taxonomy_dumps(bo);
        std::cout.rdbuf(old);
        if (info) std::cout << quiet.str();
        S.tx.taxa = std::move(t.taxa);
    }
    for (size_t i = 0; i < S.tx.taxa.size(); ++i) S.tx.byId.emplace(S.tx.taxa[i].id, (uint32_t)i);
    S.tx.build_covered();
    if (info) std::cerr << "Applied taxonomy to database.\n";
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    S.threads = o.threads > 0 ? (unsigned)o.threads : hw;
    S.workers = 1;
    if (info) std::cerr << "Merging result files.\n";
    bool anyReadable = false;
    for (const auto& f : o.infiles) { std::ifstream is(f); anyReadable = anyReadable || is.good(); }
    if (!anyReadable) throw std::runtime_error("None of the query sequence files could be opened");
    if (!o.outfile.empty() && info) std::cout << "Per-Read mappings will be written to file: " << o.outfile << std::endl;
    MergedInput M;
    M.files = o.infiles;
    const size_t maxCand = o.maxCand > 0 ? (size_t)o.maxCand : 2;               // merge_result_files, mode_merge.cpp:254-258
    for (const auto& f : o.infiles) read_results(results_file_properties(f), S.tx, o.lowest, maxCand, M, info != 0);
    if (info) std::cerr << "Completed merge. Starting classification.\n";
    run_job(S, o, {}, o.outfile, "", o.abundanceFile, &M);
    return 0;
}

// ---- info mode (mode_info.cpp), the parts that read metadata only: database properties, targets, lineage table, rank statistics ----
int info_main(const std::vector<std::string>& args)
{
    auto static_properties = [](uint32_t tb, uint64_t k, uint64_t sk, uint64_t w, uint64_t stride, uint64_t maxLocs) {   // printing.cpp:625-657
        const char* tid = tb == 2 ? "unsigned short int 16 bits" : "unsigned int 32 bits";
        std::cout << "------------------------------------------------\n"
                  << "MetaCache version  mcq / metacache_amd (MI355X)\n"
                  << "database version   20200820\n"
                  << "------------------------------------------------\n"
                  << "sequence type      mc::char_sequence\n"
                  << "target id type     " << tid << "\n"
                  << "target limit       " << (tb == 2 ? 65535ull : 4294967295ull) << "\n"
                  << "------------------------------------------------\n"
                  << "window id type     unsigned int 32 bits\n"
                  << "window limit       4294967295\n"
                  << "window length      " << w << "\n"
                  << "window stride      " << stride << "\n"
                  << "------------------------------------------------\n"
                  << "sketcher type      mc::single_function_unique_min_hasher<unsigned int, mc::same_size_hash<unsigned int> >\n"
                  << "feature type       unsigned int 32 bits\n"
                  << "feature hash       mc::same_size_hash<unsigned int>\n"
                  << "kmer size          " << k << "\n"
                  << "kmer limit         16\n"
                  << "sketch size        " << sk << "\n"
                  << "------------------------------------------------\n"
                  << "bucket size type   unsigned char 8 bits\n"
                  << "max. locations     " << maxLocs << "\n"
                  << "location limit     254\n"
                  << "------------------------------------------------" << std::endl;
    };
    auto query_config = [] {
        std::cout << "hit classifier       mc::best_distinct_matches_in_contiguous_window_ranges\n"
                  << "------------------------------------------------\n";
    };
    if (args.empty()) { static_properties(4, 0, 0, 0, 0, 254); query_config(); std::cout << std::endl; return 0; }   // show_basic_exec_info: an empty database
    std::string name = args[0];
    { auto pos = name.find(".meta"); if (pos != std::string::npos) name.erase(pos); else { pos = name.find(".cache"); if (pos != std::string::npos) name.erase(pos); } }
    const std::string what = args.size() > 1 ? args[1] : "";
    if (what == "statistics" || what == "stat" || what == "locations" || what == "loc" || what == "featuremap" || what == "features" || what == "featurecounts")
        throw std::runtime_error("'info " + what + "' describes the reference's host hash table; not available here");
    std::cerr << "Reading database '" << name << "' ... Reading database metadata ...\nCompleted database reading.\ndone." << std::endl;
    mc_ctx* ctx = nullptr;
    if (mc_open_metadata(name.c_str(), &ctx) != MC_OK) throw std::runtime_error(mc_last_error(nullptr));
    struct Guard { mc_ctx* c; ~Guard() { mc_destroy(c); } } guard{ctx};
    uint64_t info[8]; mc_db_info(ctx, info);
    Taxonomy tx;
    std::vector<std::string> srcFile; std::vector<uint64_t> srcIndex;
    uint64_t nt = 0; mc_db_num_taxa(ctx, &nt);
    tx.taxa.resize(nt); srcFile.resize(nt); srcIndex.resize(nt);
    for (uint64_t i = 0; i < nt; ++i) {
        uint32_t rk; const char *nm, *fn;
        mc_db_taxon(ctx, i, &tx.taxa[i].id, &tx.taxa[i].parent, &rk, &nm);
        tx.taxa[i].rank = int(rk); tx.taxa[i].name = nm;
        mc_db_taxon_source(ctx, i, &fn, &srcIndex[i], &tx.taxa[i].windows);
        srcFile[i] = fn;
        tx.byId.emplace(tx.taxa[i].id, (uint32_t)i);
        if (tx.taxa[i].rank == 0 && tx.taxa[i].id < 0) tx.targetByName.emplace(tx.taxa[i].name, (uint32_t)i + 1);
    }
    mc_db_lineages(ctx, &tx.targetLineages, &tx.numTargets);
    if (what.empty()) {                                                          // show_database_config
        uint32_t tb = 4;
        { std::ifstream is(name + ".meta", std::ios::binary); char hdr[10] = {}; is.read(hdr, 10); if (is.gcount() == 10) tb = (uint8_t)hdr[9]; }
        static_properties(tb, info[0], info[1], info[2], info[3], info[4]);
        query_config();
        std::cout << "database parts       " << info[6] << "\n------------------------------------------------\n";
        return 0;
    }
    auto show_target = [&](uint32_t tgt) {                                       // show_target_info, mode_info.cpp:106-121
        const Lineage lin = tx.target_ranks(tgt);
        const Taxon* t = tx.taxon(lin[0]);
        if (!t) return;
        std::cout << "Target " << t->name << "):\n    source:     " << srcFile[lin[0] - 1] << " / " << srcIndex[lin[0] - 1]
                  << "\n    length:     " << t->windows << " windows";
        for (uint32_t l : lin) {
            const Taxon* a = tx.taxon(l);
            if (!a) continue;
            std::string rn = std::string(kRankNames[a->rank]) + ":";
            rn.resize(12, ' ');
            std::cout << "\n    " << rn << "(" << a->id << ") " << a->name;
        }
        std::cout << '\n';
    };
    if (what == "targets" || what == "target" || what == "reference" || what == "references" || what == "ref" || what == "sequence" || what == "sequences" || what == "seq") {
        if (args.size() > 2) {
            for (size_t i = 2; i < args.size(); ++i) {
                const uint32_t t = tx.with_name(args[i]);
                if (t && tx.taxon(t)->id < 0) show_target((uint32_t)(-tx.taxon(t)->id - 1));
                else std::cout << "Target (reference sequence) '" << args[i] << "' not found in database.\n";
            }
        } else {
            std::cout << "Targets (reference sequences) in database:\n";
            for (uint64_t t = 0; t < tx.numTargets; ++t) show_target((uint32_t)t);
        }
        return 0;
    }
    if (what == "lineages" || what == "lineage" || what == "lin") {              // show_lineage_table, mode_info.cpp:157-186
        if (tx.numTargets < 1) return 0;
        std::cout << "name";
        for (int r = 0; r <= 19; ++r) std::cout << '\t' << kRankNames[r];
        std::cout << '\n';
        for (uint64_t t = 0; t < tx.numTargets; ++t) {
            const Lineage lin = tx.target_ranks((uint32_t)t);
            const Taxon* me = tx.taxon(lin[0]);
            std::cout << (me ? me->name : std::string());
            for (int r = 0; r <= 19; ++r) { const Taxon* a = tx.taxon(lin[r]); std::cout << '\t' << (a ? a->id : 0); }
            std::cout << '\n';
        }
        return 0;
    }
    if (what == "rank") {                                                        // show_rank_statistics, mode_info.cpp:193-232
        const int rank = args.size() > 2 ? rank_from_name(args[2]) : -1;
        if (rank < 0 || rank >= kNumRanks) {
            std::cerr << "Please specify a taxonomic rank:\n";
            for (int r = 0; r <= 19; ++r) std::cerr << "    " << kRankNames[r] << '\n';
            return 0;
        }
        std::vector<std::pair<uint32_t, size_t>> stat;                           // the reference walks a map keyed by taxon address
        for (uint64_t t = 0; t < tx.numTargets; ++t) {
            const uint32_t a = tx.target_ranks((uint32_t)t)[rank];
            if (!a) continue;
            auto it = std::find_if(stat.begin(), stat.end(), [&](const std::pair<uint32_t, size_t>& x) { return x.first == a; });
            if (it == stat.end()) stat.emplace_back(a, 1); else ++it->second;
        }
        std::cout << "Sequence distribution for rank '" << kRankNames[rank] << "':\n" << "taxid \t taxon_name \t sequences" << std::endl;
        for (const auto& x : stat) std::cout << tx.taxon(x.first)->id << " \t " << tx.taxon(x.first)->name << " \t " << x.second << '\n';
        return 0;
    }
    throw std::runtime_error("unknown info topic '" + what + "'");
}

// query mode proper / the query half of build+query: files of the command line, or the interactive loop
int query_main(Session& S, const Options& init)
{
    auto process = [&](const Options& o) {                                   // process_input_files, querying.cpp:141-222
        if (!o.splitOut) { run_job(S, o, o.infiles, o.outfile, o.targetsFile, o.abundanceFile); return; }
        const size_t stride = (o.pairing == Options::files && o.infiles.size() > 1) ? 2 : 1;
        for (size_t i = 0; i + stride <= o.infiles.size(); i += stride) {
            std::vector<std::string> in(o.infiles.begin() + i, o.infiles.begin() + i + stride);
            std::string suffix;
            for (const auto& f : in) suffix += "_" + f.substr(f.find_last_of("/\\") + 1);
            auto named = [&](const std::string& f) { return f.empty() ? std::string() : f + suffix + ".txt"; };
            run_job(S, o, in, named(o.outfile), named(o.targetsFile), named(o.abundanceFile));
        }
    };
    if (!init.infiles.empty()) { process(init); return 0; }
    // run_interactive_query_mode (querying.cpp:274-322)
    S.open(init);
    std::cout << "Running in interactive mode:\n"
                 " - Enter input file name(s) and command line options and press return.\n"
                 " - The initially given command line options will be used as defaults.\n"
                 " - All command line options that would modify the database are ignored.\n"
                 " - Each line will be processed separately.\n"
                 " - Lines starting with '#' will be ignored.\n"
                 " - Enter an empty line or press Ctrl-D to quit MetaCache.\n" << std::endl;
    for (;;) {
        std::cout << "$> " << std::flush;
        std::string line;
        std::getline(std::cin, line);
        if (line.empty() || line.find(":q") == 0) { std::cout << "Terminate." << std::endl; return 0; }
        if (line[0] == '#') continue;
        std::vector<std::string> args;
        std::istringstream iss(line);
        for (std::string w; iss >> w;) args.push_back(w);
        try {
            Options o = parse(args, init);
            // load-time settings stay those of the initial command line
            o.maxLocs = init.maxLocs; o.removeOverpopulated = init.removeOverpopulated; o.maxLoadFac = init.maxLoadFac;
            if (o.infiles.empty()) { if (init.showErrors) std::cerr << "No input filenames provided!\n"; continue; }
            process(o);
        } catch (std::exception& e) { if (init.showErrors) std::cerr << e.what() << '\n'; }
    }
}

}  // namespace

int main(int argc, char** argv)
{
    // The HIP runtime's "direct dispatch" (the calling thread writes the queue packets itself) made the query phase of this program -- many
    // worker threads, eight streams -- run at one of two speeds from process to process: hipMemcpyAsync and the kernel launches of a batch
    // took 0.2 or 3-10 ms (150 Gbp: 81-130 or 300-480 ms per 10^7 reads, the kernels' time the same).  With the runtime's own
    // submission threads every run is the fast kind.  The runtime reads the switch when it starts, i.e. before main: the program starts
    // itself again with it set (once; a value the user has set is left alone).
    if (!std::getenv("AMD_DIRECT_DISPATCH") && !std::getenv("MCQ_NO_REEXEC")) {
        setenv("AMD_DIRECT_DISPATCH", "0", 0);
        setenv("MCQ_NO_REEXEC", "1", 1);
        execv("/proc/self/exe", argv);
        // (no /proc: go on as we are)
    }
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        const std::vector<std::string> args(argv + std::min(argc, 2), argv + argc);
        if (mode == "query") {
            if (args.empty()) throw std::runtime_error("usage: mcq query <database> [<sequence file/directory>...] [options]");
            Options init;
            init.db = args[0];
            init = parse(std::vector<std::string>(args.begin() + 1, args.end()), init);
            Session S;
            return query_main(S, init);
        }
        if (mode == "merge") return merge_main(args);
        if (mode == "info") return info_main(args);
        if (mode == "build") {                                                  // main_mode_build, mode_build.cpp:93-106, :41-66
            using clock = std::chrono::steady_clock;
            std::vector<std::string> none;
            const BuildOptions bo = parse_build(args, false, none);
            const bool info = bo.info != BuildOptions::silent;
            if (info) std::cout << "Building new database '" << bo.dbfile << "' from reference sequences." << std::endl;
            const auto t0 = clock::now();
            BuiltDatabase db;
            build_database(bo, db);
            const double btime = std::chrono::duration<double>(clock::now() - t0).count();
            db.write();
            const double total = std::chrono::duration<double>(clock::now() - t0).count();
            if (info)
                std::cout << "------------------------------------------------\n"
                          << "Construction time: " << btime << " s\n"
                          << "Writing time:      " << total - btime << " s\n"
                          << "Total build time:  " << total << " s" << std::endl;
            return 0;
        }
        if (mode == "modify") {                                                 // main_mode_modify, mode_build.cpp:74-88
            std::vector<std::string> none;
            BuildOptions bo = parse_build(args, false, none);
            std::cout << "Modify database " << bo.dbfile << std::endl;
            {   // get_modify_options (options.cpp:765-795): sketching and limits of the database file are the defaults
                mc_ctx* meta = nullptr;
                if (mc_open_metadata(bo.dbfile.c_str(), &meta) != MC_OK) throw std::runtime_error(mc_last_error(nullptr));
                uint64_t info[8]; mc_db_info(meta, info);
                mc_destroy(meta);
                bo.k = (uint32_t)info[0]; bo.s = (uint32_t)info[1]; bo.w = (uint32_t)info[2]; bo.stride = (uint32_t)info[3];
                if (!bo.maxLocsGiven) bo.maxLocs = (int)info[4];
                std::ifstream is(bo.dbfile + ".meta", std::ios::binary);
                char hdr[10] = {};
                is.read(hdr, 10);
                bo.targetIdBytes = (is.gcount() == 10 && (uint8_t)hdr[9] == 2) ? 2 : 4;
            }
            bo.modify = true;
            // the reference parses the command line twice (options.cpp:772 and :786) and its value list grows both times: every new
            // file is added twice, the second copy's sequences under '<id>!1' names
            { const std::vector<std::string> once = bo.infiles; bo.infiles.insert(bo.infiles.end(), once.begin(), once.end()); }
            if (bo.info != BuildOptions::silent) std::cout << "Adding reference sequences to database..." << std::endl;
            BuiltDatabase db;
            build_database(bo, db);
            db.write();
            return 0;
        }
        if (mode == "build+query") {                                            // main_mode_build_query, mode_build_query.cpp:41-94
            std::vector<std::string> qargs;
            const BuildOptions bo = parse_build(args, true, qargs);
            if (bo.info != BuildOptions::silent) std::cout << "Building new database from reference sequences." << std::endl;
            BuiltDatabase db;
            build_database(bo, db);
            Options init;
            init.db = "<built in memory>";
            init = parse(qargs, init);
            int rc = 0;
            {
                Session S;
                S.built = &db;
                rc = query_main(S, init);
            }
            if (bo.saveDb) db.write();
            return rc;
        }
        throw std::runtime_error("usage: mcq query|build|modify|build+query|merge|info ... (see the header of mcq_main.cpp / mcq_build.h)");
    } catch (std::exception& e) {
        std::cerr << "ABORT: " << e.what() << "!" << std::endl;                  // main.cpp:65-68
        return 1;
    }
    return 0;
}
