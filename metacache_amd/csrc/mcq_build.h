// mcq_build.h -- `metacache build` (and the build half of `build+query`) on MI355X: reference sequences -> database, host C++14 above
// the C ABI's builder (mc_build_*: sketching, sorting and bucketising run on the GPU, builder.hip).  SURVEY.md §8f rank 1.  Mirrors
//   option handling       options.cpp:251-265 (database name), :298-370 (info level, id format, taxonomy), :375-485 (sketching, storage),
//                         :490-520 (augment_taxonomy_options), :535-640 (build mode), :1498-1575 (build+query: -targets / -query / -save-db)
//   taxonomy dumps        taxonomy_io.cpp:55-175 (names.dmp, merged.dmp, nodes.dmp -> non-target taxa), taxonomy.hpp:182-221 (rank names)
//   sequence -> taxon id  taxonomy_io.cpp:180-318 (assembly_summary tables), building.cpp:80-150 (*.accession2taxid after the build),
//                         :196-232 (try_to_rank_unranked_targets), :240-262 (find_taxon_id)
//   adding targets        building.cpp:283-327 (one sequence), :335-455 (all files; here: files and records in the given order = the
//                         reference's single-part build), database.cpp:36-82 (ids, names of duplicates), sequence_io.cpp:470-673 (ids)
//   post-processing       building.cpp:516-534 (-remove-overpopulated-features), :550-566 (-remove-ambig-features)
// Not offered:
// -parts > 1 (the reference spreads targets over parts in thread-schedule order; the multi-GPU modes use key shards instead).
#ifndef MCQ_BUILD_H_
#define MCQ_BUILD_H_
#include "mcq_common.h"

#include <set>

namespace mcq {

enum class IdType { smart, ncbi, genbank, filename, leading_word };

inline std::string trimmed(std::string s)
{
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}

inline std::string ncbi_accession_number(const std::string& text)             // sequence_io.cpp:541-573
{
    if (text.empty()) return "";
    std::smatch m;
    std::regex_search(text, m, accession_regex());
    return m[2];
}

inline std::string genbank_identifier(const std::string& text)                // sequence_io.cpp:581-603
{
    if (text.empty()) return "";
    auto i = text.find("gi|");
    if (i == std::string::npos) i = text.find("gi:");
    if (i == std::string::npos) i = text.find("gi=");
    if (i == std::string::npos) return "";
    i += 3;
    auto j = text.find('|', i);
    if (j == std::string::npos) { j = text.find(' ', i); if (j == std::string::npos) j = text.size(); }
    return trimmed(text.substr(i, j - i));
}

inline std::string accession_string(const std::string& text, IdType t)        // extract_accession_string, sequence_io.cpp:608-642
{
    if (text.empty()) return "";
    switch (t) {
        case IdType::ncbi: return ncbi_accession_number(text);
        case IdType::genbank: return genbank_identifier(text);
        case IdType::leading_word: return leading_word(text);
        case IdType::filename: return filename_without_extension(text);
        case IdType::smart: {
            auto s = ncbi_accession_number(text);
            if (!s.empty()) return s;
            s = genbank_identifier(text);
            if (!s.empty()) return s;
            s = filename_without_extension(text);
            if (!s.empty()) return s;
        }
    }
    return text;
}

struct BuildOptions {
    std::string dbfile;
    std::vector<std::string> infiles;
    std::string taxPath;
    std::vector<std::string> mapPost;               // -taxpostmap + the defaults of augment_taxonomy_options
    IdType idType = IdType::smart;
    enum Info { silent, moderate, verbose } info = moderate;
    uint32_t k = 16, s = 16, w = 127, stride = 0;   // options.hpp:102
    bool resetParents = false, removeOverpopulated = false, saveDb = false;
    int maxLocs = -1, parts = 1, targetIdBytes = 4;
    int shards = 0;                                 // key shards of the build (0 = from the input size: one device sort takes 2^32 pairs)
    float maxLoadFac = -1;
    int removeAmbigRank = kNumRanks;
    int maxAmbig = 1;                                        // options.hpp:78
    bool modify = false, maxLocsGiven = false;      // modify mode: the database <dbfile> is read first (mode_build.cpp:74-88)
};

// args: everything after the mode word.  build: <database> <files>... ; build+query: -targets <files>... [-query <files>...] and every
// word this parser does not know goes to the query parser (queryArgs).
inline BuildOptions parse_build(const std::vector<std::string>& args, bool buildQuery, std::vector<std::string>& queryArgs)
{
    BuildOptions o;
    bool haveDb = buildQuery;
    auto need = [&](size_t& i) -> std::string { if (i + 1 >= args.size()) throw std::runtime_error("value missing after '" + args[i] + "'"); return args[++i]; };
    auto values = [&](size_t& i, std::vector<std::string>& dst) { while (i + 1 < args.size() && !args[i + 1].empty() && args[i + 1][0] != '-') dst.push_back(args[++i]); };
    auto set_db = [&](std::string name) {                                      // sanitize_database_name, options.cpp:110-128
        auto pos = name.find(".meta");
        if (pos != std::string::npos) name.erase(pos);
        else { pos = name.find(".cache"); if (pos != std::string::npos) name.erase(pos); }
        o.dbfile = name;
    };
    for (size_t i = 0; i < args.size(); ++i) {
        const std::string& a = args[i];
        if (a.empty()) continue;
        if (a[0] != '-') {
            if (!haveDb) { set_db(a); haveDb = true; }
            else if (!buildQuery) o.infiles.push_back(a);
            else queryArgs.push_back(a);
            continue;
        }
        if (a == "-targets" && buildQuery) values(i, o.infiles);
        else if (a == "-query" && buildQuery) values(i, queryArgs);
        else if (a == "-save-db" && buildQuery) { set_db(need(i)); o.saveDb = true; }
        else if (a == "-taxonomy") o.taxPath = need(i);
        else if (a == "-taxpostmap") values(i, o.mapPost);
        else if (a == "-sequence-id-format") {
            const std::string v = need(i);
            if (v == "smart") o.idType = IdType::smart; else if (v == "ncbi") o.idType = IdType::ncbi;
            else if (v == "gi") o.idType = IdType::genbank; else if (v == "filename") o.idType = IdType::filename;
            else if (v == "leadingword") o.idType = IdType::leading_word;
            else throw std::runtime_error("unknown sequence id format '" + v + "'");
        }
        else if (a == "-silent") o.info = BuildOptions::silent;
        else if (a == "-verbose") o.info = BuildOptions::verbose;
        else if (a == "-kmerlen") o.k = (uint32_t)std::stoul(need(i));
        else if (a == "-sketchlen") o.s = (uint32_t)std::stoul(need(i));
        else if (a == "-winlen") o.w = (uint32_t)std::stoul(need(i));
        else if (a == "-winstride") o.stride = (uint32_t)std::stoul(need(i));
        else if (a == "-reset-taxa" || a == "-reset-parents") o.resetParents = true;
        else if (a == "-max-locations-per-feature") { o.maxLocs = std::stoi(need(i)); o.maxLocsGiven = true; }
        else if (a == "-remove-overpopulated-features") o.removeOverpopulated = true;
        else if (a == "-remove-ambig-features") { o.removeAmbigRank = rank_from_name(need(i)); if (o.removeAmbigRank < 0) throw std::runtime_error("unknown rank"); }
        else if (a == "-max-ambig-per-feature") o.maxAmbig = std::stoi(need(i));
        else if (a == "-max-load-fac" || a == "-max-load-factor") o.maxLoadFac = std::stof(need(i));
        else if (a == "-parts") o.parts = std::stoi(need(i));
        else if (a == "-max-part-size") (void)need(i);
        else if (a == "-target-id-type") {                                    // the reference's compile-time MC_TARGET_ID_TYPE (config.hpp:57-61)
            const std::string v = need(i);
            if (v == "uint16_t" || v == "16") o.targetIdBytes = 2; else if (v == "uint32_t" || v == "32") o.targetIdBytes = 4;
            else throw std::runtime_error("target id type must be uint16_t or uint32_t");
        }
        else if (a == "-build-shards") o.shards = std::stoi(need(i));          // not a reference option: see BuildOptions::shards
        else if (a == "-threads" && !buildQuery) (void)need(i);               // accepted: sketching and sorting run on the GPU
        else if (buildQuery) queryArgs.push_back(a);
        else throw std::runtime_error("unknown option '" + a + "'");
    }
    if (!buildQuery && o.dbfile.empty()) throw std::runtime_error("Database name is missing");
    // process_build_options (options.cpp:614-626)
    {
        std::vector<std::string> expanded;
        for (const auto& name : o.infiles) {
            auto sub = files_in_directory(name);
            if (sub.empty()) expanded.push_back(name); else expanded.insert(expanded.end(), sub.begin(), sub.end());
        }
        o.infiles.swap(expanded);
    }
    if (o.infiles.empty()) throw std::runtime_error("No reference sequence files provided or found");
    if (o.maxLocs < 0) o.maxLocs = 254;                                       // database::max_supported_locations_per_feature()
    if (o.stride == 0) o.stride = o.w - o.k + 1;
    if (o.parts > 1) throw std::runtime_error("-parts > 1 is not available: one part per build (see DESIGN.md, multi-GPU modes)");
    // augment_taxonomy_options (options.cpp:490-520)
    if (!o.taxPath.empty() && o.taxPath.back() != '/') o.taxPath += '/';
    for (const char* f : {"nucl_gb.accession2taxid", "nucl_wgs.accession2taxid", "nucl_est.accession2taxid", "nucl_gss.accession2taxid"})
        o.mapPost.push_back(o.taxPath + f);
    for (const auto& f : files_in_directory(o.taxPath))
        if (f.find(".accession2taxid") != std::string::npos && std::find(o.mapPost.begin(), o.mapPost.end(), f) == o.mapPost.end()) o.mapPost.push_back(f);
    return o;
}

// ---- taxonomy dumps (taxonomy_io.cpp:55-175): taxa above sequence level; the first record of an id wins (unordered_set::emplace) ----
struct TaxTree {
    std::vector<Taxon> taxa;
    std::unordered_map<int64_t, uint32_t> byId;
    bool emplace(int64_t id, int64_t parent, std::string name, int rank)
    {
        if (id == 0 || byId.count(id)) return false;
        Taxon t; t.id = id; t.parent = parent; t.rank = rank; t.name = std::move(name);
        byId.emplace(id, (uint32_t)taxa.size());
        taxa.push_back(std::move(t));
        return true;
    }
};

// fields of one "a\t|\tb\t|\tc\t|" row
inline std::vector<std::string> dump_fields(const std::string& line)
{
    std::vector<std::string> f;
    size_t p = 0;
    while (p <= line.size()) {
        size_t q = line.find('|', p);
        if (q == std::string::npos) q = line.size();
        std::string x = line.substr(p, q - p);
        while (!x.empty() && x.front() == '\t') x.erase(x.begin());
        while (!x.empty() && (x.back() == '\t' || x.back() == '\r')) x.pop_back();
        f.push_back(std::move(x));
        p = q + 1;
    }
    return f;
}

inline TaxTree read_taxonomy_dumps(const BuildOptions& o)
{
    const bool info = o.info != BuildOptions::silent;
    TaxTree tax;
    std::map<int64_t, std::string> names;
    {
        std::ifstream is(o.taxPath + "names.dmp");
        if (is.good()) {
            if (info) std::cout << "Reading taxon names ... " << std::flush;
            int64_t lastId = 0;
            for (std::string line; std::getline(is, line);) {
                auto f = dump_fields(line);
                if (f.size() < 4 || f[0].empty()) continue;
                int64_t id = 0;
                try { id = std::stoll(f[0]); } catch (std::exception&) { continue; }
                if (id == lastId) continue;                                    // a scientific name was already taken for this id
                if (leading_word(f[3]).find("scientific") != std::string::npos) { lastId = id; names.emplace(id, f[1]); }
            }
            if (info) std::cout << "done." << std::endl;
        } else if (info) std::cerr << "Could not read taxon names file " << o.taxPath << "names.dmp; continuing with ids only." << std::endl;
    }
    std::map<int64_t, int64_t> merged;
    {
        std::ifstream is(o.taxPath + "merged.dmp");
        if (is.good()) {
            if (info) std::cout << "Reading taxonomic node mergers ... " << std::flush;
            for (std::string line; std::getline(is, line);) {
                auto f = dump_fields(line);
                if (f.size() < 2 || f[0].empty()) continue;
                try {
                    const int64_t oldId = std::stoll(f[0]), newId = std::stoll(f[1]);
                    merged.emplace(oldId, newId);
                    tax.emplace(oldId, newId, "", kNumRanks);
                } catch (std::exception&) {}
            }
            if (info) std::cout << "done." << std::endl;
        }
    }
    {
        std::ifstream is(o.taxPath + "nodes.dmp");
        if (!is.good()) {
            if (info) std::cerr << "Could not read taxonomic nodes file " << o.taxPath << "nodes.dmp" << std::endl;
            return tax;
        }
        if (info) std::cout << "Reading taxonomic tree ... " << std::flush;
        for (std::string line; std::getline(is, line);) {
            auto f = dump_fields(line);
            if (f.size() < 3 || f[0].empty()) continue;
            int64_t id = 0, parent = 0;
            try { id = std::stoll(f[0]); parent = std::stoll(f[1]); } catch (std::exception&) { continue; }
            auto it = names.find(id);
            std::string name = it != names.end() ? it->second : std::string("--");
            if (name.empty()) name = "<" + std::to_string(id) + ">";
            auto mi = merged.find(id);
            if (mi != merged.end()) id = mi->second;
            mi = merged.find(parent);
            if (mi != merged.end()) parent = mi->second;
            tax.emplace(id, parent, std::move(name), rank_from_dump_name(f[2]));
        }
        if (info) std::cout << tax.taxa.size() << " taxa read." << std::endl;
    }
    auto root = tax.byId.find(1);
    if (root != tax.byId.end()) tax.taxa[root->second].rank = kNumRanks - 1;   // reset_rank(1, root)
    return tax;
}

// ---- sequence id -> taxon id tables (taxonomy_io.cpp:180-318) ----
inline void read_id_table(const std::string& file, std::map<std::string, int64_t>& map, bool info)
{
    std::ifstream is(file);
    if (!is.good()) return;
    if (info) std::cout << "Reading sequence to taxon mappings from " << file << std::endl;
    std::vector<std::string> lines;
    for (std::string l; std::getline(is, l);) lines.push_back(std::move(l));
    // header row = the last of the leading '#' lines (at most 10 are looked at)
    int headerRow = 0;
    for (int i = 0; i < 10; ++i, ++headerRow) { if (i >= (int)lines.size() || lines[i].empty() || lines[i][0] != '#') break; }
    if (headerRow > 0) --headerRow;
    if (headerRow >= (int)lines.size()) return;
    int keycol = 0, taxcol = 0;
    {
        std::istringstream hs(lines[headerRow]);
        int col = 0;
        for (std::string h; hs >> h; ++col) {
            if (h.size() == 1 && h[0] == '#') hs >> h;
            if (h == "taxid") taxcol = col;
            else if (h == "accession.version" || h == "assembly_accession") keycol = col;
        }
    }
    size_t first = (size_t)headerRow + 1;
    if (taxcol < 1) { taxcol = 1; first = 0; }                                // no header: 1st column = key, 2nd = taxon id, from the top
    for (size_t r = first; r < lines.size(); ++r) {
        // the reference moves 'keycol' tabs forward, reads the key, then 'taxcol' MORE tabs forward (taxonomy_io.cpp:271-277)
        const std::string& l = lines[r];
        size_t p = 0;
        bool ok = true;
        for (int i = 0; i < keycol && ok; ++i) { auto t = l.find('\t', p); if (t == std::string::npos) ok = false; else p = t + 1; }
        if (!ok) continue;
        const std::string key = leading_word(l.substr(p));
        size_t q = l.find(key, p) + key.size();
        for (int i = 0; i < taxcol && ok; ++i) { auto t = l.find('\t', q); if (t == std::string::npos) ok = false; else q = t + 1; }
        if (!ok || key.empty()) continue;
        int64_t id = 0;
        try { id = std::stoll(l.substr(q)); } catch (std::exception&) { break; }   // a failed extraction ends the reference's loop
        map.emplace(key, id);
    }
}

inline int64_t find_taxon_id(const std::map<std::string, int64_t>& m, const std::string& name)   // building.cpp:240-262
{
    if (m.empty() || name.empty()) return 0;
    auto i = m.find(name);
    if (i != m.end()) return i->second;
    i = m.upper_bound(name);
    if (i == m.end()) return 0;
    if (i->first.compare(0, name.size(), name) != 0) return 0;
    return i->second;
}

struct BuilderError : std::runtime_error { using std::runtime_error::runtime_error; };   // device / capacity failures: fatal

// ---- the built database: builder handle (device arrays) + taxonomy records (non-target taxa, then one taxon per target) ----
struct BuiltDatabase {
    std::vector<mc_builder*> bs;                    // one builder, or one per key shard (all are given every target)
    BuildOptions opt;
    std::vector<Taxon> nonTarget;
    std::vector<Taxon> targets;                     // id = -(target) - 1, rank sequence
    ~BuiltDatabase() { for (mc_builder* b : bs) mc_build_free(b); }

    void write() const                              // write_database, building.cpp:546-566
    {
        const bool info = opt.info != BuildOptions::silent;
        if (info) std::cout << "------------------------------------------------\nWriting database to file ... " << std::endl;
        std::vector<mc_taxon_rec> recs(nonTarget.size());
        for (size_t i = 0; i < nonTarget.size(); ++i) recs[i] = mc_taxon_rec{nonTarget[i].id, nonTarget[i].parent, (uint32_t)nonTarget[i].rank, nonTarget[i].name.c_str()};
        if (mc_build_write_shards(const_cast<mc_builder**>(bs.data()), (uint32_t)bs.size(), opt.dbfile.c_str(), recs.data(), recs.size()) != MC_OK) {
            if (info) std::cout << "FAIL" << std::endl;
            std::cerr << "Could not write database file!\n";
            return;
        }
        if (info) std::cout << "Completed database writing." << std::endl;
    }
};

inline void build_database(const BuildOptions& o, BuiltDatabase& db)
{
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    const bool info = o.info != BuildOptions::silent;
    db.opt = o;
    // prepare_database (building.cpp:462-508)
    if (info) std::cout << "Max locations per feature set to " << o.maxLocs << std::endl;
    if (o.maxLoadFac > 0.4f && o.maxLoadFac < 0.99f && info) std::cout << "Using custom hash table load factor of " << o.maxLoadFac << std::endl;
    if (!o.taxPath.empty()) {
        TaxTree t = read_taxonomy_dumps(o);
        db.nonTarget = std::move(t.taxa);
        if (info) std::cout << "Taxonomy applied to database." << std::endl;
    }
    if (db.nonTarget.empty() && info)
        std::cout << "The datbase doesn't contain a taxonomic hierarchy yet.\nYou can add one or update later via:\n"
                     "   metacache modify <database> -taxonomy <directory>" << std::endl;
    if (o.removeAmbigRank != kNumRanks && info) {                                // building.cpp:508-518
        if (db.nonTarget.size() > 1) std::cout << "Ambiguous features on rank " << kRankNames[o.removeAmbigRank] << " will be removed afterwards." << std::endl;
        else std::cout << "Could not determine amiguous features due to missing taxonomic information." << std::endl;
    }

    mc_config c; mc_config_default(&c);
    c.kmerlen = o.k; c.sketchlen = o.s; c.winlen = o.w; c.winstride = o.stride;
    c.target_id_bytes = (uint32_t)o.targetIdBytes;
    c.max_locations_per_feature = (uint32_t)std::max(1, std::min(o.maxLocs, 254));
    c.remove_overpopulated = o.removeOverpopulated ? 1 : 0;
    if (o.maxLoadFac > 0.4f && o.maxLoadFac < 0.99f) c.max_load_factor = o.maxLoadFac;
    // one device sort takes 2^32 (feature, location) pairs: larger inputs are built in key shards (every shard sketches every target
    // and keeps its share of the features; mc_build_finish_shards / mc_build_write_shards put them together)
    uint32_t shards = o.shards > 0 ? (uint32_t)o.shards : 1u;
    if (o.shards <= 0) {
        uint64_t bytes = 0;
        for (const auto& f : o.infiles) { struct stat st; if (stat(f.c_str(), &st) == 0) bytes += (uint64_t)st.st_size * (f.size() > 3 && f.compare(f.size() - 3, 3, ".gz") == 0 ? 4 : 1); }
        uint64_t pairs = bytes / std::max<uint32_t>(o.stride, 1) * o.s;
        if (o.modify) {                                                        // + the locations the database already holds
            std::ifstream is(o.dbfile + ".cache0", std::ios::binary);
            uint64_t hdr[3] = {0, 0, 0};
            if (is.read(reinterpret_cast<char*>(hdr), 24)) pairs += hdr[1];
        }
        shards = (uint32_t)(pairs / 3000000000ull) + 1;
    }
    for (uint32_t i = 0; i < shards; ++i) {
        mc_builder* b = nullptr;
        c.key_shard_index = i; c.key_shard_count = shards;
        if (mc_build_begin(&c, &b) != MC_OK) throw std::runtime_error(mc_last_error(nullptr));
        db.bs.push_back(b);
    }
    if (info && shards > 1) std::cout << "Building in " << shards << " key shards." << std::endl;

    // make_sequence_to_taxon_id_map (taxonomy_io.cpp:290-318)
    std::map<std::string, int64_t> seq2tax;
    {
        std::set<std::string> dirs;                                            // unique_directories, filesys_utility.cpp:82-92
        for (const auto& f : o.infiles) dirs.insert(f.substr(0, f.find_last_of("/\\")));
        for (const auto& d : dirs) read_id_table(d + "/assembly_summary.txt", seq2tax, info);
        for (const char* f : {"assembly_summary_refseq.txt", "assembly_summary_refseq_historical.txt", "assembly_summary_genbank.txt",
                              "assembly_summary_genbank_historical.txt"})
            read_id_table(o.taxPath + f, seq2tax, info);
    }
    if (info) {
        const char* idn[] = {"smart", "ncbi", "gi", "filename", "leadingword"};
        std::cout << "Sequence ID extraction method: " << idn[(int)o.idType] << "\nProcessing reference sequences." << std::endl;
    }
    // add_targets_to_database (building.cpp:335-455) with one consumer: files in the given order, records in file order
    std::map<std::string, uint32_t> name2tgt;                                  // taxonomy_cache::name2tax_ (taxonomy.hpp:1135-1160)
    if (o.modify) {
        // make_database(opt.dbfile) (mode_build.cpp:80): targets with their sources, the taxa above them unless -taxonomy replaces them
        // (reset_taxa_above_sequence_level, building.cpp:487-495), and the location lists of the (single) part
        mc_ctx* meta = nullptr;
        if (mc_open_metadata(o.dbfile.c_str(), &meta) != MC_OK) throw std::runtime_error(mc_last_error(nullptr));
        struct Guard { mc_ctx* c; ~Guard() { mc_destroy(c); } } guard{meta};
        uint64_t dbi[8]; mc_db_info(meta, dbi);
        if (dbi[6] != 1) throw std::runtime_error("modify: databases of more than one part are not supported");
        uint64_t nt = 0; mc_db_num_taxa(meta, &nt);
        std::vector<Taxon> old(dbi[5]); std::vector<std::string> oldFile(dbi[5]); std::vector<uint64_t> oldIndex(dbi[5]);
        const bool keepTaxa = o.taxPath.empty();
        for (uint64_t i = 0; i < nt; ++i) {
            Taxon t; uint32_t rk; const char *nm, *fn; uint64_t idx = 0;
            mc_db_taxon(meta, i, &t.id, &t.parent, &rk, &nm);
            t.rank = int(rk); t.name = nm;
            mc_db_taxon_source(meta, i, &fn, &idx, &t.windows);
            if (t.id < 0 && t.rank == 0) {
                const uint64_t tgt = (uint64_t)(-t.id - 1);
                if (tgt >= old.size()) throw std::runtime_error("modify: target id beyond the database's target count");
                oldFile[tgt] = fn; oldIndex[tgt] = idx; old[tgt] = std::move(t);
            }
            else if (keepTaxa) db.nonTarget.push_back(std::move(t));
        }
        for (uint64_t t = 0; t < old.size(); ++t) {
            for (mc_builder* b : db.bs)
                if (mc_build_add_existing_target(b, old[t].name.c_str(), old[t].parent, oldFile[t].c_str(), oldIndex[t], old[t].windows) != MC_OK)
                    throw BuilderError(mc_build_last_error(b));
            name2tgt.emplace(old[t].name, (uint32_t)t);
            db.targets.push_back(std::move(old[t]));
        }
        std::ifstream is(o.dbfile + ".cache0", std::ios::binary);
        uint64_t hdr[3] = {0, 0, 0};
        if (!is.read(reinterpret_cast<char*>(hdr), 24)) throw std::runtime_error("Could not read database file '" + o.dbfile + ".cache0'");
        const size_t vb = 4 + (size_t)o.targetIdBytes;
        std::vector<uint32_t> keys; std::vector<uint8_t> sizes; std::vector<char> vals;
        for (uint64_t done = 0; done < hdr[0];) {
            const uint64_t nb = std::min<uint64_t>(hdr[2], hdr[0] - done);
            keys.resize(nb); sizes.resize(nb);
            is.read(reinterpret_cast<char*>(keys.data()), nb * 4);
            is.read(reinterpret_cast<char*>(sizes.data()), nb);
            uint64_t bv = 0;
            for (uint64_t i = 0; i < nb; ++i) bv += sizes[i];
            vals.resize(bv * vb + 8);
            is.read(vals.data(), bv * vb);
            if (!is) throw std::runtime_error("truncated " + o.dbfile + ".cache0");
            for (mc_builder* b : db.bs)
                if (mc_build_add_locations(b, keys.data(), sizes.data(), vals.data(), nb, (uint32_t)o.targetIdBytes) != MC_OK)
                    throw BuilderError(mc_build_last_error(b));
            done += nb;
        }
        if (info) std::cout << "Database holds " << db.targets.size() << " reference sequences." << std::endl;
    }
    const size_t initTargets = db.targets.size();
    for (size_t fi = 0; fi < o.infiles.size(); ++fi) {
        const std::string& filename = o.infiles[fi];
        if (o.info == BuildOptions::verbose) std::cerr << "  (" << fi << '/' << o.infiles.size() << ") " << filename << std::endl;
        try {
            std::string fileAcc = accession_string(filename, o.idType);
            int64_t fileTax = find_taxon_id(seq2tax, fileAcc);
            if (fileTax == 0 && o.idType == IdType::smart) { fileAcc = accession_string(filename, IdType::filename); fileTax = find_taxon_id(seq2tax, fileAcc); }
            if (o.info == BuildOptions::verbose) std::cerr << "      accession '" << fileAcc << "' -> taxid " << fileTax << std::endl;
            SeqFile file(filename);
            file.index(std::max(1u, std::thread::hardware_concurrency()));
            std::string scratch;
            for (size_t r = 0; r < file.records(); ++r) {
                View h, s;
                file.record(r, h, s, scratch);
                if (s.empty()) continue;                                       // building.cpp:295
                const std::string header(h.p, h.n);
                std::string seqId = accession_string(header, o.idType);
                if (seqId.empty()) seqId = header;
                int64_t parent = fileTax;
                if (parent == 0) parent = find_taxon_id(seq2tax, seqId);
                if (parent == 0) parent = taxon_id_in_header(header);
                if (parent < 1) parent = 0;                                    // database.cpp:58
                // a name that is already taken gets "!1", "!1!2", ... appended until it is new (taxonomy.hpp:1141-1146)
                std::string sid = seqId;
                int dupl = 0;
                while (name2tgt.find(sid) != name2tgt.end()) { ++dupl; sid += "!" + std::to_string(dupl); }
                const uint32_t tgt = (uint32_t)db.targets.size();
                for (mc_builder* b : db.bs)
                    if (mc_build_add_target_src(b, s.p, s.n, sid.c_str(), parent, filename.c_str(), r) != MC_OK)
                        throw BuilderError(mc_build_last_error(b));
                if (dupl > 0 && info)
                    std::cerr << "Warning: duplicate sequence id! '" << seqId << "' already in database - '" << filename << "/" << r
                              << "' inserted as '" << sid << "'\n";
                name2tgt.emplace(sid, tgt);
                Taxon t; t.id = -(int64_t)tgt - 1; t.parent = parent; t.rank = 0; t.name = sid;
                mc_build_target_windows(db.bs[0], tgt, &t.windows);
                db.targets.push_back(std::move(t));
                if (o.info == BuildOptions::verbose) {
                    std::cerr << "    P0  [" << seqId;
                    if (parent > 0) std::cerr << ":" << parent;
                    std::cerr << "]  " << s.n << " bp" << (dupl > 0 ? "  --  not added to database!\n" : "\n");
                }
            }
        } catch (BuilderError&) {
            throw;
        } catch (std::exception& e) {                                          // unreadable file: the reference reports it only with -verbose
            if (o.info == BuildOptions::verbose) std::cerr << "FAIL: " << e.what() << '\n';
        }
    }
    for (mc_builder* b : db.bs) if (mc_build_finish(b, nullptr) != MC_OK) throw std::runtime_error(mc_build_last_error(b));
    if (info)
        std::cout << "Added " << db.targets.size() - initTargets << " reference sequences in "
                  << std::chrono::duration<double>(clock::now() - t0).count() << " s" << std::endl;

    // try_to_rank_unranked_targets (building.cpp:196-232)
    {
        std::set<uint32_t> unranked;
        for (uint32_t t = 0; t < db.targets.size(); ++t) if (o.resetParents || db.targets[t].parent == 0) unranked.insert(t);
        if (!unranked.empty()) {
            if (info) std::cout << unranked.size() << " targets are unranked (no taxon was assigned)." << std::endl;
            for (const auto& file : o.mapPost) {
                // rank_targets_with_mapping_file (building.cpp:80-150): rows "accession accession.version taxid gi" after one header line
                std::ifstream is(file);
                if (!is.good()) continue;
                if (info) std::cout << "Try to map sequences to taxa using '" << file << "'" << std::endl;
                std::string acc, accver, gi; uint64_t taxid = 0;
                std::getline(is, acc); acc.clear();
                auto with_name = [&](const std::string& n) -> int64_t { if (n.empty()) return -1; auto i = name2tgt.find(n); return i == name2tgt.end() ? -1 : (int64_t)i->second; };
                auto with_similar = [&](const std::string& n) -> int64_t {
                    if (n.empty()) return -1;
                    auto i = name2tgt.upper_bound(n);
                    if (i == name2tgt.end() || i->first.compare(0, n.size(), n) != 0) return -1;
                    return (int64_t)i->second; };
                while (is >> acc >> accver >> taxid >> gi) {
                    int64_t t = with_name(accver);
                    if (t < 0) { t = with_similar(acc); if (t < 0) t = with_name(gi); }
                    if (t >= 0) {
                        auto u = unranked.find((uint32_t)t);
                        if (u != unranked.end()) {
                            db.targets[t].parent = (int64_t)taxid;
                            for (mc_builder* b : db.bs) mc_build_set_parent(b, (uint64_t)t, (int64_t)taxid);
                            unranked.erase(u);
                            if (unranked.empty()) break;
                        }
                    }
                }
                if (unranked.empty()) break;
            }
        }
        size_t still = 0;
        for (const auto& t : db.targets) if (t.parent == 0) ++still;
        if (info) {
            if (!still) std::cout << "All targets are ranked (have a taxon assigned)." << std::endl;
            else std::cout << still << " targets remain unranked (no taxon was assigned)." << std::endl;
        }
    }
    // post_process_features (building.cpp:550-566): -remove-ambig-features, after the targets got their final parents
    if (o.removeAmbigRank != kNumRanks && db.nonTarget.size() > 1) {
        if (info) std::cout << "\nRemoving ambiguous features on rank " << kRankNames[o.removeAmbigRank] << "... " << std::flush;
        std::unordered_map<int64_t, uint32_t> byId;
        for (uint32_t i = 0; i < db.nonTarget.size(); ++i) byId.emplace(db.nonTarget[i].id, i);
        // the target's entry of that rank in its ranked lineage (taxonomy::make_ranks, taxonomy.hpp:576-597): the last taxon of the
        // rank on the way to the root; 0 = none
        std::vector<uint32_t> anc(db.targets.size(), 0);
        for (uint32_t t = 0; t < db.targets.size(); ++t) {
            if (o.removeAmbigRank == 0) { anc[t] = t + 1; continue; }
            int64_t id = db.targets[t].parent;
            while (id != 0) {
                auto it = byId.find(id);
                if (it == byId.end()) break;
                const Taxon& p = db.nonTarget[it->second];
                if (p.rank == o.removeAmbigRank) anc[t] = it->second + 1;
                if (p.parent == id) break;
                id = p.parent;
            }
        }
        uint64_t old = 0, rem = 0;
        for (mc_builder* b : db.bs) {
            uint64_t k = 0, v = 0, r = 0;
            mc_build_counts(b, &k, &v); old += k;
            if (mc_build_remove_ambiguous(b, anc.data(), anc.size(), (uint32_t)std::max(0, o.maxAmbig), &r) != MC_OK) throw std::runtime_error(mc_build_last_error(b));
            rem += r;
        }
        if (info) std::cout << rem << " of " << old << "." << std::endl;
    }
}

}  // namespace mcq
#endif
