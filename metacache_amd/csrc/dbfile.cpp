// metacache_amd/csrc/dbfile.cpp -- reads the reference's database files (<name>.meta + <name>.cache<p>)
// and feeds them through the loading ABI.  File layout: SURVEY.md §8a row 11
//   .meta  : database.cpp:247-290 (write) / :87-163 (read); taxonomy block taxonomy.hpp:702-728, :322-341
//   .cache : hash_multimap.hpp:1037-1082 (serialize) / :970-1030 (deserialize)
#include "context.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <memory>
#include <thread>
#include <unordered_map>

using namespace mcamd;

namespace {

struct File {
    FILE* f = nullptr;
    explicit File(const std::string& n) : f(std::fopen(n.c_str(), "rb")) {}
    ~File() { if (f) std::fclose(f); }
    bool rd(void* p, size_t n) { return std::fread(p, 1, n, f) == n; }
    bool str(std::string& s)
    {
        uint64_t n = 0;
        if (!rd(&n, 8) || n > (1ull << 30)) return false;
        s.resize(n);
        return n == 0 || rd(&s[0], n);
    }
};

struct Meta {
    uint32_t k, s, w, stride;
    uint64_t maxLocs, targetCount;
    uint32_t numParts, targetBytes;
    std::vector<Taxon> taxa;
};

int read_meta(const std::string& name, Meta& m, std::string& err)
{
    File f(name + ".meta");
    if (!f.f) { err = "Could not read database metadata file '" + name + ".meta'"; return MC_ERR_IO; }
    uint64_t ver = 0;
    uint8_t wd[7];
    if (!f.rd(&ver, 8) || !f.rd(wd, 7)) { err = "truncated .meta"; return MC_ERR_IO; }
    if (ver != 20200820ull) { err = "database version " + std::to_string(ver) + " is incompatible (need 20200820)"; return MC_ERR_UNSUPPORTED; }
    // sizeof{feature, target_id, window_id, bucket_size, part_id, taxon_id}, num_ranks (database.cpp:110-137)
    if (wd[0] != 4 || (wd[1] != 2 && wd[1] != 4) || wd[2] != 4 || wd[3] != 1 || wd[4] != 4 || wd[5] != 8 || wd[6] != MC_NUM_RANKS) {
        err = "database uses data type sizes this build does not support";
        return MC_ERR_UNSUPPORTED;
    }
    m.targetBytes = wd[1];
    uint64_t sk[8];
    if (!f.rd(sk, 64)) { err = "truncated .meta"; return MC_ERR_IO; }     // sketching options are stored twice
    m.k = (uint32_t)std::min<uint64_t>(sk[4], 16); m.s = (uint32_t)sk[5]; m.w = (uint32_t)sk[6]; m.stride = (uint32_t)sk[7];
    if (!f.rd(&m.maxLocs, 8)) { err = "truncated .meta"; return MC_ERR_IO; }
    m.maxLocs = std::min<uint64_t>(std::max<uint64_t>(m.maxLocs, 1), 254);   // host_hashmap.hpp:454-466
    if (m.targetBytes == 2) { uint16_t t; if (!f.rd(&t, 2)) return MC_ERR_IO; m.targetCount = t; }
    else { uint32_t t; if (!f.rd(&t, 4)) return MC_ERR_IO; m.targetCount = t; }
    if (!f.rd(&m.numParts, 4)) { err = "truncated .meta"; return MC_ERR_IO; }
    uint64_t nt = 0;
    if (!f.rd(&nt, 8)) { err = "truncated .meta"; return MC_ERR_IO; }
    m.taxa.resize(nt);
    for (auto& t : m.taxa) {
        if (!f.rd(&t.id, 8) || !f.rd(&t.parent, 8) || !f.rd(&t.rank, 1) || !f.str(t.name) || !f.str(t.filename) ||
            !f.rd(&t.index, 8) || !f.rd(&t.windows, 8)) { err = "truncated taxonomy block"; return MC_ERR_IO; }
    }
    return MC_OK;
}

// ranked lineage of every target as taxon index + 1 (taxonomy.hpp:576-597, :919-1030)
void make_lineages(const Meta& m, std::vector<uint32_t>& lin)
{
    std::unordered_map<int64_t, uint32_t> byId;
    byId.reserve(m.taxa.size() * 2);
    for (uint32_t i = 0; i < m.taxa.size(); ++i) byId.emplace(m.taxa[i].id, i);
    lin.assign(m.targetCount * MC_NUM_RANKS, 0);
    for (uint64_t t = 0; t < m.targetCount; ++t) {
        auto it = byId.find(-(int64_t)t - 1);
        if (it == byId.end()) continue;
        uint32_t* L = &lin[t * MC_NUM_RANKS];
        const Taxon& tx = m.taxa[it->second];
        if (tx.rank < MC_NUM_RANKS) L[tx.rank] = it->second + 1;
        int64_t id = tx.parent;
        while (id > 0) {                                   // parents live among the non-target taxa
            auto p = byId.find(id);
            if (p == byId.end()) break;
            const Taxon& px = m.taxa[p->second];
            if (px.rank < MC_NUM_RANKS) L[px.rank] = p->second + 1;
            if (px.parent == id) break;
            id = px.parent;
        }
    }
}

struct PartHeader { uint64_t nkeys = 0, nvalues = 0, batch = 0; };

// a .cache header is only believed as far as the file's size allows: every key costs at least 5 bytes (u32 key + u8 size), every
// value 4 + targetBytes; batch 0 with keys to read would never advance (hash_multimap.hpp:970-1030 loops the same way)
int read_part_header(mc_ctx* ctx, const std::string& fname, PartHeader& h, uint32_t targetBytes)
{
    File f(fname);
    if (!f.f) { ctx->err = "Could not read database file '" + fname + "'"; return MC_ERR_IO; }
    if (!f.rd(&h.nkeys, 8) || !f.rd(&h.nvalues, 8) || !f.rd(&h.batch, 8)) { ctx->err = "truncated " + fname; return MC_ERR_IO; }
    std::fseek(f.f, 0, SEEK_END);
    const long long size = ftello(f.f);
    const unsigned long long body = size > 24 ? (unsigned long long)size - 24 : 0ull;
    if ((h.nkeys && h.batch == 0) || h.nkeys > body / 5 || h.nvalues > body / (4 + targetBytes) ||
        h.nkeys * 5 + h.nvalues * (4 + targetBytes) > body) {
        ctx->err = "corrupt header in " + fname + " (key / value counts do not fit the file)";
        return MC_ERR_IO;
    }
    return MC_OK;
}

int load_part(mc_ctx* ctx, uint32_t part, const std::string& fname, uint32_t targetBytes)
{
    // a single-part context: reader threads, pinned slabs, copies and table kernels overlapped (dbload.cpp); MC_LOAD_PIPELINE=0: the
    // sequential loop below (what multi-part contexts, whose buckets are merged on the host, always take)
    static const bool pipelined = [] { const char* e = std::getenv("MC_LOAD_PIPELINE"); return !(e && e[0] == '0'); }();
    if (pipelined && ctx->parts.size() == 1 && part == 0) {
        uint64_t st[4] = {0, 0, 0, 0};
        const int rc = load_file_pipelined(ctx, fname, targetBytes, st);
        if (rc) return rc;
        for (int i = 0; i < 4; ++i) ctx->loadStats[i] += st[i];
        return mc_load_end(ctx, part);
    }
    File f(fname);
    if (!f.f) { ctx->err = "Could not read database file '" + fname + "'"; return MC_ERR_IO; }
    uint64_t nkeys = 0, nvalues = 0, batch = 0;
    if (!f.rd(&nkeys, 8) || !f.rd(&nvalues, 8) || !f.rd(&batch, 8)) { ctx->err = "truncated " + fname; return MC_ERR_IO; }
    if (nkeys && batch == 0) { ctx->err = "corrupt header in " + fname; return MC_ERR_IO; }      // validated by read_part_header before
    // the file's layout IS its batch size (keys[batch] | sizes[batch] | values, hash_multimap.hpp:898-912): reading with another one would
    // take key bytes for sizes.  The reference and this repository's writer use 2^20; anything beyond 2^26 is refused, not reinterpreted.
    if (batch > (1ull << 26)) { ctx->err = "unsupported batch size in " + fname + " (header says " + std::to_string(batch) + " keys per batch)"; return MC_ERR_UNSUPPORTED; }
    int rc;
    const size_t vb = 4 + targetBytes;
    std::vector<uint32_t> keys(std::min<uint64_t>(batch, nkeys));
    std::vector<uint8_t> sizes(keys.size());
    std::vector<uint8_t> vals;
    for (uint64_t done = 0; done < nkeys;) {
        const uint64_t nb = std::min<uint64_t>(batch, nkeys - done);
        if (!f.rd(keys.data(), nb * 4) || !f.rd(sizes.data(), nb)) { ctx->err = "truncated " + fname; return MC_ERR_IO; }
        uint64_t bv = 0;
        for (uint64_t i = 0; i < nb; ++i) bv += sizes[i];
        vals.resize(bv * vb + 8);
        if (bv && !f.rd(vals.data(), bv * vb)) { ctx->err = "truncated " + fname; return MC_ERR_IO; }
        if (ctx->cfg.target_shard_count > 1) cut_batch_to_target_range(ctx, sizes.data(), vals.data(), (uint32_t)nb, targetBytes);
        if ((rc = mc_load_batch(ctx, part, keys.data(), sizes.data(), vals.data(), nb))) return rc;
        done += nb;
    }
    return mc_load_end(ctx, part);
}

}  // namespace

extern "C" {

int mc_open_database(const char* name, const mc_config* cfgIn, mc_ctx** out)
{
    if (!name || !cfgIn || !out) return MC_ERR_INVALID;
    *out = nullptr;
    Meta m{};
    std::string err;
    const bool trace = std::getenv("MC_LOAD_TRACE") != nullptr;
    auto now_s = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
    const double tr0 = now_s();
    int rc = read_meta(name, m, err);
    if (rc) { set_global_error(err); return rc; }
    const double tr1 = now_s();
    mc_config cfg = *cfgIn;
    cfg.kmerlen = m.k;                                      // k always comes from the DB (querying.cpp:232-243)
    if (!cfg.sketchlen) cfg.sketchlen = m.s;
    if (!cfg.winlen) cfg.winlen = m.w;
    if (!cfg.winstride) cfg.winstride = cfg.winlen - m.k + 1;   // NOT the database's stride: adapt_options_to_database, querying.cpp:237-239
    cfg.target_id_bytes = m.targetBytes;
    if (m.numParts < 1 || m.numParts > 255) { set_global_error("unsupported number of database parts"); return MC_ERR_UNSUPPORTED; }
    cfg.num_parts = m.numParts;
    // read_database (mode_query.cpp:69-76): the removal limit never exceeds (the DB's own bucket cap - 1)
    if (cfg.remove_overpopulated && m.maxLocs > 1) cfg.remove_overpopulated = std::min<uint32_t>(cfg.remove_overpopulated, (uint32_t)m.maxLocs - 1);
    uint32_t firstPart = 0;
    if (cfg.single_part >= 0) {
        if ((uint32_t)cfg.single_part >= m.numParts) { set_global_error("database part is not available"); return MC_ERR_INVALID; }
        firstPart = (uint32_t)cfg.single_part; cfg.num_parts = 1;
    }
    mc_ctx* ctx = nullptr;
    // Single-part databases whose windows can be numbered in 32 bits get the compact location store (4 bytes per location: global
    // window numbers, DeviceTable::values32); every target's window count comes from the target metadata (taxonomy.hpp:264-280
    // file_source::windows; targets carry the ids -(target) - 1, taxonomy.hpp:930).  Should a file hold a location outside of what
    // its own metadata says, the load is repeated with 8-byte locations.
    double trCreate = 0, trBegin = 0, trLoad = 0;
    bool compactRefused = false;
    uint64_t exactPlain = 0, exactPadded = 0, exactKeys = 0;   // Mode T: a first load found its store or its table too small and counted them
    for (int attempt = 0; attempt < 3; ++attempt) {
        const double tc0 = now_s();
        if ((rc = mc_create(&cfg, &ctx))) return rc;
        trCreate += now_s() - tc0;
        ctx->targetSketch = SketchParams{m.k, m.s, m.w, m.stride};
        ctx->targetCount = m.targetCount;
        ctx->maxLocs = cfg.max_locations_per_feature ? std::min<uint64_t>(m.maxLocs, cfg.max_locations_per_feature) : m.maxLocs;
        ctx->taxa = std::move(m.taxa);
        m.taxa.clear();
        bool tryCompact = !compactRefused && cfg.num_parts == 1 && m.targetCount > 0 && m.targetCount < 0xFFFFFFFFull;
        if (tryCompact) {
            std::vector<uint32_t> windows((size_t)m.targetCount, 0u);
            uint64_t seen = 0;
            for (const auto& t : ctx->taxa) {
                if (t.id >= 0) continue;
                const uint64_t tgt = (uint64_t)(-(t.id + 1));
                if (tgt >= m.targetCount || t.windows > 0xFFFFFFF0ull) { tryCompact = false; break; }
                windows[(size_t)tgt] = (uint32_t)t.windows; ++seen;
            }
            tryCompact = tryCompact && seen == m.targetCount;
            if (tryCompact) rc = mc_load_target_windows(ctx, windows.data(), windows.size());
        }
        if (cfg.target_shard_count > 1) {
            // Mode T: the contiguous target range of this context, cut where the windows split into equal shares (metacache_amd.h)
            std::vector<uint64_t> win((size_t)m.targetCount, 1);
            for (const auto& t : ctx->taxa)
                if (t.id < 0 && (uint64_t)(-(t.id + 1)) < m.targetCount) win[(size_t)(-(t.id + 1))] = std::max<uint64_t>(t.windows, 1);
            uint64_t total = 0;
            for (uint64_t w : win) total += w;
            const uint64_t R = cfg.target_shard_count, me = cfg.target_shard_index;
            uint64_t before = 0, mine = 0;
            uint32_t lo = (uint32_t)m.targetCount, hi = 0;
            for (uint64_t t = 0; t < m.targetCount; ++t) {
                const uint64_t r = (uint64_t)(((unsigned __int128)before * R) / std::max<uint64_t>(total, 1));
                if (r == me) { lo = std::min<uint32_t>(lo, (uint32_t)t); hi = (uint32_t)t + 1; mine += win[(size_t)t]; }
                before += win[(size_t)t];
            }
            if (hi == 0) lo = 0;                                // (more ranges than targets: an empty one)
            ctx->tgtLo = lo; ctx->tgtHi = hi;
            ctx->tgtShare = total ? (double)mine / (double)total : 1.0;
            ctx->tgtRangeSet = true;
            ctx->tgtExactPlain = exactPlain; ctx->tgtExactPadded = exactPadded; ctx->tgtExactKeys = exactKeys;
        }
        // every part is announced first (the merged table is sized for all of them), then loaded in part order
        const double tb0 = now_s();
        for (uint32_t p = 0; p < cfg.num_parts && !rc; ++p) {
            PartHeader h;
            rc = read_part_header(ctx, std::string(name) + ".cache" + std::to_string(firstPart + p), h, m.targetBytes);
            if (!rc) rc = mc_load_begin(ctx, p, h.nkeys, h.nvalues);
        }
        const double tb1 = now_s();
        // the slot pipes' workspaces are allocated beside the file load (context.cpp size_pipe: allocations of memory that another process
        // has just given back take 0.1-0.2 s apiece; inside the first batches they stall every stream)
        std::thread reserve;
        if (!rc && cfg.num_slots > 0 && !std::getenv("MC_NO_RESERVE")) {
            uint64_t locs = 0, keys = 0;
            for (const auto& P : ctx->parts) { locs += P.expectValues; keys += P.expectKeys; }
            ctx->reserveByLoader.store(true, std::memory_order_release);
            reserve = std::thread([ctx, locs, keys] { (void)reserve_slot_pipes(ctx, locs, keys); });   // (a failure here is not one: the first batch asks again)
        }
        for (uint32_t p = 0; p < cfg.num_parts && !rc; ++p)
            rc = load_part(ctx, p, std::string(name) + ".cache" + std::to_string(firstPart + p), m.targetBytes);
        ctx->loadSettled.store(true, std::memory_order_release);
        if (reserve.joinable()) reserve.join();
        trBegin += tb1 - tb0; trLoad += now_s() - tb1;
        if (rc && tryCompact && ctx->locRangeViolated && attempt < 2) {
            compactRefused = true;
            m.taxa = std::move(ctx->taxa);
            mc_destroy(ctx); ctx = nullptr; rc = 0;
            continue;
        }
        if (rc && ctx->storeShort && !exactPlain && attempt < 2) {
            exactPlain = ctx->tgtExactPlain; exactPadded = ctx->tgtExactPadded; exactKeys = ctx->tgtExactKeys;
            m.taxa = std::move(ctx->taxa);
            mc_destroy(ctx); ctx = nullptr; rc = 0;
            continue;
        }
        break;
    }
    if (rc) {
        set_global_error(ctx->err);
        mc_destroy(ctx);
        return rc;
    }
    const double tr2 = now_s();
    Meta tmp{}; tmp.targetCount = ctx->targetCount; tmp.taxa = ctx->taxa;
    std::vector<uint32_t> lin;
    make_lineages(tmp, lin);
    if ((rc = mc_set_lineages(ctx, lin.data(), ctx->targetCount))) { set_global_error(ctx->err); mc_destroy(ctx); return rc; }
    if (trace) std::fprintf(stderr, "mc_open_database %s part %d: metadata %.3f s, mc_create %.3f s, table allocation %.3f s, files %.3f s, lineages %.3f s\n", name, cfg.single_part, tr1 - tr0,
                            trCreate, trBegin, trLoad, now_s() - tr2);
    *out = ctx;
    return MC_OK;
}

// database::read with scope::metadata_only (database.cpp:183-242; `info` mode, mode_info.cpp): header, sketching, taxa, lineages --
// no table, no device.  Every query call on such a context fails with MC_ERR_STATE.
int mc_open_metadata(const char* name, mc_ctx** out)
{
    if (!name || !out) return MC_ERR_INVALID;
    *out = nullptr;
    Meta m{};
    std::string err;
    int rc = read_meta(name, m, err);
    if (rc) { set_global_error(err); return rc; }
    auto* ctx = new mc_ctx;
    mc_config_default(&ctx->cfg);
    ctx->cfg.kmerlen = m.k; ctx->cfg.sketchlen = m.s; ctx->cfg.winlen = m.w; ctx->cfg.winstride = m.stride;
    ctx->cfg.target_id_bytes = m.targetBytes; ctx->cfg.num_parts = m.numParts;
    ctx->querySketch = ctx->targetSketch = SketchParams{m.k, m.s, m.w, m.stride};
    ctx->targetCount = m.targetCount;
    ctx->maxLocs = m.maxLocs;
    ctx->parts.resize(std::max<uint32_t>(m.numParts, 1));
    ctx->taxa = std::move(m.taxa);
    Meta tmp{}; tmp.targetCount = ctx->targetCount; tmp.taxa = ctx->taxa;
    std::vector<uint32_t> lin;
    make_lineages(tmp, lin);
    ctx->lineages = std::move(lin);
    *out = ctx;
    return MC_OK;
}

int mc_load_stats(const mc_ctx* ctx, uint64_t stats[4])
{
    if (!ctx || !stats) return MC_ERR_INVALID;
    for (int i = 0; i < 4; ++i) stats[i] = ctx->loadStats[i];
    return MC_OK;
}

int mc_db_num_taxa(const mc_ctx* ctx, uint64_t* n)
{
    if (!ctx || !n) return MC_ERR_INVALID;
    *n = ctx->taxa.size();
    return MC_OK;
}

int mc_db_taxon(const mc_ctx* ctx, uint64_t i, int64_t* id, int64_t* parent, uint32_t* rank, const char** name)
{
    if (!ctx || i >= ctx->taxa.size()) return MC_ERR_INVALID;
    const Taxon& t = ctx->taxa[i];
    if (id) *id = t.id;
    if (parent) *parent = t.parent;
    if (rank) *rank = t.rank;
    if (name) *name = t.name.c_str();
    return MC_OK;
}

int mc_db_taxon_source(const mc_ctx* ctx, uint64_t i, const char** filename, uint64_t* index, uint64_t* windows)
{
    if (!ctx || i >= ctx->taxa.size()) return MC_ERR_INVALID;
    const Taxon& t = ctx->taxa[i];
    if (filename) *filename = t.filename.c_str();
    if (index) *index = t.index;
    if (windows) *windows = t.windows;
    return MC_OK;
}

int mc_db_lineages(const mc_ctx* ctx, const uint32_t** lin, uint64_t* nt)
{
    if (!ctx || !lin || !nt) return MC_ERR_INVALID;
    *lin = ctx->lineages.data();
    *nt = ctx->lineages.size() / MC_NUM_RANKS;
    return MC_OK;
}

}  // extern "C"
