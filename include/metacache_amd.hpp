// metacache_amd.hpp -- header-only C++14 mirror of the reference's query-side interface on top of the C ABI
// (metacache_amd.h).  Same names and argument meaning as muellan/metacache so that the caller code of
// database_query.hpp:87-124 (query_gpu) carries over almost verbatim:
//
//   mc_amd::database db;  db.read("refseq");                                   // database::read        database.cpp:183-242
//   mc_amd::query_batch batch(db, numWorkers);                                  // query_batch ctor      database_query.hpp:192-202
//   auto rules = mc_amd::make_candidate_generation_rules(q, opt, db.target_sketching().winstride);
//   batch.add_paired_read(hostId, q.seq1, q.seq2, rules);                       // query_batch.cuh:383-391
//   db.query_gpu_async(batch, hostId, lowestRank);                              // database.hpp:386-397
//   batch.host_data(hostId).wait_for_results();                                 // query_batch.cu:147-152
//   batch.host_data(hostId).allhits(i) / top_candidates(i) / clear()            // query_batch.cuh:212-259
//
// Errors: the reference throws std::runtime_error (caught in main.cpp:65-77); so does this wrapper.
#ifndef METACACHE_AMD_HPP_
#define METACACHE_AMD_HPP_

#include "metacache_amd.h"

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace mc_amd {

using target_id = std::uint32_t;
using window_id = std::uint32_t;

enum class taxon_rank : int {                       // taxonomy.hpp:68-91
    Sequence = 0, Form, Variety, subSpecies, Species, subGenus, Genus, subTribe, Tribe, subFamily, Family, subOrder, Order,
    subClass, Class, subPhylum, Phylum, subKingdom, Kingdom, Domain, root, none
};

struct location { window_id win; target_id tgt; };                              // database.hpp:136-166 (widened)
struct window_range { window_id beg = 0, end = 0; };                             // candidate_structs.hpp:42-71
struct match_candidate {                                                          // candidate_structs.hpp:80-104 (tax resolved by the caller)
    target_id tgt; std::uint32_t hits; window_range pos;
};
struct candidate_generation_rules {                                               // candidate_structs.hpp:113-125
    window_id maxWindowsInRange = 3;
    std::size_t maxCandidates = 2;
    taxon_rank mergeBelow = taxon_rank::Sequence;
};
struct sketching_opt { std::uint32_t kmerlen = 16, sketchlen = 16, winlen = 127, winstride = 112; };

template <class T>
struct span {                                                                     // span.hpp
    const T* first = nullptr; const T* last = nullptr;
    const T* begin() const noexcept { return first; }
    const T* end() const noexcept { return last; }
    std::size_t size() const noexcept { return std::size_t(last - first); }
    const T& operator[](std::size_t i) const noexcept { return first[i]; }
};

// candidate_structs.hpp:134-151
template <class Query, class ClassificationOptions>
candidate_generation_rules make_candidate_generation_rules(const Query& query, const ClassificationOptions& opt, std::uint32_t targetWindowStride)
{
    candidate_generation_rules rules;
    rules.maxWindowsInRange = window_id(2 + (std::max<std::size_t>(query.seq1.size() + query.seq2.size(), opt.insertSizeMax) / targetWindowStride));
    rules.mergeBelow = taxon_rank(int(opt.lowestRank));
    rules.maxCandidates = opt.maxNumCandidatesPerQuery;
    return rules;
}

class query_batch;

class database {
public:
    database() = default;
    database(const database&) = delete;
    database& operator=(const database&) = delete;
    ~database() { if (ctx_) mc_destroy(ctx_); }

    // database::read(filename, singlePartId, replication, scope, info); cfg carries the query-time options
    void read(const std::string& filename, int singlePartId = -1, const mc_config* cfg = nullptr)
    {
        mc_config c;
        if (cfg) c = *cfg; else { mc_config_default(&c); c.kmerlen = c.sketchlen = c.winlen = c.winstride = 0; }
        c.single_part = singlePartId;
        if (mc_open_database(filename.c_str(), &c, &ctx_) != MC_OK) throw std::runtime_error(mc_last_error(nullptr));
        std::uint64_t info[8];
        mc_db_info(ctx_, info);
        sk_.kmerlen = std::uint32_t(info[0]); sk_.sketchlen = std::uint32_t(info[1]);
        sk_.winlen = std::uint32_t(info[2]); sk_.winstride = std::uint32_t(info[3]);
        targets_ = info[5]; parts_ = unsigned(info[6]);
        maxCand_ = c.max_candidates; slots_ = c.num_slots; copyAllhits_ = c.copy_allhits != 0;
    }
    const sketching_opt& target_sketching() const noexcept { return sk_; }
    std::uint64_t target_count() const noexcept { return targets_; }
    unsigned part_count() const noexcept { return parts_; }
    mc_ctx* handle() const noexcept { return ctx_; }

    // database::query_gpu_async(queryBatch, hostId, querySketching, lowestRank)  database.hpp:386-397
    void query_gpu_async(query_batch& batch, unsigned hostId, taxon_rank lowestRank) const;

private:
    friend class query_batch;
    mc_ctx* ctx_ = nullptr;
    sketching_opt sk_;
    std::uint64_t targets_ = 0;
    unsigned parts_ = 0, maxCand_ = 2, slots_ = 1;
    bool copyAllhits_ = false;
};

class query_batch {
public:
    class query_host_data {                                                      // query_batch.cuh:60-280
    public:
        std::size_t num_queries() const noexcept { return res_.num_queries; }
        void wait_for_results()                                                  // query_batch.cu:147-152
        {
            if (mc_batch_wait(ctx_, slot_, &res_) != MC_OK) throw std::runtime_error(mc_last_error(ctx_));
            const std::size_t n = res_.num_queries, K = res_.max_candidates;
            tops_.resize(n * K);
            for (std::size_t i = 0; i < n * K; ++i) {
                const mc_candidate& c = res_.cands[i];
                tops_[i].tgt = c.tgt; tops_[i].hits = c.hits; tops_[i].pos.beg = c.beg; tops_[i].pos.end = c.end;
            }
        }
        span<location> allhits(std::size_t i) const noexcept                      // query_batch.cuh:212-221
        {
            span<location> s;
            if (res_.hits) {
                s.first = reinterpret_cast<const location*>(res_.hits) + res_.hit_offsets[i];
                s.last = reinterpret_cast<const location*>(res_.hits) + res_.hit_offsets[i + 1];
            }
            return s;
        }
        span<match_candidate> top_candidates(std::size_t i) const noexcept        // query_batch.cuh:223-231; unused entries: hits == 0
        {
            const std::size_t K = res_.max_candidates;
            span<match_candidate> s;
            s.first = tops_.data() + i * K; s.last = s.first + K;
            return s;
        }
        void clear() { mc_batch_clear(ctx_, slot_); res_ = mc_results{}; }        // query_batch.cuh:255-259
    private:
        friend class query_batch;
        mc_ctx* ctx_ = nullptr; std::uint32_t slot_ = 0;
        mc_results res_{};
        std::vector<match_candidate> tops_;
    };

    query_batch(const database& db, unsigned numHostThreads) : ctx_(db.ctx_), hosts_(numHostThreads)
    {
        if (numHostThreads > db.slots_) throw std::runtime_error("query_batch: more host threads than slots (mc_config.num_slots)");
        for (unsigned i = 0; i < numHostThreads; ++i) { hosts_[i].ctx_ = ctx_; hosts_[i].slot_ = i; }
    }
    // returns false if the batch is full (submit, wait, clear, then add again)      query_batch.cuh:383-391
    template <class Sequence>
    bool add_paired_read(unsigned hostId, const Sequence& seq1, const Sequence& seq2, const candidate_generation_rules& rules)
    {
        const int rc = mc_batch_add(ctx_, hostId, seq1.data(), std::uint32_t(seq1.size()), seq2.data(), std::uint32_t(seq2.size()),
                                    rules.maxWindowsInRange);
        if (rc < 0) throw std::runtime_error(mc_last_error(ctx_));
        return rc == MC_OK;
    }
    query_host_data& host_data(unsigned hostId) noexcept { return hosts_[hostId]; }

private:
    friend class database;
    mc_ctx* ctx_;
    std::vector<query_host_data> hosts_;
};

inline void database::query_gpu_async(query_batch& batch, unsigned hostId, taxon_rank lowestRank) const
{
    (void)batch;
    if (mc_batch_submit(ctx_, hostId, int(lowestRank)) != MC_OK) throw std::runtime_error(mc_last_error(ctx_));
}

}  // namespace mc_amd

#endif
