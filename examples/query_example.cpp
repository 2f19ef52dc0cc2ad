// examples/query_example.cpp -- the reference's query_gpu loop (database_query.hpp:87-124) on metacache_amd.hpp.
//   g++ -std=c++14 -Iinclude examples/query_example.cpp -Lmetacache_amd/lib -lmetacache_amd -o query_example
//   ./query_example <database> <file with one sequence per line> [lowest rank as number]
// prints per query:  <index> TAB <tgt>:<hits>:<beg>-<end>,...   (top candidates, sequence-level ids)
#include "metacache_amd.hpp"

#include <fstream>
#include <iostream>
#include <string>
#include <vector>

struct sequence_query { std::string header, seq1, seq2; };                       // database_query.hpp:45-72
struct classification_options { int lowestRank = 0; std::size_t insertSizeMax = 0, maxNumCandidatesPerQuery = 2; };

int main(int argc, char** argv)
{
    if (argc < 3) { std::cerr << "usage: query_example <database> <sequences.txt> [lowest]\n"; return 2; }
    try {
        classification_options opt;
        if (argc > 3) opt.lowestRank = std::stoi(argv[3]);
        mc_amd::database db;
        db.read(argv[1]);
        mc_amd::query_batch batch(db, 1);
        std::vector<sequence_query> all;
        { std::ifstream is(argv[2]); std::string line; while (std::getline(is, line)) all.push_back({"q", line, ""}); }

        std::size_t done = 0;
        auto flush = [&](std::size_t upto) {
            db.query_gpu_async(batch, 0, mc_amd::taxon_rank(opt.lowestRank));
            auto& host = batch.host_data(0);
            host.wait_for_results();
            for (std::size_t s = 0; s < host.num_queries(); ++s) {
                std::cout << (done + s) << '\t';
                for (const auto& c : host.top_candidates(s)) {
                    if (c.hits == 0) break;                                   // printing.cpp:291
                    std::cout << c.tgt << ':' << c.hits << ':' << c.pos.beg << '-' << c.pos.end << ',';
                }
                std::cout << '\n';
            }
            host.clear();
            done = upto;
        };
        for (std::size_t i = 0; i < all.size(); ++i) {
            auto rules = mc_amd::make_candidate_generation_rules(all[i], opt, db.target_sketching().winstride);
            if (!batch.add_paired_read(0, all[i].seq1, all[i].seq2, rules)) {
                flush(i);
                if (!batch.add_paired_read(0, all[i].seq1, all[i].seq2, rules))
                    std::cerr << "query batch is too small for a single read!\n";     // database_query.hpp:101-105
            }
        }
        flush(all.size());
    } catch (std::exception& e) {
        std::cerr << "ABORT: " << e.what() << "!" << std::endl;                  // main.cpp:65-68
        return 1;
    }
    return 0;
}
