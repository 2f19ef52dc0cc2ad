// tools/slot_driver.cpp -- MEASUREMENT TOOL (round 6): T host threads drive the slot API of libmetacache_amd.so the way the reference's consumer
// threads do (database_query.hpp:185-252: fill a batch, submit, wait, walk through the results, clear), without an interpreter between
// the calls.  Built as libmcslotdrv.so by metacache_amd/build.py; tools/slot_path_bench.py loads it beside the product library and hands
// it the context.  Reads: one byte array, read i = seqs + i * read_len.
#include "metacache_amd.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

extern "C" int mc_slot_drive(mc_ctx* ctx, const char* seqs, uint64_t num_reads, uint32_t read_len, uint32_t batch, uint32_t threads, double seconds,
                             const mc_candidate* want, uint64_t want_reads, uint64_t out[4])
{
    std::vector<uint64_t> offs(batch + 1);
    for (uint32_t i = 0; i <= batch; ++i) offs[i] = (uint64_t)i * read_len;
    const uint64_t nb = num_reads / batch;
    if (!nb) return MC_ERR_INVALID;
    std::atomic<uint64_t> done{0}, bad{0}, failed{0};
    const auto t0 = std::chrono::steady_clock::now();
    const auto stop = t0 + std::chrono::duration<double>(seconds);
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            uint64_t i = t, mine = 0;
            while (std::chrono::steady_clock::now() < stop) {
                const uint64_t b = i % nb;
                if (mc_batch_add_bulk(ctx, t, seqs + b * batch * read_len, offs.data(), batch, 0) != (int64_t)batch) { ++failed; return; }
                mc_results r{};
                if (mc_batch_submit(ctx, t, 0) != MC_OK || mc_batch_wait(ctx, t, &r) != MC_OK) { ++failed; return; }
                // the consumer looks at every read's candidates (classification happens here in the reference)
                uint64_t sum = 0;
                for (uint32_t q = 0; q < r.num_queries; ++q) sum += r.cands[(uint64_t)q * r.max_candidates].hits;
                if (sum == 0x7fffffffffffull) ++bad;
                if (want && (b + 1) * batch <= want_reads && mine < 4)
                    if (std::memcmp(r.cands, want + b * batch * r.max_candidates, (size_t)batch * r.max_candidates * sizeof(mc_candidate)) != 0) ++bad;
                if (mc_batch_clear(ctx, t) != MC_OK) { ++failed; return; }
                ++done; ++mine;
                i += threads;
            }
        });
    for (auto& x : th) x.join();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out[0] = done; out[1] = bad; out[2] = failed; out[3] = (uint64_t)(el * 1e6);
    return failed ? MC_ERR_STATE : MC_OK;
}
