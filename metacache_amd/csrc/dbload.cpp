// metacache_amd/csrc/dbload.cpp -- a single-part database file into the device table at the speed of the host's memory and the PCIe
// link: reader THREADS fill pinned slabs with the file's batches (pread of disjoint ranges), one feeder copies them to the device on a
// copy stream and launches the table kernels (table_build.hip) behind the copies on the context's stream -- reads, H2D copies and insert
// kernels of different batches overlap, and nothing on the way waits for the device (the one number the old loader fetched from it per
// batch, how many locations the batch adds to the location store, is a pure function of the batch's keys and sizes: the readers compute it).
//
// What it replaces: the reference reads every part of a database in its own thread (database.cpp:203-226, std::async per part) with
// hash_multimap::deserialize's sequential loop inside (hash_multimap.hpp:970-1030); round 3's loader here was that loop: one fread into
// pageable memory, a blocking copy, three kernels, two synchronisations per batch -- 6.8 GB/s from /dev/shm.
//
// File layout (hash_multimap.hpp:1037-1082): {nkeys, nvalues, batch} u64, then per batch keys[nb] u32 | sizes[nb] u8 | values[sum sizes]
// of (4 + target_id_bytes) bytes each: where batch i + 1 begins follows from batch i's sizes, so an INDEX PASS reads the sizes of all
// batches first (1 byte per key: 1-2 % of the file) and the readers then know every batch's place.
#include "context.h"
#include "devcache.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

using namespace mcamd;

namespace {

// ---- pinned slabs, kept between loads (part groups load part after part; pinning memory costs about as much as reading it) ---------
struct Slab { uint8_t* p = nullptr; size_t cap = 0; };
struct SlabPool {
    std::mutex mtx;
    std::vector<Slab> idle;
    size_t idleBytes = 0;
    static constexpr size_t kKeep = 3ull << 30;              // cached at most
    Slab get(size_t need)
    {
        {
            std::lock_guard<std::mutex> l(mtx);
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].cap >= need && idle[i].cap <= 2 * need + (1u << 20)) { Slab s = idle[i]; idle.erase(idle.begin() + i); idleBytes -= s.cap; return s; }
        }
        Slab s;
        if (hipHostMalloc((void**)&s.p, need, hipHostMallocPortable) != hipSuccess) { s.p = nullptr; return s; }
        s.cap = need;
        return s;
    }
    void put(Slab s)
    {
        if (!s.p) return;
        {
            std::lock_guard<std::mutex> l(mtx);
            if (idleBytes + s.cap <= kKeep) { idle.push_back(s); idleBytes += s.cap; return; }
        }
        (void)hipHostFree(s.p);
    }
};
SlabPool& pool() { static SlabPool p; return p; }

bool pread_all(int fd, void* dst, size_t n, uint64_t off)
{
    uint8_t* d = static_cast<uint8_t*>(dst);
    while (n) {
        const ssize_t r = ::pread(fd, d, std::min<size_t>(n, 1u << 30), (off_t)off);
        if (r <= 0) return false;
        d += r; n -= (size_t)r; off += (uint64_t)r;
    }
    return true;
}

struct BatchPlace { uint64_t off = 0, fileVals = 0; uint32_t nkeys = 0; };   // byte offset of the batch's keys in the file

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// effective_size of table_build.hip on the host: what a key of `fileSize` locations adds to the location store
inline uint32_t stored_of(uint32_t key, uint32_t fileSize, const LoadFilter& lf)
{
    uint32_t size = fileSize;
    if (lf.rmOver && size > lf.rmOver) size = 0;
    if (lf.maxLocs && size > lf.maxLocs) size = lf.maxLocs;
    if (lf.shardCnt > 1 && key_owner(key, lf.shardCnt) != lf.shardIdx) size = 0;
    return list_alloc(size, lf.align);
}

}  // namespace

// One chunk (<= 2^22 keys, < 2^32 file values) whose arrays are in device memory, WITHOUT any synchronisation: `stored` (what the chunk adds
// to the location store) comes from the host.  The scratch arrays are reused chunk after chunk: the stream's order protects them.
int mcamd::load_chunk_device_async(mc_ctx* ctx, const uint32_t* dkeys, const uint8_t* dsizes, const uint8_t* dvals, uint32_t nb, uint64_t fileVals, uint64_t stored)
{
    Part& P = ctx->parts[0];
    auto fail = [&](int code, const char* msg) { ctx->err = msg; return code; };
    if (P.keysLoaded + nb > P.expectKeys) return fail(MC_ERR_INVALID, "database file: more keys than its header announces");
    if (fileVals >= (1ull << 32)) return fail(MC_ERR_INVALID, "database file: a chunk holds 2^32 or more locations");
    if (mcamd::allocate_buckets(ctx, 0) != MC_OK) return MC_ERR_NOMEM;
    if (mcamd::allocate_values(ctx) != MC_OK) return fail(MC_ERR_NOMEM, "database load: cannot allocate the location store");
    if (P.valuesStored + stored > P.dvaluesCap) return fail(MC_ERR_INVALID, "database file: more values than its header announces");
    const uint32_t tb = ctx->cfg.target_id_bytes;
    const LoadFilter lf{ctx->cfg.max_locations_per_feature, ctx->cfg.remove_overpopulated, ctx->cfg.key_shard_index, ctx->cfg.key_shard_count, ctx->parts[0].listAlign};
    hipStream_t st = ctx->stream;
    auto ensure = [&](DevBuf& b, size_t bytes) -> bool {
        if (bytes <= b.cap) return true;
        // (growing a scratch array the stream may still be reading: drain it first -- happens a few times per load, the sizes settle at once)
        if (b.p) { (void)hipStreamSynchronize(st); (void)hipFree(b.p); }
        b.p = nullptr; b.cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        if (mcamd::dev_malloc(&b.p, want) != hipSuccess) return false;
        b.cap = want;
        return true;
    };
    if (!ensure(ctx->bLdFileSz, (size_t)nb * 4) || !ensure(ctx->bLdStoreSz, (size_t)nb * 4) || !ensure(ctx->bLdFileOff, (size_t)(nb + 2) * 4) ||
        !ensure(ctx->bLdStoreOff, (size_t)(nb + 2) * 4) || !ensure(ctx->bLdScan, scan_tmp_bytes(nb + 1)))
        return fail(MC_ERR_NOMEM, "database load: cannot allocate the table-build scratch");
    auto* fileSz = (uint32_t*)ctx->bLdFileSz.p; auto* storeSz = (uint32_t*)ctx->bLdStoreSz.p;
    auto* fileOff = (uint32_t*)ctx->bLdFileOff.p; auto* storeOff = (uint32_t*)ctx->bLdStoreOff.p;
    auto* counters = (unsigned long long*)ctx->bLdCounters.p;
    launch_table_prep(dkeys, dsizes, nb, lf, fileSz, storeSz, counters, st);
    launch_scan_u32(fileSz, 1, nb, fileOff, nullptr, ctx->bLdScan.p, st);
    launch_scan_u32(storeSz, 1, nb, storeOff, nullptr, ctx->bLdScan.p, st);
    const GwLayout gwl = P.compact ? GwLayout{ctx->dGwBase, ctx->gwTargets, ctx->gwGap} : GwLayout{};
    launch_table_insert(dkeys, dsizes, nb, lf, fileOff, storeOff, dvals, tb, P.valuesStored, P.dbuckets, P.nbuckets,
                        (unsigned int*)(counters + 2), (unsigned int*)(counters + 2) + 1, st, gwl, (unsigned int*)(counters + 3));
    if (P.compact)
        launch_table_values_compact(dkeys, dsizes, nb, lf, fileOff, storeOff, dvals, tb, fileVals, reinterpret_cast<uint32_t*>(P.dvalues) + P.valuesStored,
                                    gwl, (unsigned int*)(counters + 3), st);
    else
        launch_table_values(dkeys, dsizes, nb, lf, fileOff, storeOff, dvals, tb, fileVals, P.dvalues + P.valuesStored, st);
    if (hipGetLastError() != hipSuccess) return fail(MC_ERR_HIP, "database load: a table-build kernel failed to launch");
    P.valuesStored += stored;
    P.keysLoaded += nb;
    return MC_OK;
}

// Mode T (mc_config.target_shard_*): the values of a batch that lie in the context's target range move to the front of the batch's value
// array, bucket after bucket; a bucket is first cut as the whole table would cut it (remove-overpopulated empties it, max-locations
// keeps its first n values: host_hashmap.hpp:454-495), so that the device's own size rules find nothing more to do.  A bucket none
// of whose locations are in range keeps its key with size 0 (the table build skips it).
uint64_t mcamd::cut_batch_to_target_range(const mc_ctx* ctx, uint8_t* sizes, uint8_t* vals, uint32_t nkeys, uint32_t targetBytes)
{
    const uint32_t lo = ctx->tgtLo, hi = ctx->tgtHi;
    const uint32_t rmOver = ctx->cfg.remove_overpopulated, maxLocs = ctx->cfg.max_locations_per_feature;
    const size_t vb = 4 + targetBytes;
    const uint8_t* src = vals;
    uint8_t* dst = vals;
    uint64_t kept = 0;
    for (uint32_t i = 0; i < nkeys; ++i) {
        const uint32_t fileSize = sizes[i];
        uint32_t eff = fileSize;
        if (rmOver && eff > rmOver) eff = 0;
        if (maxLocs && eff > maxLocs) eff = maxLocs;
        uint32_t n = 0;
        for (uint32_t j = 0; j < eff; ++j) {
            const uint8_t* v = src + (size_t)j * vb;
            uint32_t tgt = (uint32_t)v[4] | ((uint32_t)v[5] << 8);
            if (targetBytes == 4) tgt |= ((uint32_t)v[6] << 16) | ((uint32_t)v[7] << 24);
            if (tgt >= lo && tgt < hi) {
                if (dst != v) std::memmove(dst, v, vb);
                dst += vb; ++n;
            }
        }
        src += (size_t)fileSize * vb;
        sizes[i] = (uint8_t)n;
        kept += n;
    }
    return kept;
}

// The whole .cache file of a single-part context (between mc_load_begin and mc_load_end, which the caller issues).
int mcamd::load_file_pipelined(mc_ctx* ctx, const std::string& fname, uint32_t targetBytes, uint64_t stats[4])
{
    auto fail = [&](int code, const std::string& msg) { ctx->err = msg; return code; };
    if (ctx->parts.size() != 1 || !ctx->parts[0].loading) return fail(MC_ERR_STATE, "load_file_pipelined: single-part load only");
    const int fd = ::open(fname.c_str(), O_RDONLY);
    if (fd < 0) return fail(MC_ERR_IO, "Could not read database file '" + fname + "'");
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    uint64_t head[3] = {0, 0, 0};
    if (!pread_all(fd, head, 24, 0)) return fail(MC_ERR_IO, "truncated " + fname);
    const uint64_t nkeys = head[0], batch = head[2];
    if (nkeys && batch == 0) return fail(MC_ERR_IO, "corrupt header in " + fname);
    if (batch > (1ull << 26)) return fail(MC_ERR_UNSUPPORTED, "unsupported batch size in " + fname + " (header says " + std::to_string(batch) + " keys per batch)");
    struct stat sb{};
    if (fstat(fd, &sb) != 0) return fail(MC_ERR_IO, "cannot stat " + fname);
    const uint64_t fileSize = (uint64_t)sb.st_size;
    const size_t vb = 4 + targetBytes;
    const uint64_t t0 = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }();
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; };

    // ---- index pass: every batch's place (its sizes are read to find the next one), and what the store would take with every list on
    //      lines of its own (announce_store)
    std::vector<BatchPlace> place;
    uint64_t padded = 0;
    // Mode T: what this range's store takes is known only when the values have been read; the first estimate assumes a bucket's
    // locations fall into the range independently with the range's share of the windows: a list of e locations leaves a piece of
    // e * share entries, there at all with probability 1 - (1 - share)^e, and a piece on lines of its own wastes half a line
    const bool targetCut = ctx->cfg.target_shard_count > 1;
    double pieceEntries[256] = {0}, piecePadded[256] = {0}, pieceThere[256] = {0}, estPlain = 0, estPadded = 0, estKeys = 0;
    struct ModelSums { double keys, plain, padded; };
    std::vector<ModelSums> model;                             // the model's sums batch by batch (the sampled ones are compared with what is there)
    if (targetCut)
        for (int e = 1; e < 256; ++e) {
            const double sh = std::min(1.0, std::max(ctx->tgtShare, 0.0)), there = 1.0 - std::pow(1.0 - sh, e);
            pieceThere[e] = there;                             // (the bucket table is sized by the keys that have a piece here)
            if (e >= 2) { pieceEntries[e] = e * sh; piecePadded[e] = e * sh + 0.5 * kListAlign * there; }
        }
    const uint32_t rmOver0 = ctx->cfg.remove_overpopulated, maxLocs0 = ctx->cfg.max_locations_per_feature;
    {
        // The places are a chain (a batch's sizes say where the next batch begins): this thread reads the sizes and sums them; what else
        // is wanted of them -- the padded total, Mode T's model sums -- are functions of the batch's HISTOGRAM of sizes, taken by helper
        // threads from a ring of buffers (one thread doing both: 0.36 s per 4 x 10^8 keys, 40 % of a 19 GB part's load)
        uint32_t effOf[256];                                   // (the load-time modifiers that look at the size alone: table_build.hip effective_size)
        for (uint32_t v = 0; v < 256; ++v) {
            uint32_t e = v;
            if (rmOver0 && e > rmOver0) e = 0;
            if (maxLocs0 && e > maxLocs0) e = maxLocs0;
            effOf[v] = e;
        }
        const size_t expectBatches = (size_t)((nkeys + batch - 1) / std::max<uint64_t>(batch, 1));
        if (targetCut) model.assign(expectBatches, ModelSums{0, 0, 0});
        constexpr uint32_t kRing = 8;
        const size_t bufBytes = (size_t)std::min<uint64_t>(batch, nkeys);
        std::vector<std::vector<uint8_t>> ring(kRing, std::vector<uint8_t>(bufBytes));
        struct Slot { int state = 0; uint32_t nb = 0; size_t b = 0; } slot[kRing];   // 0 free, 1 filled, 2 taken
        std::mutex im; std::condition_variable icv;
        bool idone = false;
        struct Sums { uint64_t padded = 0; double keys = 0, plain = 0, pad = 0; };
        const uint32_t nhelp = 3;
        std::vector<Sums> sums(nhelp);
        auto helper = [&](uint32_t id) {
            for (;;) {
                uint32_t k = kRing;
                {
                    std::unique_lock<std::mutex> l(im);
                    icv.wait(l, [&] { for (uint32_t i = 0; i < kRing; ++i) if (slot[i].state == 1) { k = i; return true; } return idone; });
                    if (k == kRing) return;
                    slot[k].state = 2;
                }
                const uint8_t* sz = ring[k].data();
                const uint32_t nb = slot[k].nb;
                uint32_t h[4][256] = {};
                uint32_t i = 0;
                for (; i + 4 <= nb; i += 4) { ++h[0][sz[i]]; ++h[1][sz[i + 1]]; ++h[2][sz[i + 2]]; ++h[3][sz[i + 3]]; }
                for (; i < nb; ++i) ++h[0][sz[i]];
                Sums S;
                for (uint32_t v = 0; v < 256; ++v) {
                    const uint64_t c = (uint64_t)h[0][v] + h[1][v] + h[2][v] + h[3][v];
                    if (!c) continue;
                    const uint32_t e = effOf[v];
                    S.padded += c * list_alloc(e, kListAlign);
                    S.keys += c * pieceThere[e]; S.plain += c * pieceEntries[e]; S.pad += c * piecePadded[e];
                }
                if (targetCut) model[slot[k].b] = ModelSums{S.keys, S.plain, S.pad};
                sums[id].padded += S.padded; sums[id].keys += S.keys; sums[id].plain += S.plain; sums[id].pad += S.pad;
                { std::lock_guard<std::mutex> l(im); slot[k].state = 0; }
                icv.notify_all();
            }
        };
        std::vector<std::thread> helpers;
        for (uint32_t t = 0; t < nhelp; ++t) helpers.emplace_back(helper, t);
        auto finish = [&] { { std::lock_guard<std::mutex> l(im); idone = true; } icv.notify_all(); for (auto& t : helpers) t.join(); };
        uint64_t off = 24;
        size_t bidx = 0;
        for (uint64_t done = 0; done < nkeys; ++bidx) {
            const uint32_t nb = (uint32_t)std::min<uint64_t>(batch, nkeys - done);
            const uint32_t k = (uint32_t)(bidx % kRing);
            { std::unique_lock<std::mutex> l(im); icv.wait(l, [&] { return slot[k].state == 0; }); }
            uint8_t* sz = ring[k].data();
            if (off + (uint64_t)nb * 5 > fileSize || !pread_all(fd, sz, nb, off + (uint64_t)nb * 4)) { finish(); return fail(MC_ERR_IO, "truncated " + fname); }
            uint64_t bv = 0;
            for (uint32_t i = 0; i < nb; ++i) bv += sz[i];
            if (off + (uint64_t)nb * 5 + bv * vb > fileSize) { finish(); return fail(MC_ERR_IO, "truncated " + fname); }
            place.push_back(BatchPlace{off, bv, nb});
            { std::lock_guard<std::mutex> l(im); slot[k].nb = nb; slot[k].b = bidx; slot[k].state = 1; }
            icv.notify_all();
            off += (uint64_t)nb * 5 + bv * vb;
            done += nb;
        }
        finish();
        for (const Sums& S : sums) { padded += S.padded; estKeys += S.keys; estPlain += S.plain; estPadded += S.pad; }
        if (targetCut) model.resize(place.size());
    }
    const uint64_t tIndex = now();
    const size_t nbatches = place.size();
    if (nbatches == 0) return mcamd::allocate_buckets(ctx, 1);   // (an empty part file: mc_load_begin may have left the bucket table to the loader -- target ranges)
    {
        // (a key shard keeps about 1 / count of the lists: the margin of allocate_table's estimate for the plain store)
        const uint64_t c = std::max<uint32_t>(ctx->cfg.key_shard_count, 1);
        uint64_t want = c > 1 ? padded / c + padded / (3 * c) + (1u << 16) : padded;
        if (targetCut) {
            // (15 % margin on the corrected estimate, everything for small files; a store or a table that turns out too small is counted to
            // its end and the load repeated with the exact numbers: mc_open_database)
            Part& T = ctx->parts[0];
            if (ctx->tgtExactPlain) {
                T.dvaluesCap = ctx->tgtExactPlain + 1; want = ctx->tgtExactPadded;
                if (mcamd::allocate_buckets(ctx, std::max<uint64_t>(ctx->tgtExactKeys, 1)) != MC_OK) return MC_ERR_NOMEM;
            } else {
                // the model against the file: a few batches spread over the file are read and cut now; what they keep over what the model says they
                // keep corrects the model's totals (locations of a feature cluster in neighbouring targets -- strains of one species: fewer
                // features per range and longer pieces than independent draws give)
                const size_t nsample = std::min<size_t>(nbatches, 4);
                double got[3] = {0, 0, 0}, said[3] = {0, 0, 0};
                {
                    std::vector<std::thread> th;
                    std::vector<std::array<double, 3>> part(nsample, {0, 0, 0});
                    std::vector<char> okS(nsample, 1);
                    for (size_t k = 0; k < nsample; ++k)
                        th.emplace_back([&, k] {
                            const BatchPlace& B = place[(2 * k + 1) * nbatches / (2 * nsample)];
                            std::vector<uint8_t> sz(B.nkeys), vals((size_t)B.fileVals * vb + 8);
                            if (!pread_all(fd, sz.data(), B.nkeys, B.off + (uint64_t)B.nkeys * 4) ||
                                (B.fileVals && !pread_all(fd, vals.data(), (size_t)B.fileVals * vb, B.off + (uint64_t)B.nkeys * 5))) { okS[k] = 0; return; }
                            cut_batch_to_target_range(ctx, sz.data(), vals.data(), B.nkeys, targetBytes);
                            for (uint32_t i = 0; i < B.nkeys; ++i) {
                                part[k][0] += sz[i] ? 1 : 0; part[k][1] += list_alloc(sz[i], 1); part[k][2] += list_alloc(sz[i], kListAlign);
                            }
                        });
                    for (auto& t : th) t.join();
                    for (size_t k = 0; k < nsample; ++k) {
                        if (!okS[k]) return fail(MC_ERR_IO, "truncated " + fname);
                        const ModelSums& M = model[(2 * k + 1) * nbatches / (2 * nsample)];
                        for (int j = 0; j < 3; ++j) got[j] += part[k][j];
                        said[0] += M.keys; said[1] += M.plain; said[2] += M.padded;
                    }
                }
                auto corrected = [&](double total, int j) { return said[j] > 0 ? total * got[j] / said[j] : total; };
                estKeys = corrected(estKeys, 0); estPlain = corrected(estPlain, 1); estPadded = corrected(estPadded, 2);
                double margin = 1.15;
                bool tiny = true;
                if (const char* e = std::getenv("MC_TARGET_STORE_MARGIN")) { margin = std::atof(e); tiny = false; }   // tests: an estimate that is too small
                auto roomy = [&](double est, uint64_t all) {
                    return tiny && all <= (1ull << 24) ? all : std::min<uint64_t>(all, (uint64_t)(est * margin) + (tiny ? (1u << 20) : 0u));
                };
                T.dvaluesCap = roomy(estPlain, head[1]) + 1;
                want = roomy(estPadded, padded);
                if (mcamd::allocate_buckets(ctx, std::max<uint64_t>(roomy(estKeys, nkeys), 1)) != MC_OK) return MC_ERR_NOMEM;
            }
        }
        mcamd::announce_store(ctx, want);
    }
    auto batch_bytes = [&](const BatchPlace& b) { return align16((size_t)b.nkeys * 4) + align16(b.nkeys) + align16((size_t)b.fileVals * vb) + 16; };
    size_t slabBytes = 0;
    for (const auto& b : place) slabBytes = std::max(slabBytes, batch_bytes(b));
    uint32_t nthreads = 8;
    if (const char* e = std::getenv("MC_LOAD_THREADS")) nthreads = (uint32_t)std::max(1, std::atoi(e));
    nthreads = (uint32_t)std::min<size_t>(nthreads, nbatches);
    const uint32_t nslabs = (uint32_t)std::min<size_t>(nbatches, nthreads + 2);

    // ---- slabs: batch b lives in slab b % nslabs; a reader may fill it once the feeder has released batch b - nslabs (no deadlock: the
    // earliest batch the feeder waits for always finds its slab free or about to be freed by the feeder's own progress)
    std::vector<Slab> slabs(nslabs);
    for (auto& s : slabs) {
        s = pool().get(slabBytes);
        if (!s.p) { for (auto& t : slabs) pool().put(t); return fail(MC_ERR_NOMEM, "database load: cannot allocate pinned staging memory"); }
    }
    const uint64_t tSlabs = now();
    std::mutex mtx;
    std::condition_variable cv;
    std::vector<uint8_t> ready(nbatches, 0);                  // 1 = in its slab, 2 = the read failed
    std::vector<uint64_t> storedOf(nbatches, 0);
    std::vector<uint64_t> keptOf(nbatches, 0);                // Mode T: the values of the batch that are in this context's target range,
    std::vector<uint32_t> keysOf(targetCut ? nbatches : 0, 0);
    std::vector<uint64_t> plainOf(targetCut ? nbatches : 0, 0), paddedOf(targetCut ? nbatches : 0, 0);   // what they take in the store (plain / on lines of their own)
    size_t released = 0;                                      // batches whose slab the feeder has given back
    std::atomic<size_t> next{0};
    bool abort = false;
    const LoadFilter lf{ctx->cfg.max_locations_per_feature, ctx->cfg.remove_overpopulated, ctx->cfg.key_shard_index, ctx->cfg.key_shard_count, ctx->parts[0].listAlign};
    auto reader = [&] {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nbatches) return;
            {
                std::unique_lock<std::mutex> l(mtx);
                cv.wait(l, [&] { return abort || b < released + nslabs; });
                if (abort) return;
            }
            const BatchPlace& B = place[b];
            uint8_t* base = slabs[b % nslabs].p;
            uint8_t* keys = base; uint8_t* sizes = keys + align16((size_t)B.nkeys * 4); uint8_t* vals = sizes + align16(B.nkeys);
            // the batch is one contiguous range of the file; keys | sizes | values go to 16-byte aligned places of the slab
            bool ok = pread_all(fd, keys, (size_t)B.nkeys * 4, B.off) && pread_all(fd, sizes, B.nkeys, B.off + (uint64_t)B.nkeys * 4) &&
                      (B.fileVals == 0 || pread_all(fd, vals, (size_t)B.fileVals * vb, B.off + (uint64_t)B.nkeys * 5));
            uint64_t st = 0;
            if (ok && targetCut) {                             // (only this reader touches the batch's entries before ready[b])
                keptOf[b] = cut_batch_to_target_range(ctx, sizes, vals, B.nkeys, targetBytes);
                uint64_t pl = 0, pd = 0;
                uint32_t nk = 0;
                for (uint32_t i = 0; i < B.nkeys; ++i) { pl += list_alloc(sizes[i], 1); pd += list_alloc(sizes[i], kListAlign); nk += sizes[i] ? 1u : 0u; }
                plainOf[b] = pl; paddedOf[b] = pd; keysOf[b] = nk;
            }
            if (ok) {
                const uint32_t* k = reinterpret_cast<const uint32_t*>(keys);
                if (lf.shardCnt > 1 || lf.maxLocs || lf.rmOver) for (uint32_t i = 0; i < B.nkeys; ++i) st += stored_of(k[i], sizes[i], lf);
                else if (lf.align > 1) for (uint32_t i = 0; i < B.nkeys; ++i) st += list_alloc(sizes[i], lf.align);
                else for (uint32_t i = 0; i < B.nkeys; ++i) st += sizes[i] > 1 ? sizes[i] : 0u;
            }
            {
                std::lock_guard<std::mutex> l(mtx);
                storedOf[b] = st;
                ready[b] = ok ? 1 : 2;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> threads;
    for (uint32_t t = 0; t < nthreads; ++t) threads.emplace_back(reader);
    hipStream_t copySt = nullptr;
    auto stop = [&](int code, const std::string& msg) {         // an error: nothing stays in flight, every slab goes back
        { std::lock_guard<std::mutex> l(mtx); abort = true; }
        cv.notify_all();
        for (auto& t : threads) t.join();
        if (copySt) (void)hipStreamSynchronize(copySt);
        (void)hipStreamSynchronize(ctx->stream);
        for (auto& s : slabs) pool().put(s);
        return fail(code, msg);
    };

    // ---- feeder: slab -> device staging (two of them) on a copy stream, table kernels behind it on the context's stream
    if (hipSetDevice(ctx->device) != hipSuccess) return stop(MC_ERR_HIP, "hipSetDevice");
    hipEvent_t copied[2] = {nullptr, nullptr}, built[2] = {nullptr, nullptr};
    DevBuf stage[2];
    bool okInit = hipStreamCreateWithFlags(&copySt, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 2 && okInit; ++i)
        okInit = hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&built[i], hipEventDisableTiming) == hipSuccess;
    auto cleanup = [&] {
        for (int i = 0; i < 2; ++i) { if (copied[i]) (void)hipEventDestroy(copied[i]); if (built[i]) (void)hipEventDestroy(built[i]); if (stage[i].p) (void)hipFree(stage[i].p); }
        if (copySt) (void)hipStreamDestroy(copySt);
    };
    if (!okInit) { const int code = stop(MC_ERR_HIP, "database load: cannot create the copy stream"); cleanup(); return code; }
    uint64_t waitNs = 0, bytesIn = 24;
    uint64_t exactPlain = 0, exactPadded = 0, exactKeys = 0;   // Mode T: this range's store and keys, known when the last batch is through
    bool countOnly = false;
    int rc = MC_OK;
    std::string emsg;
    for (size_t b = 0; b < nbatches && !rc; ++b) {
        const int k = (int)(b & 1);
        {
            const uint64_t w0 = now();
            std::unique_lock<std::mutex> l(mtx);
            cv.wait(l, [&] { return ready[b] != 0; });
            waitNs += now() - w0;
            if (ready[b] == 2) { rc = MC_ERR_IO; emsg = "truncated " + fname; break; }
        }
        BatchPlace B = place[b];
        const uint64_t fileBytes = (uint64_t)B.nkeys * 5 + B.fileVals * vb;
        if (targetCut) {
            B.fileVals = keptOf[b];                            // (the slab keeps the batch's places; its values end earlier)
            exactPlain += plainOf[b]; exactPadded += paddedOf[b]; exactKeys += keysOf[b];
            Part& T = ctx->parts[0];
            // (a table for too few keys: twice the load factor it was sized for is where the load is given up)
            const bool crowded = (double)exactKeys > (double)T.nbuckets * kSlotsPerBucket * std::min(0.9, 2.0 * (double)ctx->loadFactor);
            if (!countOnly && (T.valuesStored + storedOf[b] > T.dvaluesCap || crowded)) {
                // the estimate was too small: the rest of the file is only counted (no more device work)
                countOnly = true;
                (void)hipStreamSynchronize(copySt);
            }
            if (countOnly) {
                bytesIn += fileBytes;
                { std::lock_guard<std::mutex> l(mtx); released = b + 1; }
                cv.notify_all();
                continue;
            }
        }
        const size_t bytes = batch_bytes(B);
        if (b >= 2 && hipEventSynchronize(built[k]) != hipSuccess) { rc = MC_ERR_HIP; emsg = "database load: table build failed"; break; }   // staging k is free again
        if (bytes > stage[k].cap) {
            if (stage[k].p) (void)hipFree(stage[k].p);
            stage[k].p = nullptr; stage[k].cap = 0;
            if (mcamd::dev_malloc(&stage[k].p, slabBytes) != hipSuccess) { rc = MC_ERR_NOMEM; emsg = "database load: cannot allocate the device staging"; break; }
            stage[k].cap = slabBytes;
        }
        const uint8_t* hbase = slabs[b % nslabs].p;
        if (hipMemcpyAsync(stage[k].p, hbase, bytes, hipMemcpyHostToDevice, copySt) != hipSuccess || hipEventRecord(copied[k], copySt) != hipSuccess ||
            hipStreamWaitEvent(ctx->stream, copied[k], 0) != hipSuccess) { rc = MC_ERR_HIP; emsg = "database load: copy to the device failed"; break; }
        const uint8_t* dbase = static_cast<const uint8_t*>(stage[k].p);
        const uint32_t* dkeys = reinterpret_cast<const uint32_t*>(dbase);
        const uint8_t* dsizes = dbase + align16((size_t)B.nkeys * 4);
        const uint8_t* dvals = dsizes + align16(B.nkeys);
        const uint8_t* hsizes = hbase + align16((size_t)B.nkeys * 4);
        // device chunks of at most 2^22 keys (the reference's and this repository's files: one per batch)
        const uint32_t kChunk = 1u << 22;
        if (B.nkeys <= kChunk && B.fileVals < (1ull << 32)) rc = load_chunk_device_async(ctx, dkeys, dsizes, dvals, B.nkeys, B.fileVals, storedOf[b]);
        else {
            const uint32_t* hkeys = reinterpret_cast<const uint32_t*>(hbase);
            uint64_t voff = 0;
            for (uint32_t done = 0; done < B.nkeys && !rc;) {
                uint32_t nb = 0; uint64_t fv = 0, stc = 0;
                while (done + nb < B.nkeys && nb < kChunk && fv + hsizes[done + nb] < (1ull << 32)) {
                    fv += hsizes[done + nb]; stc += stored_of(hkeys[done + nb], hsizes[done + nb], lf); ++nb;
                }
                rc = load_chunk_device_async(ctx, dkeys + done, dsizes + done, dvals + voff * vb, nb, fv, stc);
                done += nb; voff += fv;
            }
        }
        if (rc) { emsg = ctx->err; break; }
        if (hipEventRecord(built[k], ctx->stream) != hipSuccess) { rc = MC_ERR_HIP; emsg = "database load: event"; break; }
        bytesIn += fileBytes;
        // the slab goes back to the readers when its copy has left the host: the copy of the PREVIOUS batch has had a batch's time to finish
        if (b >= 1) {
            if (hipEventSynchronize(copied[k ^ 1]) != hipSuccess) { rc = MC_ERR_HIP; emsg = "database load: copy to the device failed"; break; }
            { std::lock_guard<std::mutex> l(mtx); released = b; }
            cv.notify_all();
        }
    }
    if (rc) { const int code = stop(rc, emsg); cleanup(); return code; }
    for (auto& t : threads) t.join();
    const bool synced = hipStreamSynchronize(copySt) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    cleanup();
    for (auto& s : slabs) pool().put(s);
    if (!synced) return fail(MC_ERR_HIP, "database load: table build failed");
    if (countOnly) {
        ctx->tgtExactPlain = std::max<uint64_t>(exactPlain, 1); ctx->tgtExactPadded = exactPadded; ctx->tgtExactKeys = exactKeys; ctx->storeShort = true;
        return fail(MC_ERR_NOMEM, "database load: the target range holds more locations than estimated (" + std::to_string(exactPlain) + "); load again with the exact size");
    }
    if (stats) { stats[0] = bytesIn; stats[1] = now() - t0; stats[2] = tIndex - t0; stats[3] = waitNs; }
    if (std::getenv("MC_LOAD_TRACE"))
        std::fprintf(stderr, "mc load %s: %.2f GB, %zu batches, %u threads, %u slabs of %.0f MB; index %.3f s, slabs %.3f s, total %.3f s (feeder waited %.3f s) = %.1f GB/s\n",
                     fname.c_str(), bytesIn / 1e9, nbatches, nthreads, nslabs, slabBytes / 1e6, (tIndex - t0) / 1e9, (tSlabs - tIndex) / 1e9, (now() - t0) / 1e9, waitNs / 1e9,
                     bytesIn / (double)(now() - t0));
    return MC_OK;
}
