// metacache_amd/csrc/context.h -- the object behind mc_ctx (internal).
#pragma once

#include "../../include/metacache_amd.h"
#include "kernels.h"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <thread>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace mcamd {

struct DevBuf {                 // grow-only device allocation
    void* p = nullptr;
    size_t cap = 0;
};

struct Pipe {                   // the device workspace of ONE batch in flight + the stream its work is enqueued on
    GwSortSide sortSide;            // second stream + fork / join events of the sorted path (created on first use)
    hipStream_t stream = nullptr;
    DevBuf bWinCount, bWinOff, bFeatures, bPsize, bPpay, bQstat, bHitOff, bHits, bCscr, bCscr2, bScan, bStats,
        bCands, bScanIn, bQflag, bMid, bChunkList, bBigPool, bSliceFill, bBigPool2, bSortTmp, bSide,
        bOrder,              // scratch of launch_gw_order (work lists longest first)
        bNumbers, bCounts;   // Mode K shard side: the partial lists as global window numbers + per-read counts (mc_partial_numbers)
    uint32_t lastN = 0;
    uint32_t numbersN = 0xFFFFFFFFu; uint64_t numbersTotal = 0;   // MC_WANT_PARTIAL_NUMBERS: bNumbers / bCounts hold this batch's lists already
    uint64_t* hTotal = nullptr;   // pinned: the one host round trip of a batch lands here (a pageable target makes the copy blocking)
    // MC_DEFER_TAIL: the batch's main kernels are enqueued, the rare classes (sorted lists, what is left for the exact wave kernels) wait
    // for mc_query_finish -- the host looks at their counters THEN, while the device is busy with the other pipe's batch
    struct Tail {
        bool pending = false;
        Workspace ws{}; BatchView b{}; SketchParams sp{}; DeviceTable tab{};
        uint32_t K = 0; const uint32_t* taxkey = nullptr;
        bool compact = false, sortedPath = false, fuse = false, skipWaveSketch = false;
        uint64_t poolEntries = 0;
        hipStream_t st = nullptr;
        hipEvent_t mainDone = nullptr;   // recorded behind the main kernels and the copy of the sorted-class counter
    } tail;
};

struct Part {
    // host-side build state (freed by mc_load_end)
    std::vector<TableBucket> hbuckets;
    uint64_t expectKeys = 0, expectValues = 0;
    uint64_t keysLoaded = 0;
    uint64_t valuesStored = 0;      // entries written to dvalues (buckets of size > 1 only)
    uint64_t locations = 0;         // all locations kept (incl. inline singletons)
    uint64_t keysStored = 0;
    uint32_t nbuckets = 0, maxProbe = 1;
    bool loading = false, ready = false, announced = false;
    TableBucket* dbuckets = nullptr;
    uint64_t* dvalues = nullptr;    // compact: the same allocation holds uint32_t entries (global window numbers, kernels.h DeviceTable)
    uint64_t dvaluesCap = 0;
    uint64_t* ddirect = nullptr;    // direct-address index (kernels.h DeviceTable::direct), built at mc_load_end where it pays and fits
    bool compact = false;
    uint32_t listAlign = 1;         // lists begin at multiples of this many entries (kernels.h list_alloc; kListAlign when the padded total was announced and fits)
    uint64_t expectStore = 0;       // entries of the store with the padding (0: unknown -> no alignment)
};

struct Taxon {
    int64_t id, parent;
    uint8_t rank;
    std::string name, filename;
    uint64_t index, windows;
};

struct Slot {
    // pinned host staging
    uint8_t* hseq = nullptr; uint32_t* hqinfo = nullptr; uint32_t* hmaxwin = nullptr;
    uint32_t nq = 0; uint64_t nchars = 0;
    uint32_t maxSingle = 0;         // longest single read of the batch (mc_batch_add): none beyond a lane's reach -> the chunk lanes' three launches are left out
    // device input
    uint8_t* dseq = nullptr; uint32_t* dqinfo = nullptr; uint32_t* dmaxwin = nullptr;
    // pinned results
    mc_candidate* hcands = nullptr; QueryStat* hqstat = nullptr; uint32_t* hhitcounts = nullptr;
    uint64_t* hhitoff = nullptr; mc_location* hhits = nullptr; size_t hhitsCap = 0;
    hipEvent_t done = nullptr;
    bool submitted = false;
    uint32_t submittedQueries = 0;
    Pipe* pipe = nullptr;
    // coalesced submission (context.cpp "slot coalescer"): 0 = not queued, 1 = waiting for a dispatcher, 2 = in a dispatcher's hands,
    // 3 = enqueued on the device (`done` is recorded behind its copies back) or failed (coRc)
    int coState = 0, coRc = 0, coLowest = 0;
    bool coEvent = false;
    hipEvent_t coDoneEv = nullptr;  // the united batch's event (the dispatcher's, one of four in turn): recorded behind the kernel that delivered the results
    std::string coErr;
};

// one dispatcher thread of the slot coalescer: a pipe of its own, the united batch's device input, pinned staging (twice) for the rebased qinfo rows
// query_on_pipe, internal (the slots, which see every read on the host): no single read is longer than one lane takes -- the chunk lanes'
// kernels (launched for their work list, which would be empty) are left out.  Masked out of mc_query_device's flags.
constexpr int kQueryNoLongReads = 1 << 24;

struct CoDispatcher {
    std::thread th;
    Pipe* pipe = nullptr;
    DevBuf dseq, dqinfo, dmaxwin;
    uint32_t* hq[2] = {nullptr, nullptr}; uint32_t* hmw[2] = {nullptr, nullptr};
    hipEvent_t staged[2] = {nullptr, nullptr};   // the staging set has left the host
    hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr};   // a united batch's results are in its slots' pinned buffers (ONE event for all its slots, four in turn)
    uint32_t doneTurn = 0;
    bool stagedUsed[2] = {false, false};
    uint32_t turn = 0;
};

// table_build: insert one chunk of a single-part database whose batch arrays already live in device memory
// (between mc_load_begin and mc_load_end; used by mc_load_batch after its upload and by the builder directly)
int load_chunk_device(mc_ctx* ctx, const uint32_t* dkeys, const uint8_t* dsizes, const uint8_t* dvals, uint32_t nkeys, uint64_t fileVals);

// the same without any synchronisation: what the chunk adds to the location store comes from the host (dbload.cpp)
// list alignment of the compact store (kernels.h list_alloc): a loader that knows the lists' sizes announces what the store takes with
// every list on lines of its own -- before the first chunk, with which the store is allocated
void announce_store(mc_ctx* ctx, uint64_t paddedEntries);
// what the caller of mc_open_database / mc_create on THIS thread knows about the device's other tenants (partset.cpp): list alignment
// default (-1 / 0 / 1, only where neither MC_LIST_ALIGN nor mc_set_tuning says otherwise) and the share of free memory its padding may take
struct OpenHints { int listAlign = -1; double listAlignShare = 1.0; int directIndex = -1; };
OpenHints& open_hints();
int allocate_values(mc_ctx* ctx);
int allocate_buckets(mc_ctx* ctx, uint64_t nkeys);
int reserve_slot_pipes(mc_ctx* ctx, uint64_t locs, uint64_t keys, bool waitForStores = true);
int reserve_query_pipes(mc_ctx* ctx, uint32_t n, uint64_t chars);
int load_chunk_device_async(mc_ctx* ctx, const uint32_t* dkeys, const uint8_t* dsizes, const uint8_t* dvals, uint32_t nkeys, uint64_t fileVals, uint64_t stored);
// dbload.cpp: a whole .cache file of a single-part context through reader threads, pinned slabs and a copy stream (between mc_load_begin
// and mc_load_end).  stats (may be NULL): bytes read, nanoseconds in all, of the index pass, the feeder waited for readers
int load_file_pipelined(mc_ctx* ctx, const std::string& fname, uint32_t targetBytes, uint64_t stats[4]);
// Mode T: one batch of the file in host memory (keys | sizes | packed values) cut IN PLACE to the locations of the context's target
// range, after the load-time rules that look at a bucket's size; returns the number of values that stay (sizes[] updated)
uint64_t cut_batch_to_target_range(const mc_ctx* ctx, uint8_t* sizes, uint8_t* vals, uint32_t nkeys, uint32_t targetBytes);

// error text for failures that have no context yet (mc_last_error(NULL))
void set_global_error(const std::string& msg);

struct TimedKernel { double ms = 0; uint64_t launches = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };

}  // namespace mcamd

struct mc_ctx {
    mc_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    mcamd::SketchParams querySketch{};     // used for queries
    mcamd::SketchParams targetSketch{};    // the database's own (window stride feeds maxWindowsInRange)
    uint64_t maxLocs = 254, targetCount = 0;
    std::vector<mcamd::Part> parts;
    float loadFactor = 0.5f;
    // mc_load_target_windows: every target's window count.  When all windows of the database (plus a gap per target) can be numbered
    // in 32 bits and the table is a single part, the location lists are stored as global window numbers, 4 bytes each
    // (DeviceTable::values32 / gwBase / gwDir).
    bool compactAllowed = true, locRangeViolated = false;
    std::vector<uint32_t> targetWindows;   // empty: not announced
    // cfg.target_shard_count > 1 ("Mode T", set by mc_open_database): only locations of the targets [tgtLo, tgtHi) are kept;
    // tgtShare = this range's share of the database's windows (what the location store is sized by)
    uint32_t tgtLo = 0, tgtHi = 0xFFFFFFFFu;
    double tgtShare = 1.0;
    bool tgtRangeSet = false, storeShort = false;      // storeShort: the estimate of the range's store was too small; tgtExact* hold what it takes
    uint64_t tgtExactPlain = 0, tgtExactPadded = 0, tgtExactKeys = 0;
    uint32_t* dGwBase = nullptr;           // [targets + 1]
    uint32_t* dGwDir = nullptr;
    uint32_t gwDirShift = 0, gwGap = 0, gwTargets = 0, gwBits = 32;   // gwBits: bits of the largest window number
    bool tableReady = false;
    std::vector<uint64_t> hvalues;          // multi-part load: all location lists on the host until the last part is in

    // taxonomy / lineages (host) + per-rank taxon keys (device)
    std::vector<mcamd::Taxon> taxa;
    std::vector<uint32_t> lineages;        // [targets * 21], taxon index + 1
    std::map<int, uint32_t*> taxkeyDev;    // lowest_rank -> device array [targets]

    // workspace of mc_query_device / mc_candidates_from_hits callers (pipe0.stream == stream); every host batch slot has its own
    // Pipe, so that the H2D copy, the kernels and the D2H copy of different slots overlap on the device
    mcamd::Pipe pipe0;
    mcamd::Pipe pipe1;                     // MC_SECOND_PIPE callers: a second batch in flight next to pipe0's (stream created on first use)
    std::mutex taxMtx, timerMtx;
    // host batch slots borrow a pipe from this pool between mc_batch_submit and mc_batch_wait: a few batches in flight are
    // enough to overlap H2D, kernels and D2H, and the pool bounds both the device memory and the number of threads inside the
    // HIP runtime at once (measured: 128 threads submitting on 128 streams spend 99 % of their time in runtime locks)
    std::vector<mcamd::Pipe*> pipes, freePipes;
    std::mutex pipeMtx;
    std::condition_variable pipeCv;
    // single-part tables are built on the device (table_build.hip): staging for one batch of the file
    mcamd::DevBuf bLdKeys, bLdSizes, bLdVals, bLdFileSz, bLdStoreSz, bLdFileOff, bLdStoreOff, bLdScan, bLdCounters;
    bool useLanePath = true;               // lane-parallel fast path for short reads (off: wave kernels only)
    uint32_t bigMin = 128;                 // location lists longer than this (and than 64) are filtered before they are counted (big_filter_kernel); MC_BIG_MIN.
                                           // 256 / 128 / 64 at 15 Gbp (195 locations per read): 3.92 / 3.53 / 3.48 ms per 10^6 reads; at 4.5 Gbp (100): 3.01 / 3.12 / 3.25
    int quadLookup = -1;                   // MC_QUAD_LOOKUP=0/1 forces the bucket fetch scheme of probe_cands (tests); -1 = by table size
    int listAlignWant = -1;                // mc_set_tuning "list_align" / MC_LIST_ALIGN: -1 = where the padded store stays below 1.5 x the plain one and fits, 0 / 1
    std::atomic<uint32_t> storesPlaced{0}; // single-part loads: location stores allocated (reserve_slot_pipes waits for the table before it takes memory)
    std::atomic<bool> reserveByLoader{false};  // mc_open_database reserves the slot pipes on a thread of its own beside the file load; otherwise mc_load_end does (tables built through mc_load_*)
    std::atomic<bool> loadSettled{false};  // mc_open_database: the files are through (or the load failed)
    double listAlignShare = 1.0;           // announce_store: the padding (padded - plain store) may take this share of the device's free memory; the part set driver
                                           // lowers it to 1 / (parts it still has to place on the device) -- mcamd::open_hints
    int directWant = -1;                   // direct-address index beside the buckets: -1 = for tables whose buckets take 8 GiB and more, where 34 GB + head-room are free;
                                           // 0 / 1 (mc_set_tuning "direct_index" before the table is loaded, MC_DIRECT_INDEX)
    int fuseLane = -1;                     // sketching + probing of the lane path in ONE kernel: -1 = where the lookups are quad-cooperative (tables beyond 1 GiB: the
                                           // probing waits for HBM and the sketching runs under it: 5.27 -> 5.08 ms per 5 x 10^6 reads at full scale), 0 / 1 = never / always
                                           // (MC_LANE_FUSION, mc_set_tuning "lane_fusion"); small tables: 5 % slower on configs[1] (ALU phase at the probe kernel's occupancy)

    uint32_t gwMidH = 0;      // mc_set_tuning "gw_mid_h": reads up to this many locations take the small-filter instance of the stream filter (0 = none: the default --
                              // 8 192: the filters' 3.35 -> 3.0 ms per 250 000 long reads, the false keeps' sort and counting +0.35: lab notebook r06 section 6)
    uint32_t gwBigH = 32768;  // mc_set_tuning "gw_big_h": reads beyond this many locations take the fine-block instance of the stream filter (0 = none)
    bool buildHold = false;   // mc_build_table_begin took a hold on the block cache (devcache.h) that mc_build_table_end / mc_destroy gives back
    int filterBpc = 0, countBpc = 0, gwFuse = 1;   // mc_set_tuning: grids' blocks per CU (0 = default), counting inside the filter kernel -- this context only

    uint64_t loadStats[4] = {0, 0, 0, 0};  // mc_load_stats: bytes read from the database files, nanoseconds of the load, of its index pass, the feeder waited for the readers
    uint64_t ownerStats[4] = {0, 0, 0, 0}; // mc_owner_stats: reads, reads on the filtered path, numbers received, locations decoded for the sort

    // timing
    bool timing = false;
    std::map<std::string, mcamd::TimedKernel> timers;
    std::vector<hipEvent_t> eventPool;

    // slots
    std::vector<mcamd::Slot> slots;
    // slot coalescer: slots that are submitted while the dispatchers are busy go to the device TOGETHER, as one batch (a batch of the
    // reference's size -- 4 096 reads -- is ~30 kernel launches and three host round trips for 0.1 ms of device work)
    bool coalesce = false;
    std::mutex coMu;
    std::condition_variable coCv, coDoneCv;
    std::deque<uint32_t> coPending;
    std::vector<mcamd::CoDispatcher*> coDisp, coFree;   // all dispatchers' state (a pipe + staging each) / those nobody is using
    bool coStop = false;
    uint32_t coMaxQueries = 0; uint64_t coMaxChars = 0;
    uint64_t coBatches = 0, coSlots = 0;   // united batches sent, slots they held (mc_slot_stats)
};
