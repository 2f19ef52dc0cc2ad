// metacache_amd/csrc/context.cpp -- host side of the C ABI: context, table loading, the per-batch
// pipeline, host slots, timing.  (Compiled with hipcc; the kernels live in kernels.hip.)
#include "context.h"
#include "devcache.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>

using namespace mcamd;

namespace {

thread_local std::string g_createError;

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail((ctx), MC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                                                                                           \
    } while (0)

// a helper thread that works on a context beside its owner (reserve_slot_pipes) must not write ctx->err: its failures stay its own
static thread_local bool t_quietErrors = false;
static thread_local std::string* t_errSink = nullptr;         // a dispatcher thread of the slot coalescer keeps its errors for the slots it carries
int fail(mc_ctx* ctx, int code, const std::string& msg)
{
    if (t_errSink) { *t_errSink = msg; return code; }
    if (t_quietErrors) return code;
    if (ctx) ctx->err = msg; else g_createError = msg;
    return code;
}

int ensure(mc_ctx* ctx, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return MC_OK;
    static const bool trace = std::getenv("MC_ALLOC_TRACE") != nullptr;      // allocations of 20 ms (MC_ALLOC_TRACE=<ms>) and more go to stderr
    static const double traceMs = [] { const char* e = std::getenv("MC_ALLOC_TRACE"); const double v = e ? std::atof(e) : 0; return v > 1.0 ? v : (e && e[0] == '0' ? 0.0 : 20.0); }();
    timespec t0{}, t1{}, t2{};
    if (trace) clock_gettime(CLOCK_MONOTONIC, &t0);
    if (b.p) HIP_TRY(ctx, hipFree(b.p));
    if (trace) clock_gettime(CLOCK_MONOTONIC, &t1);
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + std::min<size_t>(bytes / 4, (size_t)512 << 20) + 256;   // head-room: fewer re-allocations (a quarter, at most 512 MB: the pools of a 2.5 M-pair batch are 11 GB each)
    HIP_TRY(ctx, dev_malloc(&b.p, want));
    b.cap = want;
    if (trace) {
        clock_gettime(CLOCK_MONOTONIC, &t2);
        const double f = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6, m = (t2.tv_sec - t1.tv_sec) * 1e3 + (t2.tv_nsec - t1.tv_nsec) / 1e6;
        if (f + m >= traceMs) std::fprintf(stderr, "mc ensure: %.1f MB: hipFree of the old buffer %.1f ms, hipMalloc %.1f ms\n", want / 1e6, f, m);
    }
    return MC_OK;
}

// ---- MC_SUBMIT_TRACE=1: where the host spends a slot batch (sums over all threads, printed by mc_destroy) -----------------------
static const bool g_submitTrace = std::getenv("MC_SUBMIT_TRACE") != nullptr;
static std::atomic<uint64_t> g_trace[8];   // ns: [0] waiting for a free pipe, [1] enqueueing H2D, [2] query_on_pipe in all, [3] of it inside hipStreamSynchronize,
                                           //     [4] enqueueing D2H; [5] batches, [6] hipStreamSynchronize calls
static inline uint64_t trace_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }

// ---- the runtime's slow submission state: calls that only ENQUEUE -- hipMemcpyAsync of a pinned buffer -- take milliseconds
// apiece once the HIP runtime's direct dispatch (AMD_DIRECT_DISPATCH, on by default) has gone into it, process by process, under many
// submitting threads.  The slot paths time exactly these calls (two clock reads per batch); a window of 32 batches whose enqueues
// averaged more than 0.5 ms each leaves a warning that names the switch: mc_runtime_warning(), and one line on stderr.
static std::atomic<uint64_t> g_enqNs{0}, g_enqCalls{0}, g_enqBatches{0};
static std::atomic<bool> g_enqWarned{false};
static std::mutex g_warnMu;
static std::string g_warning;
static void note_enqueues(uint64_t ns, uint32_t calls)
{
    if (g_enqWarned.load(std::memory_order_relaxed)) return;
    const uint64_t n = g_enqNs.fetch_add(ns) + ns, c = g_enqCalls.fetch_add(calls) + calls, b = g_enqBatches.fetch_add(1) + 1;
    if (b < 32) return;
    g_enqNs = 0; g_enqCalls = 0; g_enqBatches = 0;                 // (windows: racing updates only blur a window's edge)
    if (c == 0 || n / c < 500000ull) return;
    const char* dd = std::getenv("AMD_DIRECT_DISPATCH");
    if (dd && dd[0] == '0') return;                               // (already off: something else is slow)
    if (g_enqWarned.exchange(true)) return;
    char msg[400];
    std::snprintf(msg, sizeof msg, "metacache_amd: HIP calls that only enqueue (hipMemcpyAsync of pinned batches) take %.1f ms each in this process -- the HIP runtime's direct "
                  "dispatch does this under many submitting threads; start the process with AMD_DIRECT_DISPATCH=0 (the runtime reads it before main)", (double)(n / c) / 1e6);
    { std::lock_guard<std::mutex> l(g_warnMu); g_warning = msg; }
    std::fprintf(stderr, "%s\n", msg);
}
static hipError_t traced_sync(hipStream_t st)
{
    if (!g_submitTrace) return hipStreamSynchronize(st);
    const uint64_t t0 = trace_now();
    const hipError_t e = hipStreamSynchronize(st);
    g_trace[3] += trace_now() - t0; ++g_trace[6];
    return e;
}

// ---- timing ------------------------------------------------------------------------------
hipEvent_t get_event(mc_ctx* ctx)
{
    if (!ctx->eventPool.empty()) { hipEvent_t e = ctx->eventPool.back(); ctx->eventPool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ScopedTimer {
    mc_ctx* ctx; const char* name; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(mc_ctx* c, const char* n, hipStream_t s) : ctx(c), name(n), st(s)
    {
        if (ctx->timing) { std::lock_guard<std::mutex> l(ctx->timerMtx); a = get_event(ctx); b = get_event(ctx); (void)hipEventRecord(a, st); }
    }
    ~ScopedTimer()
    {
        if (a) { std::lock_guard<std::mutex> l(ctx->timerMtx); (void)hipEventRecord(b, st); ctx->timers[name].pending.emplace_back(a, b); }
    }
};

void collect_timers(mc_ctx* ctx)
{
    for (auto& kv : ctx->timers) {
        for (auto& pr : kv.second.pending) {
            (void)hipEventSynchronize(pr.second);
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { kv.second.ms += ms; kv.second.launches++; }
            ctx->eventPool.push_back(pr.first); ctx->eventPool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

}  // namespace

void mcamd::set_global_error(const std::string& msg) { g_createError = msg; }
mcamd::OpenHints& mcamd::open_hints() { static thread_local OpenHints h; return h; }

// The loader knows what the lists would take with every list on a line of its own (kernels.h list_alloc): alignment is switched on where
// that is wanted ("list_align"), the table has the compact store, and the padded store is affordable.  Before the first chunk.
void mcamd::announce_store(mc_ctx* ctx, uint64_t paddedEntries)
{
    Part& T = ctx->parts[0];
    if (T.dvalues || !T.compact || ctx->listAlignWant == 0 || paddedEntries == 0) return;
    const uint64_t plain = T.dvaluesCap;
    if (ctx->listAlignWant != 1 && paddedEntries > plain + plain / 2 + (1u << 20)) return;
    // Decision and allocation are ONE step per process: several loads run side by side on a device (a part group loads two parts at a
    // time while the previous group is still resident), and each would judge the padded store affordable against the same free memory.
    // A padded store that cannot be had is not an error -- the table falls back to the plain one.
    static std::mutex decide;
    std::lock_guard<std::mutex> lk(decide);
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return;
    // (inside a part group the other parts of the group still have to fit: only the padding beyond the plain store counts against a
    // share of what is free: listAlignShare, 1 outside part sets -- partset.cpp sets it through open_hints)
    const uint64_t extra = paddedEntries > plain ? (paddedEntries - plain) * 4 : 0;
    const bool affordable = (paddedEntries + 4) * 4 + (4ull << 30) < freeB && extra <= (uint64_t)((double)freeB * ctx->listAlignShare);
    if (!affordable) return;
    void* p = nullptr;
    if (big_malloc(&p, (paddedEntries + 1 + 4) * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return; }
    T.dvalues = reinterpret_cast<uint64_t*>(p);
    T.listAlign = kListAlign; T.expectStore = paddedEntries; T.dvaluesCap = paddedEntries + 1;
    ctx->storesPlaced.fetch_add(1, std::memory_order_release);
}
// the bucket table of a single-part context whose mc_load_begin left it open (Mode T), for `nkeys` keys at the context's load factor;
// nkeys = 0: for all keys the part announced
int mcamd::allocate_buckets(mc_ctx* ctx, uint64_t nkeys)
{
    Part& T = ctx->parts[0];
    if (T.dbuckets) return MC_OK;
    if (nkeys == 0 || nkeys > T.expectKeys) nkeys = T.expectKeys;
    uint64_t nb = (uint64_t)((double)nkeys / (kSlotsPerBucket * (double)ctx->loadFactor)) + 2;
    nb += nb & 1;
    if (nb > 0xFFFFFFF0ull) return fail(ctx, MC_ERR_UNSUPPORTED, "table too large for 32-bit bucket index");
    T.nbuckets = (uint32_t)nb;
    HIP_TRY(ctx, big_malloc((void**)&T.dbuckets, (size_t)nb * sizeof(TableBucket)));
    HIP_TRY(ctx, hipMemsetAsync(T.dbuckets, 0, (size_t)nb * sizeof(TableBucket), ctx->stream));
    return MC_OK;
}
// the location store, with the first chunk that is loaded
int mcamd::allocate_values(mc_ctx* ctx)
{
    Part& T = ctx->parts[0];
    if (T.dvalues) return MC_OK;
    // (+ 4 entries: the filter reads the compact lists 16 bytes at a time, a list's last load may reach 3 entries past its end)
    HIP_TRY(ctx, big_malloc((void**)&T.dvalues, (T.dvaluesCap + 4) * (T.compact ? sizeof(uint32_t) : sizeof(uint64_t))));
    ctx->storesPlaced.fetch_add(1, std::memory_order_release);
    return MC_OK;
}

// One chunk of a single-part database whose batch arrays are already in device memory (keys u32 | sizes u8 | packed values).
// counters: [0] keys stored, [1] locations kept, [2] (as two u32) longest probe sequence | table-full flag.
int mcamd::load_chunk_device(mc_ctx* ctx, const uint32_t* dkeys, const uint8_t* dsizes, const uint8_t* dvals, uint32_t nb, uint64_t fileVals)
{
    if (!ctx || ctx->parts.size() != 1 || !ctx->parts[0].loading) return fail(ctx, MC_ERR_STATE, "load_chunk_device: single-part load only");
    Part& P = ctx->parts[0];
    if (P.keysLoaded + nb > P.expectKeys) return fail(ctx, MC_ERR_INVALID, "mc_load_batch: more keys than announced");
    if (fileVals >= (1ull << 32)) return fail(ctx, MC_ERR_INVALID, "load_chunk_device: chunk too large");
    if (int rcB = mcamd::allocate_buckets(ctx, 0)) return rcB;
    const uint32_t tb = ctx->cfg.target_id_bytes;
    const LoadFilter lf{ctx->cfg.max_locations_per_feature, ctx->cfg.remove_overpopulated, ctx->cfg.key_shard_index, ctx->cfg.key_shard_count, ctx->parts[0].listAlign};
    hipStream_t st = ctx->stream;
    int rc = 0;
    if ((rc = ensure(ctx, ctx->bLdFileSz, (size_t)nb * 4)) || (rc = ensure(ctx, ctx->bLdStoreSz, (size_t)nb * 4)) ||
        (rc = ensure(ctx, ctx->bLdFileOff, (size_t)(nb + 2) * 4)) || (rc = ensure(ctx, ctx->bLdStoreOff, (size_t)(nb + 2) * 4)) ||
        (rc = ensure(ctx, ctx->bLdScan, scan_tmp_bytes(nb + 1))))
        return rc;
    auto* fileSz = (uint32_t*)ctx->bLdFileSz.p; auto* storeSz = (uint32_t*)ctx->bLdStoreSz.p;
    auto* fileOff = (uint32_t*)ctx->bLdFileOff.p; auto* storeOff = (uint32_t*)ctx->bLdStoreOff.p;
    auto* counters = (unsigned long long*)ctx->bLdCounters.p;
    launch_table_prep(dkeys, dsizes, nb, lf, fileSz, storeSz, counters, st);
    launch_scan_u32(fileSz, 1, nb, fileOff, nullptr, ctx->bLdScan.p, st);
    launch_scan_u32(storeSz, 1, nb, storeOff, nullptr, ctx->bLdScan.p, st);
    uint32_t stored = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&stored, storeOff + nb, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, traced_sync(st));
    if ((rc = allocate_values(ctx))) return rc;
    if (P.valuesStored + stored > P.dvaluesCap) return fail(ctx, MC_ERR_INVALID, "mc_load_batch: more values than announced");
    const GwLayout gwl = P.compact ? GwLayout{ctx->dGwBase, ctx->gwTargets, ctx->gwGap} : GwLayout{};
    launch_table_insert(dkeys, dsizes, nb, lf, fileOff, storeOff, dvals, tb, P.valuesStored, P.dbuckets, P.nbuckets,
                        (unsigned int*)(counters + 2), (unsigned int*)(counters + 2) + 1, st, gwl, (unsigned int*)(counters + 3));
    if (P.compact)
        launch_table_values_compact(dkeys, dsizes, nb, lf, fileOff, storeOff, dvals, tb, fileVals, reinterpret_cast<uint32_t*>(P.dvalues) + P.valuesStored,
                                    gwl, (unsigned int*)(counters + 3), st);
    else
        launch_table_values(dkeys, dsizes, nb, lf, fileOff, storeOff, dvals, tb, fileVals, P.dvalues + P.valuesStored, st);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, traced_sync(st));                   // staging buffers are reused by the next chunk
    P.valuesStored += stored;
    P.keysLoaded += nb;
    return MC_OK;
}

extern "C" {

void mc_config_default(mc_config* c)
{
    std::memset(c, 0, sizeof(*c));
    c->device = 0;
    c->kmerlen = 16; c->sketchlen = 16; c->winlen = 127; c->winstride = 112;   // options.hpp:102, options.cpp:625
    c->max_candidates = 2;                                                      // options.hpp:257
    c->target_id_bytes = 4;
    c->num_parts = 1;
    c->max_locations_per_feature = 0;
    c->remove_overpopulated = 0;
    c->max_load_factor = 0.3f;   // few full buckets => most lookups end after one line (measured: 0.5 -> 0.3 is 8 % on the probing kernel)
                                 // (the reference CPU map uses 0.8, host_hashmap.hpp:159-161; -max-load-fac overrides)
    c->num_slots = 1;
    c->slot_max_queries = 1u << 16;
    c->slot_max_chars = 1u << 24;
    c->copy_allhits = 0;
    c->single_part = -1;
    c->key_shard_index = 0; c->key_shard_count = 1;
    c->target_shard_index = 0; c->target_shard_count = 1;
}

const char* mc_last_error(const mc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

static void co_dispatch(mc_ctx* ctx, mcamd::CoDispatcher* D);
static void co_run(mc_ctx* ctx, mcamd::CoDispatcher* D, const std::vector<uint32_t>& mine, int lowest);

int mc_create(const mc_config* cfg, mc_ctx** out)
{
    if (!cfg || !out) return fail(nullptr, MC_ERR_INVALID, "mc_create: null argument");
    *out = nullptr;
    if (cfg->kmerlen < 1 || cfg->kmerlen > 16) return fail(nullptr, MC_ERR_UNSUPPORTED, "kmerlen must be 1..16 (kmer_type = uint32_t)");
    if (cfg->sketchlen < 1 || cfg->sketchlen > kMaxSketch) return fail(nullptr, MC_ERR_UNSUPPORTED, "sketchlen must be 1..32");
    if (cfg->winlen < cfg->kmerlen || cfg->winlen > kMaxWinLen) return fail(nullptr, MC_ERR_UNSUPPORTED, "winlen must be kmerlen..1024");
    if (cfg->winstride < 1) return fail(nullptr, MC_ERR_INVALID, "winstride must be >= 1");
    if (cfg->target_id_bytes != 2 && cfg->target_id_bytes != 4) return fail(nullptr, MC_ERR_INVALID, "target_id_bytes must be 2 or 4");
    if (cfg->num_parts < 1 || cfg->num_parts > 255) return fail(nullptr, MC_ERR_UNSUPPORTED, "num_parts must be 1..255");
    if (cfg->max_candidates < 1) return fail(nullptr, MC_ERR_INVALID, "max_candidates must be >= 1");
    if (cfg->key_shard_count > 1 && cfg->key_shard_index >= cfg->key_shard_count) return fail(nullptr, MC_ERR_INVALID, "key_shard_index out of range");
    if (cfg->target_shard_count > 1) {
        if (cfg->target_shard_index >= cfg->target_shard_count) return fail(nullptr, MC_ERR_INVALID, "target_shard_index out of range");
        if (cfg->key_shard_count > 1) return fail(nullptr, MC_ERR_UNSUPPORTED, "target shards and key shards cannot be combined");
        if (cfg->num_parts > 1) return fail(nullptr, MC_ERR_UNSUPPORTED, "target shards need a single part per context (single_part)");
    }

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) return fail(nullptr, MC_ERR_HIP, "no usable HIP device (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MC_ERR_INVALID, "device ordinal out of range");

    auto* ctx = new mc_ctx;
    ctx->cfg = *cfg;
    ctx->device = cfg->device;
    ctx->querySketch = SketchParams{cfg->kmerlen, cfg->sketchlen, cfg->winlen, cfg->winstride};
    ctx->targetSketch = ctx->querySketch;
    ctx->loadFactor = (cfg->max_load_factor > 0.05f && cfg->max_load_factor <= 0.99f) ? cfg->max_load_factor : 0.3f;
    ctx->parts.resize(cfg->num_parts);
    if (const char* e = std::getenv("MC_NO_LANE_PATH")) ctx->useLanePath = !(e[0] == '1');   // debugging aid
    if (const char* e = std::getenv("MC_LANE_FUSION")) ctx->fuseLane = e[0] == '1' ? 1 : 0;   // (default: by table size)
    ctx->listAlignWant = mcamd::open_hints().listAlign; ctx->listAlignShare = mcamd::open_hints().listAlignShare;
    ctx->directWant = mcamd::open_hints().directIndex;
    if (const char* e = std::getenv("MC_DIRECT_INDEX")) ctx->directWant = e[0] == '1' ? 1 : 0;
    if (const char* e = std::getenv("MC_LIST_ALIGN")) ctx->listAlignWant = e[0] == '1' ? 1 : 0;   // (default: where the padded store is affordable)
    if (const char* e = std::getenv("MC_GW_FUSE")) ctx->gwFuse = e[0] == '5' ? 5 : e[0] == '6' ? 6 : e[0] == '1' ? 1 : 0;   // (5 / 6: the fused kernel's five- / six-waves-per-SIMD instances)
    if (const char* e = std::getenv("MC_QUAD_LOOKUP")) ctx->quadLookup = e[0] == '1' ? 1 : 0;   // tests
    if (const char* e = std::getenv("MC_COMPACT_LOCATIONS")) ctx->compactAllowed = e[0] != '0';   // tests / tuning
    if (const char* e = std::getenv("MC_BIG_MIN")) ctx->bigMin = (uint32_t)std::max(0, std::atoi(e));   // tests / tuning
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail(nullptr, MC_ERR_HIP, "cannot create HIP stream");
    }
    ctx->pipe0.stream = ctx->stream;
    // host slots
    ctx->slots.resize(cfg->num_slots);
    const size_t K = cfg->max_candidates;
    for (auto& s : ctx->slots) {
        const size_t nq = cfg->slot_max_queries, nc = (size_t)cfg->slot_max_chars + 16;
        bool ok = hipHostMalloc((void**)&s.hseq, nc) == hipSuccess && hipHostMalloc((void**)&s.hqinfo, nq * 16) == hipSuccess &&
                  hipHostMalloc((void**)&s.hmaxwin, nq * 4) == hipSuccess && hipMalloc((void**)&s.dseq, nc) == hipSuccess &&
                  hipMalloc((void**)&s.dqinfo, nq * 16) == hipSuccess && hipMalloc((void**)&s.dmaxwin, nq * 4) == hipSuccess &&
                  hipHostMalloc((void**)&s.hcands, nq * K * sizeof(mc_candidate)) == hipSuccess &&
                  hipHostMalloc((void**)&s.hqstat, nq * sizeof(QueryStat)) == hipSuccess &&
                  hipHostMalloc((void**)&s.hhitcounts, nq * 4) == hipSuccess &&
                  hipHostMalloc((void**)&s.hhitoff, (nq + 1) * 8) == hipSuccess && hipEventCreate(&s.done) == hipSuccess;
        if (!ok) { mc_destroy(ctx); return fail(nullptr, MC_ERR_NOMEM, "cannot allocate slot buffers"); }
    }
    uint32_t npipes = std::min<uint32_t>(cfg->num_slots, 8);
    if (const char* e = std::getenv("MC_PIPES")) npipes = std::max(1, std::atoi(e));
    // Slots that are submitted side by side go to the device as ONE batch (slot coalescer, below): several slots, top candidates only.
    // MC_SLOT_COALESCE=0: every slot its own batch on a pipe it borrows, as before round 6.
    // Slots of 8 192 reads and fewer (the reference's 4 096): there a batch per slot is bound by its launches (at 16 384 reads per slot united and
    // apart are level: 4 133 / 5 402 / 6 136 / 7 279 against 3 479 / 6 000 / 6 537 / 6 535 Mreads/min with 4 / 8 / 16 / 32 threads).  Larger slots keep a batch
    // each on a pipe they borrow -- united they gain nothing (10 554 against 10 618 Mreads/min at 65 536 reads per slot) and `mcq`, whose 32
    // workers run under AMD_DIRECT_DISPATCH=0, lost a third of its query phase (97 against 68 ms per 10^7 reads at 30 Gbp, profiles/r06_e2e_matrix02.json).
    // MC_SLOT_COALESCE=1 / 0 forces either.
    ctx->coalesce = cfg->num_slots >= 2 && !cfg->copy_allhits && cfg->slot_max_queries <= 8192;
    if (const char* e = std::getenv("MC_SLOT_COALESCE")) ctx->coalesce = cfg->num_slots >= 2 && !cfg->copy_allhits && e[0] != '0';
    if (ctx->coalesce) {
        // dispatchers: one united batch each in flight: few submitters find a free one at once (their slots go alone), many find them busy and
        // their slots wait together.  FIVE: the runtime runs a process' streams on four hardware queues, and pipes that share one take turns
        // (a united batch's chain of ~25 steps: 715 / 770 / 1 014 us with 4 / 6 / 8 pipes).  Same box, 4 096-read slots, 8 / 16 / 32 threads:
        // 4 pipes 2 960 / 4 010-4 070 / 5 400-5 470, five 3 350 / 4 530-4 550 / 5 790-5 920, six 3 070-3 140 / 3 940-4 120 / 5 380-5 490,
        // eight 3 030-3 060 / 3 930-3 960 / 5 320-5 400 Mreads/min (docs/LAB_NOTEBOOK_r06.md section 5b).  MC_PIPES / MC_SLOT_DISPATCHERS override.
        if (!std::getenv("MC_PIPES")) npipes = std::min<uint32_t>(npipes, 5);
        if (const char* e = std::getenv("MC_SLOT_DISPATCHERS")) npipes = (uint32_t)std::min(8, std::max(1, std::atoi(e)));
        // what a united batch may hold: up to 2^18 reads (beyond that the kernels run at their large-batch rate anyway), character offsets are 32 bits
        ctx->coMaxQueries = (uint32_t)std::max<uint64_t>(cfg->slot_max_queries, std::min<uint64_t>(1u << 18, (uint64_t)cfg->num_slots * cfg->slot_max_queries));
        ctx->coMaxChars = std::max<uint64_t>(cfg->slot_max_chars, std::min<uint64_t>(1ull << 30, (uint64_t)cfg->num_slots * cfg->slot_max_chars));
    }
    for (uint32_t i = 0; i < npipes && i < cfg->num_slots; ++i) {
        Pipe* p = new Pipe;
        ctx->pipes.push_back(p);
        if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) { mc_destroy(ctx); return fail(nullptr, MC_ERR_HIP, "cannot create HIP stream"); }
        if (!ctx->coalesce) ctx->freePipes.push_back(p);
    }
    if (ctx->coalesce) {
        for (Pipe* p : ctx->pipes) {
            auto* d = new mcamd::CoDispatcher;
            d->pipe = p;
            bool ok = true;
            for (int k = 0; k < 2 && ok; ++k)
                ok = hipHostMalloc((void**)&d->hq[k], (size_t)ctx->coMaxQueries * 20) == hipSuccess &&
                     hipEventCreateWithFlags(&d->staged[k], hipEventDisableTiming) == hipSuccess;
            for (int k = 0; k < 4 && ok; ++k) ok = hipEventCreateWithFlags(&d->done[k], hipEventDisableTiming) == hipSuccess;
            ctx->coDisp.push_back(d);
            if (!ok) { mc_destroy(ctx); return fail(nullptr, MC_ERR_NOMEM, "cannot allocate the slot coalescer's staging"); }
        }
        ctx->coFree = ctx->coDisp;
        for (auto* d : ctx->coDisp) d->th = std::thread(co_dispatch, ctx, d);
    }
    *out = ctx;
    return MC_OK;
}

void mc_destroy(mc_ctx* ctx)
{
    if (g_submitTrace && ctx && g_trace[5].load())
        std::fprintf(stderr, "mc submit trace: %llu batches; ms summed over the submitting threads: waiting for a pipe %.1f, H2D enqueue %.1f, query %.1f (of it %.1f in %llu hipStreamSynchronize calls), D2H enqueue %.1f\n",
                     (unsigned long long)g_trace[5].load(), g_trace[0] / 1e6, g_trace[1] / 1e6, g_trace[2] / 1e6, g_trace[3] / 1e6, (unsigned long long)g_trace[6].load(), g_trace[4] / 1e6);
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (!ctx->coDisp.empty()) {                                     // the slot coalescer's dispatchers: let them finish what they carry, then go
        { std::lock_guard<std::mutex> l(ctx->coMu); ctx->coStop = true; }
        ctx->coCv.notify_all();
        for (auto* d : ctx->coDisp) if (d->th.joinable()) d->th.join();
        for (auto* d : ctx->coDisp) {
            if (d->pipe && d->pipe->stream) (void)hipStreamSynchronize(d->pipe->stream);
            for (int k = 0; k < 2; ++k) { if (d->hq[k]) (void)hipHostFree(d->hq[k]); if (d->hmw[k]) (void)hipHostFree(d->hmw[k]); if (d->staged[k]) (void)hipEventDestroy(d->staged[k]); }
            for (int k = 0; k < 4; ++k) if (d->done[k]) (void)hipEventDestroy(d->done[k]);
            for (DevBuf* b : {&d->dseq, &d->dqinfo, &d->dmaxwin}) if (b->p) (void)hipFree(b->p);
            delete d;
        }
        ctx->coDisp.clear();
    }
    // every stream that may still run kernels on the tables: the context's, both pipes', the slots' (callers' own streams: theirs to wait for)
    if (ctx->pipe0.tail.pending || ctx->pipe1.tail.pending) { ctx->pipe0.tail.pending = false; ctx->pipe1.tail.pending = false; }   // (a deferred tail nobody asked for is dropped)
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->pipe0.stream) (void)hipStreamSynchronize(ctx->pipe0.stream);
    if (ctx->pipe1.stream) (void)hipStreamSynchronize(ctx->pipe1.stream);
    for (Pipe* p : ctx->pipes) if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (ctx->buildHold) { big_cache_hold(-1); ctx->buildHold = false; }   // (a table build that was abandoned before mc_build_table_end)
    for (auto& p : ctx->parts) { if (p.dbuckets) (void)big_free(p.dbuckets); if (p.dvalues) (void)big_free(p.dvalues); if (p.ddirect) (void)big_free(p.ddirect); }
    for (auto& kv : ctx->taxkeyDev) (void)hipFree(kv.second);
    if (ctx->dGwBase) (void)hipFree(ctx->dGwBase);
    if (ctx->dGwDir) (void)hipFree(ctx->dGwDir);
    auto free_pipe = [](Pipe& P, bool ownStream) {
        DevBuf* pb[] = {&P.bWinCount, &P.bWinOff, &P.bFeatures, &P.bPsize, &P.bPpay, &P.bQstat, &P.bHitOff, &P.bHits, &P.bCscr, &P.bCscr2,
                        &P.bScan, &P.bStats, &P.bCands, &P.bScanIn, &P.bQflag, &P.bMid, &P.bChunkList, &P.bBigPool, &P.bSliceFill, &P.bBigPool2, &P.bSortTmp, &P.bSide, &P.bNumbers, &P.bCounts, &P.bOrder};
        if (P.stream) (void)hipStreamSynchronize(P.stream);
        if (P.hTotal) (void)hipHostFree(P.hTotal);
        if (P.tail.mainDone) (void)hipEventDestroy(P.tail.mainDone);
        if (P.sortSide.stream) { (void)hipStreamSynchronize(P.sortSide.stream); (void)hipStreamDestroy(P.sortSide.stream); }
        if (P.sortSide.fork) (void)hipEventDestroy(P.sortSide.fork);
        if (P.sortSide.join) (void)hipEventDestroy(P.sortSide.join);
        for (auto* b : pb) if (b->p) (void)hipFree(b->p);
        if (ownStream && P.stream) (void)hipStreamDestroy(P.stream);
    };
    free_pipe(ctx->pipe0, false);
    free_pipe(ctx->pipe1, true);
    for (Pipe* p : ctx->pipes) { free_pipe(*p, true); delete p; }
    ctx->pipes.clear(); ctx->freePipes.clear();
    DevBuf* bufs[] = {&ctx->bLdKeys, &ctx->bLdSizes, &ctx->bLdVals, &ctx->bLdFileSz, &ctx->bLdStoreSz, &ctx->bLdFileOff, &ctx->bLdStoreOff,
                      &ctx->bLdScan, &ctx->bLdCounters};
    for (auto* b : bufs) if (b->p) (void)hipFree(b->p);
    for (auto& s : ctx->slots) {
        if (s.hseq) (void)hipHostFree(s.hseq); if (s.hqinfo) (void)hipHostFree(s.hqinfo); if (s.hmaxwin) (void)hipHostFree(s.hmaxwin);
        if (s.dseq) (void)hipFree(s.dseq); if (s.dqinfo) (void)hipFree(s.dqinfo); if (s.dmaxwin) (void)hipFree(s.dmaxwin);
        if (s.hcands) (void)hipHostFree(s.hcands); if (s.hqstat) (void)hipHostFree(s.hqstat);
        if (s.hhitcounts) (void)hipHostFree(s.hhitcounts); if (s.hhitoff) (void)hipHostFree(s.hhitoff);
        if (s.hhits) (void)hipHostFree(s.hhits); if (s.done) (void)hipEventDestroy(s.done);
    }
    collect_timers(ctx);
    for (auto e : ctx->eventPool) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ------------------------------------------------------------------------------------------------
// table loading
// ------------------------------------------------------------------------------------------------
// All parts of a database share ONE device table (state kept in parts[0]): a feature that occurs in several
// parts gets the concatenation of its per-part buckets, with the part number folded into the target id
// ((part << 24) | target), so that sorting by the stored key yields "per-part sorted lists concatenated in
// part order" -- the intended result of host_hashmap.hpp:695-723 -- with a single lookup per feature.
static int allocate_table(mc_ctx* ctx)
{
    Part& T = ctx->parts[0];
    uint64_t nkeys = 0, nvalues = 0;
    for (auto& p : ctx->parts) { nkeys += p.expectKeys; nvalues += p.expectValues; }
    if (ctx->cfg.key_shard_count > 1) {
        // Mode K: this context keeps ~1/count of the keys (hashed: near uniform) and of the locations (lumpier: wider margin;
        // running out fails loudly in load_chunk_device)
        if (ctx->parts.size() > 1) return fail(ctx, MC_ERR_UNSUPPORTED, "key sharding needs a single-part database");
        const uint64_t c = ctx->cfg.key_shard_count;
        nkeys = std::min<uint64_t>(nkeys, nkeys / c + nkeys / (8 * c) + 4096);
        nvalues = std::min<uint64_t>(nvalues, nvalues / c + nvalues / (3 * c) + (1u << 16));
    }
    // (Mode T: the buckets wait for the loader's estimate of the keys that have a location in the range, mcamd::allocate_buckets)
    const bool bucketsLater = ctx->parts.size() == 1 && ctx->cfg.target_shard_count > 1;
    uint64_t nb = (uint64_t)((double)nkeys / (kSlotsPerBucket * (double)ctx->loadFactor)) + 2;
    nb += nb & 1;                                            // two buckets per line
    if (nb > 0xFFFFFFF0ull) return fail(ctx, MC_ERR_UNSUPPORTED, "table too large for 32-bit bucket index");
    T.nbuckets = (uint32_t)nb;
    if (ctx->parts.size() == 1) {
        // one part: the table is built on the device, batch by batch (table_build.hip)
        T.dvaluesCap = nvalues + 1;
        T.compact = false;
        if (!ctx->targetWindows.empty() && ctx->compactAllowed) {
            // global window numbers: gwBase[0] = gap, gwBase[t + 1] = gwBase[t] + windows(t) + gap; 0xFFFFFFFF stays free (the kernels' "no location")
            uint32_t gap = kGwGap;
            if (const char* e = std::getenv("MC_GW_GAP")) gap = (uint32_t)std::max(8, std::atoi(e));   // tests: reads whose window range exceeds the gap
            const size_t nt = ctx->targetWindows.size();
            uint64_t total = gap;
            for (uint32_t w : ctx->targetWindows) total += (uint64_t)w + gap;
            if (total < 0xFFFFFFFFull) {
                std::vector<uint32_t> base(nt + 1);
                base[0] = gap;
                for (size_t t = 0; t < nt; ++t) base[t + 1] = base[t] + ctx->targetWindows[t] + gap;
                uint32_t shift = 6;
                while (((total >> shift) + 2) > (1ull << 22)) ++shift;            // directory of at most 4 M entries
                const size_t nd = (size_t)(total >> shift) + 2;
                std::vector<uint32_t> dir(nd);
                size_t t = 0;
                for (size_t blk = 0; blk < nd; ++blk) {                            // dir[blk] = target whose numbers (gap included) hold max(blk << shift, gap)
                    const uint64_t g = std::max<uint64_t>((uint64_t)blk << shift, gap);
                    while (t + 1 < nt && g >= base[t + 1]) ++t;
                    dir[blk] = (uint32_t)t;
                }
                HIP_TRY(ctx, hipMalloc((void**)&ctx->dGwBase, (nt + 1) * 4));
                HIP_TRY(ctx, hipMalloc((void**)&ctx->dGwDir, nd * 4));
                HIP_TRY(ctx, hipMemcpy(ctx->dGwBase, base.data(), (nt + 1) * 4, hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy(ctx->dGwDir, dir.data(), nd * 4, hipMemcpyHostToDevice));
                ctx->gwDirShift = shift; ctx->gwGap = gap; ctx->gwTargets = (uint32_t)nt; ctx->gwBits = bits_for((uint32_t)total);
                T.compact = true;
            }
        }
        // (the location store is allocated with the first chunk -- allocate_values: a loader that knows the lists' sizes announces the
        // padded total first, mcamd::announce_store)
        // (big_malloc: the tables of a part group come back from the tables of the group before it, devcache.h)
        if (!bucketsLater) {
            HIP_TRY(ctx, big_malloc((void**)&T.dbuckets, (size_t)nb * sizeof(TableBucket)));
            HIP_TRY(ctx, hipMemsetAsync(T.dbuckets, 0, (size_t)nb * sizeof(TableBucket), ctx->stream));
        }
        int rc = ensure(ctx, ctx->bLdCounters, 4 * sizeof(unsigned long long));
        if (rc) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->bLdCounters.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
    } else {
        T.hbuckets.assign((size_t)nb, TableBucket{});
    }
    return MC_OK;
}

// One batch of the file (host memory): upload in chunks, insert on the device.
static int load_batch_device(mc_ctx* ctx, const uint32_t* keys, const uint8_t* sizes, const uint8_t* values, uint64_t n)
{
    const uint32_t vb = 4 + ctx->cfg.target_id_bytes;
    hipStream_t st = ctx->stream;
    const uint64_t kChunk = 1ull << 22;                       // keys per launch: file values of a chunk stay below 2^32
    for (uint64_t done = 0; done < n;) {
        const uint32_t nb = (uint32_t)std::min<uint64_t>(kChunk, n - done);
        uint64_t fileVals = 0;
        for (uint32_t i = 0; i < nb; ++i) fileVals += sizes[done + i];
        int rc = 0;
        if ((rc = ensure(ctx, ctx->bLdKeys, (size_t)nb * 4)) || (rc = ensure(ctx, ctx->bLdSizes, nb)) ||
            (rc = ensure(ctx, ctx->bLdVals, fileVals * vb + 16)))
            return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->bLdKeys.p, keys + done, (size_t)nb * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->bLdSizes.p, sizes + done, nb, hipMemcpyHostToDevice, st));
        if (fileVals) HIP_TRY(ctx, hipMemcpyAsync(ctx->bLdVals.p, values, fileVals * vb, hipMemcpyHostToDevice, st));
        rc = load_chunk_device(ctx, (const uint32_t*)ctx->bLdKeys.p, (const uint8_t*)ctx->bLdSizes.p, (const uint8_t*)ctx->bLdVals.p, nb, fileVals);
        if (rc) return rc;
        values += fileVals * vb;
        done += nb;
    }
    return MC_OK;
}

int mc_load_target_windows(mc_ctx* ctx, const uint32_t* windows, uint64_t numTargets)
{
    if (!ctx || (!windows && numTargets)) return MC_ERR_INVALID;
    for (auto& p : ctx->parts) if (p.announced) return fail(ctx, MC_ERR_STATE, "mc_load_target_windows: call it before mc_load_begin");
    if (numTargets >= 0xFFFFFFFFull) return fail(ctx, MC_ERR_INVALID, "mc_load_target_windows: too many targets");
    ctx->targetWindows.assign(windows, windows + numTargets);
    return MC_OK;
}

// the same announcement when only bounds are known: targets 0 .. maxTarget, each with maxWindow + 1 windows
int mc_load_location_range(mc_ctx* ctx, uint32_t maxTarget, uint32_t maxWindow)
{
    if (!ctx) return MC_ERR_INVALID;
    for (auto& p : ctx->parts) if (p.announced) return fail(ctx, MC_ERR_STATE, "mc_load_location_range: call it before mc_load_begin");
    ctx->targetWindows.clear();
    if (((uint64_t)maxTarget + 1) * ((uint64_t)maxWindow + 1 + kGwGap) < 0xFFFFFFFFull) ctx->targetWindows.assign((size_t)maxTarget + 1, maxWindow + 1);
    return MC_OK;
}

int mc_table_layout(const mc_ctx* ctx, uint64_t layout[4])
{
    if (!ctx || !layout || ctx->parts.empty()) return MC_ERR_INVALID;
    const Part& T = ctx->parts[0];
    layout[0] = (T.compact ? 4 : 8) | (T.ddirect ? 1ull << 32 : 0); layout[1] = (T.compact ? ctx->gwGap : 0) | ((uint64_t)T.listAlign << 32); layout[2] = T.nbuckets; layout[3] = T.valuesStored;
    return MC_OK;
}

int mc_target_range(const mc_ctx* ctx, uint64_t range[4])
{
    if (!ctx || !range) return MC_ERR_INVALID;
    range[0] = ctx->tgtRangeSet ? ctx->tgtLo : 0;
    range[1] = ctx->tgtRangeSet ? ctx->tgtHi : ctx->targetCount;
    range[2] = ctx->parts.empty() ? 0 : ctx->parts[0].keysStored;
    range[3] = 0;
    for (const auto& p : ctx->parts) range[3] += p.locations;
    return MC_OK;
}

int mc_load_begin(mc_ctx* ctx, uint32_t part, uint64_t nkeys, uint64_t nvalues)
{
    if (!ctx) return MC_ERR_INVALID;
    if (part >= ctx->parts.size()) return fail(ctx, MC_ERR_INVALID, "mc_load_begin: part out of range");
    Part& P = ctx->parts[part];
    if (P.announced) return fail(ctx, MC_ERR_STATE, "mc_load_begin: part already announced");
    if (ctx->cfg.target_shard_count > 1 && !ctx->tgtRangeSet)
        return fail(ctx, MC_ERR_UNSUPPORTED, "target shards are cut from a database file (mc_open_database), not from mc_load_batch arrays");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    P.expectKeys = nkeys; P.expectValues = nvalues;
    P.announced = true; P.loading = true;
    if (ctx->parts.size() == 1) return allocate_table(ctx);
    return MC_OK;
}

int mc_load_batch(mc_ctx* ctx, uint32_t part, const uint32_t* keys, const uint8_t* sizes, const void* values, uint64_t n)
{
    if (!ctx) return MC_ERR_INVALID;
    if (part >= ctx->parts.size() || !ctx->parts[part].loading) return fail(ctx, MC_ERR_STATE, "mc_load_batch: call mc_load_begin first");
    Part& P = ctx->parts[part];
    Part& T = ctx->parts[0];
    const bool multi = ctx->parts.size() > 1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (multi && T.hbuckets.empty()) {
        for (auto& q : ctx->parts)
            if (!q.announced) return fail(ctx, MC_ERR_STATE, "multi-part load: call mc_load_begin for EVERY part before the first mc_load_batch");
        int rc = allocate_table(ctx);
        if (rc) return rc;
    }
    if (P.keysLoaded + n > P.expectKeys) return fail(ctx, MC_ERR_INVALID, "mc_load_batch: more keys than announced");
    if (!multi) return load_batch_device(ctx, keys, sizes, static_cast<const uint8_t*>(values), n);
    const uint32_t tb = ctx->cfg.target_id_bytes, vb = 4 + tb;
    const uint32_t maxLocs = ctx->cfg.max_locations_per_feature;
    const uint32_t rmOver = ctx->cfg.remove_overpopulated;
    const uint8_t* vp = static_cast<const uint8_t*>(values);
    // several parts: buckets of one feature are merged across parts on the host, the table goes up in mc_load_end
    std::vector<uint64_t>& store = ctx->hvalues;
    const uint64_t storeBase = 0;
    bool badTarget = false;
    auto decode = [&](const uint8_t* p) -> uint64_t {
        uint32_t win; std::memcpy(&win, p, 4);
        uint32_t tgt = 0;
        if (tb == 2) { uint16_t t; std::memcpy(&t, p + 4, 2); tgt = t; } else std::memcpy(&tgt, p + 4, 4);
        if (multi) { badTarget = badTarget || tgt >= (1u << 24); tgt |= part << 24; }
        return ((uint64_t)tgt << 32) | win;
    };
    if (P.keysLoaded + n > P.expectKeys) return fail(ctx, MC_ERR_INVALID, "mc_load_batch: more keys than announced");
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t fileSize = sizes[i];
        uint32_t size = fileSize;
        if (rmOver && size > rmOver) size = 0;                 // bucket emptied, key kept without values = never matches
        if (maxLocs && size > maxLocs) size = maxLocs;         // keep the FIRST n values
        if (size > 0) {
            // walk the key's probe sequence: an earlier part may already hold the key; otherwise it goes into
            // the first bucket with a free slot
            const uint32_t key = keys[i];
            const uint32_t home = (uint32_t)(((uint64_t)mix32(key) * T.nbuckets) >> 32);
            uint32_t cur = home, probe = 1;
            TableBucket* grp = nullptr;
            uint32_t slot = kSlotsPerBucket;
            bool found = false;
            for (;; ++probe) {
                grp = &T.hbuckets[cur];
                uint32_t freeSlot = kSlotsPerBucket;
                for (uint32_t j = 0; j < kSlotsPerBucket; ++j) {
                    if (!grp->size[j]) { if (freeSlot == kSlotsPerBucket) freeSlot = j; }
                    else if (multi && grp->key[j] == key) { slot = j; found = true; break; }
                }
                if (found) break;
                if (freeSlot < kSlotsPerBucket) { slot = freeSlot; break; }
                if (probe > T.nbuckets) return fail(ctx, MC_ERR_NOMEM, "hash table full");
                cur = next_bucket(home, cur, probe, T.nbuckets);
            }
            if (probe > T.maxProbe) T.maxProbe = probe;
            if (!found) {
                grp->key[slot] = key;
                grp->size[slot] = (uint16_t)size;
                if (size == 1) grp->payload[slot] = decode(vp);
                else {
                    grp->payload[slot] = storeBase + store.size();
                    for (uint32_t t = 0; t < size; ++t) store.push_back(decode(vp + (size_t)t * vb));
                }
            } else {
                // same feature in an earlier part: new bucket = old locations followed by this part's
                const uint32_t s0 = grp->size[slot];
                if (s0 + size > 0xFFFFu) return fail(ctx, MC_ERR_UNSUPPORTED, "merged bucket exceeds 65535 locations");
                const uint64_t p0 = grp->payload[slot];
                const uint64_t off = store.size();
                if (s0 == 1) store.push_back(p0);
                else for (uint32_t t = 0; t < s0; ++t) { const uint64_t v = store[p0 + t]; store.push_back(v); }
                for (uint32_t t = 0; t < size; ++t) store.push_back(decode(vp + (size_t)t * vb));
                grp->size[slot] = (uint16_t)(s0 + size);
                grp->payload[slot] = off;
                T.keysStored--;                                   // counted again below
            }
            P.locations += size;
            T.keysStored++;
        }
        vp += (size_t)fileSize * vb;
    }
    if (badTarget) return fail(ctx, MC_ERR_UNSUPPORTED, "multi-part databases need target ids < 2^24");
    P.keysLoaded += n;
    return MC_OK;
}

// The direct-address index of a finished single-part table (kernels.h DeviceTable::direct): 2^32 entries of 8 bytes = 32 GiB beside the
// buckets.  Built where the lookups are bound by random requests -- bucket tables of 8 GiB and more -- and the device has the room
// (the index, and 40 GB more for the batches' workspaces); "direct_index" 1 / MC_DIRECT_INDEX=1 asks for it whatever the table's
// size, 0 never.  A payload that does not fit its 48 bits (targets or windows beyond 2^24 in a single location) drops it again.
static int build_direct_index(mc_ctx* ctx)
{
    Part& T = ctx->parts[0];
    if (T.ddirect || !T.dbuckets || ctx->directWant == 0) return MC_OK;
    if (ctx->directWant < 0 && (uint64_t)T.nbuckets * sizeof(TableBucket) < (8ull << 30)) return MC_OK;
    const size_t bytes = kDirectEntries * 8;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return MC_OK;
    if (freeB < bytes + (ctx->directWant > 0 ? (4ull << 30) : (40ull << 30))) return MC_OK;
    void* p = nullptr;
    if (big_malloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return MC_OK; }
    unsigned int* dflag = nullptr;
    unsigned int flag = 1;
    hipStream_t st = ctx->stream;
    if (hipMalloc((void**)&dflag, 4) == hipSuccess && hipMemsetAsync(dflag, 0, 4, st) == hipSuccess && hipMemsetAsync(p, 0, bytes, st) == hipSuccess) {
        launch_direct_index(T.dbuckets, T.nbuckets, (uint64_t*)p, dflag, st);
        if (hipMemcpyAsync(&flag, dflag, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) flag = 1;
    }
    if (dflag) (void)hipFree(dflag);
    if (flag) { (void)hipGetLastError(); (void)big_free(p); return MC_OK; }
    T.ddirect = (uint64_t*)p;
    return MC_OK;
}

// A table that came through mc_load_* (not mc_open_database, whose loader reserves beside the file load): the slot pipes' workspaces are
// sized now, once, for full (united) batches.  Left to the first batches every pipe grew its ~25 buffers batch by batch -- a hipFree (which waits
// for the whole device) and a hipMalloc each: the first seconds of 32 threads on a fresh context ran at 3-100 Mreads/min instead of 5 000.
// (A failure here is not one: the first batch asks again.)
static void reserve_pipes_after_load(mc_ctx* ctx)
{
    if (ctx->pipes.empty() || ctx->reserveByLoader.load(std::memory_order_acquire) || std::getenv("MC_NO_RESERVE")) return;
    uint64_t locs = 0;
    for (auto& p : ctx->parts) locs += p.locations;
    std::string keep = ctx->err;
    (void)mcamd::reserve_slot_pipes(ctx, locs, ctx->parts[0].keysStored, false);
    (void)hipGetLastError();
    ctx->err = keep;
}

int mc_load_end(mc_ctx* ctx, uint32_t part)
{
    if (!ctx) return MC_ERR_INVALID;
    if (part >= ctx->parts.size() || !ctx->parts[part].loading) return fail(ctx, MC_ERR_STATE, "mc_load_end: nothing being loaded");
    Part& P = ctx->parts[part];
    Part& T = ctx->parts[0];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    P.loading = false; P.ready = true;
    for (auto& q : ctx->parts) if (!q.ready) return MC_OK;     // the table goes to the device when the last part is in
    if (ctx->parts.size() == 1) {                              // built on the device: fetch the counters, drop the staging
        unsigned long long c[4] = {0, 0, 0, 0};
        HIP_TRY(ctx, hipMemcpy(c, ctx->bLdCounters.p, sizeof(c), hipMemcpyDeviceToHost));
        T.keysStored = c[0]; P.locations = c[1];
        T.maxProbe = std::max<uint32_t>(1u, (uint32_t)(c[2] & 0xFFFFFFFFull));
        const bool full = (c[2] >> 32) != 0;
        DevBuf* st[] = {&ctx->bLdKeys, &ctx->bLdSizes, &ctx->bLdVals, &ctx->bLdFileSz, &ctx->bLdStoreSz, &ctx->bLdFileOff, &ctx->bLdStoreOff,
                        &ctx->bLdScan};
        for (auto* b : st) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
        if (full) return fail(ctx, MC_ERR_NOMEM, "hash table full");
        if (c[3]) ctx->locRangeViolated = true;
        if (c[3]) return fail(ctx, MC_ERR_INVALID, "a location lies outside the range announced with mc_load_target_windows / mc_load_location_range");
        ctx->tableReady = true;
        (void)build_direct_index(ctx);                            // (a lookup structure beside the buckets: not having it is not an error)
        reserve_pipes_after_load(ctx);
        return MC_OK;
    }
    if (T.hbuckets.empty()) { int rc = allocate_table(ctx); if (rc) return rc; }
    if (ctx->parts.size() > 1) {
        T.dvaluesCap = ctx->hvalues.size() + 1;
        HIP_TRY(ctx, hipMalloc((void**)&T.dvalues, T.dvaluesCap * sizeof(uint64_t)));
        if (!ctx->hvalues.empty())
            HIP_TRY(ctx, hipMemcpy(T.dvalues, ctx->hvalues.data(), ctx->hvalues.size() * 8, hipMemcpyHostToDevice));
        T.valuesStored = ctx->hvalues.size();
        std::vector<uint64_t>().swap(ctx->hvalues);
    }
    const size_t bytes = T.hbuckets.size() * sizeof(TableBucket);
    HIP_TRY(ctx, hipMalloc((void**)&T.dbuckets, bytes));
    HIP_TRY(ctx, hipMemcpy(T.dbuckets, T.hbuckets.data(), bytes, hipMemcpyHostToDevice));
    std::vector<TableBucket>().swap(T.hbuckets);
    ctx->tableReady = true;
    reserve_pipes_after_load(ctx);
    return MC_OK;
}

int mc_set_lineages(mc_ctx* ctx, const uint32_t* lin, uint64_t numTargets)
{
    if (!ctx || !lin) return MC_ERR_INVALID;
    for (uint64_t i = 0; i < numTargets * MC_NUM_RANKS; ++i)
        if (lin[i] >= (1u << 24)) return fail(ctx, MC_ERR_UNSUPPORTED, "taxon index does not fit 24 bits");
    ctx->lineages.assign(lin, lin + numTargets * MC_NUM_RANKS);
    if (ctx->targetCount < numTargets) ctx->targetCount = numTargets;
    for (auto& kv : ctx->taxkeyDev) (void)hipFree(kv.second);
    ctx->taxkeyDev.clear();
    return MC_OK;
}

int mc_db_info(const mc_ctx* ctx, uint64_t info[8])
{
    if (!ctx) return MC_ERR_INVALID;
    info[0] = ctx->targetSketch.k; info[1] = ctx->targetSketch.s; info[2] = ctx->targetSketch.w; info[3] = ctx->targetSketch.stride;
    info[4] = ctx->maxLocs; info[5] = ctx->targetCount; info[6] = ctx->parts.size();
    uint64_t loc = 0;
    for (auto& p : ctx->parts) loc += p.locations;
    info[7] = loc;
    return MC_OK;
}

// taxkey[tgt] = lowest_ranked_ancestor(tgt, rank) as taxon index + 1 (taxonomy.hpp:1260-1267)
static int taxkey_for_rank(mc_ctx* ctx, int rank, const uint32_t** out)
{
    *out = nullptr;
    if (rank <= 0) return MC_OK;
    if (rank >= MC_NUM_RANKS) return fail(ctx, MC_ERR_INVALID, "lowest_rank out of range");
    std::lock_guard<std::mutex> lock(ctx->taxMtx);             // slots submit concurrently
    auto it = ctx->taxkeyDev.find(rank);
    if (it != ctx->taxkeyDev.end()) { *out = it->second; return MC_OK; }
    if (ctx->lineages.empty()) return fail(ctx, MC_ERR_STATE, "lowest_rank > sequence needs mc_set_lineages first");
    const uint64_t nt = ctx->lineages.size() / MC_NUM_RANKS;
    std::vector<uint32_t> tk(nt, 0);
    for (uint64_t t = 0; t < nt; ++t)
        for (int r = rank; r < MC_NUM_RANKS; ++r)
            if (ctx->lineages[t * MC_NUM_RANKS + r]) { tk[t] = ctx->lineages[t * MC_NUM_RANKS + r]; break; }
    uint32_t* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, std::max<uint64_t>(nt, 1) * 4));
    HIP_TRY(ctx, hipMemcpy(d, tk.data(), nt * 4, hipMemcpyHostToDevice));
    ctx->taxkeyDev[rank] = d;
    *out = d;
    return MC_OK;
}

// ------------------------------------------------------------------------------------------------
// the per-batch pipeline
// ------------------------------------------------------------------------------------------------
// the sorted class of the filtered path: filtered lists the counting kernels do not take (long reads: thousands of numbers, wide window
// ranges) are sorted -- one segmented sort over the pool -- and scanned (gw_sorted_cands_kernel); the filter kernels counted them.
// The one place of the filtered path where the host looks at a device counter (how many such lists: the sort's segment count).
static int run_sorted_tail(mc_ctx* ctx, Pipe& P, const BatchView& b, const SketchParams& sp, const DeviceTable& tab, Workspace& ws, uint32_t K,
                           const uint32_t* taxkey, uint64_t poolEntries, bool counterCopied, hipStream_t st)
{
    int rc = MC_OK;
    const uint32_t n = b.n;
    if (!P.hTotal) HIP_TRY(ctx, hipHostMalloc((void**)&P.hTotal, 128));
    uint32_t* nsorted = reinterpret_cast<uint32_t*>(P.hTotal + 9);
    if (!counterCopied) launch_words_to_host(nsorted, ws.midCount + 13, 1, st);
    HIP_TRY(ctx, traced_sync(st));
    {   // MC_GW_DIAG=1: the batch's work-list counters on stderr (the classes of the filtered path)
        static const bool diag = [] { const char* e = std::getenv("MC_GW_DIAG"); return e && e[0] == '1'; }();
        if (diag) {
            uint32_t mc[32];
            HIP_TRY(ctx, hipMemcpy(mc, ws.midCount, sizeof mc, hipMemcpyDeviceToHost));
            std::fprintf(stderr, "[gw diag] n %u filtered %u | stream filter %u | counted apart: 257..512 %u, 513..1024 %u | sorted %u\n",
                         n, mc[9], mc[12], mc[14], mc[15], mc[13]);
        }
    }
    if (!*nsorted) return MC_OK;
    if ((rc = ensure(ctx, P.bBigPool2, poolEntries * 4))) return rc;
    ws.bigPool2 = (uint32_t*)P.bBigPool2.p;
    const uint32_t nseg = std::min(*nsorted, n);
    // the sorted class longest list first: the segmented sort (a block per segment) and the scan (a wave per list) take them in this order
    size_t ordBytes = 0;
    if (launch_gw_order(3, ws, n, nseg, nullptr, ordBytes, st) != 0) return fail(ctx, MC_ERR_HIP, "ordering of the sorted lists: size query failed");
    if ((rc = ensure(ctx, P.bOrder, (size_t)3 * std::max<uint32_t>(n, 1) * 4 + ordBytes + 256))) return rc;
    size_t tmpBytes = 0;
    if (launch_gw_segsort(nullptr, tmpBytes, (const uint32_t*)ws.bigPool, ws.bigPool2, poolEntries, ws, n, nseg, ctx->gwBits, st) != 0)
        return fail(ctx, MC_ERR_HIP, "segmented sort: size query failed");
    if ((rc = ensure(ctx, P.bSortTmp, tmpBytes + 256))) return rc;
    {
        ScopedTimer t(ctx, "gw_sort", st);
        if (launch_gw_order(3, ws, n, nseg, (uint32_t*)P.bOrder.p, ordBytes, st) != 0) return fail(ctx, MC_ERR_HIP, "ordering of the sorted lists failed");
        if (!P.sortSide.stream) {
            if (hipStreamCreateWithFlags(&P.sortSide.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&P.sortSide.fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&P.sortSide.join, hipEventDisableTiming) != hipSuccess) P.sortSide = GwSortSide{};
        }
        if (launch_gw_segsort(P.bSortTmp.p, tmpBytes, (const uint32_t*)ws.bigPool, ws.bigPool2, poolEntries, ws, n, nseg, ctx->gwBits, st, &P.sortSide) != 0)
            return fail(ctx, MC_ERR_HIP, "segmented sort failed");
    }
    { ScopedTimer t(ctx, "gw_sorted_cands", st); launch_big_cands(4, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
    return MC_OK;
}

// The filtered candidate path on the work list the probing kernels (or, on the owner side of Mode K, owner_entries_kernel) left
// in list 6: filter -> counting -> [segmented sort -> scan of the sorted lists].  poolEntries: entries of ws.bigPool (slices + overflow).
// deferSorted: the sorted class is left to the caller (mc_query_finish runs run_sorted_tail).
static int run_filtered_path(mc_ctx* ctx, Pipe& P, const BatchView& b, const SketchParams& sp, const DeviceTable& tab, Workspace& ws, uint32_t K,
                             const uint32_t* taxkey, bool compact, bool second, uint64_t poolEntries, hipStream_t st, bool deferSorted = false)
{
    // timers carry the kernels' own names, one kernel each: compact store gw_filter_count_kernel (or gw_filter_kernel with "gw_fuse" 0), gw_filter2,
    // gw_compact (+ the ordering of the stream filter's reads), gw_filter_stream<fine>, gw_filter_stream (+ the second compaction),
    // gw_count_kernel<9>, <10>, <11>; 8-byte store: big_*
    { ScopedTimer t(ctx, compact ? (ws.gwFuse ? "gw_filter_count" : "gw_filter") : "big_filter", st); launch_big_cands(0, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
    // (compact store: gw_filter_kernel itself may leave reads to the second kernel -- it counts them on the device, after the
    // host's look at the counters: always launched, returns at once with nothing to do)
    if (compact) {                                             // (timers by kernel: a bench line's dominant "kernel" must be one)
        { ScopedTimer t(ctx, "gw_filter2", st); launch_big_cands(3, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
        { ScopedTimer t(ctx, "gw_compact", st); launch_big_cands(7, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
        { ScopedTimer t(ctx, "gw_filter_stream_fine", st); launch_big_cands(8, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
        { ScopedTimer t(ctx, "gw_filter_stream_mid", st); launch_big_cands(11, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
        { ScopedTimer t(ctx, "gw_filter_stream", st); launch_big_cands(9, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
    } else if (second) { ScopedTimer t(ctx, "big_filter_2", st); launch_big_cands(3, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
    { ScopedTimer t(ctx, compact ? "gw_count" : "big_count", st); launch_big_cands(1, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
    if (compact) { ScopedTimer t(ctx, "gw_count_512", st); launch_big_cands(10, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
    { ScopedTimer t(ctx, compact ? "gw_count_1024" : "big_count_2", st); launch_big_cands(2, b, sp, tab, ws, K, taxkey, P.bCands.p, st); }
    if (compact && !deferSorted) return run_sorted_tail(ctx, P, b, sp, tab, ws, K, taxkey, poolEntries, false, st);
    return MC_OK;
}

// What the lane path did not finish goes through the exact wave kernels (long reads, duplicate hashes, reads the filtered path handed
// back, -allhits, ...): sketch + probe unless done, segments for their location lists (the host sizes them: one round trip), sort + candidates.
// sortedPool (small batches whose filtered path left its sorted class to this call): the host's look at that class's counter shares this
// call's one round trip -- the wave kernels' sketching and the scan go out first; in the rare batch that has sorted lists they run again
// behind run_sorted_tail (query_kernel takes the reads still flagged for it: a second pass finds only what the sorted class handed back).
static int run_wave_tail(mc_ctx* ctx, Pipe& P, const BatchView& b, const SketchParams& sp, const DeviceTable& tab, Workspace& ws, uint32_t K,
                         const uint32_t* taxkey, bool fuse, bool skipWaveSketch, bool wantAllhits, bool wantPartial, bool wantNumbers, bool lanePath, hipStream_t st,
                         const uint64_t* sortedPool = nullptr)
{
    int rc = MC_OK;
    const uint32_t n = b.n;
    if (!P.hTotal) HIP_TRY(ctx, hipHostMalloc((void**)&P.hTotal, 128));
    auto sketch_and_scan = [&]() {
        if (!skipWaveSketch) {
            ScopedTimer t(ctx, "query_wave", st);
            launch_query(b, sp, tab, fuse, wantAllhits, ws, K, P.bCands.p, st);
        }
        ScopedTimer t(ctx, "scan", st);
        // (how many locations need a segment in HBM: the total goes to pinned host memory with the scan)
        launch_scan_u32(ws.hitScan, 1, n, nullptr, ws.hitOff, ws.scanTmp, st, P.hTotal);
    };
    sketch_and_scan();
    uint32_t* nsorted = reinterpret_cast<uint32_t*>(P.hTotal + 9);
    if (sortedPool) launch_words_to_host(nsorted, ws.midCount + 13, 1, st);
    HIP_TRY(ctx, traced_sync(st));
    if (sortedPool && *nsorted) {
        if ((rc = run_sorted_tail(ctx, P, b, sp, tab, ws, K, taxkey, *sortedPool, true, st))) return rc;
        sketch_and_scan();
        HIP_TRY(ctx, traced_sync(st));
    }
    const uint64_t totalHits = *P.hTotal;
    const size_t hb = (size_t)(totalHits + 1) * 8;
    if ((rc = ensure(ctx, P.bHits, hb))) return rc;
    if ((rc = ensure(ctx, P.bCscr, hb))) return rc;
    if (taxkey && (rc = ensure(ctx, P.bCscr2, hb))) return rc;
    ws.hits = (uint64_t*)P.bHits.p; ws.cscr = (uint64_t*)P.bCscr.p; ws.cscr2 = (uint64_t*)P.bCscr2.p;
    if (wantNumbers && ((rc = ensure(ctx, P.bNumbers, (size_t)(totalHits + 8) * 4)) || (rc = ensure(ctx, P.bCounts, (size_t)(n + 1) * 4)))) return rc;
    if (wantPartial && lanePath && !wantNumbers) { ScopedTimer t(ctx, "gather_lists", st); launch_gather_lists(b, sp, tab, ws, nullptr, st); }
    {
        ScopedTimer t(ctx, "sort_candidates", st);
        launch_sort_candidates(b, sp, tab, ws, taxkey, K, wantAllhits, P.bCands.p, st);
    }
    if (wantNumbers) {
        // what the wave kernels left in ws.hits -> numbers, then the lane path's (and the chunk lanes') lists straight from the table
        { ScopedTimer t(ctx, "pack_numbers", st); launch_pack_other_reads(b, tab, ws, (uint32_t*)P.bNumbers.p, (uint32_t*)P.bCounts.p, st); }
        if (lanePath) { ScopedTimer t(ctx, "gather_lists", st); launch_gather_lists(b, sp, tab, ws, (uint32_t*)P.bNumbers.p, st); }
        P.numbersN = n; P.numbersTotal = totalHits;
    }
    return MC_OK;
}

static int query_on_pipe(mc_ctx* ctx, Pipe& P, const mc_device_batch* in, int lowestRank, int flags, mc_device_results* out, hipStream_t st);
static int finish_on_pipe(mc_ctx* ctx, Pipe& P);

int mc_query_device(mc_ctx* ctx, const mc_device_batch* in, int lowestRank, int flags, mc_device_results* out, void* streamv)
{
    if (!ctx || !in || !out) return MC_ERR_INVALID;
    flags &= ~mcamd::kQueryNoLongReads;                           // (internal: the caller of this entry point has not seen the reads)
    if (flags & MC_SECOND_PIPE) {
        if (!ctx->pipe1.stream) {
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->pipe1.stream, hipStreamNonBlocking));
        }
        return query_on_pipe(ctx, ctx->pipe1, in, lowestRank, flags, out, streamv ? (hipStream_t)streamv : ctx->pipe1.stream);
    }
    return query_on_pipe(ctx, ctx->pipe0, in, lowestRank, flags, out, streamv ? (hipStream_t)streamv : ctx->stream);
}

// A pipe's buffers whose sizes follow from the batch's size (n queries, numChars characters) and the table's mean list length alone --
// everything mc_query_device needs before its first host round trip.  mc_open_database calls this for every slot pipe WHILE the file
// loads (reserve_slot_pipes): a hipMalloc of memory another process has just given back takes 100-200 ms per 600 MB, and eight pipes
// growing their pools inside the first batches made the same `mcq query` run take 81 or 390 ms per 10^7 reads.
struct PipeSizes { uint64_t maxWindows = 0, poolCap = 0, ovfCap = 0; size_t nfeat = 0; };
static int size_pipe(mc_ctx* ctx, Pipe& P, uint32_t n, uint64_t numChars, bool wantFeatures, bool lanePath, uint64_t locs, uint64_t keys, PipeSizes& out)
{
    int rc = MC_OK;
    const SketchParams sp = ctx->querySketch;
    const uint32_t K = ctx->cfg.max_candidates;
    // windows <= chars/stride + 2 per sequence (row 1), so the sketch buffers can be sized without a sync
    const uint64_t maxWindows = numChars / sp.stride + 4ull * n + 1;
    if (maxWindows * sp.s > 0xFFFFFFF0ull) return fail(ctx, MC_ERR_UNSUPPORTED, "batch too large (feature index exceeds 32 bits)");
    const size_t nfeat = (size_t)maxWindows * sp.s;
    if ((rc = ensure(ctx, P.bWinCount, (size_t)(n + 1) * 4))) return rc;
    if ((rc = ensure(ctx, P.bWinOff, (size_t)(n + 2) * 4))) return rc;
    // (the lane path delivers top candidates only: -allhits and K > 4 go through the wave kernels)
    if ((wantFeatures || lanePath) && (rc = ensure(ctx, P.bFeatures, nfeat * 4))) return rc;
    if ((rc = ensure(ctx, P.bPsize, nfeat * 4))) return rc;
    if ((rc = ensure(ctx, P.bPpay, nfeat * 8))) return rc;
    if ((rc = ensure(ctx, P.bQstat, (size_t)(n + 1) * sizeof(QueryStat)))) return rc;
    if ((rc = ensure(ctx, P.bScanIn, (size_t)(n + 1) * 4))) return rc;
    if ((rc = ensure(ctx, P.bQflag, (size_t)(n + 1) * 4))) return rc;
    if (lanePath && (rc = ensure(ctx, P.bMid, 128 + (size_t)8 * std::max<uint32_t>(n, 1) * 16))) return rc;
    // pool of the filtered location lists (big_filter_kernel -> big_count_kernel): 384 per query on average, at least 4 MB
    // (tables whose features have few locations each never produce such lists: a token pool; a full pool sends lists to the wave kernel)
    const Part& T0 = ctx->parts[0];
    const bool longLists = (double)locs / (double)std::max<uint64_t>(keys, 1) * 2.0 * sp.s > 64.0;   // mean list of a 2-window read
    // (448 per 150 bp read = 3 per base: longer reads collect -- and keep -- in proportion)
    const uint64_t poolCap = std::min<uint64_t>(0xFFFFFFF0ull, std::max<uint64_t>(longLists ? std::max<uint64_t>((uint64_t)n * 448, numChars * 3) : (uint64_t)n * 8,
                                                                                  (uint64_t)big_filter_grid(n, T0.compact, ctx->filterBpc) * 4 * std::min<uint64_t>(131072, std::max<uint64_t>(4096, 64 * (numChars / std::max<uint32_t>(n, 1))))));   // per wave: one full batch of gw_filter_kernel (2 112 numbers); long reads keep up to 65 535
    // compact store: behind the waves' slices an OVERFLOW region for filtered lists that may not fit their wave's slice (the longest
    // reads of a batch keep 10^5 numbers): reserved with one atomic per such read (midCount[16..17])
    const uint64_t ovfCap = T0.compact ? std::min<uint64_t>(0xFFFFFFF0ull - poolCap, std::max<uint64_t>(poolCap / 4, 8ull << 20)) : 0;
    if (lanePath && (rc = ensure(ctx, P.bBigPool, (poolCap + ovfCap) * (T0.compact ? 4 : 8)))) return rc;   // the pool holds the table's location form
    if (lanePath && (rc = ensure(ctx, P.bSliceFill, (size_t)big_filter_grid(n, T0.compact, ctx->filterBpc) * 4 * 4 + 64))) return rc;
    if (lanePath && T0.compact && (rc = ensure(ctx, P.bSide, (size_t)5 * std::max<uint32_t>(n, 1) * 4))) return rc;
    if (lanePath && (rc = ensure(ctx, P.bChunkList, (size_t)(maxWindows + n + 1) * 8))) return rc;
    if ((rc = ensure(ctx, P.bHitOff, (size_t)(n + 2) * 8))) return rc;
    if ((rc = ensure(ctx, P.bScan, scan_tmp_bytes(n + 1)))) return rc;
    if ((rc = ensure(ctx, P.bStats, 64))) return rc;
    if ((rc = ensure(ctx, P.bCands, (size_t)std::max<uint32_t>(n, 1) * K * sizeof(mc_candidate)))) return rc;
    out.maxWindows = maxWindows; out.poolCap = poolCap; out.ovfCap = ovfCap; out.nfeat = nfeat;
    return MC_OK;
}

// every slot pipe sized for a full slot of reads of the usual length (see size_pipe): called by mc_open_database from a thread of its own
// once the table is announced; `locs` / `keys`: the part headers' counts
}  // extern "C" (the loader's helper below has C++ linkage)
int mcamd::reserve_slot_pipes(mc_ctx* ctx, uint64_t locs, uint64_t keys, bool waitForStores)
{
    if (!ctx || ctx->pipes.empty() || ctx->parts.empty()) return MC_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return MC_ERR_HIP;
    t_quietErrors = true;                                        // (the loader thread owns ctx->err)
    // The table comes first: a single-part table's location store is allocated when the index pass is through (announce_store /
    // allocate_values), and on a device the table nearly fills the pipes must not have taken its memory by then.
    for (; waitForStores;) {
        if (ctx->loadSettled.load(std::memory_order_acquire)) break;
        if (ctx->storesPlaced.load(std::memory_order_acquire) >= ctx->parts.size()) break;
        std::this_thread::sleep_for(std::chrono::microseconds(500));
    }
    const SketchParams sp = ctx->querySketch;
    const uint32_t K = ctx->cfg.max_candidates, n = ctx->coalesce ? ctx->coMaxQueries : ctx->cfg.slot_max_queries;   // (coalescer: a dispatcher's pipe takes a united batch)
    const bool wantAll = ctx->cfg.copy_allhits != 0;
    const bool lanePath = lane_path_supported(sp) && ctx->useLanePath && !wantAll && lane_candidates_supported(K);
    // (a slot of short reads: 152 characters each -- a slot filled with longer reads has fewer of them and grows its buffers as before)
    const uint64_t chars = std::min<uint64_t>(ctx->coalesce ? ctx->coMaxChars : ctx->cfg.slot_max_chars, (uint64_t)n * 152);
    int rc = MC_OK;
    for (Pipe* P : ctx->pipes) {
        PipeSizes sz{};
        if ((rc = size_pipe(ctx, *P, n, chars, false, lanePath, locs, keys, sz))) break;
    }
    t_quietErrors = false;
    return rc;
}
// the same for the context's own two pipes (mc_query_device with and without MC_SECOND_PIPE): the part set driver sizes them for its batches
// right after a part has loaded -- on the group loader's thread, beside the other parts' loads -- instead of inside the first batches
int mcamd::reserve_query_pipes(mc_ctx* ctx, uint32_t n, uint64_t chars)
{
    if (!ctx || ctx->parts.empty() || !ctx->tableReady) return MC_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return MC_ERR_HIP;
    const SketchParams sp = ctx->querySketch;
    const bool lanePath = lane_path_supported(sp) && ctx->useLanePath && lane_candidates_supported(ctx->cfg.max_candidates);
    uint64_t locs = 0;
    for (auto& p : ctx->parts) locs += p.locations;
    if (!ctx->pipe1.stream && hipStreamCreateWithFlags(&ctx->pipe1.stream, hipStreamNonBlocking) != hipSuccess) return MC_ERR_HIP;
    for (Pipe* P : {&ctx->pipe0, &ctx->pipe1}) {
        PipeSizes sz{};
        if (const int rc = size_pipe(ctx, *P, n, chars, false, lanePath, locs, ctx->parts[0].keysStored, sz)) return rc;
    }
    return MC_OK;
}
extern "C" {

static int query_on_pipe(mc_ctx* ctx, Pipe& P, const mc_device_batch* in, int lowestRank, int flags, mc_device_results* out, hipStream_t st)
{
    // MC_WANT_PARTIAL_HITS: the location lists as they are (unsorted), lane path allowed -- a key shard's side of Mode K; the queries the
    // lane path does not take go through the wave kernels as with MC_WANT_ALLHITS (their lists come out sorted, which is allowed)
    // MC_WANT_PARTIAL_NUMBERS: the same, and the lists are left as the 4-byte global window numbers they are stored as (mc_partial_numbers)
    const bool wantNumbers = (flags & MC_WANT_PARTIAL_NUMBERS) != 0 && !(flags & MC_WANT_ALLHITS);
    const bool wantPartial = ((flags & MC_WANT_PARTIAL_HITS) != 0 || wantNumbers) && !(flags & MC_WANT_ALLHITS);
    const int wantAllhits = (flags & MC_WANT_ALLHITS) | (wantPartial ? 1 : 0);
    const bool wantFeatures = (flags & MC_WANT_FEATURES) != 0;
    if (P.tail.pending) { int rcf = finish_on_pipe(ctx, P); if (rcf) return rcf; }   // (a caller that never asked for the last batch's tail: run it, the workspace is reused now)
    if (!ctx->tableReady) return fail(ctx, MC_ERR_STATE, "no database loaded (every part needs mc_load_begin .. mc_load_end)");
    if (!in->max_win && in->max_win_uniform < 1) return fail(ctx, MC_ERR_INVALID, "max_win or max_win_uniform required");
    if (wantNumbers && (ctx->parts.size() != 1 || !ctx->parts[0].compact || !ctx->dGwBase))
        return fail(ctx, MC_ERR_UNSUPPORTED, "MC_WANT_PARTIAL_NUMBERS: the database has no global window numbers (compact location store)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = in->num_queries;
    P.numbersN = 0xFFFFFFFFu;
    const SketchParams sp = ctx->querySketch;
    const uint32_t K = ctx->cfg.max_candidates;
    const uint32_t* taxkey = nullptr;
    int rc = taxkey_for_rank(ctx, lowestRank, &taxkey);
    if (rc) return rc;

    // the pipe's buffers that depend on the batch's size alone (size_pipe; mc_open_database reserves them beside the file load)
    const bool lanePath = lane_path_supported(sp) && ctx->useLanePath && (!wantAllhits || wantPartial) && lane_candidates_supported(K);
    const Part& T0 = ctx->parts[0];
    uint64_t locs = 0;
    for (auto& p : ctx->parts) locs += p.locations;
    PipeSizes sz{};
    if ((rc = size_pipe(ctx, P, n, in->num_chars, wantFeatures, lanePath, locs, T0.keysStored, sz))) return rc;
    const uint64_t maxWindows = sz.maxWindows, poolCap = sz.poolCap, ovfCap = sz.ovfCap;
    const size_t nfeat = sz.nfeat;
    (void)maxWindows;

    Workspace ws{};
    ws.filterBpc = ctx->filterBpc; ws.countBpc = ctx->countBpc; ws.gwFuse = ctx->gwFuse; ws.gwBigH = ctx->gwBigH; ws.gwMidH = ctx->gwMidH;
    ws.winCount = (uint32_t*)P.bWinCount.p; ws.winOff = (uint32_t*)P.bWinOff.p;
    ws.features = (wantFeatures || lanePath) ? (uint32_t*)P.bFeatures.p : nullptr; ws.psize = (uint32_t*)P.bPsize.p; ws.ppay = (uint64_t*)P.bPpay.p;
    ws.qstat = (QueryStat*)P.bQstat.p; ws.hitScan = (uint32_t*)P.bScanIn.p; ws.qflag = (uint32_t*)P.bQflag.p; ws.hitOff = (uint64_t*)P.bHitOff.p;
    ws.scanTmp = P.bScan.p; ws.stats = (uint64_t*)P.bStats.p;
    if (lanePath) {
        ws.midCount = (uint32_t*)P.bMid.p; ws.midList = ws.midCount + 32;
        ws.bigMin = ctx->bigMin;
        ws.partialLists = wantPartial ? 1u : 0u;
        ws.bigPool = (uint64_t*)P.bBigPool.p; ws.bigPoolCap = (uint32_t)poolCap; ws.bigOvfCap = (uint32_t)ovfCap;
        ws.sliceFill = (uint32_t*)P.bSliceFill.p;
        ws.sideList = (uint32_t*)P.bSide.p;
        ws.chunkList = (flags & kQueryNoLongReads) ? nullptr : (uint2*)P.bChunkList.p;   // (null: no read is cut into chunks, launch_chunk_lanes launches nothing)
    }

    BatchView b{in->seq, in->qinfo, in->max_win, in->max_win_uniform, n};
    const Part& T = ctx->parts[0];
    const bool multiPart = ctx->parts.size() > 1;
    DeviceTable tab{T.dbuckets, T.dvalues, T.nbuckets, multiPart ? 0x00FFFFFFu : 0xFFFFFFFFu, T.maxProbe};
    if (T.compact) {
        tab.values = nullptr; tab.values32 = reinterpret_cast<const uint32_t*>(T.dvalues);
        tab.gwBase = ctx->dGwBase; tab.gwDir = ctx->dGwDir; tab.gwDirShift = ctx->gwDirShift; tab.gwGap = ctx->gwGap; tab.gwTargets = ctx->gwTargets;
    }
    tab.direct = T.ddirect;                                       // (the lane path's lookups; every other kernel goes through the buckets)

    bool planSmall = false;                                      // small batches: plan + scan + the work-list counters' clearing in one launch
    {
        ScopedTimer t(ctx, "plan", st);
        planSmall = launch_plan_scan_small(b, sp, ws.winCount, ws.winOff, lanePath ? ws.midCount : nullptr, st);
        if (!planSmall) {
            launch_plan(b, sp, ws.winCount, st);
            launch_scan_u32(ws.winCount, 1, n, ws.winOff, nullptr, ws.scanTmp, st);
        }
    }
    const bool fuse = !wantAllhits && !taxkey;
    bool waveWork = true;                                        // wave kernels needed (always without the lane path)
    bool skipWaveSketch = false;                                 // ... their sketching and probing has run already
    bool sortedInTail = false;                                   // the filtered path's sorted class is run_wave_tail's to look at
    if (lanePath) {
        // short reads: one lane per query for sketching and candidates, cooperative probing in between
        if (!planSmall) HIP_TRY(ctx, hipMemsetAsync(ws.midCount, 0, 128, st));
        // sketching + probing in ONE kernel where the lookups wait for HBM (tables beyond the infinity cache: quad-cooperative fetches) -- the
        // sketching of some waves runs under the waiting of others (5.27 -> 5.08 ms per 5 x 10^6 reads at full scale); small tables keep
        // the two kernels (the ALU phase at the probe kernel's occupancy cost 5 % on configs[1]).  "lane_fusion" / MC_LANE_FUSION: 0 / 1 force it.
        // (A key shard's side of Mode K masks the features it does not own between the two: not fused.)
        const bool maskFeatures = wantPartial && !wantFeatures && ctx->cfg.key_shard_count > 1;
        const bool quadTable = ctx->quadLookup >= 0 ? ctx->quadLookup != 0 : (uint64_t)tab.nbuckets * 64ull > (1ull << 30);
        const bool fuseSketch = !maskFeatures && (ctx->fuseLane >= 0 ? ctx->fuseLane != 0 : quadTable);
        if (fuseSketch) {
            { ScopedTimer t(ctx, "sketch_probe", st); launch_sketch_probe_lane(b, sp, tab, ws, K, taxkey, P.bCands.p, ctx->quadLookup, st); }
            { ScopedTimer t(ctx, "chunk_sketch", st); launch_chunk_lanes(0, b, sp, tab, ws, ctx->quadLookup, st); }
            { ScopedTimer t(ctx, "chunk_probe", st); launch_chunk_lanes(1, b, sp, tab, ws, ctx->quadLookup, st); }
        } else {
            { ScopedTimer t(ctx, "sketch_lane", st); launch_sketch_lane(b, sp, ws, st); }
            { ScopedTimer t(ctx, "chunk_sketch", st); launch_chunk_lanes(0, b, sp, tab, ws, ctx->quadLookup, st); }
            // a key shard's side of Mode K: only the features this shard owns are looked up (the others cannot be in its table)
            if (maskFeatures) {
                ScopedTimer t(ctx, "mask_features", st);
                launch_mask_foreign_features(ws.features, ws.winOff + n, sp.s, nfeat, ctx->cfg.key_shard_index, ctx->cfg.key_shard_count, st);
            }
            { ScopedTimer t(ctx, "chunk_probe", st); launch_chunk_lanes(1, b, sp, tab, ws, ctx->quadLookup, st); }
            { ScopedTimer t(ctx, "probe_cands", st); launch_probe_cands(b, sp, tab, ws, K, taxkey, P.bCands.p, ctx->quadLookup, st); }
        }
        // small batches take one look at the work lists and launch only the kernels with work (a launch costs as much as such a
        // batch's kernel: 0.37 -> 0.32 ms per 65 536 reads); large ones skip the round trip and launch everything
        uint32_t all[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        uint32_t none[16] = {0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0};      // partial lists: no candidate kernels, the wave kernels for the rest
        uint32_t* hcnt = wantPartial ? none : all;
        // (MC_DEFER_TAIL: no look at the counters either -- everything is launched, the caller has another batch to enqueue)
        if (!wantPartial && n <= (1u << 20) && !(flags & MC_DEFER_TAIL)) {
            if (!P.hTotal) HIP_TRY(ctx, hipHostMalloc((void**)&P.hTotal, 128));
            hcnt = reinterpret_cast<uint32_t*>(P.hTotal + 1);
            launch_flag_count_host(ws, n, hcnt, st);
            HIP_TRY(ctx, traced_sync(st));
        }
        auto mid_and_hash = [&]() {
            if (hcnt[0]) { ScopedTimer t(ctx, "mid_cands_64", st); launch_mid_cands(0, b, tab, ws, K, taxkey, P.bCands.p, st); }
            if (hcnt[1]) { ScopedTimer t(ctx, "mid_cands_128", st); launch_mid_cands(1, b, tab, ws, K, taxkey, P.bCands.p, st); }
            if (hcnt[2]) { ScopedTimer t(ctx, "mid_cands_256", st); launch_mid_cands(2, b, tab, ws, K, taxkey, P.bCands.p, st); }
            if (hcnt[8]) { ScopedTimer t(ctx, "hash_cands_256", st); launch_hash_cands(5, b, tab, ws, K, taxkey, P.bCands.p, st); }
            if (hcnt[3]) { ScopedTimer t(ctx, "hash_cands_512", st); launch_hash_cands(3, b, tab, ws, K, taxkey, P.bCands.p, st); }
            if (hcnt[4]) { ScopedTimer t(ctx, "hash_cands_1024", st); launch_hash_cands(4, b, tab, ws, K, taxkey, P.bCands.p, st); }
        };
        mid_and_hash();
        bool waveDone = false;
        if (T.compact && !wantPartial && hcnt[6]) {
            // compact store: the wave kernel's sketching and probing first, so that its reads can join the filtered path
            { ScopedTimer t(ctx, "query_wave", st); launch_query(b, sp, tab, fuse, false, ws, K, P.bCands.p, st); }
            launch_wave_rejoin(b, sp, tab, ws, st);
            waveDone = true;
        }
        if (T.compact && !wantPartial && n <= (1u << 20) && (hcnt == all || hcnt[10])) {
            // reads beyond kGwSmallH locations (long reads): the stream filter takes them longest first (launch_gw_order)
            size_t ordBytes = 0;
            if (launch_gw_order(0, ws, n, n, nullptr, ordBytes, st) != 0) return fail(ctx, MC_ERR_HIP, "ordering of the stream filter's reads: size query failed");
            if ((rc = ensure(ctx, P.bOrder, (size_t)3 * std::max<uint32_t>(n, 1) * 4 + ordBytes + 256))) return rc;
            ws.orderScratch = (uint32_t*)P.bOrder.p; ws.orderTemp = ordBytes;
        }
        // MC_DEFER_TAIL (large batches on the lane path): everything the host has to look at device counters for -- the sorted class of the
        // filtered path, the segment sizes of the exact wave kernels' leftovers -- waits for mc_query_finish; this call returns with
        // the main kernels enqueued and NO synchronisation, so that the caller can enqueue the next batch on the other pipe first
        const bool defer = (flags & MC_DEFER_TAIL) != 0 && hcnt == all && !wantFeatures;
        waveWork = wantPartial || hcnt[6] != 0 || hcnt[7] != 0 || hcnt[9] != 0;   // big_cands hands a few queries on to the wave kernels
        // small batches: the sorted class's counter is looked at together with the wave tail's total (run_wave_tail: one round trip for both)
        sortedInTail = hcnt != all && !defer && T.compact && waveWork && (hcnt[9] || waveDone);
        if (hcnt[9] || waveDone) {
            if ((rc = run_filtered_path(ctx, P, b, sp, tab, ws, K, taxkey, T.compact, hcnt[10] != 0, poolCap + ovfCap, st, defer || sortedInTail))) return rc;
        }
        skipWaveSketch = waveDone;
        if (defer) {
            Pipe::Tail& tl = P.tail;
            if (!P.hTotal) HIP_TRY(ctx, hipHostMalloc((void**)&P.hTotal, 128));
            if (!tl.mainDone) HIP_TRY(ctx, hipEventCreateWithFlags(&tl.mainDone, hipEventDisableTiming));
            tl.sortedPath = T.compact && (hcnt[9] || waveDone);
            if (tl.sortedPath) HIP_TRY(ctx, hipMemcpyAsync(reinterpret_cast<uint32_t*>(P.hTotal + 9), ws.midCount + 13, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipEventRecord(tl.mainDone, st));
            tl.ws = ws; tl.b = b; tl.sp = sp; tl.tab = tab; tl.K = K; tl.taxkey = taxkey; tl.compact = T.compact; tl.fuse = fuse;
            tl.skipWaveSketch = skipWaveSketch; tl.poolEntries = poolCap + ovfCap; tl.st = st;
            tl.pending = true;
            HIP_TRY(ctx, hipGetLastError());
            P.lastN = n;
            out->cands = (const mc_candidate*)P.bCands.p;
            out->hit_counts = (const uint32_t*)P.bQstat.p;
            out->hit_offsets = nullptr; out->hits = nullptr;
            out->features = ws.features;
            out->win_offsets = ws.winOff;
            return MC_OK;
        }
    } else {
        HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)ws.qflag, 1, n, st));     // every query: needs sketch + probe
    }
    const uint64_t sortedPoolEntries = poolCap + ovfCap;
    if (waveWork && (rc = run_wave_tail(ctx, P, b, sp, tab, ws, K, taxkey, fuse, skipWaveSketch, wantAllhits != 0, wantPartial, wantNumbers, lanePath, st,
                                        sortedInTail ? &sortedPoolEntries : nullptr))) return rc;
    HIP_TRY(ctx, hipGetLastError());
    P.lastN = n;
    out->cands = (const mc_candidate*)P.bCands.p;
    out->hit_counts = (const uint32_t*)P.bQstat.p;       // QueryStat.hits: stride 4 words
    out->hit_offsets = wantAllhits ? ws.hitOff : nullptr;
    out->hits = wantAllhits ? (const mc_location*)ws.hits : nullptr;
    out->features = ws.features;
    out->win_offsets = ws.winOff;
    return MC_OK;
}

// the rare classes of a batch whose main kernels mc_query_device(MC_DEFER_TAIL) enqueued: sorted lists, then the exact wave kernels
static int finish_on_pipe(mc_ctx* ctx, Pipe& P)
{
    Pipe::Tail& tl = P.tail;
    if (!tl.pending) return MC_OK;
    tl.pending = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(tl.mainDone));
    int rc = MC_OK;
    if (tl.sortedPath && (rc = run_sorted_tail(ctx, P, tl.b, tl.sp, tl.tab, tl.ws, tl.K, tl.taxkey, tl.poolEntries, true, tl.st))) return rc;
    if ((rc = run_wave_tail(ctx, P, tl.b, tl.sp, tl.tab, tl.ws, tl.K, tl.taxkey, tl.fuse, tl.skipWaveSketch, false, false, false, true, tl.st))) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return MC_OK;
}

int mc_query_finish(mc_ctx* ctx, int flags)
{
    if (!ctx) return MC_ERR_INVALID;
    return finish_on_pipe(ctx, (flags & MC_SECOND_PIPE) ? ctx->pipe1 : ctx->pipe0);
}

uint32_t mc_key_owner(uint32_t feature, uint32_t shardCount) { return key_owner(feature, shardCount); }

int mc_candidates_from_hits(mc_ctx* ctx, const mc_device_hits* in, int lowestRank, mc_device_results* out, void* streamv)
{
    if (!ctx || !in || !out || !in->hit_offsets) return MC_ERR_INVALID;
    if (!in->max_win && in->max_win_uniform < 1) return fail(ctx, MC_ERR_INVALID, "max_win or max_win_uniform required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Pipe& P = ctx->pipe0;
    if (P.tail.pending) { const int rcf = finish_on_pipe(ctx, P); if (rcf) return rcf; }   // (a deferred tail still owns the workspace this call is about to reuse)
    hipStream_t st = streamv ? (hipStream_t)streamv : ctx->stream;
    const uint32_t n = in->num_queries;
    const uint32_t K = ctx->cfg.max_candidates;
    const uint32_t* taxkey = nullptr;
    int rc = taxkey_for_rank(ctx, lowestRank, &taxkey);
    if (rc) return rc;
    uint64_t total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, in->hit_offsets + n, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, traced_sync(st));
    const size_t hb = (size_t)(total + 1) * 8;
    if ((rc = ensure(ctx, P.bHits, hb)) || (rc = ensure(ctx, P.bCscr, hb)) || (taxkey && (rc = ensure(ctx, P.bCscr2, hb)))) return rc;
    if ((rc = ensure(ctx, P.bHitOff, (size_t)(n + 2) * 8)) || (rc = ensure(ctx, P.bQstat, (size_t)(n + 1) * sizeof(QueryStat))) ||
        (rc = ensure(ctx, P.bCands, (size_t)std::max<uint32_t>(n, 1) * K * sizeof(mc_candidate))))
        return rc;
    // the lists are sorted in place: work on a copy inside the context
    if (total) HIP_TRY(ctx, hipMemcpyAsync(P.bHits.p, in->hits, total * 8, hipMemcpyDeviceToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(P.bHitOff.p, in->hit_offsets, (size_t)(n + 1) * 8, hipMemcpyDeviceToDevice, st));
    Workspace ws{};
    ws.filterBpc = ctx->filterBpc; ws.countBpc = ctx->countBpc; ws.gwFuse = ctx->gwFuse; ws.gwBigH = ctx->gwBigH; ws.gwMidH = ctx->gwMidH;
    ws.hits = (uint64_t*)P.bHits.p; ws.cscr = (uint64_t*)P.bCscr.p; ws.cscr2 = (uint64_t*)P.bCscr2.p;
    ws.hitOff = (uint64_t*)P.bHitOff.p; ws.qstat = (QueryStat*)P.bQstat.p;
    BatchView b{nullptr, nullptr, in->max_win, in->max_win_uniform, n};
    DeviceTable tab{nullptr, nullptr, 0, 0xFFFFFFFFu, 1};
    {
        ScopedTimer t(ctx, "cands_from_hits", st);
        launch_cands_from_hits(b, tab, ws, taxkey, K, P.bCands.p, st);
    }
    HIP_TRY(ctx, hipGetLastError());
    P.lastN = 0;
    out->cands = (const mc_candidate*)P.bCands.p;
    out->hit_counts = (const uint32_t*)P.bQstat.p;
    out->hit_offsets = ws.hitOff;
    out->hits = (const mc_location*)ws.hits;
    out->features = nullptr; out->win_offsets = nullptr;
    return MC_OK;
}

// Mode K, owner side in one call: union of the sources' partial lists (device side, no host round trip), then rows 8-10 on it
int mc_candidates_from_partial_hits(mc_ctx* ctx, const mc_device_partial_hits* in, int lowestRank, mc_device_results* out, void* streamv)
{
    if (!ctx || !in || !out || !in->counts || (!in->hits && in->total_hits) || in->num_sources < 1) return MC_ERR_INVALID;
    if (!in->max_win && in->max_win_uniform < 1) return fail(ctx, MC_ERR_INVALID, "max_win or max_win_uniform required");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Pipe& P = ctx->pipe0;
    if (P.tail.pending) { const int rcf = finish_on_pipe(ctx, P); if (rcf) return rcf; }   // (a deferred tail still owns the workspace this call is about to reuse)
    hipStream_t st = streamv ? (hipStream_t)streamv : ctx->stream;
    const uint32_t n = in->num_queries, S = in->num_sources;
    const uint32_t K = ctx->cfg.max_candidates;
    const uint32_t* taxkey = nullptr;
    int rc = taxkey_for_rank(ctx, lowestRank, &taxkey);
    if (rc) return rc;
    const size_t hb = (size_t)(in->total_hits + 1) * 8;
    if ((rc = ensure(ctx, P.bHits, hb)) || (rc = ensure(ctx, P.bCscr, hb)) || (taxkey && (rc = ensure(ctx, P.bCscr2, hb)))) return rc;
    if ((rc = ensure(ctx, P.bHitOff, (size_t)(n + 2) * 8)) || (rc = ensure(ctx, P.bQstat, (size_t)(n + 1) * sizeof(QueryStat))) ||
        (rc = ensure(ctx, P.bCands, (size_t)std::max<uint32_t>(n, 1) * K * sizeof(mc_candidate))) ||
        (rc = ensure(ctx, P.bScanIn, (size_t)(n + 1) * 4)) || (rc = ensure(ctx, P.bScan, scan_tmp_bytes(n + 1))) ||
        (rc = ensure(ctx, P.bPpay, ((size_t)S * (n + 2) + n + 1) * 8)) || (rc = ensure(ctx, P.bPsize, (size_t)(n + 1) * 4)) ||
        (rc = ensure(ctx, P.bWinOff, (size_t)(n + 2) * 4)) || (rc = ensure(ctx, P.bWinCount, (size_t)(n + 1) * 4)) ||
        (rc = ensure(ctx, P.bQflag, (size_t)(n + 1) * 4)) || (rc = ensure(ctx, P.bMid, 128 + (size_t)8 * std::max<uint32_t>(n, 1) * 16)))
        return rc;
    const uint64_t poolCap = std::min<uint64_t>(0xFFFFFFF0ull, std::max<uint64_t>((uint64_t)n * 448, (uint64_t)big_filter_grid(n, false, ctx->filterBpc) * 4 * 1024));
    if ((rc = ensure(ctx, P.bBigPool, poolCap * 8))) return rc;
    // srcStart (union) and the one-entry-per-read tables of the filtered path share bPpay: [S * (n + 2)] u64 | [n] u64
    uint64_t* srcStart = (uint64_t*)P.bPpay.p;
    launch_union_partial(in->counts, S, n, reinterpret_cast<const uint64_t*>(in->hits), (uint32_t*)P.bScanIn.p, srcStart, (uint64_t*)P.bHitOff.p,
                         (uint64_t*)P.bHits.p, P.bScan.p, st);
    Workspace ws{};
    ws.filterBpc = ctx->filterBpc; ws.countBpc = ctx->countBpc; ws.gwFuse = ctx->gwFuse; ws.gwBigH = ctx->gwBigH; ws.gwMidH = ctx->gwMidH;
    ws.hits = (uint64_t*)P.bHits.p; ws.cscr = (uint64_t*)P.bCscr.p; ws.cscr2 = (uint64_t*)P.bCscr2.p;
    ws.hitOff = (uint64_t*)P.bHitOff.p; ws.qstat = (QueryStat*)P.bQstat.p;
    BatchView b{nullptr, nullptr, in->max_win, in->max_win_uniform, n};
    // Long united lists (RefSeq scale: 1 300 locations per read) take the filtered path of the replicated mode instead of a sort: the union
    // buffer stands in for the table's location store, a read's whole list is ONE entry of it (rounds of 16 / 64 locations); what the
    // filter cannot take (more than 16 384 locations, wide window ranges) and what it hands back goes through the sort as before.
    const bool filtered = lane_candidates_supported(K) && ctx->useLanePath;
    if (filtered) {
        ws.midCount = (uint32_t*)P.bMid.p; ws.midList = ws.midCount + 32;
        ws.bigMin = ctx->bigMin; ws.bigPool = (uint64_t*)P.bBigPool.p; ws.bigPoolCap = (uint32_t)poolCap;
        ws.psize = (uint32_t*)P.bPsize.p; ws.ppay = srcStart + (size_t)S * (n + 2);
        ws.winOff = (uint32_t*)P.bWinOff.p; ws.qflag = (uint32_t*)P.bQflag.p; ws.hitScan = (uint32_t*)P.bWinCount.p;
        HIP_TRY(ctx, hipMemsetAsync(ws.midCount, 0, 128, st));
        launch_owner_classify(b, ws, std::max<uint32_t>(ctx->bigMin, 256u), st);
        DeviceTable utab{nullptr, ws.hits, 0, 0xFFFFFFFFu, 1};
        const SketchParams one{16, 1, 16, 1};                                  // step D finds a read's entry at winOff[q] * s = q
        { ScopedTimer t(ctx, "big_filter", st); launch_big_cands(0, b, one, utab, ws, K, taxkey, P.bCands.p, st); }
        { ScopedTimer t(ctx, "big_count", st); launch_big_cands(1, b, one, utab, ws, K, taxkey, P.bCands.p, st); }
        { ScopedTimer t(ctx, "big_count_2", st); launch_big_cands(2, b, one, utab, ws, K, taxkey, P.bCands.p, st); }
    }
    DeviceTable tab{nullptr, nullptr, 0, 0xFFFFFFFFu, 1};
    {
        ScopedTimer t(ctx, "cands_from_hits", st);
        launch_cands_from_hits(b, tab, ws, taxkey, K, P.bCands.p, st);
    }
    HIP_TRY(ctx, hipGetLastError());
    P.lastN = 0;
    out->cands = (const mc_candidate*)P.bCands.p;
    out->hit_counts = (const uint32_t*)P.bQstat.p;
    out->hit_offsets = ws.hitOff;
    out->hits = (const mc_location*)ws.hits;
    out->features = nullptr; out->win_offsets = nullptr;
    return MC_OK;
}

// ---- Mode K with 4-byte locations on the wire (keyshard.hip) --------------------------------------------------------------------------
// shard side: the partial lists of the last mc_query_device(MC_WANT_PARTIAL_HITS) call as global window numbers + per-read counts
int mc_partial_numbers(mc_ctx* ctx, const mc_device_results* res, uint32_t n, const uint32_t* cutQueries, uint32_t numCuts, uint64_t* cutOffsets,
                       mc_device_partial_numbers* out, void* streamv)
{
    if (!ctx || !res || !out || (numCuts && (!cutQueries || !cutOffsets))) return MC_ERR_INVALID;
    if (!res->hit_offsets) return fail(ctx, MC_ERR_STATE, "mc_partial_numbers: the results hold no location lists (MC_WANT_PARTIAL_HITS)");
    if (ctx->parts.size() != 1 || !ctx->parts[0].compact || !ctx->dGwBase)
        return fail(ctx, MC_ERR_UNSUPPORTED, "mc_partial_numbers: the database has no global window numbers (compact location store)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // (the pipe whose batch these results are: a caller with two batches in flight -- keyset.cpp's lanes -- runs the shards' lookups on either)
    Pipe& P = (ctx->pipe1.bHitOff.p && res->hit_offsets == (const uint64_t*)ctx->pipe1.bHitOff.p) ? ctx->pipe1 : ctx->pipe0;
    if (P.tail.pending) { const int rcf = finish_on_pipe(ctx, P); if (rcf) return rcf; }   // (a deferred tail still owns the workspace this call is about to reuse)
    hipStream_t st = streamv ? (hipStream_t)streamv : (&P == &ctx->pipe1 && ctx->pipe1.stream) ? ctx->pipe1.stream : ctx->stream;
    for (uint32_t i = 0; i < numCuts; ++i) {
        if (cutQueries[i] > n) return fail(ctx, MC_ERR_INVALID, "mc_partial_numbers: cut beyond the batch");
        HIP_TRY(ctx, hipMemcpyAsync(&cutOffsets[i], res->hit_offsets + cutQueries[i], 8, hipMemcpyDeviceToHost, st));
    }
    uint64_t total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, res->hit_offsets + n, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, traced_sync(st));                    // the one host round trip of the exchange: its split sizes
    int rc;
    if (P.numbersN == n && P.numbersTotal == total && res->hit_offsets == (const uint64_t*)P.bHitOff.p) {
        // mc_query_device(MC_WANT_PARTIAL_NUMBERS) left the numbers and the counts where they belong
    } else {
        if ((rc = ensure(ctx, P.bNumbers, (size_t)(total + 8) * 4)) || (rc = ensure(ctx, P.bCounts, (size_t)(n + 1) * 4))) return rc;
        DeviceTable tab{nullptr, nullptr, 0, 0xFFFFFFFFu, 1};
        tab.gwBase = ctx->dGwBase; tab.gwDir = ctx->dGwDir; tab.gwDirShift = ctx->gwDirShift; tab.gwGap = ctx->gwGap; tab.gwTargets = ctx->gwTargets;
        {
            ScopedTimer t(ctx, "pack_numbers", st);
            launch_pack_numbers(reinterpret_cast<const uint64_t*>(res->hits), res->hit_offsets, total, n, tab, (uint32_t*)P.bNumbers.p, (uint32_t*)P.bCounts.p, st);
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    out->counts = (const uint32_t*)P.bCounts.p;
    out->numbers = (const uint32_t*)P.bNumbers.p;
    out->total = total;
    return MC_OK;
}

// owner side: rows 8-10 on the pieces the key shards sent for this rank's reads, where they lie in the receive buffer
int mc_candidates_from_partial_numbers(mc_ctx* ctx, const mc_device_partial_numbers_in* in, int lowestRank, mc_device_results* out, void* streamv)
{
    return mc_candidates_from_partial_numbers_on(ctx, in, lowestRank, 0, out, streamv);
}

int mc_candidates_from_partial_numbers_on(mc_ctx* ctx, const mc_device_partial_numbers_in* in, int lowestRank, int flags, mc_device_results* out, void* streamv)
{
    if (!ctx || !in || !out || !in->counts || !in->source_offsets || in->num_sources < 1 || in->num_sources > 64) return MC_ERR_INVALID;
    if (!in->max_win && in->max_win_uniform < 1) return fail(ctx, MC_ERR_INVALID, "max_win or max_win_uniform required");
    if (ctx->parts.size() != 1 || !ctx->parts[0].compact || !ctx->dGwBase)
        return fail(ctx, MC_ERR_UNSUPPORTED, "mc_candidates_from_partial_numbers: the database has no global window numbers (compact location store)");
    const uint32_t n = in->num_queries, S = in->num_sources;
    const uint32_t K = ctx->cfg.max_candidates;
    if (!lane_candidates_supported(K)) return fail(ctx, MC_ERR_UNSUPPORTED, "mc_candidates_from_partial_numbers: max_candidates above 4");
    if ((uint64_t)n * S > 0xFFFFFFF0ull) return fail(ctx, MC_ERR_UNSUPPORTED, "mc_candidates_from_partial_numbers: batch too large");
    const uint64_t totalIn = in->source_offsets[S];
    if (totalIn && !in->numbers) return MC_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((flags & MC_SECOND_PIPE) && !ctx->pipe1.stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->pipe1.stream, hipStreamNonBlocking));
    Pipe& P = (flags & MC_SECOND_PIPE) ? ctx->pipe1 : ctx->pipe0;
    if (P.tail.pending) { const int rcf = finish_on_pipe(ctx, P); if (rcf) return rcf; }   // (a deferred tail still owns the workspace this call is about to reuse)
    hipStream_t st = streamv ? (hipStream_t)streamv : (flags & MC_SECOND_PIPE) ? ctx->pipe1.stream : ctx->stream;
    const uint32_t* taxkey = nullptr;
    int rc = taxkey_for_rank(ctx, lowestRank, &taxkey);
    if (rc) return rc;
    const uint64_t avg = totalIn / std::max<uint32_t>(n, 1);
    const uint64_t poolCap = std::min<uint64_t>(0xFFFFFFF0ull, std::max<uint64_t>(std::max<uint64_t>((uint64_t)n * 448, totalIn / 2),
                                                                                  (uint64_t)big_filter_grid(n, true, ctx->filterBpc) * 4 * std::min<uint64_t>(131072, std::max<uint64_t>(4096, avg))));
    const uint64_t ovfCap = std::min<uint64_t>(0xFFFFFFF0ull - poolCap, std::max<uint64_t>(poolCap / 4, 8ull << 20));
    if ((rc = ensure(ctx, P.bPsize, ((size_t)n * S + 4) * 4)) || (rc = ensure(ctx, P.bPpay, ((size_t)n * S + (size_t)S * (n + 2) + 4) * 8)) ||
        (rc = ensure(ctx, P.bQstat, (size_t)(n + 1) * sizeof(QueryStat))) || (rc = ensure(ctx, P.bQflag, (size_t)(n + 1) * 4)) ||
        (rc = ensure(ctx, P.bScanIn, (size_t)(n + 1) * 4)) || (rc = ensure(ctx, P.bScan, scan_tmp_bytes(n + 1))) ||
        (rc = ensure(ctx, P.bHitOff, (size_t)(n + 2) * 8)) || (rc = ensure(ctx, P.bMid, 128 + (size_t)8 * std::max<uint32_t>(n, 1) * 16)) ||
        (rc = ensure(ctx, P.bBigPool, (poolCap + ovfCap) * 4)) || (rc = ensure(ctx, P.bSliceFill, (size_t)big_filter_grid(n, true, ctx->filterBpc) * 4 * 4 + 64)) ||
        (rc = ensure(ctx, P.bSide, (size_t)5 * std::max<uint32_t>(n, 1) * 4)) ||
        (rc = ensure(ctx, P.bCands, (size_t)std::max<uint32_t>(n, 1) * K * sizeof(mc_candidate))))
        return rc;
    Workspace ws{};
    ws.filterBpc = ctx->filterBpc; ws.countBpc = ctx->countBpc; ws.gwFuse = ctx->gwFuse; ws.gwBigH = ctx->gwBigH; ws.gwMidH = ctx->gwMidH;
    ws.psize = (uint32_t*)P.bPsize.p; ws.ppay = (uint64_t*)P.bPpay.p;
    uint64_t* srcStart = ws.ppay + (size_t)n * S + 2;                 // [S][n + 1] exclusive scans of the sources' counts
    ws.qstat = (QueryStat*)P.bQstat.p; ws.qflag = (uint32_t*)P.bQflag.p; ws.hitScan = (uint32_t*)P.bScanIn.p; ws.hitOff = (uint64_t*)P.bHitOff.p;
    ws.scanTmp = P.bScan.p;
    ws.midCount = (uint32_t*)P.bMid.p; ws.midList = ws.midCount + 32;
    ws.bigMin = ctx->bigMin;
    ws.bigPool = (uint64_t*)P.bBigPool.p; ws.bigPoolCap = (uint32_t)poolCap; ws.bigOvfCap = (uint32_t)ovfCap;
    ws.sliceFill = (uint32_t*)P.bSliceFill.p; ws.sideList = (uint32_t*)P.bSide.p;
    BatchView b{nullptr, nullptr, in->max_win, in->max_win_uniform, n};
    // the receive buffer stands in for the table's location store
    DeviceTable tab{nullptr, nullptr, 0, 0xFFFFFFFFu, 1};
    tab.values32 = in->numbers;
    tab.gwBase = ctx->dGwBase; tab.gwDir = ctx->dGwDir; tab.gwDirShift = ctx->gwDirShift; tab.gwGap = ctx->gwGap; tab.gwTargets = ctx->gwTargets;
    KeyshardBases bases{};
    for (uint32_t s = 0; s < S; ++s) bases.b[s] = in->source_offsets[s];
    HIP_TRY(ctx, hipMemsetAsync(ws.midCount, 0, 128, st));
    {
        ScopedTimer t(ctx, "owner_entries", st);
        for (uint32_t s = 0; s < S; ++s) launch_scan_u32(in->counts + (size_t)s * n, 1, n, nullptr, srcStart + (size_t)s * (n + 1), ws.scanTmp, st);
        launch_owner_entries(b, tab, ws, in->counts, srcStart, bases, S, st);
    }
    const SketchParams one{16, 1, 16, 1};
    if ((rc = run_filtered_path(ctx, P, b, one, tab, ws, K, taxkey, true, true, poolCap + ovfCap, st))) return rc;
    // what is left (short lists, reads the filtered path handed back): decoded to (target, window) lists and sorted
    launch_scan_u32(ws.hitScan, 1, n, nullptr, ws.hitOff, ws.scanTmp, st);
    if (!P.hTotal) HIP_TRY(ctx, hipHostMalloc((void**)&P.hTotal, 128));
    HIP_TRY(ctx, hipMemcpyAsync(P.hTotal, ws.hitOff + n, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(P.hTotal + 10, ws.midCount + 9, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, traced_sync(st));
    ctx->ownerStats[0] += n; ctx->ownerStats[1] += *reinterpret_cast<const uint32_t*>(P.hTotal + 10); ctx->ownerStats[2] += totalIn; ctx->ownerStats[3] += *P.hTotal;
    const size_t hb = (size_t)(*P.hTotal + 1) * 8;
    if ((rc = ensure(ctx, P.bHits, hb)) || (rc = ensure(ctx, P.bCscr, hb)) || (taxkey && (rc = ensure(ctx, P.bCscr2, hb)))) return rc;
    ws.hits = (uint64_t*)P.bHits.p; ws.cscr = (uint64_t*)P.bCscr.p; ws.cscr2 = (uint64_t*)P.bCscr2.p;
    { ScopedTimer t(ctx, "decode_union", st); launch_decode_union(b, tab, ws, in->counts, srcStart, bases, S, st); }
    DeviceTable none{nullptr, nullptr, 0, 0xFFFFFFFFu, 1};
    { ScopedTimer t(ctx, "cands_from_hits", st); launch_cands_from_hits(b, none, ws, taxkey, K, P.bCands.p, st); }
    HIP_TRY(ctx, hipGetLastError());
    P.lastN = 0;
    out->cands = (const mc_candidate*)P.bCands.p;
    out->hit_counts = (const uint32_t*)P.bQstat.p;
    out->hit_offsets = nullptr; out->hits = nullptr; out->features = nullptr; out->win_offsets = nullptr;
    return MC_OK;
}

int mc_owner_stats(const mc_ctx* ctx, uint64_t stats[4])
{
    if (!ctx || !stats) return MC_ERR_INVALID;
    for (int i = 0; i < 4; ++i) stats[i] = ctx->ownerStats[i];
    return MC_OK;
}

// Mode P / part groups: per-part candidate lists of the same reads (device pointers, [n][max_candidates] each, in part order) -> one list
// per read, as if the parts had been queried one after the other with one candidate list (candidate_generation.hpp:172-231).  Target ids
// must be the database's own (mc_open_database with single_part keeps them), lowest_rank > 0 uses this context's lineages.
int mc_merge_part_candidates(mc_ctx* ctx, const mc_candidate* const* lists, uint32_t numLists, uint32_t n, int lowestRank, mc_candidate* out, void* streamv)
{
    if (!ctx || !lists || !out || numLists == 0) return MC_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = streamv ? (hipStream_t)streamv : ctx->stream;
    const uint32_t* taxkey = nullptr;
    int rc = taxkey_for_rank(ctx, lowestRank, &taxkey);
    if (rc) return rc;
    const uint32_t K = ctx->cfg.max_candidates;
    // more than 16 lists: in rounds, the merged list of a round leads the next one (the insert is sequential anyway)
    const void* ptrs[16];
    uint32_t done = 0;
    bool lead = false;
    while (done < numLists) {
        uint32_t m = 0;
        if (lead) ptrs[m++] = out;
        while (m < 16 && done < numLists) ptrs[m++] = lists[done++];
        if (launch_merge_parts(ptrs, m, n, K, taxkey, out, st) != 0) return fail(ctx, MC_ERR_UNSUPPORTED, "mc_merge_part_candidates: max_candidates above 4");
        lead = true;
    }
    HIP_TRY(ctx, hipGetLastError());
    return MC_OK;
}

// tuning / test hook: the switches the MC_* environment variables set at mc_create, changeable on a live context (no batch in flight)
int mc_set_tuning(mc_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return MC_ERR_INVALID;
    const std::string n(name);
    if (n == "big_min") ctx->bigMin = (uint32_t)std::max<int64_t>(0, value);
    else if (n == "quad_lookup") ctx->quadLookup = value < 0 ? -1 : (value != 0);
    else if (n == "lane_path") ctx->useLanePath = value != 0;
    else if (n == "compact_locations") ctx->compactAllowed = value != 0;      // before mc_load_begin
    else if (n == "filter_bpc") ctx->filterBpc = (int)value;                  // blocks per CU of the filter kernels' persistent grids (0 = default); this context only
    else if (n == "count_bpc") ctx->countBpc = (int)value;
    else if (n == "direct_index") {                                // before the table is loaded; on a loaded table: 0 drops the index, 1 / -1 builds it now (by the rules above)
        ctx->directWant = value < 0 ? -1 : (value != 0);
        if (ctx->tableReady && ctx->parts.size() == 1) {
            Part& T = ctx->parts[0];
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            const bool wanted = value > 0 || (value < 0 && (uint64_t)T.nbuckets * sizeof(TableBucket) >= (8ull << 30));
            if (!wanted && T.ddirect) { HIP_TRY(ctx, hipDeviceSynchronize()); (void)big_free(T.ddirect); T.ddirect = nullptr; }
            else if (wanted) (void)build_direct_index(ctx);
        }
    }
    else if (n == "list_align") ctx->listAlignWant = value < 0 ? -1 : (value != 0);   // before the table is loaded: lists of the compact store on lines of their own
    else if (n == "lane_fusion") ctx->fuseLane = value < 0 ? -1 : (value != 0);   // sketch + probe of the lane path in one kernel (-1: where the lookups are quad-cooperative)
    else if (n == "gw_mid_h") ctx->gwMidH = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 32768));   // reads up to this many locations: the stream filter's small-filter instance (0 = none; default 8 192)
    else if (n == "gw_big_h") ctx->gwBigH = value <= 0 ? 0xFFFFFFFFu : (uint32_t)std::min<int64_t>(value, 0xFFFFFFFFll);   // reads beyond this many locations: the stream filter's fine-block instance (0 = none; default 32 768)
    else if (n == "gw_fuse") ctx->gwFuse = (value == 5 || value == 6) ? (int)value : (value != 0);                         // counting of short filtered lists inside the filter kernel: 1 (default) = fused, 0 = the two kernels apart
    else return fail(ctx, MC_ERR_INVALID, "mc_set_tuning: unknown switch '" + n + "'");
    return MC_OK;
}

int mc_synchronize(mc_ctx* ctx)
{
    if (!ctx) return MC_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pipe1.stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->pipe1.stream));
    return MC_OK;
}

int mc_query_wait(mc_ctx* ctx, int flags)
{
    if (!ctx) return MC_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (flags & MC_SECOND_PIPE) ? ctx->pipe1.stream : ctx->stream;
    if (st) HIP_TRY(ctx, traced_sync(st));
    return MC_OK;
}

int mc_copy_results(mc_ctx* ctx, void* dst, const void* src, uint64_t bytes, int kind) { return mc_copy_results_on(ctx, dst, src, bytes, kind, nullptr); }

int mc_copy_results_on(mc_ctx* ctx, void* dst, const void* src, uint64_t bytes, int kind, void* stream)
{
    if (!ctx || !dst || !src) return MC_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // kind: bit 0 = device -> host, bit 1 = host -> device (else device -> device); MC_SECOND_PIPE: on the second pipe's stream when no stream is given
    if ((kind & MC_SECOND_PIPE) && !stream && !ctx->pipe1.stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->pipe1.stream, hipStreamNonBlocking));
    hipStream_t st = stream ? (hipStream_t)stream : (kind & MC_SECOND_PIPE) ? ctx->pipe1.stream : ctx->stream;
    const hipMemcpyKind how = (kind & 1) ? hipMemcpyDeviceToHost : (kind & 2) ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, how, st));
    return MC_OK;
}

int mc_last_batch_stats(mc_ctx* ctx, uint64_t stats[8])
{
    if (!ctx) return MC_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::memset(stats, 0, 64);
    Pipe& P = ctx->pipe0;
    if (P.tail.pending) { const int rcf = finish_on_pipe(ctx, P); if (rcf) return rcf; }   // (a deferred tail still owns the workspace this call is about to reuse)
    if (!P.bStats.p || !P.bQstat.p) return MC_OK;
    Workspace ws{};
    ws.filterBpc = ctx->filterBpc; ws.countBpc = ctx->countBpc; ws.gwFuse = ctx->gwFuse; ws.gwBigH = ctx->gwBigH; ws.gwMidH = ctx->gwMidH;
    ws.qstat = (QueryStat*)P.bQstat.p; ws.winOff = (uint32_t*)P.bWinOff.p; ws.stats = (uint64_t*)P.bStats.p;
    if (P.bMid.p) { ws.midCount = (uint32_t*)P.bMid.p; ws.midList = ws.midCount + 32; }
    launch_batch_stats(ws, P.lastN, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(stats, ws.stats, 64, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MC_OK;
}

int mc_timing_enable(mc_ctx* ctx, int on) { if (!ctx) return MC_ERR_INVALID; ctx->timing = on != 0; return MC_OK; }
int mc_timing_reset(mc_ctx* ctx)
{
    if (!ctx) return MC_ERR_INVALID;
    collect_timers(ctx);
    for (auto& kv : ctx->timers) { kv.second.ms = 0; kv.second.launches = 0; }
    return MC_OK;
}
int mc_timing_get(mc_ctx* ctx, const char* kernel, double* ms, uint64_t* launches)
{
    if (!ctx || !kernel) return MC_ERR_INVALID;
    collect_timers(ctx);
    auto it = ctx->timers.find(kernel);
    if (ms) *ms = it == ctx->timers.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->timers.end() ? 0 : it->second.launches;
    return MC_OK;
}

// ------------------------------------------------------------------------------------------------
// host batch slots
// ------------------------------------------------------------------------------------------------
int mc_batch_add(mc_ctx* ctx, uint32_t slot, const char* s1, uint32_t l1, const char* s2, uint32_t l2, uint32_t maxWin)
{
    if (!ctx || slot >= ctx->slots.size()) return MC_ERR_INVALID;
    Slot& S = ctx->slots[slot];
    if (S.submitted) return fail(ctx, MC_ERR_STATE, "mc_batch_add: slot is in flight, call mc_batch_wait + mc_batch_clear");
    if (l2 == kNoTail) return fail(ctx, MC_ERR_INVALID, "sequence too long");
    const uint64_t need = ((uint64_t)l1 + 3) / 4 * 4 + ((uint64_t)l2 + 3) / 4 * 4;
    if (need > ctx->cfg.slot_max_chars) return fail(ctx, MC_ERR_INVALID, "query longer than the slot's character capacity");
    if (S.nq >= ctx->cfg.slot_max_queries || S.nchars + need > ctx->cfg.slot_max_chars) return MC_BATCH_FULL;
    uint32_t* qi = S.hqinfo + (size_t)S.nq * 4;
    qi[0] = (uint32_t)S.nchars; qi[1] = l1;
    if (l1) std::memcpy(S.hseq + S.nchars, s1, l1);
    S.nchars += ((uint64_t)l1 + 3) / 4 * 4;
    qi[2] = (uint32_t)S.nchars; qi[3] = l2;
    if (l2) std::memcpy(S.hseq + S.nchars, s2, l2);
    S.nchars += ((uint64_t)l2 + 3) / 4 * 4;
    S.hmaxwin[S.nq] = maxWin;
    if (l2 == 0 && l1 > S.maxSingle) S.maxSingle = l1;
    S.nq++;
    return MC_OK;
}

int64_t mc_batch_add_bulk(mc_ctx* ctx, uint32_t slot, const char* seqs, const uint64_t* offs, uint64_t n, uint64_t insertMax)
{
    if (!ctx || slot >= ctx->slots.size() || !seqs || !offs) return MC_ERR_INVALID;
    const uint64_t stride = ctx->targetSketch.stride ? ctx->targetSketch.stride : 1;
    uint64_t i = 0;
    for (; i < n; ++i) {
        const uint64_t len = offs[i + 1] - offs[i];
        if (len >= 0xFFFFFFF0ull) return fail(ctx, MC_ERR_INVALID, "sequence too long");
        const uint32_t maxWin = (uint32_t)(2 + std::max<uint64_t>(len, insertMax) / stride);
        const int rc = mc_batch_add(ctx, slot, seqs + offs[i], (uint32_t)len, nullptr, 0, maxWin);
        if (rc == MC_BATCH_FULL) break;
        if (rc < 0) return rc;
    }
    return (int64_t)i;
}

int mc_batch_submit(mc_ctx* ctx, uint32_t slot, int lowestRank)
{
    if (!ctx || slot >= ctx->slots.size()) return MC_ERR_INVALID;
    Slot& S = ctx->slots[slot];
    if (S.submitted) return fail(ctx, MC_ERR_STATE, "mc_batch_submit: slot already submitted");
    // every slot has its own stream and device workspace: batches of different slots overlap on the device (the reference orders
    // submissions with a mutex and overlaps through per-batch CUDA streams, database_query.hpp:110-113, query_batch.cu)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->coalesce) {
        // hand the slot to the dispatchers: whatever is waiting when one of them comes free goes to the device as ONE batch
        S.submittedQueries = S.nq; S.coLowest = lowestRank; S.coRc = MC_OK; S.coEvent = false; S.coErr.clear();
        S.submitted = true;
        mcamd::CoDispatcher* D = nullptr;
        {
            std::lock_guard<std::mutex> l(ctx->coMu);
            if (S.nq == 0) S.coState = 3;
            else if (ctx->coPending.empty() && !ctx->coFree.empty()) {
                // nothing is waiting and a pipe is free: this slot goes out now, on the submitter's own thread -- no hand-over (a lone
                // submitter's batch takes what it took before there was a coalescer)
                D = ctx->coFree.back(); ctx->coFree.pop_back();
                S.coState = 2;
                ctx->coBatches++; ctx->coSlots++;
            }
            else { S.coState = 1; ctx->coPending.push_back(slot); }
        }
        if (D) {
            co_run(ctx, D, std::vector<uint32_t>{slot}, lowestRank);
            { std::lock_guard<std::mutex> l(ctx->coMu); ctx->coFree.push_back(D); }
            ctx->coCv.notify_one();
        } else if (S.nq) ctx->coCv.notify_one();
        return MC_OK;
    }
    const uint64_t tt0 = g_submitTrace ? trace_now() : 0;
    {
        std::unique_lock<std::mutex> lk(ctx->pipeMtx);
        ctx->pipeCv.wait(lk, [&] { return !ctx->freePipes.empty(); });
        S.pipe = ctx->freePipes.back();
        ctx->freePipes.pop_back();
    }
    const uint64_t tt1 = g_submitTrace ? trace_now() : 0;
    struct Giveback {                                           // an error on the way returns the pipe at once
        mc_ctx* c; Slot& s; bool armed = true;
        ~Giveback() { if (armed && s.pipe) { (void)hipStreamSynchronize(s.pipe->stream); std::lock_guard<std::mutex> l(c->pipeMtx); c->freePipes.push_back(s.pipe); s.pipe = nullptr; c->pipeCv.notify_one(); } }
    } giveback{ctx, S};
    Pipe& P = *S.pipe;
    hipStream_t st = P.stream;
    const uint32_t n = S.nq;
    S.submittedQueries = n;
    if (n == 0) { HIP_TRY(ctx, hipEventRecord(S.done, st)); S.submitted = true; return MC_OK; }
    const uint64_t te0 = trace_now();
    HIP_TRY(ctx, hipMemcpyAsync(S.dseq, S.hseq, S.nchars + 16, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(S.dqinfo, S.hqinfo, (size_t)n * 16, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(S.dmaxwin, S.hmaxwin, (size_t)n * 4, hipMemcpyHostToDevice, st));
    note_enqueues(trace_now() - te0, 3);
    mc_device_batch in{S.dseq, S.dqinfo, S.dmaxwin, 0, n, S.nchars};
    mc_device_results res{};
    const int wantAll = ctx->cfg.copy_allhits ? 1 : 0;
    const uint64_t tt2 = g_submitTrace ? trace_now() : 0;
    int rc = query_on_pipe(ctx, P, &in, lowestRank, wantAll | (S.maxSingle <= mcamd::lane_max_len() ? mcamd::kQueryNoLongReads : 0), &res, st);
    if (rc) return rc;
    const uint64_t tt3 = g_submitTrace ? trace_now() : 0;
    const size_t K = ctx->cfg.max_candidates;
    {   // candidates and statistics to the slot's pinned buffers by one kernel (launch_deliver) instead of two copies
        DeliverTable dt{};
        dt.e[0] = DeliverEntry{S.hcands, S.hqstat, 0u, n}; dt.n = 1;
        launch_deliver(dt, res.cands, P.bQstat.p, (uint32_t)K, st);
    }
    if (wantAll) {
        HIP_TRY(ctx, hipMemcpyAsync(S.hhitoff, res.hit_offsets, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, traced_sync(st));
        const uint64_t total = S.hhitoff[n];
        if (total > S.hhitsCap) {
            if (S.hhits) HIP_TRY(ctx, hipHostFree(S.hhits));
            S.hhits = nullptr; S.hhitsCap = 0;
            HIP_TRY(ctx, hipHostMalloc((void**)&S.hhits, (total + total / 4 + 64) * sizeof(mc_location)));
            S.hhitsCap = total + total / 4 + 64;
        }
        if (total) HIP_TRY(ctx, hipMemcpyAsync(S.hhits, res.hits, total * sizeof(mc_location), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipEventRecord(S.done, st));
    if (g_submitTrace) { const uint64_t tt4 = trace_now(); g_trace[0] += tt1 - tt0; g_trace[1] += tt2 - tt1; g_trace[2] += tt3 - tt2; g_trace[4] += tt4 - tt3; ++g_trace[5]; }
    S.submitted = true;
    giveback.armed = false;                                     // mc_batch_wait returns the pipe
    return MC_OK;
}

// ---- slot coalescer -----------------------------------------------------------------------------------------------------------------
// The reference's consumer threads submit batches of 4 096 reads (options.hpp:229-232) and order their submissions with a mutex
// (database_query.hpp:110-113).  Here a batch of that size is ~30 kernel launches and three host round trips for ~0.1 ms of device
// work: one batch per slot keeps the device waiting for the host.  So submissions are QUEUED, and a few dispatcher threads -- a pipe
// each -- take whatever is waiting (same lowest rank, up to coMaxQueries reads) as ONE device batch: every slot's characters go to
// their place in the united input (an H2D each, out of the slot's pinned buffer), the qinfo rows are rebased on the host, one
// query_on_pipe, and every slot's candidates and statistics come back into its own pinned buffers (`done` behind them).  Under load
// the united batches grow by themselves; a lone submitter pays a thread hand-over (~20 us).  Only these threads make HIP calls for
// the slots -- dozens of host threads enqueueing on eight streams is what sends the runtime's direct dispatch into its slow state.
// takes what is waiting (front of the queue, one lowest rank, up to the united batch's limits); coMu held
static void co_take(mc_ctx* ctx, std::vector<uint32_t>& mine, int& lowest)
{
    uint64_t nq = 0, nc = 0;
    lowest = ctx->slots[ctx->coPending.front()].coLowest;
    while (!ctx->coPending.empty()) {
        Slot& S = ctx->slots[ctx->coPending.front()];
        if (!mine.empty() && (S.coLowest != lowest || nq + S.nq > ctx->coMaxQueries || nc + S.nchars + 16 > ctx->coMaxChars)) break;
        nq += S.nq; nc += S.nchars;
        S.coState = 2;
        mine.push_back(ctx->coPending.front());
        ctx->coPending.pop_front();
    }
    ctx->coBatches++; ctx->coSlots += mine.size();
}
static void co_run(mc_ctx* ctx, mcamd::CoDispatcher* D, const std::vector<uint32_t>& mine, int lowest);

static void co_dispatch(mc_ctx* ctx, mcamd::CoDispatcher*)
{
    (void)hipSetDevice(ctx->device);
    std::vector<uint32_t> mine;
    for (;;) {
        mine.clear();
        int lowest = 0;
        mcamd::CoDispatcher* D = nullptr;
        {
            std::unique_lock<std::mutex> l(ctx->coMu);
            ctx->coCv.wait(l, [&] { return (ctx->coStop && ctx->coPending.empty()) || (!ctx->coPending.empty() && !ctx->coFree.empty()); });
            if (ctx->coPending.empty()) return;                    // (stop: nothing is waiting any more)
            D = ctx->coFree.back(); ctx->coFree.pop_back();
            co_take(ctx, mine, lowest);
        }
        co_run(ctx, D, mine, lowest);
        { std::lock_guard<std::mutex> l(ctx->coMu); ctx->coFree.push_back(D); }
        ctx->coCv.notify_one();                                     // (a pipe is free again: whoever waits for one)
    }
}

// one united batch on D's pipe, by the calling thread: a dispatcher, or the submitter itself when nothing was waiting and a pipe was free
static void co_run(mc_ctx* ctx, mcamd::CoDispatcher* D, const std::vector<uint32_t>& mine, int lowest)
{
    std::string err;
    std::string* const sinkBefore = t_errSink;
    t_errSink = &err;
    Pipe& P = *D->pipe;
    hipStream_t st = P.stream;
    const size_t K = ctx->cfg.max_candidates;
    {
        uint64_t nq = 0, nc = 0;
        uint32_t maxSingle = 0;
        for (uint32_t s : mine) { nq += ctx->slots[s].nq; nc += ctx->slots[s].nchars; }
        int rc = MC_OK;
        err.clear();
        const uint32_t k = D->turn++ & 1u;
        auto hip = [&](hipError_t e, const char* what) { if (e != hipSuccess && !rc) { rc = MC_ERR_HIP; err = std::string(what) + ": " + hipGetErrorString(e); } };
        if (D->stagedUsed[k]) hip(hipEventSynchronize(D->staged[k]), "hipEventSynchronize");   // (two united batches ago: long through)
        if (!rc) rc = ensure(ctx, D->dseq, nc + 64);
        if (!rc) rc = ensure(ctx, D->dqinfo, nq * 20);
        if (!rc) {
            const uint64_t ttA = g_submitTrace ? trace_now() : 0;
            uint64_t qb = 0, cb = 0, enq = 0;
            // the rows of the queries and their window limits in ONE pinned buffer and one copy: [nq x 16 bytes | nq x 4 bytes]
            uint32_t* hq = D->hq[k]; uint32_t* hmw = hq + nq * 4;
            for (uint32_t s : mine) {
                Slot& S = ctx->slots[s];
                const bool last = s == mine.back();
                if (last) std::memset(S.hseq + S.nchars, 0, 16);   // (the united input ends with 16 zero bytes: they travel with the last slot's characters)
                const uint64_t te0 = trace_now();
                hip(hipMemcpyAsync((uint8_t*)D->dseq.p + cb, S.hseq, S.nchars + (last ? 16 : 0), hipMemcpyHostToDevice, st), "hipMemcpyAsync");
                enq += trace_now() - te0;
                if (S.maxSingle > maxSingle) maxSingle = S.maxSingle;
                for (uint32_t j = 0; j < S.nq; ++j) {
                    const uint32_t* q = S.hqinfo + (size_t)j * 4;
                    uint32_t* o = hq + (qb + j) * 4;
                    o[0] = q[0] + (uint32_t)cb; o[1] = q[1]; o[2] = q[2] + (uint32_t)cb; o[3] = q[3];
                }
                std::memcpy(hmw + qb, S.hmaxwin, (size_t)S.nq * 4);
                qb += S.nq; cb += S.nchars;
            }
            hip(hipMemcpyAsync(D->dqinfo.p, hq, nq * 20, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
            hip(hipEventRecord(D->staged[k], st), "hipEventRecord");
            D->stagedUsed[k] = true;
            note_enqueues(enq, (uint32_t)mine.size());
            if (g_submitTrace) g_trace[1] += trace_now() - ttA;
        }
        mc_device_results res{};
        if (!rc) {
            mc_device_batch in{(const uint8_t*)D->dseq.p, (const uint32_t*)D->dqinfo.p, (const uint32_t*)D->dqinfo.p + nq * 4, 0, (uint32_t)nq, nc};
            // (MC_SLOT_DEFER=1: deferred tail, finished at once -- the main kernels go out without a look at the work lists' counters: two
            // host round trips less per united batch, every kernel of the path launched whether it has work or not.  Measured at 4 096
            // reads per slot, 32 threads: 4 019 against 4 189 Mreads/min for the synchronous call, profiles/r06_slot_path.json: off)
            static const bool defer = [] { const char* e = std::getenv("MC_SLOT_DEFER"); return e && e[0] == '1'; }();
            const uint64_t ttB = g_submitTrace ? trace_now() : 0;
            rc = query_on_pipe(ctx, P, &in, lowest, (defer ? MC_DEFER_TAIL : 0) | (maxSingle <= mcamd::lane_max_len() ? mcamd::kQueryNoLongReads : 0), &res, st);
            if (!rc && defer) rc = finish_on_pipe(ctx, P);
            if (g_submitTrace) { g_trace[2] += trace_now() - ttB; ++g_trace[5]; }
        }
        if (!rc) {
            const uint64_t ttC = g_submitTrace ? trace_now() : 0;
            // every slot's candidates and statistics into its own pinned buffers: one kernel per sixteen slots (launch_deliver) -- two copies
            // per slot at ~15 us each were half a united batch's time on the stream --, then the slots' events behind it
            uint64_t qb = 0;
            DeliverTable dt{};
            for (uint32_t s : mine) {
                Slot& S = ctx->slots[s];
                dt.e[dt.n++] = DeliverEntry{S.hcands, S.hqstat, (uint32_t)qb, S.nq};
                if (dt.n == kDeliverMax) { launch_deliver(dt, res.cands, P.bQstat.p, (uint32_t)K, st); dt.n = 0; }
                qb += S.nq;
            }
            launch_deliver(dt, res.cands, P.bQstat.p, (uint32_t)K, st);
            hip(hipGetLastError(), "deliver_kernel");
            hipEvent_t ev = D->done[D->doneTurn++ & 3u];
            hip(hipEventRecord(ev, st), "hipEventRecord");
            for (uint32_t s : mine) ctx->slots[s].coDoneEv = ev;
            if (g_submitTrace) g_trace[4] += trace_now() - ttC;
        }
        if (rc) (void)hipStreamSynchronize(st);                     // (whatever was enqueued is through before the slots' buffers go back to their owners)
        {
            std::lock_guard<std::mutex> l(ctx->coMu);
            for (uint32_t s : mine) { Slot& S = ctx->slots[s]; S.coRc = rc; S.coErr = err; S.coEvent = rc == MC_OK; S.coState = 3; }
        }
        ctx->coDoneCv.notify_all();
    }
    t_errSink = sinkBefore;
}

static void release_pipe(mc_ctx* ctx, Slot& S)
{
    if (!S.pipe) return;
    std::lock_guard<std::mutex> l(ctx->pipeMtx);
    ctx->freePipes.push_back(S.pipe);
    S.pipe = nullptr;
    ctx->pipeCv.notify_one();
}

int mc_batch_wait(mc_ctx* ctx, uint32_t slot, mc_results* out)
{
    if (!ctx || slot >= ctx->slots.size() || !out) return MC_ERR_INVALID;
    Slot& S = ctx->slots[slot];
    if (!S.submitted) return fail(ctx, MC_ERR_STATE, "mc_batch_wait: slot not submitted");
    if (ctx->coalesce) {
        { std::unique_lock<std::mutex> l(ctx->coMu); ctx->coDoneCv.wait(l, [&] { return S.coState == 3; }); }
        if (S.coRc) return fail(ctx, S.coRc, S.coErr);
        if (S.coEvent) HIP_TRY(ctx, hipEventSynchronize(S.coDoneEv));
    } else
    HIP_TRY(ctx, hipEventSynchronize(S.done));
    release_pipe(ctx, S);                                       // the results are in the slot's pinned buffers
    const uint32_t n = S.submittedQueries;
    int status = MC_OK;
    for (uint32_t i = 0; i < n; ++i) {
        S.hhitcounts[i] = S.hqstat[i].hits;
        if (S.hqstat[i].hits > kMaxHitsPerQuery) status = MC_ERR_UNSUPPORTED;
    }
    out->num_queries = n;
    out->max_candidates = ctx->cfg.max_candidates;
    out->cands = S.hcands;
    out->hit_counts = S.hhitcounts;
    out->hit_offsets = ctx->cfg.copy_allhits ? S.hhitoff : nullptr;
    out->hits = ctx->cfg.copy_allhits ? S.hhits : nullptr;
    if (status) return fail(ctx, status, "a query produced more than 2^20-1 location hits (unsupported)");
    return MC_OK;
}

const char* mc_runtime_warning(void)
{
    static thread_local std::string copy;
    { std::lock_guard<std::mutex> l(g_warnMu); copy = g_warning; }
    return copy.c_str();
}

int mc_slot_stats(mc_ctx* ctx, uint64_t stats[4])
{
    if (!ctx || !stats) return MC_ERR_INVALID;
    std::lock_guard<std::mutex> l(ctx->coMu);
    stats[0] = ctx->coalesce ? 1 : 0; stats[1] = ctx->coBatches; stats[2] = ctx->coSlots; stats[3] = ctx->coDisp.size();
    return MC_OK;
}

int mc_batch_clear(mc_ctx* ctx, uint32_t slot)
{
    if (!ctx || slot >= ctx->slots.size()) return MC_ERR_INVALID;
    Slot& S = ctx->slots[slot];
    if (S.submitted && ctx->coalesce) {
        { std::unique_lock<std::mutex> l(ctx->coMu); ctx->coDoneCv.wait(l, [&] { return S.coState == 3; }); }
        if (S.coEvent) (void)hipEventSynchronize(S.coDoneEv);
        S.coState = 0; S.coEvent = false;
    } else if (S.submitted) (void)hipEventSynchronize(S.done);
    release_pipe(ctx, S);
    S.submitted = false; S.nq = 0; S.nchars = 0; S.maxSingle = 0;
    return MC_OK;
}

}  // extern "C"
