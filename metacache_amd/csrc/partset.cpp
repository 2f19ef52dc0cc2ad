// metacache_amd/csrc/partset.cpp -- a partitioned database queried PART GROUP BY PART GROUP, parts spread over the GPUs of the node
// (C ABI: mc_partset_*, include/metacache_amd.h).
//
// What it replaces: (1) the reference's parts-over-GPUs query inside one process (gpu_hashmap.cu:1255-1290 query_hashtables_async over
// the parts' tables, query_batch.cu:464-527 forwarding the sketches GPU -> GPU), (2) its documented workflow for databases that do not
// fit the node (docs/partitioning.md:116-153: `query` one part at a time, `merge` the result files, mode_merge.cpp:247-296).
// Here: `resident` parts are in HBM at a time, one context each (mc_open_database with single_part), dealt out round-robin over the
// devices; while the reads run against group g a loader thread opens the parts of group g + 1 (their H2D copies and insert kernels
// run on those contexts' own streams: two table slots, copy behind compute).  Per batch every device queries its parts; the reads
// are dealt out to the devices as OWNERS (read i of a batch of m belongs to device i * owners / m), every device sends every owner
// its parts' top lists of that owner's reads (one grouped ncclSend / ncclRecv round: RCCL, one communicator rank per device, xGMI
// between GPUs -- 1 / owners of what an all-gather moves) and every owner merges its reads' lists IN PART ORDER (merge_parts_kernel
// -- the CPU's list insert, candidate_generation.hpp:172-231 -- behind the list the earlier groups left for the read) and copies its
// share to the host: the reference forwards the running top candidates GPU -> GPU (query_batch.cu:638-652); here the merge is spread
// over the devices instead of chained through them.  The result is what one candidate list fed by all parts in order holds: the
// intended semantics of host_hashmap.hpp:695-723 (the in-process reference itself is history dependent for more than one part,
// SURVEY 8a row 8).
//
// Two batches are in flight (round 6): a batch takes one of four pinned STAGING slots on the host and one of two LANES on the devices
// (a lane = per device a stream, the batch's input, the parts' lists, the exchanged lists -- and pipe 0 / 1 of every part's context).
// Its upload and main kernels are enqueued without any host synchronisation (mc_query_device MC_DEFER_TAIL); what needs the host --
// the tails of the parts' queries, exchange, merge, copy back -- is enqueued one batch later, when the next batch's upload and kernels
// are already queued behind it on the other lane.  Callers may be several threads (mcq's workers: a batch each): they share the lanes.
//
// RCCL is loaded at run time (rccl_dl.h).
#include "context.h"
#include "devcache.h"
#include "rccl_dl.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace mcamd;

namespace {

Rccl& g_rccl = rccl();                 // rccl_dl.h

constexpr uint32_t kLanes = 2, kStaging = 4, kInputs = 3;

struct DevInput {                      // one device's copy of a batch's input: uploaded on the device's upload stream, read by a lane
    uint8_t* dseq = nullptr; uint32_t* dqinfo = nullptr; uint32_t* dmaxwin = nullptr;
    mc_candidate* dprior = nullptr;    // the earlier groups' list of the reads this device owns
    hipEvent_t upDone = nullptr;       // upload stream: the batch is here
    hipEvent_t inFree = nullptr;       // lane stream: the batch that used this input has read it to its end (main kernels, tails, merge)
    bool used = false;
};
struct DevLane {                       // one device's side of a lane
    hipStream_t stream = nullptr;
    mc_candidate* dmine = nullptr;     // [slotsPerDev][maxQ][K]: this device's parts of the resident group, all reads of the batch
    mc_candidate* dall = nullptr;      // [ndev][slotsPerDev][maxQ][K]: every part's lists of the reads this device OWNS (rows 0 .. its share)
    mc_candidate* dout = nullptr;      // the merged lists of the owned reads
    std::vector<const mc_candidate*> cands;              // [slot]: where the part's context leaves the batch's top lists (its pipe's result buffer)
};
struct DevState {
    int device = 0;
    void* comm = nullptr;
    hipStream_t up = nullptr;          // uploads: never behind a lane's small kernels, which wait for CUs while the other lane's batch fills the device
    DevInput in[kInputs];
    DevLane lane[kLanes];
};

}  // namespace

struct mc_partset {
    std::string db, err;
    mc_config cfg{};
    uint32_t nparts = 0, resident = 1, K = 2, stride = 112;
    std::vector<int> devices;
    std::vector<DevState> dev;
    uint32_t slotsPerDev = 1;
    std::vector<mc_ctx*> cur, next;    // contexts of the resident group / of the group being loaded (part order)
    uint32_t curFirst = 0, nextFirst = 0;
    std::thread loader;
    int loaderRc = MC_OK;
    std::string loaderErr;
    size_t maxQ = 0, maxChars = 0;
    // host staging of a batch in pinned memory (pageable memory would make every hipMemcpyAsync a synchronous staged copy): a batch is
    // packed into a free slot while earlier ones are on the devices, its merged lists come back into the same slot
    struct Staging {
        uint8_t* seq = nullptr; uint32_t* q = nullptr; uint32_t* mw = nullptr;
        mc_candidate* prior = nullptr; mc_candidate* out = nullptr;
        std::vector<hipEvent_t> outDone;  // per device: the owner's share of the merged lists is in `out`
        uint64_t chars = 0;               // characters of the batch in `seq`
        uint32_t count = 0, owners = 0;   // reads of the batch, devices that own a share of them
        bool hasPrior = false;
        bool busy = false;
    } stg[kStaging];
    bool laneBusy[kLanes] = {false, false};
    bool inputBusy[kInputs] = {false, false, false};
    std::mutex poolMu;                 // staging slots and lanes
    std::condition_variable poolCv;
    std::mutex orderMu;                // the exchange's RCCL calls: one group at a time, the same order on every device
    uint64_t loadNs = 0, waitNs = 0;   // time the loader spent / the queries waited for it
    std::atomic<uint64_t> loadBytes{0}; // bytes of .cache files read by the group loads
    uint32_t packThreads = 8;          // host threads that copy a batch's characters into the staging buffer (MC_PARTSET_PACK_THREADS)
    bool ranges = false;               // the parts are target ranges of one file (cfg.target_shard_count > 1)
    bool rccl = false;                 // several devices (or MC_PARTSET_RCCL=1: the same calls with a single rank, tests): exchange over RCCL
};

namespace {

int ps_fail(mc_partset* ps, int code, const std::string& msg)
{
    static std::mutex mu;                                          // (several callers may fail at once: the last one's text stays)
    std::lock_guard<std::mutex> l(mu);
    if (ps) ps->err = msg; else set_global_error(msg);
    return code;
}

uint64_t now_ns() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }

void drop(mc_partset* ps) { delete ps; mcamd::big_cache_hold(-1); }   // (mc_partset_open before any device state exists: the hold goes back too)

void close_group(std::vector<mc_ctx*>& g) { for (mc_ctx* c : g) if (c) mc_destroy(c); g.clear(); }

// opens the parts [first, first + count) as one context each, part p on device devices[(p - first) % ndev] -- two loading threads per
// DEVICE (every load has its reader threads, its own copy stream and its device's PCIe link: dbload.cpp), as the reference reads every
// part in a thread of its own (database.cpp:203-226)
int open_group(mc_partset* ps, uint32_t first, std::vector<mc_ctx*>& out, std::string& err)
{
    const uint32_t count = std::min(ps->resident, ps->nparts - first);
    const uint32_t nd = (uint32_t)ps->devices.size();
    out.assign(count, nullptr);
    std::vector<int> rcs(nd, MC_OK);
    std::vector<std::string> errs(nd);
    std::mutex errMu;
    // a device's parts: two at a time (one part's index pass, table allocation and first batches run under the other's copies; more
    // than two only share the one PCIe link) -- MC_PARTSET_LOADS_PER_DEVICE
    const uint32_t perDevice = [] { const char* e = std::getenv("MC_PARTSET_LOADS_PER_DEVICE"); return (uint32_t)std::max(1, e ? std::atoi(e) : 2); }();
    std::vector<std::atomic<uint32_t>> nextOf(nd);
    for (uint32_t d = 0; d < nd; ++d) nextOf[d] = d;
    auto load_device = [&](uint32_t d) {
        for (;;) {
            const uint32_t i = nextOf[d].fetch_add(nd);
            if (i >= count) return;
            { std::lock_guard<std::mutex> l(errMu); if (rcs[d] != MC_OK) return; }
            mc_config c = ps->cfg;
            if (ps->ranges) { c.single_part = std::max(ps->cfg.single_part, 0); c.target_shard_index = first + i; c.target_shard_count = ps->nparts; }
            else c.single_part = (int32_t)(first + i);
            c.device = ps->devices[d];
            c.num_slots = 0; c.copy_allhits = 0;                   // (no host slots: the set has its own staging; the contexts' two pipes are sized below)
            // list alignment (up to 1.5 x the plain store) is a table's own decision against the device's free memory: with several tenants
            // per device it would be made against memory the group's later parts need.  Groups that stream (the next one loads beside the
            // resident one): plain stores; all parts resident: the padding may take its share of what is free.
            const uint32_t mine = (count - d + nd - 1) / nd, placed = i / nd;
            mcamd::open_hints().listAlign = ps->resident < ps->nparts ? 0 : -1;
            mcamd::open_hints().directIndex = 0;                     // (32 GiB per table: not beside other tenants)
            mcamd::open_hints().listAlignShare = 1.0 / (double)std::max<uint32_t>(1, mine - std::min(placed, mine - 1));
            const int rc = mc_open_database(ps->db.c_str(), &c, &out[i]);
            mcamd::open_hints() = mcamd::OpenHints{};
            if (rc != MC_OK) { std::lock_guard<std::mutex> l(errMu); rcs[d] = rc; errs[d] = mc_last_error(nullptr); return; }
            // the contexts' two pipes sized here, beside the other parts' loads -- where the device has room to spare (a group that fills it
            // leaves the workspaces to the first batches: they grow on demand)
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB > (24ull << 30))
                (void)mcamd::reserve_query_pipes(out[i], (uint32_t)ps->maxQ, std::min<uint64_t>(ps->maxChars, (uint64_t)ps->maxQ * 152));   // (a failure here is not one: the first batch asks again)
            uint64_t st[4] = {0, 0, 0, 0};
            if (mc_load_stats(out[i], st) == MC_OK) ps->loadBytes += st[0];
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t d = 0; d < std::min(nd, count); ++d) {
            const uint32_t mine = (count - d + nd - 1) / nd;       // parts of the group on this device
            for (uint32_t k = 0; k < std::min(perDevice, mine); ++k) th.emplace_back(load_device, d);
        }
        for (auto& t : th) t.join();
    }
    for (uint32_t d = 0; d < nd; ++d)
        if (rcs[d]) { err = errs[d]; close_group(out); return rcs[d]; }
    return MC_OK;
}

void start_loader(mc_partset* ps, uint32_t first)
{
    if (ps->loader.joinable()) ps->loader.join();              // (never assign to a joinable thread: std::terminate)
    close_group(ps->next);
    ps->nextFirst = first;
    ps->loaderRc = MC_OK;
    ps->loader = std::thread([ps, first] {
        const uint64_t t0 = now_ns();
        ps->loaderRc = open_group(ps, first, ps->next, ps->loaderErr);
        ps->loadNs += now_ns() - t0;
    });
}

}  // namespace

extern "C" {

const char* mc_partset_last_error(const mc_partset* ps) { return ps ? ps->err.c_str() : mc_last_error(nullptr); }

int mc_partset_open(const char* name, const mc_config* cfg, uint32_t residentParts, const int32_t* devices, uint32_t numDevices, mc_partset** out)
{
    if (!name || !cfg || !out) return MC_ERR_INVALID;
    *out = nullptr;
    mc_ctx* meta = nullptr;
    int rc = mc_open_metadata(name, &meta);
    if (rc) return rc;
    uint64_t info[8];
    mc_db_info(meta, info);
    mc_destroy(meta);
    auto* ps = new mc_partset;
    mcamd::big_cache_hold(+1);                                    // (the tables of a closed group are the next group's: devcache.h; released in mc_partset_close)
    ps->db = name; ps->cfg = *cfg;
    ps->nparts = (uint32_t)info[6];
    // Mode T: the "parts" are the target ranges of ONE part file (cfg.single_part, or part 0), cut at load (metacache_amd.h mc_config)
    ps->ranges = cfg->target_shard_count > 1;
    if (ps->ranges) {
        if (cfg->single_part >= (int32_t)ps->nparts) { drop(ps); return ps_fail(nullptr, MC_ERR_INVALID, "database part is not available"); }
        ps->nparts = cfg->target_shard_count;
    }
    ps->stride = (uint32_t)(info[3] ? info[3] : 112);
    ps->K = cfg->max_candidates;
    ps->resident = std::max<uint32_t>(1, std::min<uint32_t>(residentParts ? residentParts : ps->nparts, ps->nparts));
    if (ps->K > 4) { drop(ps); return ps_fail(nullptr, MC_ERR_UNSUPPORTED, "mc_partset_open: max_candidates above 4"); }
    int ndevAvail = 0;
    if (hipGetDeviceCount(&ndevAvail) != hipSuccess || ndevAvail < 1) { drop(ps); return ps_fail(nullptr, MC_ERR_HIP, "no usable HIP device (this library has no CPU fallback)"); }
    if (devices && numDevices) ps->devices.assign(devices, devices + numDevices); else ps->devices.assign(1, cfg->device);
    for (size_t i = 0; i < ps->devices.size(); ++i) {
        if (ps->devices[i] < 0 || ps->devices[i] >= ndevAvail) { drop(ps); return ps_fail(nullptr, MC_ERR_INVALID, "mc_partset_open: device ordinal out of range"); }
        for (size_t j = 0; j < i; ++j)
            if (ps->devices[j] == ps->devices[i]) { drop(ps); return ps_fail(nullptr, MC_ERR_INVALID, "mc_partset_open: a device is listed twice"); }
    }
    const uint32_t nd = (uint32_t)ps->devices.size();
    ps->slotsPerDev = (ps->resident + nd - 1) / nd;
    if (const char* e = std::getenv("MC_PARTSET_PACK_THREADS")) ps->packThreads = (uint32_t)std::max(1, std::atoi(e));
    ps->packThreads = std::min<uint32_t>(ps->packThreads, std::max(1u, std::thread::hardware_concurrency()));
    ps->maxQ = std::max<uint32_t>(cfg->slot_max_queries, 1);
    ps->maxChars = std::max<uint32_t>(cfg->slot_max_chars, 1u << 16);
    // RCCL: one communicator rank per device of this process (ncclCommInitAll).  A single device needs no gather (loading and
    // initialising RCCL takes a stand-alone process 25 s); MC_PARTSET_RCCL=1 runs the same calls with one rank (tests)
    const char* force = std::getenv("MC_PARTSET_RCCL");
    ps->rccl = nd > 1 || (force && force[0] == '1');
    std::vector<void*> comms(nd, nullptr);
    if (ps->rccl) {
        if (!g_rccl.load()) { const std::string e = g_rccl.err; drop(ps); return ps_fail(nullptr, MC_ERR_UNSUPPORTED, e); }
        if (int r = g_rccl.CommInitAll(comms.data(), (int)nd, ps->devices.data())) {
            const std::string e = std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
            drop(ps);
            return ps_fail(nullptr, MC_ERR_HIP, e);
        }
    }
    ps->dev.resize(nd);
    const size_t listBytes = ps->maxQ * ps->K * sizeof(mc_candidate);
    bool ok = true;
    for (uint32_t d = 0; d < nd && ok; ++d) {
        DevState& D = ps->dev[d];
        D.device = ps->devices[d]; D.comm = comms[d];
        ok = hipSetDevice(D.device) == hipSuccess && hipStreamCreateWithFlags(&D.up, hipStreamNonBlocking) == hipSuccess;
        for (DevInput& I : D.in) {
            if (!ok) break;
            ok = mcamd::dev_malloc((void**)&I.dseq, ps->maxChars + 64) == hipSuccess && mcamd::dev_malloc((void**)&I.dqinfo, ps->maxQ * 16) == hipSuccess &&
                 mcamd::dev_malloc((void**)&I.dmaxwin, ps->maxQ * 4) == hipSuccess && mcamd::dev_malloc((void**)&I.dprior, listBytes) == hipSuccess &&
                 hipEventCreateWithFlags(&I.upDone, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&I.inFree, hipEventDisableTiming) == hipSuccess;
        }
        for (DevLane& L : D.lane) {
            if (!ok) break;
            ok = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess &&
                 mcamd::dev_malloc((void**)&L.dmine, ps->slotsPerDev * listBytes) == hipSuccess && mcamd::dev_malloc((void**)&L.dout, listBytes) == hipSuccess;
            // (without RCCL -- one device -- the exchanged lists ARE the device's own)
            if (ok && ps->rccl) ok = mcamd::dev_malloc((void**)&L.dall, (size_t)nd * ps->slotsPerDev * listBytes) == hipSuccess;
        }
    }
    for (auto& H : ps->stg) {
        if (!ok) break;
        ok = hipHostMalloc((void**)&H.seq, ps->maxChars + 64) == hipSuccess && hipHostMalloc((void**)&H.q, ps->maxQ * 16) == hipSuccess &&
             hipHostMalloc((void**)&H.mw, ps->maxQ * 4) == hipSuccess && hipHostMalloc((void**)&H.prior, listBytes) == hipSuccess &&
             hipHostMalloc((void**)&H.out, listBytes) == hipSuccess;
        H.outDone.assign(nd, nullptr);
        for (uint32_t d = 0; d < nd && ok; ++d)
            ok = hipSetDevice(ps->devices[d]) == hipSuccess && hipEventCreateWithFlags(&H.outDone[d], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { mc_partset_close(ps); return ps_fail(nullptr, MC_ERR_NOMEM, "mc_partset_open: cannot allocate the batch buffers"); }
    std::string err;
    const uint64_t tl = now_ns();
    rc = open_group(ps, 0, ps->cur, err);
    ps->loadNs += now_ns() - tl;
    if (rc) { mc_partset_close(ps); return ps_fail(nullptr, rc, err); }
    ps->curFirst = 0;
    *out = ps;
    return MC_OK;
}

void mc_partset_close(mc_partset* ps)
{
    if (!ps) return;
    if (ps->loader.joinable()) ps->loader.join();
    close_group(ps->cur); close_group(ps->next);
    for (DevState& D : ps->dev) {
        (void)hipSetDevice(D.device);
        if (D.up) (void)hipStreamSynchronize(D.up);
        for (DevLane& L : D.lane) {
            if (L.stream) (void)hipStreamSynchronize(L.stream);
            void* bufs[] = {L.dmine, L.dall, L.dout};
            for (void* b : bufs) if (b) (void)hipFree(b);
        }
        for (DevInput& I : D.in) {
            void* bufs[] = {I.dseq, I.dqinfo, I.dmaxwin, I.dprior};
            for (void* b : bufs) if (b) (void)hipFree(b);
            if (I.upDone) (void)hipEventDestroy(I.upDone);
            if (I.inFree) (void)hipEventDestroy(I.inFree);
        }
        if (D.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(D.comm);
        for (DevLane& L : D.lane) if (L.stream) (void)hipStreamDestroy(L.stream);
        if (D.up) (void)hipStreamDestroy(D.up);
    }
    for (auto& H : ps->stg) {
        void* bufs[] = {H.seq, H.q, H.mw, H.prior, H.out};
        for (void* b : bufs) if (b) (void)hipHostFree(b);
        for (hipEvent_t e : H.outDone) if (e) (void)hipEventDestroy(e);
    }
    delete ps;
    mcamd::big_cache_hold(-1);
}

int mc_partset_info(const mc_partset* ps, uint64_t info[6])
{
    if (!ps || !info) return MC_ERR_INVALID;
    info[0] = ps->nparts; info[1] = ps->resident; info[2] = (ps->nparts + ps->resident - 1) / ps->resident; info[3] = ps->devices.size();
    info[4] = ps->loadNs; info[5] = ps->waitNs;
    return MC_OK;
}

int mc_partset_load_bytes(const mc_partset* ps, uint64_t* bytes)
{
    if (!ps || !bytes) return MC_ERR_INVALID;
    *bytes = ps->loadBytes.load();
    return MC_OK;
}

// Makes part group g (parts g * resident ...) the resident one: waits for the loader if it is loading exactly that group, else opens it;
// then starts loading group g + 1 behind the caller's queries.  Going through the groups in order 0, 1, ... is what overlaps every load
// with the previous group's queries.
int mc_partset_select_group(mc_partset* ps, uint32_t g)
{
    if (!ps) return MC_ERR_INVALID;
    const uint32_t first = g * ps->resident;
    if (first >= ps->nparts) return ps_fail(ps, MC_ERR_INVALID, "mc_partset_select_group: no such part group");
    if (ps->curFirst != first || ps->cur.empty()) {
        const uint64_t t0 = now_ns();
        if (ps->loader.joinable()) ps->loader.join();
        ps->waitNs += now_ns() - t0;
        if (ps->nextFirst == first && !ps->next.empty() && ps->loaderRc == MC_OK) {
            close_group(ps->cur);
            ps->cur.swap(ps->next);
        } else {
            // (a group that could not be loaded BESIDE the resident one -- the two did not fit the devices together -- is loaded in its place:
            // the caller is through with the resident group when it selects the next)
            const bool retry = ps->nextFirst == first && (ps->loaderRc == MC_ERR_NOMEM || (ps->loaderRc == MC_ERR_HIP && ps->loaderErr.find("out of memory") != std::string::npos));
            if (ps->nextFirst == first && ps->loaderRc != MC_OK && !retry) { close_group(ps->next); return ps_fail(ps, ps->loaderRc, ps->loaderErr); }
            close_group(ps->next);
            close_group(ps->cur);
            if (retry) { for (DevState& D : ps->dev) { (void)hipSetDevice(D.device); (void)hipDeviceSynchronize(); } mcamd::big_cache_trim(); }
            std::string err;
            const uint64_t t1 = now_ns();
            const int rc = open_group(ps, first, ps->cur, err);
            ps->loadNs += now_ns() - t1; ps->waitNs += now_ns() - t1;
            if (rc) return ps_fail(ps, rc, err);
        }
        ps->curFirst = first;
    }
    // the next group loads behind this group's queries (a loader that is already at it is left alone)
    const uint32_t nf = first + ps->resident;
    if (nf < ps->nparts && !(ps->loader.joinable() && ps->nextFirst == nf) && !(ps->nextFirst == nf && !ps->next.empty())) start_loader(ps, nf);
    return MC_OK;
}

}  // extern "C"

// ---- a batch's way through the set ----------------------------------------------------------------------------------------------------
namespace {

struct BatchRef { uint64_t first = 0, count = 0, chars = 0; };
struct CallArgs { const char* seqs; const uint64_t* offs; const char* seqs2; const uint64_t* offs2; int lowestRank; uint64_t insertMax; bool hasPrior; mc_candidate* out; };

int take_staging(mc_partset* ps)
{
    std::unique_lock<std::mutex> l(ps->poolMu);
    int got = -1;
    ps->poolCv.wait(l, [&] { for (uint32_t i = 0; i < kStaging; ++i) if (!ps->stg[i].busy) { got = (int)i; return true; } return false; });
    ps->stg[got].busy = true;
    return got;
}
void give_staging(mc_partset* ps, int i) { { std::lock_guard<std::mutex> l(ps->poolMu); ps->stg[i].busy = false; } ps->poolCv.notify_all(); }
// block = false: -1 when both lanes are taken (a caller that holds one finishes its own batch first: no two callers ever wait for each other)
int take_lane(mc_partset* ps, bool block)
{
    std::unique_lock<std::mutex> l(ps->poolMu);
    for (;;) {
        for (uint32_t i = 0; i < kLanes; ++i) if (!ps->laneBusy[i]) { ps->laneBusy[i] = true; return (int)i; }
        if (!block) return -1;
        ps->poolCv.wait(l);
    }
}
int take_input(mc_partset* ps, bool block)
{
    std::unique_lock<std::mutex> l(ps->poolMu);
    for (;;) {
        for (uint32_t i = 0; i < kInputs; ++i) if (!ps->inputBusy[i]) { ps->inputBusy[i] = true; return (int)i; }
        if (!block) return -1;
        ps->poolCv.wait(l);
    }
}
void give_input(mc_partset* ps, int i) { { std::lock_guard<std::mutex> l(ps->poolMu); ps->inputBusy[i] = false; } ps->poolCv.notify_all(); }
void give_lane(mc_partset* ps, int i) { { std::lock_guard<std::mutex> l(ps->poolMu); ps->laneBusy[i] = false; } ps->poolCv.notify_all(); }

template <class F>
void on_threads(uint32_t nt, F&& f)                                 // f(t) for t in [0, nt): t = 0 on the caller's thread
{
    if (nt <= 1) { f(0u); return; }
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nt; ++t) th.emplace_back(f, t);
    f(0u);
    for (auto& t : th) t.join();
}

// the batch into a staging slot: where every read goes (a sequence starts 4-byte aligned, mc_batch_add), then the characters by a few
// threads (one thread packs 60 M reads of 150 bp a second: less than the devices take)
void pack_batch(mc_partset* ps, const CallArgs& A, const BatchRef& B, mc_partset::Staging& H)
{
    const uint32_t m = (uint32_t)B.count, K = ps->K;
    const uint32_t nt = m >= (1u << 15) ? ps->packThreads : 1;
    // every thread a contiguous share of the reads: what its share takes, then (the shares' offsets known) places, window ranges, characters
    std::vector<uint64_t> base(nt + 1, 0);
    auto padded = [&](uint64_t i) { return (A.offs[i + 1] - A.offs[i] + 3) / 4 * 4 + (A.seqs2 ? (A.offs2[i + 1] - A.offs2[i] + 3) / 4 * 4 : 0); };
    if (nt > 1)
        on_threads(nt, [&](uint32_t t) {
            uint64_t sum = 0;
            for (uint64_t i = B.first + (uint64_t)m * t / nt, e = B.first + (uint64_t)m * (t + 1) / nt; i < e; ++i) sum += padded(i);
            base[t + 1] = sum;
        });
    for (uint32_t t = 0; t < nt; ++t) base[t + 1] += base[t];
    on_threads(nt, [&](uint32_t t) {
        uint64_t at = base[t];
        for (uint32_t j = (uint32_t)((uint64_t)m * t / nt), j1 = (uint32_t)((uint64_t)m * (t + 1) / nt); j < j1; ++j) {
            const uint64_t i = B.first + j, l1 = A.offs[i + 1] - A.offs[i], l2 = A.seqs2 ? A.offs2[i + 1] - A.offs2[i] : 0;
            H.q[4 * j] = (uint32_t)at; H.q[4 * j + 1] = (uint32_t)l1;
            if (l1) std::memcpy(H.seq + at, A.seqs + A.offs[i], l1);
            at += (l1 + 3) / 4 * 4;
            H.q[4 * j + 2] = (uint32_t)at; H.q[4 * j + 3] = (uint32_t)l2;
            if (l2) std::memcpy(H.seq + at, A.seqs2 + A.offs2[i], l2);
            at += (l2 + 3) / 4 * 4;
            H.mw[j] = (uint32_t)(2 + std::max<uint64_t>(l1 + l2, A.insertMax) / ps->stride);   // candidate_structs.hpp:143-145
        }
        if (t + 1 == nt) H.chars = at;
        if (A.hasPrior) {                                           // the earlier groups' lists of the share
            const uint32_t j0 = (uint32_t)((uint64_t)m * t / nt), j1 = (uint32_t)((uint64_t)m * (t + 1) / nt);
            std::memcpy(H.prior + (size_t)j0 * K, A.out + (B.first + j0) * K, (size_t)(j1 - j0) * K * sizeof(mc_candidate));
        }
    });
    H.count = m; H.hasPrior = A.hasPrior;
}

// rows [lo, hi) of a batch of m reads belong to owner o of `owners`
inline uint32_t share_lo(uint32_t m, uint32_t o, uint32_t owners) { return (uint32_t)((uint64_t)m * o / owners); }

// The batch in staging slot H to device input k of every device that holds a part of the resident group (they are the owners of the
// reads, too), on the devices' upload streams; an input's last batch must have read it to its end (a device-side wait).
int upload_batch(mc_partset* ps, mc_partset::Staging& H, int k)
{
    const uint32_t nd = (uint32_t)ps->devices.size(), np = (uint32_t)ps->cur.size(), K = ps->K, m = H.count;
    const uint32_t owners = std::min(nd, np);
    H.owners = owners;
    for (uint32_t d = 0; d < owners; ++d) {
        DevState& D = ps->dev[d];
        DevInput& I = D.in[k];
        if (hipSetDevice(D.device) != hipSuccess) return ps_fail(ps, MC_ERR_HIP, "hipSetDevice");
        if (I.used && hipStreamWaitEvent(D.up, I.inFree, 0) != hipSuccess) return ps_fail(ps, MC_ERR_HIP, "hipStreamWaitEvent");
        I.used = true;
        const uint32_t lo = share_lo(m, d, owners), hi = share_lo(m, d + 1, owners);
        if (hipMemcpyAsync(I.dseq, H.seq, H.chars + 16, hipMemcpyHostToDevice, D.up) != hipSuccess ||
            hipMemcpyAsync(I.dqinfo, H.q, (size_t)m * 16, hipMemcpyHostToDevice, D.up) != hipSuccess ||
            hipMemcpyAsync(I.dmaxwin, H.mw, (size_t)m * 4, hipMemcpyHostToDevice, D.up) != hipSuccess ||
            (H.hasPrior && hi > lo &&
             hipMemcpyAsync(I.dprior, H.prior + (size_t)lo * K, (size_t)(hi - lo) * K * sizeof(mc_candidate), hipMemcpyHostToDevice, D.up) != hipSuccess) ||
            hipEventRecord(I.upDone, D.up) != hipSuccess)
            return ps_fail(ps, MC_ERR_HIP, "copy of a batch to the device failed");
    }
    return MC_OK;
}

// Main kernels of the batch (input k) on lane ln: every owner device runs its parts on it, all on the lane's stream of that device --
// nothing here waits for a device (MC_DEFER_TAIL).
int submit_batch(mc_partset* ps, mc_partset::Staging& H, int ln, int k, int lowestRank)
{
    const uint32_t nd = (uint32_t)ps->devices.size(), np = (uint32_t)ps->cur.size(), m = H.count, owners = H.owners;
    std::vector<int> rcs(owners, MC_OK);
    std::vector<std::string> errs(owners);
    on_threads(owners, [&](uint32_t d) {
        DevState& D = ps->dev[d];
        DevLane& L = D.lane[ln];
        DevInput& I = D.in[k];
        if (hipSetDevice(D.device) != hipSuccess || hipStreamWaitEvent(L.stream, I.upDone, 0) != hipSuccess) { rcs[d] = MC_ERR_HIP; errs[d] = "hipStreamWaitEvent"; return; }
        L.cands.clear();
        for (uint32_t p = d; p < np; p += nd) {
            mc_device_batch in{I.dseq, I.dqinfo, I.dmaxwin, 0, m, H.chars};
            mc_device_results res{};
            const int rc = mc_query_device(ps->cur[p], &in, lowestRank, MC_DEFER_TAIL | (ln ? MC_SECOND_PIPE : 0), &res, L.stream);
            if (rc) { rcs[d] = rc; errs[d] = mc_last_error(ps->cur[p]); return; }
            L.cands.push_back(res.cands);
        }
    });
    for (uint32_t d = 0; d < owners; ++d) if (rcs[d]) return ps_fail(ps, rcs[d], errs[d]);
    return MC_OK;
}

// The rest of the batch on lane ln, enqueued once its main kernels are through (mc_query_finish waits for them; the NEXT batch is
// queued on the other lane by then): tails of the parts' queries, their top lists side by side in dmine, the exchange, the owners'
// merges, every owner's share to the staging slot.
int finish_batch(mc_partset* ps, mc_partset::Staging& H, int ln, int k, int lowestRank)
{
    const uint32_t nd = (uint32_t)ps->devices.size(), np = (uint32_t)ps->cur.size(), K = ps->K, m = H.count, owners = H.owners;
    const size_t listBytes = ps->maxQ * K * sizeof(mc_candidate), row = K * sizeof(mc_candidate);
    std::vector<int> rcs(owners, MC_OK);
    std::vector<std::string> errs(owners);
    on_threads(owners, [&](uint32_t d) {
        DevState& D = ps->dev[d];
        DevLane& L = D.lane[ln];
        if (hipSetDevice(D.device) != hipSuccess) { rcs[d] = MC_ERR_HIP; errs[d] = "hipSetDevice"; return; }
        uint32_t slot = 0;
        for (uint32_t p = d; p < np; p += nd, ++slot) {
            mc_ctx* c = ps->cur[p];
            int rc = mc_query_finish(c, ln ? MC_SECOND_PIPE : 0);
            if (!rc) rc = mc_copy_results_on(c, reinterpret_cast<char*>(L.dmine) + slot * listBytes, L.cands[slot], (uint64_t)m * row, 0, L.stream);
            if (rc) { rcs[d] = rc; errs[d] = mc_last_error(c); return; }
        }
    });
    for (uint32_t d = 0; d < owners; ++d) if (rcs[d]) return ps_fail(ps, rcs[d], errs[d]);
    // every device's lists of owner o's reads -> o: dall[source device][slot], rows 0 .. o's share
    if (ps->rccl) {
        std::lock_guard<std::mutex> order(ps->orderMu);
        g_rccl.GroupStart();
        int r = 0;
        for (uint32_t d = 0; d < owners && !r; ++d) {
            DevState& D = ps->dev[d];
            DevLane& L = D.lane[ln];
            (void)hipSetDevice(D.device);
            const uint32_t myLo = share_lo(m, d, owners), myN = share_lo(m, d + 1, owners) - myLo;
            for (uint32_t o = 0; o < owners && !r; ++o) {
                const uint32_t lo = share_lo(m, o, owners), cnt = share_lo(m, o + 1, owners) - lo;
                uint32_t slot = 0;
                for (uint32_t pp = d; pp < np && !r; pp += nd, ++slot)          // my parts' lists of o's reads
                    if (cnt) r = g_rccl.Send(reinterpret_cast<const char*>(L.dmine) + slot * listBytes + (size_t)lo * row, (size_t)cnt * row, Rccl::kChar, (int)o, D.comm, L.stream);
                slot = 0;
                for (uint32_t pp = o; pp < np && !r; pp += nd, ++slot)          // o's parts' lists of my reads
                    if (myN) r = g_rccl.Recv(reinterpret_cast<char*>(L.dall) + ((size_t)o * ps->slotsPerDev + slot) * listBytes, (size_t)myN * row, Rccl::kChar, (int)o, D.comm, L.stream);
            }
        }
        const int e2 = g_rccl.GroupEnd();
        if (r || e2) return ps_fail(ps, MC_ERR_HIP, std::string("ncclSend / ncclRecv of the per-part candidates: ") + g_rccl.text(r ? r : e2));
    }
    // every owner: the earlier groups' list of its reads first, then this group's parts in part order (part p: device p % nd, slot p / nd)
    for (uint32_t d = 0; d < owners; ++d) {
        DevState& D = ps->dev[d];
        DevLane& L = D.lane[ln];
        const uint32_t lo = share_lo(m, d, owners), cnt = share_lo(m, d + 1, owners) - lo;
        if (hipSetDevice(D.device) != hipSuccess) return ps_fail(ps, MC_ERR_HIP, "hipSetDevice");
        if (cnt) {
            std::vector<const mc_candidate*> lists;
            if (H.hasPrior) lists.push_back(D.in[k].dprior);
            for (uint32_t p = 0; p < np; ++p)
                lists.push_back(ps->rccl ? reinterpret_cast<const mc_candidate*>(reinterpret_cast<const char*>(L.dall) + ((size_t)(p % nd) * ps->slotsPerDev + p / nd) * listBytes)
                                         : reinterpret_cast<const mc_candidate*>(reinterpret_cast<const char*>(L.dmine) + (size_t)(p / nd) * listBytes + (size_t)lo * row));
            const int rc = mc_merge_part_candidates(ps->cur[d], lists.data(), (uint32_t)lists.size(), cnt, lowestRank, L.dout, L.stream);
            if (rc) return ps_fail(ps, rc, mc_last_error(ps->cur[d]));
            if (hipMemcpyAsync(H.out + (size_t)lo * K, L.dout, (size_t)cnt * row, hipMemcpyDeviceToHost, L.stream) != hipSuccess)
                return ps_fail(ps, MC_ERR_HIP, "copy of the merged candidates failed");
        }
        if (hipEventRecord(H.outDone[d], L.stream) != hipSuccess || hipEventRecord(D.in[k].inFree, L.stream) != hipSuccess) return ps_fail(ps, MC_ERR_HIP, "hipEventRecord");
    }
    return MC_OK;
}

// the merged lists of the batch in slot H -> the caller's array, once every owner has delivered its share
int collect_batch(mc_partset* ps, mc_partset::Staging& H, const CallArgs& A, const BatchRef& B)
{
    for (uint32_t d = 0; d < H.owners; ++d)
        if (hipEventSynchronize(H.outDone[d]) != hipSuccess) return ps_fail(ps, MC_ERR_HIP, "copy of the merged candidates failed");
    const size_t bytes = (size_t)B.count * ps->K * sizeof(mc_candidate);
    char* dst = reinterpret_cast<char*>(A.out + B.first * ps->K);
    const char* src = reinterpret_cast<const char*>(H.out);
    const uint32_t nt = bytes >= (4u << 20) ? std::min<uint32_t>(ps->packThreads, 4) : 1;
    on_threads(nt, [&](uint32_t t) { const size_t a = bytes * t / nt, b = bytes * (t + 1) / nt; std::memcpy(dst + a, src + a, b - a); });
    return MC_OK;
}

void idle_devices(mc_partset* ps)
{
    for (DevState& D : ps->dev) { (void)hipSetDevice(D.device); (void)hipStreamSynchronize(D.up); for (DevLane& L : D.lane) (void)hipStreamSynchronize(L.stream); }
}

}  // namespace

extern "C" {

// n reads (pairs: mate i = seqs2 + offs2[i] .. offs2[i + 1]; seqs2 == NULL: single reads) against the parts of the RESIDENT group only.
// inout [n][max_candidates] (host): with hasPrior the list the earlier groups left for these reads -- it leads the merge, as if its parts had
// been queried before this group's (candidate_generation.hpp:172-231) --, on return the merged list.  Callers stream their batches through
// group after group and keep 16 bytes x max_candidates per read between groups (mcq -resident-parts; mc_partset_classify below).
// Thread-safe: several callers (a batch each) share the set's two lanes; ONE caller with several batches keeps two of them in flight itself.
int mc_partset_classify_resident(mc_partset* ps, const char* seqs, const uint64_t* offs, const char* seqs2, const uint64_t* offs2, uint64_t n, int lowestRank,
                                 uint64_t insertMax, int hasPrior, mc_candidate* out)
{
    if (!ps || !seqs || !offs || !out || (seqs2 && !offs2)) return MC_ERR_INVALID;
    if (ps->cur.empty()) return ps_fail(ps, MC_ERR_STATE, "mc_partset_classify_resident: no part group is resident (mc_partset_select_group)");
    const CallArgs A{seqs, offs, seqs2, offs2, lowestRank, insertMax, hasPrior != 0, out};
    // the batches: as many reads as fit the slot limits (a sequence starts 4-byte aligned, mc_batch_add)
    std::vector<BatchRef> batches;
    auto need = [&](uint64_t i) {
        const uint64_t l1 = offs[i + 1] - offs[i], l2 = seqs2 ? offs2[i + 1] - offs2[i] : 0;
        return (l1 + 3) / 4 * 4 + (l2 + 3) / 4 * 4;
    };
    // (a call of several batches starts with a quarter and a half batch: the devices begin after a quarter of a batch's packing, and the
    // first full batch is packed under work that is already there)
    const bool ramp = n >= 3 * ps->maxQ;
    for (uint64_t i = 0; i < n;) {
        BatchRef b{i, 0, 0};
        const uint64_t cap = ramp && batches.size() < 2 ? std::max<uint64_t>(ps->maxQ >> (2 - batches.size()), 1) : ps->maxQ;
        while (i < n && b.count < cap && b.chars + need(i) <= ps->maxChars) { b.chars += need(i); ++b.count; ++i; }
        if (b.count == 0) return ps_fail(ps, MC_ERR_INVALID, "mc_partset_classify: a read is longer than slot_max_chars");
        batches.push_back(b);
    }
    const size_t nbt = batches.size();
    if (nbt == 0) return MC_OK;
    static const bool trace = std::getenv("MC_PARTSET_TRACE") != nullptr;
    uint64_t tPackWait = 0, tSubmit = 0, tFinish = 0, tCollect = 0;
    // the packer runs ahead of the devices: batch b + 1 (b + 2) is in its staging slot when batch b is submitted
    std::vector<int> slotOf(nbt, -1);
    std::mutex qMu;
    std::condition_variable qCv;
    size_t packed = 0;                                             // batches [0, packed) are in their slots
    bool stop = false;
    std::thread packer;
    if (nbt > 1)
        packer = std::thread([&] {
            for (size_t b = 0; b < nbt; ++b) {
                { std::lock_guard<std::mutex> l(qMu); if (stop) return; }
                const int s = take_staging(ps);
                pack_batch(ps, A, batches[b], ps->stg[s]);
                { std::lock_guard<std::mutex> l(qMu); slotOf[b] = s; packed = b + 1; }
                qCv.notify_all();
            }
        });
    else { slotOf[0] = take_staging(ps); pack_batch(ps, A, batches[0], ps->stg[slotOf[0]]); packed = 1; }
    // collecting (event wait + copy into the caller's array) by a helper of its own: the submitting thread only enqueues
    std::vector<int> rcOf(nbt, MC_OK);
    size_t finished = 0, collected = 0;                            // batches [0, finished) have their copy back enqueued
    std::thread collector;
    if (nbt > 1)
        collector = std::thread([&] {
            for (size_t b = 0; b < nbt; ++b) {
                { std::unique_lock<std::mutex> l(qMu); qCv.wait(l, [&] { return stop || finished > b; }); if (finished <= b) return; }
                rcOf[b] = collect_batch(ps, ps->stg[slotOf[b]], A, batches[b]);
                give_staging(ps, slotOf[b]);
                { std::lock_guard<std::mutex> l(qMu); collected = b + 1; }
                qCv.notify_all();
            }
        });
    int rc = MC_OK;
    int laneOf[2] = {-1, -1}, inputOf[2] = {-1, -1};               // lane and device input of the batches in flight: [b & 1]
    auto finish = [&](size_t b) -> int {                          // tail, exchange, merge, copy back of batch b; its lane and input go back
        const uint64_t t0 = now_ns();
        const int r = finish_batch(ps, ps->stg[slotOf[b]], laneOf[b & 1], inputOf[b & 1], lowestRank);
        give_lane(ps, laneOf[b & 1]); laneOf[b & 1] = -1;
        give_input(ps, inputOf[b & 1]); inputOf[b & 1] = -1;      // (its next batch's upload waits for this one's reads on the device: inFree)
        tFinish += now_ns() - t0;
        if (!r) { { std::lock_guard<std::mutex> l(qMu); finished = b + 1; } qCv.notify_all(); }
        return r;
    };
    size_t inFlightFrom = 0;                                       // batches [inFlightFrom, b) are submitted and not finished
    // The upload of batch b + 1 is ENQUEUED right behind the submission of batch b, before this thread goes into finish(b - 1): that call
    // waits for batch b - 1's tail, whose few small kernels get their CUs only when batch b's persistent kernels end -- an upload enqueued
    // after it started when the device had nothing left to run beside it, and a batch took upload + kernels (the stream's timeline at
    // 15 Gbp: 1.06 + 1.1 ms per 250 000 reads; lab notebook r06 section 2c).  preInput: the device input that upload went to.
    int preInput = -1;
    size_t preFor = (size_t)-1;
    for (size_t b = 0; b < nbt && !rc; ++b) {
        const uint64_t t0 = now_ns();
        { std::unique_lock<std::mutex> l(qMu); qCv.wait(l, [&] { return packed > b; }); }
        const uint64_t t1 = now_ns();
        // nothing is waited for while this caller holds a lane: its own oldest batch gives a lane and an input back, else another caller will
        int k = -1;
        if (preFor == b) { k = preInput; preInput = -1; preFor = (size_t)-1; }
        else {
            k = take_input(ps, false);
            while (k < 0 && !rc) {
                if (inFlightFrom < b) { rc = finish(inFlightFrom++); if (!rc) k = take_input(ps, false); }
                else k = take_input(ps, true);
            }
            if (rc) break;
            rc = upload_batch(ps, ps->stg[slotOf[b]], k);          // (enqueued before anything below waits: it runs under the batch before)
        }
        int ln = rc ? -1 : take_lane(ps, false);
        while (ln < 0 && !rc) {
            if (inFlightFrom < b) { rc = finish(inFlightFrom++); if (!rc) ln = take_lane(ps, false); }
            else ln = take_lane(ps, true);
        }
        if (rc) { give_input(ps, k); if (ln >= 0) give_lane(ps, ln); break; }
        laneOf[b & 1] = ln; inputOf[b & 1] = k;
        rc = submit_batch(ps, ps->stg[slotOf[b]], ln, k, lowestRank);
        tPackWait += t1 - t0; tSubmit += now_ns() - t1;
        if (rc) { give_lane(ps, ln); laneOf[b & 1] = -1; give_input(ps, k); inputOf[b & 1] = -1; break; }
        if (b + 1 < nbt) {                                          // the next batch's upload, if it is packed and an input is free -- no waiting here
            bool ready;
            { std::lock_guard<std::mutex> l(qMu); ready = packed > b + 1; }
            const int k2 = ready ? take_input(ps, false) : -1;
            if (k2 >= 0) {
                rc = upload_batch(ps, ps->stg[slotOf[b + 1]], k2);
                if (rc) { give_input(ps, k2); break; }
                preInput = k2; preFor = b + 1;
            }
        }
        if (inFlightFrom < b) rc = finish(inFlightFrom++);          // the batch before this one: its kernels ran while this one was enqueued
    }
    if (preInput >= 0) { idle_devices(ps); give_input(ps, preInput); preInput = -1; }   // (an error after an upload ahead: the copy is through before its input goes back)
    const size_t submitted = rc ? inFlightFrom : nbt;
    while (!rc && inFlightFrom < submitted) rc = finish(inFlightFrom++);
    if (nbt > 1) {
        const uint64_t t0 = now_ns();
        { std::lock_guard<std::mutex> l(qMu); stop = true; }
        qCv.notify_all();
        if (rc) {                                                   // an error leaves nothing in flight and no slot taken
            idle_devices(ps);
            for (int& l : laneOf) if (l >= 0) { give_lane(ps, l); l = -1; }
            for (int& k : inputOf) if (k >= 0) { give_input(ps, k); k = -1; }
        }
        collector.join();
        // (an error: the packer may be waiting for a slot -- the ones nobody will collect go back first; it packs one batch more at most)
        std::vector<char> given(nbt, 0);
        auto give_uncollected = [&] {
            size_t upTo;
            { std::lock_guard<std::mutex> l(qMu); upTo = packed; }
            for (size_t b = collected; b < upTo; ++b) if (!given[b] && slotOf[b] >= 0) { given[b] = 1; give_staging(ps, slotOf[b]); }
        };
        give_uncollected();
        packer.join();
        give_uncollected();
        for (size_t b = 0; b < nbt; ++b) if (!rc && rcOf[b]) rc = rcOf[b];
        if (!rc && collected < nbt) rc = ps_fail(ps, MC_ERR_STATE, "mc_partset_classify_resident: internal: a batch was not collected");
        tCollect += now_ns() - t0;
    } else {
        const uint64_t t0 = now_ns();
        if (!rc) rc = collect_batch(ps, ps->stg[slotOf[0]], A, batches[0]);
        else { idle_devices(ps); for (int& l : laneOf) if (l >= 0) { give_lane(ps, l); l = -1; } for (int& k : inputOf) if (k >= 0) { give_input(ps, k); k = -1; } }
        give_staging(ps, slotOf[0]);
        tCollect += now_ns() - t0;
    }
    if (trace)
        std::fprintf(stderr, "mc_partset_classify_resident: %zu batches, %zu parts; host ms: waiting for the packer %.2f, submit %.2f, finish %.2f, waiting for the last copies %.2f\n",
                     nbt, ps->cur.size(), tPackWait / 1e6, tSubmit / 1e6, tFinish / 1e6, tCollect / 1e6);
    return rc;
}

// All n reads against every part of the database, part group by part group.  out: [n][max_candidates] in host memory.
int mc_partset_classify(mc_partset* ps, const char* seqs, const uint64_t* offs, const char* seqs2, const uint64_t* offs2, uint64_t n, int lowestRank,
                        uint64_t insertMax, mc_candidate* out)
{
    if (!ps || !seqs || !offs || !out || (seqs2 && !offs2)) return MC_ERR_INVALID;
    const uint32_t groups = (ps->nparts + ps->resident - 1) / ps->resident;
    for (uint32_t g = 0; g < groups; ++g) {
        int rc = mc_partset_select_group(ps, g);
        if (!rc) rc = mc_partset_classify_resident(ps, seqs, offs, seqs2, offs2, n, lowestRank, insertMax, g > 0, out);
        if (rc) return rc;
    }
    return MC_OK;
}

}  // extern "C"
