// metacache_amd/csrc/partset.cpp -- a partitioned database queried PART GROUP BY PART GROUP, parts spread over the GPUs of the node
// (C ABI: mc_partset_*, include/metacache_amd.h).
//
// What it replaces: (1) the reference's parts-over-GPUs query inside one process (gpu_hashmap.cu:1255-1290 query_hashtables_async over
// the parts' tables, query_batch.cu:464-527 forwarding the sketches GPU -> GPU), (2) its documented workflow for databases that do not
// fit the node (docs/partitioning.md:116-153: `query` one part at a time, `merge` the result files, mode_merge.cpp:247-296).
// Here: `resident` parts are in HBM at a time, one context each (mc_open_database with single_part), dealt out round-robin over the
// devices; while the reads run against group g a loader thread opens the parts of group g + 1 (their H2D copies and insert kernels
// run on those contexts' own streams: two table slots, copy behind compute).  Per batch every device queries its parts, the
// per-part top lists are gathered on device 0 with ncclAllGather (RCCL, one communicator rank per device; over xGMI between GPUs)
// and merged IN PART ORDER by merge_parts_kernel -- the CPU's list insert, candidate_generation.hpp:172-231 -- together with the
// list the earlier groups left for the read.  The result is what one candidate list fed by all parts in order holds: the intended
// semantics of host_hashmap.hpp:695-723 (the in-process reference itself is history dependent for more than one part).
//
// RCCL is loaded at run time (rccl_dl.h).
#include "context.h"
#include "devcache.h"
#include "rccl_dl.h"

#include <algorithm>
#include <atomic>
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

struct DevState {                      // per device: the batch's input, the parts' candidate lists, the gathered lists
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t* dseq = nullptr; uint32_t* dqinfo = nullptr; uint32_t* dmaxwin = nullptr;
    mc_candidate* dmine = nullptr;     // [slotsPerDev][maxQ][K]: this device's parts of the resident group
    mc_candidate* dall = nullptr;      // [ndev][slotsPerDev][maxQ][K]: everybody's (device 0 merges)
    mc_candidate* dprior = nullptr, *dout = nullptr;   // device 0: the earlier groups' list of the batch's reads, the merged one
    void* comm = nullptr;
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
    // host staging of a batch, twice, in pinned memory: batch b + 1 is packed while the devices run batch b, the copies in both
    // directions are asynchronous (pageable memory would make every hipMemcpyAsync a synchronous staged copy)
    struct HostSlot {
        uint8_t* seq = nullptr; uint32_t* q = nullptr; uint32_t* mw = nullptr;
        mc_candidate* prior = nullptr; mc_candidate* out = nullptr;
        std::vector<hipEvent_t> inDone;   // per device: the slot's input has left the host
        hipEvent_t outDone = nullptr;     // device 0: the merged lists are in `out`
        bool inFlight = false;
        uint64_t chars = 0;               // characters of the batch in `seq`
        uint64_t first = 0, count = 0;    // the batch whose result `out` will hold
    } hs[2];
    uint64_t loadNs = 0, waitNs = 0;   // time the loader spent / the queries waited for it
    std::atomic<uint64_t> loadBytes{0}; // bytes of .cache files read by the group loads
    uint32_t packThreads = 8;          // host threads that copy a batch's characters into the staging buffer (MC_PARTSET_PACK_THREADS)
    bool ranges = false;               // the parts are target ranges of one file (cfg.target_shard_count > 1)
    bool rccl = false;                 // several devices (or MC_PARTSET_RCCL=1: the same calls with a single rank, tests): gather over RCCL
};

namespace {

int ps_fail(mc_partset* ps, int code, const std::string& msg) { if (ps) ps->err = msg; else set_global_error(msg); return code; }

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
            c.num_slots = 1; c.copy_allhits = 0;
            // list alignment (up to 1.5 x the plain store) is a table's own decision against the device's free memory: with several tenants
            // per device it would be made against memory the group's later parts need.  Groups that stream (the next one loads beside the
            // resident one): plain stores; all parts resident: the padding may take its share of what is free.
            const uint32_t mine = (count - d + nd - 1) / nd, placed = i / nd;
            mcamd::open_hints().listAlign = ps->resident < ps->nparts ? 0 : -1;
            mcamd::open_hints().listAlignShare = 1.0 / (double)std::max<uint32_t>(1, mine - std::min(placed, mine - 1));
            const int rc = mc_open_database(ps->db.c_str(), &c, &out[i]);
            mcamd::open_hints() = mcamd::OpenHints{};
            if (rc != MC_OK) { std::lock_guard<std::mutex> l(errMu); rcs[d] = rc; errs[d] = mc_last_error(nullptr); return; }
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
        ok = hipSetDevice(D.device) == hipSuccess && hipStreamCreateWithFlags(&D.stream, hipStreamNonBlocking) == hipSuccess &&
             mcamd::dev_malloc((void**)&D.dseq, ps->maxChars + 64) == hipSuccess && mcamd::dev_malloc((void**)&D.dqinfo, ps->maxQ * 16) == hipSuccess &&
             mcamd::dev_malloc((void**)&D.dmaxwin, ps->maxQ * 4) == hipSuccess && mcamd::dev_malloc((void**)&D.dmine, ps->slotsPerDev * listBytes) == hipSuccess &&
             mcamd::dev_malloc((void**)&D.dall, (size_t)nd * ps->slotsPerDev * listBytes) == hipSuccess;
        if (ok && d == 0) ok = mcamd::dev_malloc((void**)&D.dprior, listBytes) == hipSuccess && mcamd::dev_malloc((void**)&D.dout, listBytes) == hipSuccess;
    }
    for (auto& H : ps->hs) {
        if (!ok) break;
        ok = hipHostMalloc((void**)&H.seq, ps->maxChars + 64) == hipSuccess && hipHostMalloc((void**)&H.q, ps->maxQ * 16) == hipSuccess &&
             hipHostMalloc((void**)&H.mw, ps->maxQ * 4) == hipSuccess && hipHostMalloc((void**)&H.prior, listBytes) == hipSuccess &&
             hipHostMalloc((void**)&H.out, listBytes) == hipSuccess;
        H.inDone.assign(nd, nullptr);
        for (uint32_t d = 0; d < nd && ok; ++d)
            ok = hipSetDevice(ps->devices[d]) == hipSuccess && hipEventCreateWithFlags(&H.inDone[d], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipSetDevice(ps->devices[0]) == hipSuccess && hipEventCreateWithFlags(&H.outDone, hipEventDisableTiming) == hipSuccess;
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
        if (D.stream) (void)hipStreamSynchronize(D.stream);
        void* bufs[] = {D.dseq, D.dqinfo, D.dmaxwin, D.dmine, D.dall, D.dprior, D.dout};
        for (void* b : bufs) if (b) (void)hipFree(b);
        if (D.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(D.comm);
        if (D.stream) (void)hipStreamDestroy(D.stream);
    }
    for (auto& H : ps->hs) {
        void* bufs[] = {H.seq, H.q, H.mw, H.prior, H.out};
        for (void* b : bufs) if (b) (void)hipHostFree(b);
        for (hipEvent_t e : H.inDone) if (e) (void)hipEventDestroy(e);
        if (H.outDone) (void)hipEventDestroy(H.outDone);
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
            if (ps->nextFirst == first && ps->loaderRc != MC_OK) { close_group(ps->next); return ps_fail(ps, ps->loaderRc, ps->loaderErr); }
            close_group(ps->next);
            close_group(ps->cur);
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

// n reads (pairs: mate i = seqs2 + offs2[i] .. offs2[i + 1]; seqs2 == NULL: single reads) against the parts of the RESIDENT group only.
// inout [n][max_candidates] (host): with hasPrior the list the earlier groups left for these reads -- it leads the merge, as if its parts had
// been queried before this group's (candidate_generation.hpp:172-231) --, on return the merged list.  Callers stream their batches through
// group after group and keep 16 bytes x max_candidates per read between groups (mcq -resident-parts; mc_partset_classify below).
int mc_partset_classify_resident(mc_partset* ps, const char* seqs, const uint64_t* offs, const char* seqs2, const uint64_t* offs2, uint64_t n, int lowestRank,
                                 uint64_t insertMax, int hasPrior, mc_candidate* out)
{
    if (!ps || !seqs || !offs || !out || (seqs2 && !offs2)) return MC_ERR_INVALID;
    if (ps->cur.empty()) return ps_fail(ps, MC_ERR_STATE, "mc_partset_classify_resident: no part group is resident (mc_partset_select_group)");
    const uint32_t K = ps->K, nd = (uint32_t)ps->devices.size();
    const size_t listBytes = ps->maxQ * K * sizeof(mc_candidate);
    // the batches: as many reads as fit the slot limits (a sequence starts 4-byte aligned, mc_batch_add)
    struct Batch { uint64_t first, count, chars; };
    std::vector<Batch> batches;
    auto need = [&](uint64_t i) {
        const uint64_t l1 = offs[i + 1] - offs[i], l2 = seqs2 ? offs2[i + 1] - offs2[i] : 0;
        return (l1 + 3) / 4 * 4 + (l2 + 3) / 4 * 4;
    };
    for (uint64_t i = 0; i < n;) {
        Batch b{i, 0, 0};
        while (i < n && b.count < ps->maxQ && b.chars + need(i) <= ps->maxChars) { b.chars += need(i); ++b.count; ++i; }
        if (b.count == 0) return ps_fail(ps, MC_ERR_INVALID, "mc_partset_classify: a read is longer than slot_max_chars");
        batches.push_back(b);
    }
    const uint32_t np = (uint32_t)ps->cur.size();
    auto idle = [&](int code, const std::string& msg) {            // an error leaves nothing in flight
        for (uint32_t d = 0; d < nd; ++d) { (void)hipSetDevice(ps->dev[d].device); (void)hipStreamSynchronize(ps->dev[d].stream); }
        for (auto& H : ps->hs) H.inFlight = false;
        return ps_fail(ps, code, msg);
    };
    // the merged lists of a slot's batch -> the caller's array, once device 0 has written them
    auto collect = [&](mc_partset::HostSlot& H) -> bool {
        if (!H.inFlight) return true;
        H.inFlight = false;
        if (hipEventSynchronize(H.outDone) != hipSuccess) return false;
        std::memcpy(out + H.first * K, H.out, (size_t)H.count * K * sizeof(mc_candidate));
        return true;
    };
    static const bool trace = std::getenv("MC_PARTSET_TRACE") != nullptr;
    uint64_t tCollect = 0, tPack = 0, tEnqueue = 0, tGather = 0;
    // a batch into its slot's pinned buffers: where every read goes (a sequence starts 4-byte aligned), then the characters by a few
    // threads (one thread packs 60 M reads of 150 bp a second: less than the devices take)
    auto pack_batch = [&](size_t bi) {
        const Batch& B = batches[bi];
        mc_partset::HostSlot& H = ps->hs[bi & 1];
        const uint32_t m = (uint32_t)B.count;
        uint64_t at = 0;
        for (uint32_t j = 0; j < m; ++j) {
            const uint64_t i = B.first + j, l1 = offs[i + 1] - offs[i], l2 = seqs2 ? offs2[i + 1] - offs2[i] : 0;
            H.q[4 * j] = (uint32_t)at; H.q[4 * j + 1] = (uint32_t)l1;
            at += (l1 + 3) / 4 * 4;
            H.q[4 * j + 2] = (uint32_t)at; H.q[4 * j + 3] = (uint32_t)l2;
            at += (l2 + 3) / 4 * 4;
            H.mw[j] = (uint32_t)(2 + std::max<uint64_t>(l1 + l2, insertMax) / ps->stride);   // candidate_structs.hpp:143-145
        }
        H.chars = at;
        auto pack = [&](uint32_t j0, uint32_t j1) {
            for (uint32_t j = j0; j < j1; ++j) {
                const uint64_t i = B.first + j;
                if (H.q[4 * j + 1]) std::memcpy(H.seq + H.q[4 * j], seqs + offs[i], H.q[4 * j + 1]);
                if (H.q[4 * j + 3]) std::memcpy(H.seq + H.q[4 * j + 2], seqs2 + offs2[i], H.q[4 * j + 3]);
            }
        };
        const uint32_t nt = m >= (1u << 15) ? ps->packThreads : 1;
        if (nt <= 1) pack(0, m);
        else {
            std::vector<std::thread> th;
            for (uint32_t t = 1; t < nt; ++t) th.emplace_back(pack, (uint32_t)((uint64_t)m * t / nt), (uint32_t)((uint64_t)m * (t + 1) / nt));
            pack(0, (uint32_t)((uint64_t)m / nt));
            for (auto& t : th) t.join();
        }
        if (hasPrior) std::memcpy(H.prior, out + B.first * K, (size_t)m * K * sizeof(mc_candidate));
    };
    // mc_query_device comes back when a batch's kernels are nearly through (its lists' sizes make a host round trip): batch b + 1 is packed
    // by a helper thread meanwhile
    std::thread packer;
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{packer};
    if (!batches.empty()) pack_batch(0);
    for (size_t bi = 0; bi < batches.size(); ++bi) {
        const Batch& B = batches[bi];
        mc_partset::HostSlot& H = ps->hs[bi & 1];
        const uint64_t tr0 = now_ns();
        if (packer.joinable()) packer.join();                       // this batch is in its slot
        const uint64_t tr1 = now_ns();
        // the OTHER slot is packed next: its last batch (bi - 1) must have left the host on every device, and its result goes to the caller
        mc_partset::HostSlot& O = ps->hs[(bi + 1) & 1];
        if (O.inFlight)
            for (uint32_t d = 0; d < nd; ++d) if (hipEventSynchronize(O.inDone[d]) != hipSuccess) return idle(MC_ERR_HIP, "copy of a batch to the device failed");
        if (!collect(O)) return idle(MC_ERR_HIP, "copy of the merged candidates failed");
        const uint64_t tr2 = now_ns();
        if (bi + 1 < batches.size()) packer = std::thread(pack_batch, bi + 1);
        const uint32_t m = (uint32_t)B.count;
        const uint64_t at = H.chars;
        // every device: the batch, then its parts of the group, their top lists side by side in dmine.  Everything of a device is in
        // order on its one stream: batch b + 1's input overwrites the device buffers only after batch b's kernels have read them
        std::vector<int> rcs(nd, MC_OK);
        std::vector<std::string> errs(nd);
        auto run_device = [&](uint32_t d) {
            DevState& D = ps->dev[d];
            if (hipSetDevice(D.device) != hipSuccess) { rcs[d] = MC_ERR_HIP; errs[d] = "hipSetDevice"; return; }
            (void)hipMemcpyAsync(D.dseq, H.seq, at + 16, hipMemcpyHostToDevice, D.stream);
            (void)hipMemcpyAsync(D.dqinfo, H.q, (size_t)m * 16, hipMemcpyHostToDevice, D.stream);
            (void)hipMemcpyAsync(D.dmaxwin, H.mw, (size_t)m * 4, hipMemcpyHostToDevice, D.stream);
            if (d == 0 && hasPrior) (void)hipMemcpyAsync(D.dprior, H.prior, (size_t)m * K * sizeof(mc_candidate), hipMemcpyHostToDevice, D.stream);
            (void)hipEventRecord(H.inDone[d], D.stream);
            (void)hipMemsetAsync(D.dmine, 0, ps->slotsPerDev * listBytes, D.stream);     // slots without a part: empty lists (hits = 0)
            uint32_t slot = 0;
            for (uint32_t p = d; p < np; p += nd, ++slot) {
                mc_device_batch in{D.dseq, D.dqinfo, D.dmaxwin, 0, m, at};
                mc_device_results res{};
                int rc = mc_query_device(ps->cur[p], &in, lowestRank, 0, &res, D.stream);
                if (!rc) rc = mc_copy_results_on(ps->cur[p], reinterpret_cast<char*>(D.dmine) + slot * listBytes, res.cands, (uint64_t)m * K * sizeof(mc_candidate), 0, D.stream);
                if (rc) { rcs[d] = rc; errs[d] = mc_last_error(ps->cur[p]); return; }
            }
        };
        if (nd == 1) run_device(0);
        else {
            std::vector<std::thread> th;
            for (uint32_t d = 0; d < nd; ++d) th.emplace_back(run_device, d);
            for (auto& t : th) t.join();
        }
        for (uint32_t d = 0; d < nd; ++d) if (rcs[d]) return idle(rcs[d], errs[d]);
        const uint64_t tr3 = now_ns();
        // per-rank partial lists gathered over RCCL: every rank's slotsPerDev lists -> dall[rank][slot] on every device
        if (!ps->rccl) {
            (void)hipSetDevice(ps->dev[0].device);
            (void)hipMemcpyAsync(ps->dev[0].dall, ps->dev[0].dmine, ps->slotsPerDev * listBytes, hipMemcpyDeviceToDevice, ps->dev[0].stream);
        } else {
            g_rccl.GroupStart();
            int r = 0;
            for (uint32_t d = 0; d < nd && !r; ++d) {
                DevState& D = ps->dev[d];
                (void)hipSetDevice(D.device);
                r = g_rccl.AllGather(D.dmine, D.dall, ps->slotsPerDev * listBytes, /*ncclChar*/ 0, D.comm, D.stream);
            }
            const int e2 = g_rccl.GroupEnd();
            if (r || e2) return idle(MC_ERR_HIP, std::string("ncclAllGather of the per-part candidates: ") + g_rccl.text(r ? r : e2));
        }
        // device 0: the earlier groups' list of these reads first, then this group's parts in part order (part p: rank p % nd, slot p / nd)
        DevState& D0 = ps->dev[0];
        if (hipSetDevice(D0.device) != hipSuccess) return idle(MC_ERR_HIP, "hipSetDevice");
        std::vector<const mc_candidate*> lists;
        if (hasPrior) lists.push_back(D0.dprior);
        for (uint32_t p = 0; p < np; ++p)
            lists.push_back(reinterpret_cast<const mc_candidate*>(reinterpret_cast<const char*>(D0.dall) + ((size_t)(p % nd) * ps->slotsPerDev + p / nd) * listBytes));
        int rc = mc_merge_part_candidates(ps->cur[0], lists.data(), (uint32_t)lists.size(), m, lowestRank, D0.dout, D0.stream);
        if (rc) return idle(rc, mc_last_error(ps->cur[0]));
        if (hipMemcpyAsync(H.out, D0.dout, (size_t)m * K * sizeof(mc_candidate), hipMemcpyDeviceToHost, D0.stream) != hipSuccess ||
            hipEventRecord(H.outDone, D0.stream) != hipSuccess)
            return idle(MC_ERR_HIP, "copy of the merged candidates failed");
        H.inFlight = true; H.first = B.first; H.count = m;
        const uint64_t tr4 = now_ns();
        tPack += tr1 - tr0; tCollect += tr2 - tr1; tEnqueue += tr3 - tr2; tGather += tr4 - tr3;
    }
    if (trace)
        std::fprintf(stderr, "mc_partset_classify_resident: %zu batches, %u parts; host ms: waiting for the packer %.2f, collect %.2f, enqueue %.2f, gather+merge %.2f\n",
                     batches.size(), np, tPack / 1e6, tCollect / 1e6, tEnqueue / 1e6, tGather / 1e6);
    // the last two batches; then nothing is left in flight (the callers' next call may come from another thread or select another group)
    const size_t nbt = batches.size();
    for (size_t k = nbt >= 2 ? nbt - 2 : 0; k < nbt; ++k)
        if (!collect(ps->hs[k & 1])) return idle(MC_ERR_HIP, "copy of the merged candidates failed");
    for (uint32_t d = 0; d < nd; ++d) { (void)hipSetDevice(ps->dev[d].device); (void)hipStreamSynchronize(ps->dev[d].stream); }
    return MC_OK;
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
