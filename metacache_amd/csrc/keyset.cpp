// metacache_amd/csrc/keyset.cpp -- ONE database whose features are key-sharded over the GPUs of the node (SURVEY 8e "Mode K"), driven
// from C++ (C ABI: mc_keyset_*, include/metacache_amd.h).
//
// What it replaces: the reference's query of a database that is spread over its GPUs inside one process -- gpu_hashmap.cu:1255-1290
// (query_hashtables_async over every GPU's table), query_batch.cu:464-527 (the sketches forwarded GPU -> GPU, per-part results
// accumulated), :638-652 (candidates from the accumulated list).  Here the split is by FEATURE (key_owner), not by target, so every
// shard holds 1 / S of every bucket list's keys and the united lists are exactly the single table's: the result is the single-part
// CPU result bit for bit (no per-part top lists to merge).
//
// Per batch:
//   1. every shard: the batch to its device, mc_query_device(MC_WANT_PARTIAL_NUMBERS) -- the shard sketches ALL reads (ALU work, cheap) and
//      looks up only the features it owns, so the lookups of a batch are done once, spread over the shards -- then mc_partial_numbers:
//      the partial lists as 4-byte global window numbers, back to back in read order, and where the read shards' pieces begin (the
//      one host round trip: S + 1 offsets per shard);
//   2. the exchange: shard s' piece for the reads of owner o goes to o (contiguous read shards) -- counts and numbers, one grouped
//      ncclSend / ncclRecv round = all-to-all-v over RCCL (xGMI between devices); the split sizes of all ranks are in this process'
//      memory, so no size exchange is needed.  Shards that share a device (tests on a one-GPU box) copy device-to-device;
//   3. every owner: mc_candidates_from_partial_numbers on what it received (no union copy: the receive buffer is the location store),
//      its reads' top candidates to the host.
// Round 6: TWO batches in flight.  A LANE is a full set of what a batch needs -- per shard a stream, the receive buffers, pinned results,
// and pipe 0 / 1 of the shard's context for BOTH its shard side and its owner side; per device the batch's input; a pinned staging slot --
// so the three steps of a batch on lane 0 run beside those of a batch on lane 1: one batch's lookups under the other's exchange and
// candidates, uploads under kernels, host round trips under device work.  A call with several batches drives both lanes (a helper thread
// takes the odd batches); several callers (mcq's workers, a batch each) take a lane each.  The exchanges' RCCL groups are issued one at a time.
#include "context.h"
#include "devcache.h"
#include "rccl_dl.h"

#include <algorithm>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace mcamd;

namespace {

constexpr uint32_t kKsLanes = 2;

struct KsLaneRank {                                  // a shard's side of a lane
    hipStream_t stream = nullptr;
    uint8_t* dseq = nullptr; uint32_t* dqinfo = nullptr; uint32_t* dmaxwin = nullptr;   // the batch (all reads): ONE copy per device -- the first shard of a device
                                                                                        // owns the buffers and uploads, the others (tests: several shards on one GPU) read them
    hipEvent_t upDone = nullptr;                     // owner of the input: the batch is on the device
    uint32_t* drecvCounts = nullptr;                 // [S][mMax]
    uint32_t* drecvNumbers = nullptr; uint64_t recvCap = 0;
    mc_candidate* hout = nullptr;                    // pinned: this owner's candidates come back here (a pageable target makes the copy a blocking staged one)
    // of the batch in flight
    mc_device_partial_numbers part{};
    std::vector<uint64_t> cuts;                      // [S + 1]: where owner o's piece begins in part.numbers
    std::vector<uint64_t> srcOff;                    // [S + 1]: where source s' block begins in drecvNumbers
    int rc = MC_OK; std::string err;
};
struct KsRank {
    int device = 0;
    mc_ctx* ctx = nullptr;
    void* comm = nullptr;
    bool ownsInput = false;
    KsLaneRank lane[kKsLanes];
    int rc = MC_OK; std::string err;                 // of the open
};

}  // namespace

struct mc_keyset {
    std::string db, err;
    mc_config cfg{};
    uint32_t S = 1, K = 2, stride = 112;
    std::vector<int> devices;
    std::vector<KsRank> rank;
    size_t maxQ = 0, maxChars = 0;
    bool rccl = false;
    uint64_t locations = 0;
    std::atomic<uint64_t> numbersSent{0}, batches{0};
    // pinned staging of a batch (pageable memory: 120 ms per 10^6 reads and shard), one per lane
    struct HostSlot { uint8_t* seq = nullptr; uint32_t* q = nullptr; uint32_t* mw = nullptr; uint64_t chars = 0; } hs[kKsLanes];
    bool laneBusy[kKsLanes] = {false, false};
    std::mutex poolMu; std::condition_variable poolCv;   // the lanes
    std::mutex exMu;                                 // the exchange's RCCL group: one at a time, the same order on every communicator
    uint32_t packThreads = 8;          // threads that copy a batch's characters (MC_KEYSET_PACK_THREADS)
};

namespace {

int ks_fail(mc_keyset* ks, int code, const std::string& msg)
{
    static std::mutex mu;                                          // (two lanes may fail at once: the last one's text stays)
    std::lock_guard<std::mutex> l(mu);
    if (ks) ks->err = msg; else set_global_error(msg);
    return code;
}

// contiguous, balanced read shards (as metacache_amd/distributed.py shard_bounds)
uint32_t shard_lo(uint32_t n, uint32_t r, uint32_t world) { const uint32_t base = n / world, rem = n % world; return r * base + std::min(r, rem); }

// an error leaves nothing in flight: the next call (or mc_keyset_close) finds idle streams and free staging buffers
int ks_fail_idle(mc_keyset* ks, int code, const std::string& msg)
{
    for (KsRank& R : ks->rank) { (void)hipSetDevice(R.device); for (KsLaneRank& L : R.lane) if (L.stream) (void)hipStreamSynchronize(L.stream); }
    return ks_fail(ks, code, msg);
}

template <class F>
void for_each_rank(mc_keyset* ks, F&& f)
{
    if (ks->S == 1) { f(0u); return; }
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < ks->S; ++r) th.emplace_back(f, r);
    for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

const char* mc_keyset_last_error(const mc_keyset* ks) { return ks ? ks->err.c_str() : mc_last_error(nullptr); }

int mc_keyset_open(const char* name, const mc_config* cfg, uint32_t numShards, const int32_t* devices, uint32_t numDevices, mc_keyset** out)
{
    if (!name || !cfg || !out) return MC_ERR_INVALID;
    *out = nullptr;
    int ndevAvail = 0;
    if (hipGetDeviceCount(&ndevAvail) != hipSuccess || ndevAvail < 1) return ks_fail(nullptr, MC_ERR_HIP, "no usable HIP device (this library has no CPU fallback)");
    auto* ks = new mc_keyset;
    ks->db = name; ks->cfg = *cfg; ks->K = cfg->max_candidates;
    if (devices && numDevices) ks->devices.assign(devices, devices + numDevices); else ks->devices.assign(1, cfg->device);
    const uint32_t nd = (uint32_t)ks->devices.size();
    ks->S = numShards ? numShards : nd;
    auto bail = [&](int code, const std::string& msg) { mc_keyset_close(ks); return ks_fail(nullptr, code, msg); };
    if (ks->K > 4) return bail(MC_ERR_UNSUPPORTED, "mc_keyset_open: max_candidates above 4");
    if (ks->S > 64) return bail(MC_ERR_UNSUPPORTED, "mc_keyset_open: more than 64 key shards");
    for (size_t i = 0; i < nd; ++i) {
        if (ks->devices[i] < 0 || ks->devices[i] >= ndevAvail) return bail(MC_ERR_INVALID, "mc_keyset_open: device ordinal out of range");
        for (size_t j = 0; j < i; ++j) if (ks->devices[j] == ks->devices[i]) return bail(MC_ERR_INVALID, "mc_keyset_open: a device is listed twice");
    }
    if (nd > 1 && ks->S != nd) return bail(MC_ERR_INVALID, "mc_keyset_open: with several devices, one key shard per device");
    ks->maxQ = std::max<uint32_t>(cfg->slot_max_queries, 1);
    ks->maxChars = std::max<uint32_t>(cfg->slot_max_chars, 1u << 16);
    // RCCL: one communicator rank per device of this process; a single device needs none (MC_KEYSET_RCCL=1 runs the same calls with
    // one rank -- shard 0 sends to itself -- for the tests of a one-GPU box)
    const char* force = std::getenv("MC_KEYSET_RCCL");
    ks->rccl = nd > 1 || (ks->S == 1 && force && force[0] == '1');
    std::vector<void*> comms(nd, nullptr);
    ks->rank.resize(ks->S);
    for (uint32_t r = 0; r < ks->S; ++r) ks->rank[r].device = ks->devices[r % nd];
    if (ks->rccl) {
        Rccl& R = rccl();
        if (!R.load()) return bail(MC_ERR_UNSUPPORTED, R.err);
        if (int r = R.CommInitAll(comms.data(), (int)nd, ks->devices.data())) return bail(MC_ERR_HIP, "ncclCommInitAll: " + R.text(r));
        for (uint32_t r = 0; r < ks->S; ++r) ks->rank[r].comm = comms[r % nd];      // (owned by the ranks from here on: mc_keyset_close destroys them)
    }
    const uint32_t mMax = (uint32_t)((ks->maxQ + ks->S - 1) / ks->S);
    // the shards load side by side (each reads the whole file and keeps its keys)
    for_each_rank(ks, [&](uint32_t r) {
        KsRank& R = ks->rank[r];
        mc_config c = ks->cfg;
        c.device = R.device; c.key_shard_index = r; c.key_shard_count = ks->S; c.num_slots = 0; c.copy_allhits = 0;   // (no host slots: the set has its own staging)
        R.rc = mc_open_database(ks->db.c_str(), &c, &R.ctx);
        if (R.rc) { R.err = mc_last_error(nullptr); return; }
        R.ownsInput = r < nd;
        bool ok = hipSetDevice(R.device) == hipSuccess;
        for (KsLaneRank& L : R.lane) {
            if (!ok) break;
            ok = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess &&
                 mcamd::dev_malloc((void**)&L.drecvCounts, (size_t)ks->S * std::max<uint32_t>(mMax, 1) * 4) == hipSuccess &&
                 hipHostMalloc((void**)&L.hout, (size_t)std::max<uint32_t>(mMax, 1) * ks->K * sizeof(mc_candidate)) == hipSuccess;
            if (ok && R.ownsInput)
                ok = mcamd::dev_malloc((void**)&L.dseq, ks->maxChars + 64) == hipSuccess && mcamd::dev_malloc((void**)&L.dqinfo, ks->maxQ * 16) == hipSuccess &&
                     mcamd::dev_malloc((void**)&L.dmaxwin, ks->maxQ * 4) == hipSuccess && hipEventCreateWithFlags(&L.upDone, hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) { R.rc = MC_ERR_NOMEM; R.err = "mc_keyset_open: cannot allocate the batch buffers"; }
    });
    for (uint32_t r = nd; r < ks->S; ++r)
        for (uint32_t l = 0; l < kKsLanes; ++l) {
            KsLaneRank& O = ks->rank[r % nd].lane[l]; KsLaneRank& L = ks->rank[r].lane[l];
            L.dseq = O.dseq; L.dqinfo = O.dqinfo; L.dmaxwin = O.dmaxwin;
        }
    for (KsRank& R : ks->rank) if (R.rc) { const int rc = R.rc; const std::string e = R.err; mc_keyset_close(ks); return ks_fail(nullptr, rc, e); }
    for (auto& H : ks->hs)
        if (hipHostMalloc((void**)&H.seq, ks->maxChars + 64) != hipSuccess || hipHostMalloc((void**)&H.q, ks->maxQ * 16) != hipSuccess ||
            hipHostMalloc((void**)&H.mw, ks->maxQ * 4) != hipSuccess)
            return bail(MC_ERR_NOMEM, "mc_keyset_open: cannot allocate the pinned batch staging");
    if (const char* e = std::getenv("MC_KEYSET_PACK_THREADS")) ks->packThreads = (uint32_t)std::max(1, std::atoi(e));
    ks->packThreads = std::min<uint32_t>(ks->packThreads, std::max(1u, std::thread::hardware_concurrency()));
    uint64_t info[8];
    for (KsRank& R : ks->rank) {
        mc_db_info(R.ctx, info); ks->locations += info[7]; ks->stride = (uint32_t)(info[3] ? info[3] : 112);
        // what travels is the compact store's global window number: a database without that numbering (several parts, more than 2^32
        // windows with their gaps, MC_COMPACT_LOCATIONS=0) keeps the per-process 8-byte form (metacache_amd/distributed.py, wire = 8)
        uint64_t lay[4] = {0, 0, 0, 0};
        if (mc_table_layout(R.ctx, lay) != MC_OK || lay[0] != 4)
            return bail(MC_ERR_UNSUPPORTED, "mc_keyset_open: the database has no 32-bit global window numbers (single part, compact location store) to send between the shards");
    }
    *out = ks;
    return MC_OK;
}

void mc_keyset_close(mc_keyset* ks)
{
    if (!ks) return;
    std::vector<void*> destroyed;
    for (KsRank& R : ks->rank) {
        (void)hipSetDevice(R.device);
        for (KsLaneRank& L : R.lane) if (L.stream) (void)hipStreamSynchronize(L.stream);
        if (R.ctx) mc_destroy(R.ctx);
        for (KsLaneRank& L : R.lane) {
            void* bufs[] = {R.ownsInput ? L.dseq : nullptr, R.ownsInput ? (void*)L.dqinfo : nullptr, R.ownsInput ? (void*)L.dmaxwin : nullptr, L.drecvCounts, L.drecvNumbers};
            for (void* b : bufs) if (b) (void)hipFree(b);
            if (L.hout) (void)hipHostFree(L.hout);
            if (L.upDone) (void)hipEventDestroy(L.upDone);
        }
        if (R.comm && std::find(destroyed.begin(), destroyed.end(), R.comm) == destroyed.end()) { rccl().CommDestroy(R.comm); destroyed.push_back(R.comm); }
        for (KsLaneRank& L : R.lane) if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    for (auto& H : ks->hs) {
        if (H.seq) (void)hipHostFree(H.seq);
        if (H.q) (void)hipHostFree(H.q);
        if (H.mw) (void)hipHostFree(H.mw);
    }
    delete ks;
}

int mc_keyset_info(const mc_keyset* ks, uint64_t info[8])
{
    if (!ks || !info) return MC_ERR_INVALID;
    info[0] = ks->S; info[1] = ks->devices.size(); info[2] = ks->rccl ? 1 : 0; info[3] = ks->locations; info[4] = ks->numbersSent.load(); info[5] = ks->batches.load();
    info[6] = info[7] = 0;
    for (const KsRank& R : ks->rank) {
        uint64_t st[4];
        if (R.ctx && mc_owner_stats(R.ctx, st) == MC_OK) { info[6] += st[1]; info[7] += st[3]; }
    }
    return MC_OK;
}

}  // extern "C"

namespace {

struct KsBatch { uint64_t first, count, chars; };
struct KsCall { const char* seqs; const uint64_t* offs; const char* seqs2; const uint64_t* offs2; int lowestRank; uint64_t insertMax; mc_candidate* out; };

int ks_take_lane(mc_keyset* ks)
{
    std::unique_lock<std::mutex> l(ks->poolMu);
    int got = -1;
    ks->poolCv.wait(l, [&] { for (uint32_t i = 0; i < kKsLanes; ++i) if (!ks->laneBusy[i]) { got = (int)i; return true; } return false; });
    ks->laneBusy[got] = true;
    return got;
}
void ks_give_lane(mc_keyset* ks, int i) { { std::lock_guard<std::mutex> l(ks->poolMu); ks->laneBusy[i] = false; } ks->poolCv.notify_all(); }

uint64_t ks_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }

// a batch into the lane's pinned buffers: where every read goes (a sequence starts 4-byte aligned), then the characters by a few threads
void ks_pack(mc_keyset* ks, const KsCall& A, const KsBatch& B, mc_keyset::HostSlot& H)
{
    const uint32_t m = (uint32_t)B.count;
    uint64_t at = 0;
    for (uint32_t j = 0; j < m; ++j) {
        const uint64_t i = B.first + j, l1 = A.offs[i + 1] - A.offs[i], l2 = A.seqs2 ? A.offs2[i + 1] - A.offs2[i] : 0;
        H.q[4 * j] = (uint32_t)at; H.q[4 * j + 1] = (uint32_t)l1;
        at += (l1 + 3) / 4 * 4;
        H.q[4 * j + 2] = (uint32_t)at; H.q[4 * j + 3] = (uint32_t)l2;
        at += (l2 + 3) / 4 * 4;
        H.mw[j] = (uint32_t)(2 + std::max<uint64_t>(l1 + l2, A.insertMax) / ks->stride);   // candidate_structs.hpp:143-145
    }
    H.chars = at;
    auto pack = [&](uint32_t j0, uint32_t j1) {
        for (uint32_t j = j0; j < j1; ++j) {
            const uint64_t i = B.first + j;
            if (H.q[4 * j + 1]) std::memcpy(H.seq + H.q[4 * j], A.seqs + A.offs[i], H.q[4 * j + 1]);
            if (H.q[4 * j + 3]) std::memcpy(H.seq + H.q[4 * j + 2], A.seqs2 + A.offs2[i], H.q[4 * j + 3]);
        }
    };
    const uint32_t nt = m >= (1u << 15) ? ks->packThreads : 1;
    if (nt <= 1) pack(0, m);
    else {
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < nt; ++t) th.emplace_back(pack, (uint32_t)((uint64_t)m * t / nt), (uint32_t)((uint64_t)m * (t + 1) / nt));
        pack(0, (uint32_t)((uint64_t)m / nt));
        for (auto& t : th) t.join();
    }
}

// ONE batch through the three steps on lane ln (the caller holds the lane); tns: host nanoseconds [packing, shards, exchange, owners]
int ks_run_batch(mc_keyset* ks, const KsCall& A, const KsBatch& B, int ln, uint64_t tns[4])
{
    const uint32_t K = ks->K, S = ks->S, nd = (uint32_t)ks->devices.size();
    const int pipe = ln ? MC_SECOND_PIPE : 0;
    Rccl& R = rccl();
    mc_keyset::HostSlot& H = ks->hs[ln];
    const uint64_t t0 = ks_now();
    ks_pack(ks, A, B, H);
    const uint64_t t1 = ks_now();
    const uint32_t m = (uint32_t)B.count;
    const uint64_t at = H.chars;
    std::vector<uint32_t> bounds(S + 1);
    for (uint32_t o = 0; o <= S; ++o) bounds[o] = o < S ? shard_lo(m, o, S) : m;
    auto idle = [&](int code, const std::string& msg) {            // an error leaves nothing of this lane in flight
        for (KsRank& Rk : ks->rank) { (void)hipSetDevice(Rk.device); (void)hipStreamSynchronize(Rk.lane[ln].stream); }
        return ks_fail(ks, code, msg);
    };
    // ---- 1. every shard: its features' locations for ALL reads, as numbers.  The batch crosses the link once per DEVICE (the lane's last
    // batch ended with the lane's streams idle: the buffers are free)
    for (uint32_t d = 0; d < std::min(nd, S); ++d) {
        KsRank& O = ks->rank[d]; KsLaneRank& L = O.lane[ln];
        if (hipSetDevice(O.device) != hipSuccess) return idle(MC_ERR_HIP, "hipSetDevice");
        if (hipMemcpyAsync(L.dseq, H.seq, at + 16, hipMemcpyHostToDevice, L.stream) != hipSuccess ||
            hipMemcpyAsync(L.dqinfo, H.q, (size_t)m * 16, hipMemcpyHostToDevice, L.stream) != hipSuccess ||
            hipMemcpyAsync(L.dmaxwin, H.mw, (size_t)m * 4, hipMemcpyHostToDevice, L.stream) != hipSuccess ||
            hipEventRecord(L.upDone, L.stream) != hipSuccess)
            return idle(MC_ERR_HIP, "copy of a batch to the device failed");
    }
    for_each_rank(ks, [&](uint32_t r) {
        KsRank& Rk = ks->rank[r]; KsLaneRank& L = Rk.lane[ln];
        L.rc = MC_OK;
        if (hipSetDevice(Rk.device) != hipSuccess) { L.rc = MC_ERR_HIP; L.err = "hipSetDevice"; return; }
        if (!Rk.ownsInput && hipStreamWaitEvent(L.stream, ks->rank[r % nd].lane[ln].upDone, 0) != hipSuccess) { L.rc = MC_ERR_HIP; L.err = "hipStreamWaitEvent"; return; }
        mc_device_batch in{L.dseq, L.dqinfo, L.dmaxwin, 0, m, at};
        mc_device_results res{};
        L.cuts.assign(S + 1, 0);
        int rc = mc_query_device(Rk.ctx, &in, A.lowestRank, MC_WANT_PARTIAL_NUMBERS | pipe, &res, L.stream);
        if (!rc) rc = mc_partial_numbers(Rk.ctx, &res, m, bounds.data(), S + 1, L.cuts.data(), &L.part, L.stream);
        if (!rc && !ks->rccl && hipStreamSynchronize(L.stream) != hipSuccess) rc = MC_ERR_HIP;   // (copies below run on the owners' streams)
        if (rc) { L.rc = rc; L.err = rc == MC_ERR_HIP && L.err.empty() ? "HIP error" : mc_last_error(Rk.ctx); }
    });
    for (KsRank& Rk : ks->rank) if (Rk.lane[ln].rc) return idle(Rk.lane[ln].rc, Rk.lane[ln].err);
    const uint64_t t2 = ks_now();
    // ---- 2. the exchange.  Owner o receives from source s the numbers [cuts_s[o], cuts_s[o + 1]) and the counts of its reads
    for (uint32_t o = 0; o < S; ++o) {
        KsLaneRank& O = ks->rank[o].lane[ln];
        O.srcOff.assign(S + 1, 0);
        for (uint32_t s = 0; s < S; ++s) O.srcOff[s + 1] = O.srcOff[s] + (ks->rank[s].lane[ln].cuts[o + 1] - ks->rank[s].lane[ln].cuts[o]);
        if (O.srcOff[S] + 8 > O.recvCap) {
            if (hipSetDevice(ks->rank[o].device) != hipSuccess) return idle(MC_ERR_HIP, "hipSetDevice");
            if (O.drecvNumbers) { (void)hipStreamSynchronize(O.stream); (void)hipFree(O.drecvNumbers); O.drecvNumbers = nullptr; }
            O.recvCap = O.srcOff[S] + O.srcOff[S] / 4 + 1024;
            if (mcamd::dev_malloc((void**)&O.drecvNumbers, O.recvCap * 4) != hipSuccess) { O.recvCap = 0; return idle(MC_ERR_NOMEM, "mc_keyset_classify: receive buffer"); }
        }
        ks->numbersSent += O.srcOff[S];
    }
    if (ks->rccl) {
        std::lock_guard<std::mutex> order(ks->exMu);
        int r = R.GroupStart();
        for (uint32_t a = 0; a < S && !r; ++a) {
            KsRank& Ar = ks->rank[a]; KsLaneRank& L = Ar.lane[ln];
            (void)hipSetDevice(Ar.device);
            const uint32_t ma = bounds[a + 1] - bounds[a];
            for (uint32_t p = 0; p < S && !r; ++p) {
                const uint32_t mp = bounds[p + 1] - bounds[p];
                const uint64_t sendN = L.cuts[p + 1] - L.cuts[p], recvN = L.srcOff[p + 1] - L.srcOff[p];
                if (mp) r = R.Send(L.part.counts + bounds[p], mp, Rccl::kUint32, (int)p, Ar.comm, L.stream);
                if (!r && ma) r = R.Recv(L.drecvCounts + (size_t)p * ma, ma, Rccl::kUint32, (int)p, Ar.comm, L.stream);
                if (!r && sendN) r = R.Send(L.part.numbers + L.cuts[p], sendN, Rccl::kUint32, (int)p, Ar.comm, L.stream);
                if (!r && recvN) r = R.Recv(L.drecvNumbers + L.srcOff[p], recvN, Rccl::kUint32, (int)p, Ar.comm, L.stream);
            }
        }
        const int e = R.GroupEnd();
        if (r || e) return idle(MC_ERR_HIP, "RCCL exchange of the partial lists: " + R.text(r ? r : e));
    } else {
        for (uint32_t o = 0; o < S; ++o) {
            KsRank& Or = ks->rank[o]; KsLaneRank& O = Or.lane[ln];
            (void)hipSetDevice(Or.device);
            const uint32_t mo = bounds[o + 1] - bounds[o];
            for (uint32_t s = 0; s < S; ++s) {
                KsLaneRank& Sr = ks->rank[s].lane[ln];
                const uint64_t cnt = Sr.cuts[o + 1] - Sr.cuts[o];
                if (mo) (void)hipMemcpyAsync(O.drecvCounts + (size_t)s * mo, Sr.part.counts + bounds[o], (size_t)mo * 4, hipMemcpyDeviceToDevice, O.stream);
                if (cnt) (void)hipMemcpyAsync(O.drecvNumbers + O.srcOff[s], Sr.part.numbers + Sr.cuts[o], cnt * 4, hipMemcpyDeviceToDevice, O.stream);
            }
        }
    }
    const uint64_t t3 = ks_now();
    // ---- 3. every owner: rows 8-10 on what it received (on the lane's pipe of its context), its reads' candidates to the host
    for_each_rank(ks, [&](uint32_t o) {
        KsRank& Or = ks->rank[o]; KsLaneRank& O = Or.lane[ln];
        const uint32_t mo = bounds[o + 1] - bounds[o];
        if (!mo) return;
        if (hipSetDevice(Or.device) != hipSuccess) { O.rc = MC_ERR_HIP; O.err = "hipSetDevice"; return; }
        mc_device_partial_numbers_in in{O.drecvCounts, O.drecvNumbers, O.srcOff.data(), O.dmaxwin + bounds[o], 0, mo, S};
        mc_device_results res{};
        int rc = mc_candidates_from_partial_numbers_on(Or.ctx, &in, A.lowestRank, pipe, &res, O.stream);
        if (!rc) rc = mc_copy_results_on(Or.ctx, O.hout, res.cands, (uint64_t)mo * K * sizeof(mc_candidate), 1, O.stream);
        if (rc) { O.rc = rc; O.err = mc_last_error(Or.ctx); return; }
        if (hipStreamSynchronize(O.stream) != hipSuccess) { O.rc = MC_ERR_HIP; O.err = "copy of the candidates failed"; return; }
        std::memcpy(A.out + (B.first + bounds[o]) * K, O.hout, (size_t)mo * K * sizeof(mc_candidate));
    });
    for (KsRank& Rk : ks->rank) if (Rk.lane[ln].rc) return idle(Rk.lane[ln].rc, Rk.lane[ln].err);
    // (a rank without reads of its own still took part in the exchange: its sends must be done before the lane's buffers are reused)
    for (KsRank& Rk : ks->rank) { (void)hipSetDevice(Rk.device); (void)hipStreamSynchronize(Rk.lane[ln].stream); }
    ++ks->batches;
    const uint64_t t4 = ks_now();
    tns[0] += t1 - t0; tns[1] += t2 - t1; tns[2] += t3 - t2; tns[3] += t4 - t3;
    return MC_OK;
}

}  // namespace

extern "C" {

// Thread-safe: a caller's batch takes one of the set's two lanes; a call with several batches drives both (a helper thread takes every
// second batch), so two batches are in flight either way.
int mc_keyset_classify(mc_keyset* ks, const char* seqs, const uint64_t* offs, const char* seqs2, const uint64_t* offs2, uint64_t n, int lowestRank,
                       uint64_t insertMax, mc_candidate* out)
{
    if (!ks || !seqs || !offs || !out || (seqs2 && !offs2)) return MC_ERR_INVALID;
    const KsCall A{seqs, offs, seqs2, offs2, lowestRank, insertMax, out};
    std::vector<KsBatch> batches;
    auto need = [&](uint64_t i) {
        const uint64_t l1 = offs[i + 1] - offs[i], l2 = seqs2 ? offs2[i + 1] - offs2[i] : 0;
        return (l1 + 3) / 4 * 4 + (l2 + 3) / 4 * 4;
    };
    for (uint64_t i = 0; i < n;) {
        KsBatch b{i, 0, 0};
        while (i < n && b.count < ks->maxQ && b.chars + need(i) <= ks->maxChars) { b.chars += need(i); ++b.count; ++i; }
        if (b.count == 0) return ks_fail(ks, MC_ERR_INVALID, "mc_keyset_classify: a read is longer than slot_max_chars");
        batches.push_back(b);
    }
    static const bool trace = std::getenv("MC_KEYSET_TRACE") != nullptr;
    const uint64_t tAll = ks_now();
    uint64_t tns[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    std::atomic<int> rcAll{MC_OK};
    auto drive = [&](size_t from, size_t step, uint64_t* t) {       // batches from, from + step, ...: each takes a lane of its own
        for (size_t bi = from; bi < batches.size() && rcAll.load() == MC_OK; bi += step) {
            const int ln = ks_take_lane(ks);
            const int rc = ks_run_batch(ks, A, batches[bi], ln, t);
            ks_give_lane(ks, ln);
            if (rc) { int ok = MC_OK; rcAll.compare_exchange_strong(ok, rc); }
        }
    };
    if (batches.size() > 1) {
        std::thread helper(drive, 1, 2, tns[1]);
        drive(0, 2, tns[0]);
        helper.join();
    } else drive(0, 1, tns[0]);
    if (trace)
        std::fprintf(stderr, "mc_keyset_classify: %zu batches, %u shards, %.2f ms; host ms summed over the two drivers: packing %.2f, upload + shards' lookups + split sizes %.2f, "
                     "exchange enqueued %.2f, owners' candidates + copy back %.2f\n", batches.size(), ks->S, (ks_now() - tAll) / 1e6,
                     (tns[0][0] + tns[1][0]) / 1e6, (tns[0][1] + tns[1][1]) / 1e6, (tns[0][2] + tns[1][2]) / 1e6, (tns[0][3] + tns[1][3]) / 1e6);
    return rcAll.load();
}

}  // extern "C"
