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
#include "context.h"
#include "devcache.h"
#include "rccl_dl.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace mcamd;

namespace {

struct KsRank {
    int device = 0;
    mc_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    void* comm = nullptr;
    uint8_t* dseq = nullptr; uint32_t* dqinfo = nullptr; uint32_t* dmaxwin = nullptr;   // the batch (all reads): ONE copy per device -- the first shard of a device
                                                                                        // owns the buffers and uploads, the others (tests: several shards on one GPU) read them
    bool ownsInput = false;
    hipEvent_t upDone = nullptr;                     // owner: the batch is on the device
    mc_candidate* hout = nullptr;                    // pinned: this owner's candidates come back here (a pageable target makes the copy a blocking staged one)
    uint32_t* drecvCounts = nullptr;                 // [S][mMax]
    uint32_t* drecvNumbers = nullptr; uint64_t recvCap = 0;
    // of the batch in flight
    mc_device_partial_numbers part{};
    std::vector<uint64_t> cuts;                      // [S + 1]: where owner o's piece begins in part.numbers
    std::vector<uint64_t> srcOff;                    // [S + 1]: where source s' block begins in drecvNumbers
    int rc = MC_OK; std::string err;
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
    uint64_t locations = 0, numbersSent = 0, batches = 0;
    // pinned staging of a batch (pageable memory: 120 ms per 10^6 reads and shard), twice: batch b + 1 is packed by a helper thread while
    // the shards work on batch b (every batch ends with all streams idle: a slot's last batch is long through when it is packed again)
    struct HostSlot { uint8_t* seq = nullptr; uint32_t* q = nullptr; uint32_t* mw = nullptr; uint64_t chars = 0; } hs[2];
    uint32_t packThreads = 8;          // threads that copy a batch's characters (MC_KEYSET_PACK_THREADS)
};

namespace {

int ks_fail(mc_keyset* ks, int code, const std::string& msg) { if (ks) ks->err = msg; else set_global_error(msg); return code; }

// contiguous, balanced read shards (as metacache_amd/distributed.py shard_bounds)
uint32_t shard_lo(uint32_t n, uint32_t r, uint32_t world) { const uint32_t base = n / world, rem = n % world; return r * base + std::min(r, rem); }

// an error leaves nothing in flight: the next call (or mc_keyset_close) finds idle streams and free staging buffers
int ks_fail_idle(mc_keyset* ks, int code, const std::string& msg)
{
    for (KsRank& R : ks->rank) { (void)hipSetDevice(R.device); if (R.stream) (void)hipStreamSynchronize(R.stream); }
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
        c.device = R.device; c.key_shard_index = r; c.key_shard_count = ks->S; c.num_slots = 1; c.copy_allhits = 0;
        R.rc = mc_open_database(ks->db.c_str(), &c, &R.ctx);
        if (R.rc) { R.err = mc_last_error(nullptr); return; }
        R.ownsInput = r < nd;
        bool ok = hipSetDevice(R.device) == hipSuccess && hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking) == hipSuccess &&
                  mcamd::dev_malloc((void**)&R.drecvCounts, (size_t)ks->S * std::max<uint32_t>(mMax, 1) * 4) == hipSuccess &&
                  hipHostMalloc((void**)&R.hout, (size_t)std::max<uint32_t>(mMax, 1) * ks->K * sizeof(mc_candidate)) == hipSuccess;
        if (ok && R.ownsInput)
            ok = mcamd::dev_malloc((void**)&R.dseq, ks->maxChars + 64) == hipSuccess && mcamd::dev_malloc((void**)&R.dqinfo, ks->maxQ * 16) == hipSuccess &&
                 mcamd::dev_malloc((void**)&R.dmaxwin, ks->maxQ * 4) == hipSuccess && hipEventCreateWithFlags(&R.upDone, hipEventDisableTiming) == hipSuccess;
        if (!ok) { R.rc = MC_ERR_NOMEM; R.err = "mc_keyset_open: cannot allocate the batch buffers"; }
    });
    for (uint32_t r = nd; r < ks->S; ++r) { KsRank& O = ks->rank[r % nd]; ks->rank[r].dseq = O.dseq; ks->rank[r].dqinfo = O.dqinfo; ks->rank[r].dmaxwin = O.dmaxwin; }
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
        if (R.stream) (void)hipStreamSynchronize(R.stream);
        if (R.ctx) mc_destroy(R.ctx);
        void* bufs[] = {R.ownsInput ? R.dseq : nullptr, R.ownsInput ? (void*)R.dqinfo : nullptr, R.ownsInput ? (void*)R.dmaxwin : nullptr, R.drecvCounts, R.drecvNumbers};
        for (void* b : bufs) if (b) (void)hipFree(b);
        if (R.hout) (void)hipHostFree(R.hout);
        if (R.upDone) (void)hipEventDestroy(R.upDone);
        if (R.comm && std::find(destroyed.begin(), destroyed.end(), R.comm) == destroyed.end()) { rccl().CommDestroy(R.comm); destroyed.push_back(R.comm); }
        if (R.stream) (void)hipStreamDestroy(R.stream);
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
    info[0] = ks->S; info[1] = ks->devices.size(); info[2] = ks->rccl ? 1 : 0; info[3] = ks->locations; info[4] = ks->numbersSent; info[5] = ks->batches;
    info[6] = info[7] = 0;
    for (const KsRank& R : ks->rank) {
        uint64_t st[4];
        if (R.ctx && mc_owner_stats(R.ctx, st) == MC_OK) { info[6] += st[1]; info[7] += st[3]; }
    }
    return MC_OK;
}

int mc_keyset_classify(mc_keyset* ks, const char* seqs, const uint64_t* offs, const char* seqs2, const uint64_t* offs2, uint64_t n, int lowestRank,
                       uint64_t insertMax, mc_candidate* out)
{
    if (!ks || !seqs || !offs || !out || (seqs2 && !offs2)) return MC_ERR_INVALID;
    const uint32_t K = ks->K, S = ks->S;
    struct Batch { uint64_t first, count, chars; };
    std::vector<Batch> batches;
    auto need = [&](uint64_t i) {
        const uint64_t l1 = offs[i + 1] - offs[i], l2 = seqs2 ? offs2[i + 1] - offs2[i] : 0;
        return (l1 + 3) / 4 * 4 + (l2 + 3) / 4 * 4;
    };
    for (uint64_t i = 0; i < n;) {
        Batch b{i, 0, 0};
        while (i < n && b.count < ks->maxQ && b.chars + need(i) <= ks->maxChars) { b.chars += need(i); ++b.count; ++i; }
        if (b.count == 0) return ks_fail(ks, MC_ERR_INVALID, "mc_keyset_classify: a read is longer than slot_max_chars");
        batches.push_back(b);
    }
    std::vector<uint32_t> bounds(S + 1);
    Rccl& R = rccl();
    // a batch into its slot's pinned buffers: where every read goes (a sequence starts 4-byte aligned), then the characters by a few threads
    auto pack_batch = [&](size_t bi) {
        const Batch& B = batches[bi];
        mc_keyset::HostSlot& H = ks->hs[bi & 1];
        const uint32_t m = (uint32_t)B.count;
        uint64_t at = 0;
        for (uint32_t j = 0; j < m; ++j) {
            const uint64_t i = B.first + j, l1 = offs[i + 1] - offs[i], l2 = seqs2 ? offs2[i + 1] - offs2[i] : 0;
            H.q[4 * j] = (uint32_t)at; H.q[4 * j + 1] = (uint32_t)l1;
            at += (l1 + 3) / 4 * 4;
            H.q[4 * j + 2] = (uint32_t)at; H.q[4 * j + 3] = (uint32_t)l2;
            at += (l2 + 3) / 4 * 4;
            H.mw[j] = (uint32_t)(2 + std::max<uint64_t>(l1 + l2, insertMax) / ks->stride);   // candidate_structs.hpp:143-145
        }
        H.chars = at;
        auto pack = [&](uint32_t j0, uint32_t j1) {
            for (uint32_t j = j0; j < j1; ++j) {
                const uint64_t i = B.first + j;
                if (H.q[4 * j + 1]) std::memcpy(H.seq + H.q[4 * j], seqs + offs[i], H.q[4 * j + 1]);
                if (H.q[4 * j + 3]) std::memcpy(H.seq + H.q[4 * j + 2], seqs2 + offs2[i], H.q[4 * j + 3]);
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
    };
    std::thread packer;
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{packer};
    static const bool trace = std::getenv("MC_KEYSET_TRACE") != nullptr;
    auto now_ns = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; };
    uint64_t tPack = 0, tShards = 0, tExchange = 0, tOwners = 0;
    const uint64_t tp0 = now_ns();
    if (!batches.empty()) pack_batch(0);
    tPack += now_ns() - tp0;
    for (size_t bi = 0; bi < batches.size(); ++bi) {
        const Batch& B = batches[bi];
        const uint64_t tb0 = now_ns();
        if (packer.joinable()) packer.join();                       // this batch is in its slot
        tPack += now_ns() - tb0;
        const uint64_t tb1 = now_ns();
        if (bi + 1 < batches.size()) packer = std::thread(pack_batch, bi + 1);   // (the other slot's batch ended with all streams idle)
        const mc_keyset::HostSlot& H = ks->hs[bi & 1];
        uint8_t* const hseq = H.seq; uint32_t* const hq = H.q; uint32_t* const hmw = H.mw;
        const uint32_t m = (uint32_t)B.count;
        const uint64_t at = H.chars;
        for (uint32_t o = 0; o <= S; ++o) bounds[o] = o < S ? shard_lo(m, o, S) : m;
        // ---- 1. every shard: its features' locations for ALL reads, as numbers.  The batch crosses the link once per DEVICE
        // (the last batch ended with every stream idle: the buffers are free)
        const uint32_t nd = (uint32_t)ks->devices.size();
        for (uint32_t d = 0; d < std::min(nd, S); ++d) {
            KsRank& O = ks->rank[d];
            if (hipSetDevice(O.device) != hipSuccess) return ks_fail_idle(ks, MC_ERR_HIP, "hipSetDevice");
            if (hipMemcpyAsync(O.dseq, hseq, at + 16, hipMemcpyHostToDevice, O.stream) != hipSuccess ||
                hipMemcpyAsync(O.dqinfo, hq, (size_t)m * 16, hipMemcpyHostToDevice, O.stream) != hipSuccess ||
                hipMemcpyAsync(O.dmaxwin, hmw, (size_t)m * 4, hipMemcpyHostToDevice, O.stream) != hipSuccess ||
                hipEventRecord(O.upDone, O.stream) != hipSuccess)
                return ks_fail_idle(ks, MC_ERR_HIP, "copy of a batch to the device failed");
        }
        for_each_rank(ks, [&](uint32_t r) {
            KsRank& Rk = ks->rank[r];
            Rk.rc = MC_OK;
            if (hipSetDevice(Rk.device) != hipSuccess) { Rk.rc = MC_ERR_HIP; Rk.err = "hipSetDevice"; return; }
            if (!Rk.ownsInput && hipStreamWaitEvent(Rk.stream, ks->rank[r % nd].upDone, 0) != hipSuccess) { Rk.rc = MC_ERR_HIP; Rk.err = "hipStreamWaitEvent"; return; }
            mc_device_batch in{Rk.dseq, Rk.dqinfo, Rk.dmaxwin, 0, m, at};
            mc_device_results res{};
            Rk.cuts.assign(S + 1, 0);
            int rc = mc_query_device(Rk.ctx, &in, lowestRank, MC_WANT_PARTIAL_NUMBERS, &res, Rk.stream);
            if (!rc) rc = mc_partial_numbers(Rk.ctx, &res, m, bounds.data(), S + 1, Rk.cuts.data(), &Rk.part, Rk.stream);
            if (!rc && !ks->rccl && hipStreamSynchronize(Rk.stream) != hipSuccess) rc = MC_ERR_HIP;   // (copies below run on the owners' streams)
            if (rc) { Rk.rc = rc; Rk.err = rc == MC_ERR_HIP && Rk.err.empty() ? "HIP error" : mc_last_error(Rk.ctx); }
        });
        for (KsRank& Rk : ks->rank) if (Rk.rc) return ks_fail_idle(ks, Rk.rc, Rk.err);
        const uint64_t tb2 = now_ns();
        // ---- 2. the exchange.  Owner o receives from source s the numbers [cuts_s[o], cuts_s[o + 1]) and the counts of its reads
        for (uint32_t o = 0; o < S; ++o) {
            KsRank& O = ks->rank[o];
            O.srcOff.assign(S + 1, 0);
            for (uint32_t s = 0; s < S; ++s) O.srcOff[s + 1] = O.srcOff[s] + (ks->rank[s].cuts[o + 1] - ks->rank[s].cuts[o]);
            if (O.srcOff[S] + 8 > O.recvCap) {
                if (hipSetDevice(O.device) != hipSuccess) return ks_fail_idle(ks, MC_ERR_HIP, "hipSetDevice");
                if (O.drecvNumbers) { (void)hipStreamSynchronize(O.stream); (void)hipFree(O.drecvNumbers); O.drecvNumbers = nullptr; }
                O.recvCap = O.srcOff[S] + O.srcOff[S] / 4 + 1024;
                if (mcamd::dev_malloc((void**)&O.drecvNumbers, O.recvCap * 4) != hipSuccess) { O.recvCap = 0; return ks_fail_idle(ks, MC_ERR_NOMEM, "mc_keyset_classify: receive buffer"); }
            }
            ks->numbersSent += O.srcOff[S];
        }
        if (ks->rccl) {
            int r = R.GroupStart();
            for (uint32_t a = 0; a < S && !r; ++a) {
                KsRank& A = ks->rank[a];
                (void)hipSetDevice(A.device);
                const uint32_t ma = bounds[a + 1] - bounds[a];
                for (uint32_t p = 0; p < S && !r; ++p) {
                    const uint32_t mp = bounds[p + 1] - bounds[p];
                    const uint64_t sendN = A.cuts[p + 1] - A.cuts[p], recvN = A.srcOff[p + 1] - A.srcOff[p];
                    if (mp) r = R.Send(A.part.counts + bounds[p], mp, Rccl::kUint32, (int)p, A.comm, A.stream);
                    if (!r && ma) r = R.Recv(A.drecvCounts + (size_t)p * ma, ma, Rccl::kUint32, (int)p, A.comm, A.stream);
                    if (!r && sendN) r = R.Send(A.part.numbers + A.cuts[p], sendN, Rccl::kUint32, (int)p, A.comm, A.stream);
                    if (!r && recvN) r = R.Recv(A.drecvNumbers + A.srcOff[p], recvN, Rccl::kUint32, (int)p, A.comm, A.stream);
                }
            }
            const int e = R.GroupEnd();
            if (r || e) return ks_fail_idle(ks, MC_ERR_HIP, "RCCL exchange of the partial lists: " + R.text(r ? r : e));
        } else {
            for (uint32_t o = 0; o < S; ++o) {
                KsRank& O = ks->rank[o];
                (void)hipSetDevice(O.device);
                const uint32_t mo = bounds[o + 1] - bounds[o];
                for (uint32_t s = 0; s < S; ++s) {
                    KsRank& Sr = ks->rank[s];
                    const uint64_t cnt = Sr.cuts[o + 1] - Sr.cuts[o];
                    if (mo) (void)hipMemcpyAsync(O.drecvCounts + (size_t)s * mo, Sr.part.counts + bounds[o], (size_t)mo * 4, hipMemcpyDeviceToDevice, O.stream);
                    if (cnt) (void)hipMemcpyAsync(O.drecvNumbers + O.srcOff[s], Sr.part.numbers + Sr.cuts[o], cnt * 4, hipMemcpyDeviceToDevice, O.stream);
                }
            }
        }
        const uint64_t tb3 = now_ns();
        // ---- 3. every owner: rows 8-10 on what it received, its reads' candidates to the host
        for_each_rank(ks, [&](uint32_t o) {
            KsRank& O = ks->rank[o];
            const uint32_t mo = bounds[o + 1] - bounds[o];
            if (!mo) return;
            if (hipSetDevice(O.device) != hipSuccess) { O.rc = MC_ERR_HIP; O.err = "hipSetDevice"; return; }
            mc_device_partial_numbers_in in{O.drecvCounts, O.drecvNumbers, O.srcOff.data(), O.dmaxwin + bounds[o], 0, mo, S};
            mc_device_results res{};
            int rc = mc_candidates_from_partial_numbers(O.ctx, &in, lowestRank, &res, O.stream);
            if (!rc) rc = mc_copy_results_on(O.ctx, O.hout, res.cands, (uint64_t)mo * K * sizeof(mc_candidate), 1, O.stream);
            if (rc) { O.rc = rc; O.err = mc_last_error(O.ctx); return; }
            if (hipStreamSynchronize(O.stream) != hipSuccess) { O.rc = MC_ERR_HIP; O.err = "copy of the candidates failed"; return; }
            std::memcpy(out + (B.first + bounds[o]) * K, O.hout, (size_t)mo * K * sizeof(mc_candidate));
        });
        for (KsRank& Rk : ks->rank) if (Rk.rc) return ks_fail_idle(ks, Rk.rc, Rk.err);
        // (a rank without reads of its own still took part in the exchange: its sends must be done before its buffers are reused)
        for (KsRank& Rk : ks->rank) { (void)hipSetDevice(Rk.device); (void)hipStreamSynchronize(Rk.stream); }
        ++ks->batches;
        const uint64_t tb4 = now_ns();
        tShards += tb2 - tb1; tExchange += tb3 - tb2; tOwners += tb4 - tb3;
    }
    if (trace)
        std::fprintf(stderr, "mc_keyset_classify: %zu batches, %u shards; host ms: packing not hidden %.2f, upload + shards' lookups + split sizes %.2f, exchange enqueued %.2f, owners' candidates + copy back %.2f\n",
                     batches.size(), S, tPack / 1e6, tShards / 1e6, tExchange / 1e6, tOwners / 1e6);
    return MC_OK;
}

}  // extern "C"
