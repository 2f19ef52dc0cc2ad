// metacache_amd/csrc/builder.hip -- minimal GPU database builder.
// Target sequences are cut into window-aligned chunks, sketched by the same sketch kernel the query
// path uses (probe disabled), turned into (feature, location) pairs, sorted by feature with a stable
// device radix sort (insertion order = (target, window) order survives inside a bucket) and cut to
// the first max_locations_per_feature locations per feature.  Run-length encoding, truncation and -- for
// mc_build_finish(load) -- the query table itself (table_build.hip) stay on the device; the host copy of the
// file arrays is made only by mc_build_write.
#include <cstring>
#include <string.h>

#include "context.h"

#include "devcache.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace mcamd;

struct TargetRec { std::string name, filename; int64_t parent = 0; uint64_t windows = 0, fileIndex = 0; bool existing = false; };

struct mc_builder {
    mc_config cfg{};
    SketchParams sp{};
    uint32_t maxLocs = 254;
    bool rmOver = false;              // -remove-overpopulated-features at build time (building.cpp:516-534)
    hipStream_t st = nullptr;
    std::string err;
    std::vector<TargetRec> targets;
    // staged chunk records: their characters are either copied into hseq (host sources) or stay where the caller put them in
    // device memory (mc_build_add_target_device: offsets relative to devBase, valid until the next flush)
    std::vector<uint8_t> hseq;
    std::vector<uint32_t> hqinfo, hqtgt, hqfirst;
    const uint8_t* devBase = nullptr; uint64_t devChars = 0;   // device sources of the staged records; characters they cover
    // all pairs so far (device, grow-only)
    uint32_t* dkeys = nullptr; uint64_t* dvals = nullptr; uint64_t npairs = 0, cap = 0;
    // result: the file's arrays -- keys, bucket sizes, location lists ((tgt << 32) | win = {u32 win; u32 tgt}) -- on the device
    bool finished = false, failed = false;
    uint32_t* rK = nullptr; uint8_t* rS = nullptr; uint64_t* rV = nullptr; uint64_t* rVoff = nullptr;   // rVoff[nkeys + 1]
    uint64_t nkeys = 0, nvals = 0;
};

namespace {

constexpr uint32_t kChunkWindows = 256;
constexpr uint64_t kFlushChars = 96ull << 20;          // host sources: staged and uploaded in pieces of this size
constexpr uint64_t kFlushCharsDevice = 3ull << 30;   // device sources: characters per flush (feature scratch = 64 B per window)

#define B_TRY(b, expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) { (b)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return MC_ERR_HIP; } \
    } while (0)

// The window sketches of a flush become (feature, location) pairs behind the pairs so far -- only the valid features (the padding
// 0xFFFFFFFF of short sketches never leaves) and, in a key-sharded builder, only this shard's.  Two passes, one wave per chunk
// record, order kept (the sort that follows is stable: the locations of a feature stay in (target, window) order):
//   own_count : features of the record that are kept            -> scan -> first pair index of every record
//   own_emit  : pairs[(first + rank)] = (feature, (tgt << 32) | window id)
__device__ __forceinline__ bool own_feature(uint32_t f, uint32_t shardIdx, uint32_t shardCnt)
{
    return f != 0xFFFFFFFFu && (shardCnt <= 1 || key_owner(f, shardCnt) == shardIdx);
}
__global__ __launch_bounds__(256) void own_count_kernel(const uint32_t* __restrict__ features, const uint32_t* __restrict__ winOff, uint32_t nq,
                                                        uint32_t s, uint32_t shardIdx, uint32_t shardCnt, uint32_t* __restrict__ recCount)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const uint32_t w0 = winOff[q], total = (winOff[q + 1] - w0) * s;
    uint32_t cnt = 0;
    for (uint32_t i = lane; i < total; i += 64) cnt += own_feature(features[(size_t)w0 * s + i], shardIdx, shardCnt) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if (lane == 0) recCount[q] = cnt;
}
__global__ __launch_bounds__(256) void own_emit_kernel(const uint32_t* __restrict__ features, const uint32_t* __restrict__ winOff,
                                                       const uint32_t* __restrict__ qtgt, const uint32_t* __restrict__ qfirst, uint32_t nq,
                                                       uint32_t s, uint32_t shardIdx, uint32_t shardCnt, const uint32_t* __restrict__ recPos,
                                                       uint32_t* __restrict__ keys, uint64_t* __restrict__ vals)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const uint32_t w0 = winOff[q], total = (winOff[q + 1] - w0) * s;
    const uint64_t tgt = qtgt[q];
    const uint32_t first = qfirst[q];
    uint32_t at = recPos[q];
    for (uint32_t i0 = 0; i0 < total; i0 += 64) {
        const uint32_t i = i0 + lane;
        const uint32_t f = i < total ? features[(size_t)w0 * s + i] : 0xFFFFFFFFu;
        const bool own = own_feature(f, shardIdx, shardCnt);
        const uint64_t mask = __ballot(own);
        if (own) {
            const uint32_t o = at + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            keys[o] = f;
            vals[o] = (tgt << 32) | (first + i / s);
        }
        at += (uint32_t)__popcll(mask);
    }
}

// number of runs of equal keys in a sorted array
__global__ __launch_bounds__(256) void count_runs_kernel(const uint32_t* __restrict__ keys, uint64_t n, unsigned long long* __restrict__ out)
{
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) c += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

// per run (= feature): kept size and its u32 copy for the scan
__global__ __launch_bounds__(256) void keep_sizes_kernel(const uint32_t* __restrict__ counts, uint32_t nruns, uint32_t maxLocs,
                                                         uint8_t* __restrict__ sizes, uint32_t* __restrict__ keep32)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nruns) return;
    const uint32_t k = counts[r] < maxLocs ? counts[r] : maxLocs;
    sizes[r] = (uint8_t)k; keep32[r] = k;
}

// per run: the FIRST 'keep' locations in insertion order (host_hashmap.hpp:593-605) move to their place in the file's value array
__global__ __launch_bounds__(256) void compact_values_kernel(const uint64_t* __restrict__ sorted, const uint64_t* __restrict__ runOff,
                                                             const uint64_t* __restrict__ keepOff, const uint8_t* __restrict__ sizes,
                                                             uint32_t nruns, uint64_t* __restrict__ out)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nruns) return;
    const uint64_t src = runOff[r], dst = keepOff[r];
    const uint32_t k = sizes[r];
    for (uint32_t t = 0; t < k; ++t) out[dst + t] = sorted[src + t];
}

// -remove-ambig-features (host_hashmap.hpp:499-540): a feature whose locations lie in more than maxAmbig different taxa on the chosen
// rank goes.  anc[target] = the target's ancestor on that rank (0 = none: counts as one taxon like the reference's null pointer;
// rank 'sequence': the target itself).  One thread per feature; targets come sorted, so runs of one target cost one comparison.
__global__ __launch_bounds__(256) void ambig_flags_kernel(const uint64_t* __restrict__ vals, const uint64_t* __restrict__ voff,
                                                          const uint8_t* __restrict__ sizes, uint32_t nkeys,
                                                          const uint32_t* __restrict__ anc, uint32_t maxAmbig,
                                                          uint32_t* __restrict__ keep, uint32_t* __restrict__ keepSize)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nkeys) return;
    const uint64_t* v = vals + voff[i];
    const uint32_t sz = sizes[i];
    uint32_t distinct = 0, prevTgt = 0xFFFFFFFFu;
    for (uint32_t t = 0; t < sz && distinct <= maxAmbig; ++t) {
        const uint32_t tgt = (uint32_t)(v[t] >> 32);
        if (tgt == prevTgt) continue;
        prevTgt = tgt;
        const uint32_t a = anc[tgt];
        bool seen = false;
        for (uint32_t j = 0; j < t && !seen; ++j) seen = anc[(uint32_t)(v[j] >> 32)] == a;
        distinct += seen ? 0u : 1u;
    }
    const uint32_t k = (sz != 0 && distinct <= maxAmbig) ? 1u : 0u;
    keep[i] = k; keepSize[i] = k ? sz : 0u;
}
__global__ __launch_bounds__(256) void ambig_compact_kernel(const uint32_t* __restrict__ keys, const uint8_t* __restrict__ sizes,
                                                            const uint64_t* __restrict__ vals, const uint64_t* __restrict__ voff,
                                                            uint32_t nkeys, const uint32_t* __restrict__ keep,
                                                            const uint64_t* __restrict__ kpos, const uint64_t* __restrict__ nvoff,
                                                            uint32_t* __restrict__ okeys, uint8_t* __restrict__ osizes,
                                                            uint64_t* __restrict__ ovals)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nkeys || !keep[i]) return;
    const uint64_t at = kpos[i];
    okeys[at] = keys[i]; osizes[at] = sizes[i];
    const uint64_t* src = vals + voff[i];
    uint64_t* dst = ovals + nvoff[i];
    for (uint32_t t = 0, sz = sizes[i]; t < sz; ++t) dst[t] = src[t];
}

__global__ __launch_bounds__(256) void ambig_offsets_kernel(const uint32_t* __restrict__ keep, const uint64_t* __restrict__ kpos,
                                                            const uint64_t* __restrict__ nvoff, uint32_t nkeys, uint64_t nvNew, uint64_t nkNew,
                                                            uint64_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) out[nkNew] = nvNew;
    if (i < nkeys && keep[i]) out[kpos[i]] = nvoff[i];
}

int grow_pairs(mc_builder* b, uint64_t need, bool exact = false)
{
    if (need <= b->cap) return MC_OK;
    uint64_t ncap = exact ? need : std::max<uint64_t>(need + need / 2, 1u << 20);
    uint32_t* nk = nullptr; uint64_t* nv = nullptr;
    B_TRY(b, big_malloc((void**)&nk, ncap * 4));
    if (big_malloc((void**)&nv, ncap * 8) != hipSuccess) { (void)big_free(nk); b->err = "out of device memory for (feature, location) pairs"; return MC_ERR_NOMEM; }
    if (b->npairs) {
        B_TRY(b, hipMemcpyAsync(nk, b->dkeys, b->npairs * 4, hipMemcpyDeviceToDevice, b->st));
        B_TRY(b, hipMemcpyAsync(nv, b->dvals, b->npairs * 8, hipMemcpyDeviceToDevice, b->st));
        B_TRY(b, hipStreamSynchronize(b->st));
    }
    if (b->dkeys) (void)big_free(b->dkeys);
    if (b->dvals) (void)big_free(b->dvals);
    b->dkeys = nk; b->dvals = nv; b->cap = ncap;
    return MC_OK;
}

// scratch of one flush: freed on every way out
struct FlushBufs {
    std::vector<void*> p;
    // (big_malloc / big_free: while a table is being built shard after shard the scratch of one shard's sort is the next one's, devcache.h)
    ~FlushBufs() { for (void* q : p) if (q) (void)big_free(q); }
    void drop(void* q) { for (auto& x : p) if (x == q && q) { (void)big_free(q); x = nullptr; } }
    template <class T> hipError_t get(T** out, size_t bytes) { void* q = nullptr; hipError_t e = big_malloc(&q, bytes ? bytes : 16); if (e == hipSuccess) { p.push_back(q); *out = (T*)q; } return e; }
};

int flush(mc_builder* b)
{
    const uint32_t nq = (uint32_t)(b->hqinfo.size() / 4);
    if (nq == 0) return MC_OK;
    const bool fromDevice = b->devBase != nullptr;
    const uint64_t nchars = fromDevice ? b->devChars : b->hseq.size();
    const SketchParams sp = b->sp;
    const uint64_t maxWindows = nchars / sp.stride + 4ull * nq + 1;
    FlushBufs fb;
    uint8_t* dseq = nullptr; uint32_t *dq = nullptr, *dtgt = nullptr, *dfirst = nullptr, *dwc = nullptr, *dwo = nullptr, *dfeat = nullptr;
    uint32_t *dflag = nullptr, *dcnt = nullptr, *dpos = nullptr; void* dscan = nullptr;
    if (!fromDevice) {
        B_TRY(b, fb.get(&dseq, nchars + 16));
        B_TRY(b, hipMemsetAsync(dseq + nchars, 0, 16, b->st));
        B_TRY(b, hipMemcpyAsync(dseq, b->hseq.data(), nchars, hipMemcpyHostToDevice, b->st));
    }
    B_TRY(b, fb.get(&dq, (size_t)nq * 16));
    B_TRY(b, fb.get(&dtgt, (size_t)nq * 4));
    B_TRY(b, fb.get(&dfirst, (size_t)nq * 4));
    B_TRY(b, fb.get(&dwc, (size_t)(nq + 1) * 4));
    B_TRY(b, fb.get(&dwo, (size_t)(nq + 2) * 4));
    B_TRY(b, fb.get(&dfeat, (size_t)maxWindows * sp.s * 4));
    B_TRY(b, fb.get(&dflag, (size_t)(nq + 1) * 4));
    B_TRY(b, fb.get(&dcnt, (size_t)(nq + 1) * 4));
    B_TRY(b, fb.get(&dpos, (size_t)(nq + 2) * 4));
    B_TRY(b, fb.get(&dscan, scan_tmp_bytes(nq + 1)));
    B_TRY(b, hipMemcpyAsync(dq, b->hqinfo.data(), (size_t)nq * 16, hipMemcpyHostToDevice, b->st));
    B_TRY(b, hipMemcpyAsync(dtgt, b->hqtgt.data(), (size_t)nq * 4, hipMemcpyHostToDevice, b->st));
    B_TRY(b, hipMemcpyAsync(dfirst, b->hqfirst.data(), (size_t)nq * 4, hipMemcpyHostToDevice, b->st));

    BatchView bv{fromDevice ? b->devBase : dseq, dq, nullptr, 1, nq};
    Workspace ws{};
    ws.winCount = dwc; ws.winOff = dwo; ws.features = dfeat; ws.qflag = dflag; ws.scanTmp = dscan;
    launch_plan(bv, sp, dwc, b->st);
    launch_scan_u32(dwc, 1, nq, dwo, nullptr, dscan, b->st);
    launch_build_sketch(bv, sp, ws, b->st);
    const uint32_t shardIdx = b->cfg.key_shard_index, shardCnt = std::max<uint32_t>(b->cfg.key_shard_count, 1);
    hipLaunchKernelGGL(own_count_kernel, dim3((nq + 3) / 4), dim3(256), 0, b->st, dfeat, dwo, nq, sp.s, shardIdx, shardCnt, dcnt);
    launch_scan_u32(dcnt, 1, nq, dpos, nullptr, dscan, b->st);
    uint32_t W = 0, kept = 0;
    B_TRY(b, hipMemcpyAsync(&W, dwo + nq, 4, hipMemcpyDeviceToHost, b->st));
    B_TRY(b, hipMemcpyAsync(&kept, dpos + nq, 4, hipMemcpyDeviceToHost, b->st));
    B_TRY(b, hipStreamSynchronize(b->st));
    if ((uint64_t)W > maxWindows) { b->err = "flush: more windows than planned"; return MC_ERR_STATE; }
    int rc = grow_pairs(b, b->npairs + kept);
    if (rc) return rc;
    if (kept) hipLaunchKernelGGL(own_emit_kernel, dim3((nq + 3) / 4), dim3(256), 0, b->st, dfeat, dwo, dtgt, dfirst, nq, sp.s, shardIdx, shardCnt, dpos,
                                 b->dkeys + b->npairs, b->dvals + b->npairs);
    B_TRY(b, hipGetLastError());
    B_TRY(b, hipStreamSynchronize(b->st));
    b->npairs += kept;
    b->hseq.clear(); b->hqinfo.clear(); b->hqtgt.clear(); b->hqfirst.clear();
    b->devBase = nullptr; b->devChars = 0;
    return MC_OK;
}

// file output: the first failed write is remembered, the caller checks once per file (and removes what it began)
struct OutFile {
    FILE* f = nullptr; bool bad = false; std::string name;
    explicit OutFile(const std::string& n) : f(std::fopen(n.c_str(), "wb")), name(n) {}
    ~OutFile() { if (f) std::fclose(f); }
    bool close() { if (f) { if (std::fflush(f) != 0 || std::fclose(f) != 0) bad = true; f = nullptr; } return !bad; }
};
void wr(OutFile& o, const void* p, size_t n) { if (n && std::fwrite(p, 1, n, o.f) != n) o.bad = true; }
void wr_str(OutFile& o, const std::string& s) { uint64_t n = s.size(); wr(o, &n, 8); if (n) wr(o, s.data(), n); }

}  // namespace

extern "C" {

int mc_build_begin(const mc_config* cfg, mc_builder** out)
{
    if (!cfg || !out) return MC_ERR_INVALID;
    *out = nullptr;
    if (cfg->kmerlen < 1 || cfg->kmerlen > 16 || cfg->sketchlen < 1 || cfg->sketchlen > kMaxSketch || cfg->winlen < cfg->kmerlen ||
        cfg->winlen > kMaxWinLen || cfg->winstride < 1 || (cfg->target_id_bytes != 2 && cfg->target_id_bytes != 4)) {
        set_global_error("mc_build_begin: unsupported sketching parameters");
        return MC_ERR_UNSUPPORTED;
    }
    if (cfg->key_shard_count > 1 && cfg->key_shard_index >= cfg->key_shard_count) { set_global_error("mc_build_begin: key_shard_index out of range"); return MC_ERR_INVALID; }
    if (build_record_windows() != kChunkWindows) { set_global_error("mc_build_begin: chunk records and the lane sketcher disagree about their size"); return MC_ERR_STATE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device >= ndev) {
        set_global_error("no usable HIP device (this library has no CPU fallback)");
        return MC_ERR_HIP;
    }
    auto* b = new mc_builder;
    b->cfg = *cfg;
    b->sp = SketchParams{cfg->kmerlen, cfg->sketchlen, cfg->winlen, cfg->winstride};
    b->maxLocs = cfg->max_locations_per_feature ? std::min<uint32_t>(cfg->max_locations_per_feature, 254) : 254;
    b->rmOver = cfg->remove_overpopulated != 0 && b->maxLocs > 1;      // buckets of one location always stay
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking) != hipSuccess) {
        delete b;
        set_global_error("cannot create HIP stream");
        return MC_ERR_HIP;
    }
    *out = b;
    return MC_OK;
}

int mc_build_add_target(mc_builder* b, const char* seq, uint64_t len, const char* name, int64_t parentTaxid, const char* filename)
{
    return mc_build_add_target_src(b, seq, len, name, parentTaxid, filename, 0);
}

int mc_build_set_parent(mc_builder* b, uint64_t target, int64_t parentTaxid)
{
    if (!b || target >= b->targets.size()) return MC_ERR_INVALID;
    b->targets[target].parent = parentTaxid < 1 ? 0 : parentTaxid;
    return MC_OK;
}

int mc_build_target_windows(const mc_builder* b, uint64_t target, uint64_t* windows)
{
    if (!b || !windows || target >= b->targets.size()) return MC_ERR_INVALID;
    *windows = b->targets[target].windows;
    return MC_OK;
}

// the chunk records of one target: window-aligned pieces of <= kChunkWindows windows, all but the last full windows only (tail
// suppressed); host sources are copied into the staging buffer, device sources stay where they are
static int add_target_impl(mc_builder* b, const uint8_t* seq, bool onDevice, uint64_t len, const char* name, int64_t parentTaxid,
                           const char* filename, uint64_t fileIndex)
{
    if (!b || (!seq && len)) return MC_ERR_INVALID;
    if (b->finished) { b->err = "builder already finished"; return MC_ERR_STATE; }
    const uint64_t maxTargets = b->cfg.target_id_bytes == 2 ? 0xFFFFull : 0xFFFFFFFFull;   // database.hpp:340-343
    if (b->targets.size() >= maxTargets) { b->err = "target count limit exceeded"; return MC_ERR_UNSUPPORTED; }
    if (len >= (1ull << 32) - 16) { b->err = "target sequence too long for one call"; return MC_ERR_UNSUPPORTED; }
    B_TRY(b, hipSetDevice(b->cfg.device));
    if (onDevice) {
        if (reinterpret_cast<uintptr_t>(seq) & 3u) { b->err = "mc_build_add_target_device: sequence must start 4-byte aligned"; return MC_ERR_INVALID; }
        // one flush covers one 32-bit offset range of device memory; host and device sources are not mixed in a flush
        const bool fits = b->devBase && seq >= b->devBase && (uint64_t)(seq - b->devBase) + len + 16 < 0xFFFFFFF0ull && b->devChars < kFlushCharsDevice;
        if (!b->hseq.empty() || (b->devBase && !fits)) { int rc = flush(b); if (rc) return rc; }
        if (!b->devBase) b->devBase = seq;
    } else if (b->devBase) { int rc = flush(b); if (rc) return rc; }
    const SketchParams sp = b->sp;
    const uint32_t tgt = (uint32_t)b->targets.size();
    const uint32_t L = (uint32_t)len;
    const uint32_t total = windows_of_host(L, sp);
    uint32_t firstWin = 0;
    uint64_t pos = 0;
    const uint32_t fullWins = L > sp.w ? (L - sp.w) / sp.stride + 1 : 0;
    while (true) {
        const bool last = fullWins <= firstWin + kChunkWindows;
        uint32_t clen;
        if (last) clen = (uint32_t)(L - pos);
        else clen = (kChunkWindows - 1) * sp.stride + sp.w;
        uint64_t off;
        if (onDevice) off = (uint64_t)(seq - b->devBase) + pos;
        else {
            off = b->hseq.size();
            if (off + clen + 8 > 0xFFFFFFF0ull) { int rc = flush(b); if (rc) return rc; continue; }
            b->hseq.insert(b->hseq.end(), seq + pos, seq + pos + clen);
            b->hseq.resize((b->hseq.size() + 3) / 4 * 4, 0);
        }
        b->hqinfo.push_back((uint32_t)off); b->hqinfo.push_back(clen); b->hqinfo.push_back((uint32_t)off);
        b->hqinfo.push_back(last ? 0u : kNoTail);
        b->hqtgt.push_back(tgt); b->hqfirst.push_back(firstWin);
        if (onDevice) b->devChars += clen;
        else if (b->hseq.size() >= kFlushChars) { int rc = flush(b); if (rc) return rc; }
        if (last) break;
        firstWin += kChunkWindows;
        pos += (uint64_t)kChunkWindows * sp.stride;
    }
    TargetRec r;
    r.name = name ? name : ""; r.filename = filename ? filename : ""; r.parent = parentTaxid < 1 ? 0 : parentTaxid; r.windows = total; r.fileIndex = fileIndex;
    b->targets.push_back(std::move(r));
    return MC_OK;
}

int mc_build_add_target_src(mc_builder* b, const char* seq, uint64_t len, const char* name, int64_t parentTaxid, const char* filename,
                            uint64_t fileIndex)
{
    return add_target_impl(b, reinterpret_cast<const uint8_t*>(seq), false, len, name, parentTaxid, filename, fileIndex);
}

// a target whose characters already are in device memory (4-byte aligned start, 16 readable bytes behind the last character): no
// host staging, no copy.  The memory must stay as it is until mc_build_flush / mc_build_finish returns.
int mc_build_add_target_device(mc_builder* b, const void* dseq, uint64_t len, const char* name, int64_t parentTaxid, const char* filename,
                               uint64_t fileIndex)
{
    return add_target_impl(b, static_cast<const uint8_t*>(dseq), true, len, name, parentTaxid, filename, fileIndex);
}

// sketches everything staged so far; afterwards the sources of mc_build_add_target_device calls may be reused
int mc_build_flush(mc_builder* b)
{
    if (!b) return MC_ERR_INVALID;
    if (b->finished) return MC_OK;
    B_TRY(b, hipSetDevice(b->cfg.device));
    return flush(b);
}

// room for this many (feature, location) pairs up front: a builder that grows step by step holds two copies while it grows
int mc_build_reserve(mc_builder* b, uint64_t pairs)
{
    if (!b) return MC_ERR_INVALID;
    if (b->finished) { b->err = "builder already finished"; return MC_ERR_STATE; }
    B_TRY(b, hipSetDevice(b->cfg.device));
    return grow_pairs(b, pairs, true);
}

// modify mode (mode_build.cpp:74-88): the targets and location lists of an existing database come first, new targets after them
int mc_build_add_existing_target(mc_builder* b, const char* name, int64_t parentTaxid, const char* filename, uint64_t fileIndex, uint64_t windows)
{
    if (!b) return MC_ERR_INVALID;
    if (b->finished) { b->err = "builder is finished"; return MC_ERR_STATE; }
    // (may stand between new targets: a target's number is its place in the order of the add calls -- one PART of a partitioned database
    // names the other parts' targets this way; it is the existing LOCATION LISTS that must precede everything sketched, mc_build_add_locations)
    TargetRec r;
    r.name = name ? name : ""; r.filename = filename ? filename : ""; r.parent = parentTaxid < 1 ? 0 : parentTaxid;
    r.windows = windows; r.fileIndex = fileIndex; r.existing = true;
    b->targets.push_back(std::move(r));
    return MC_OK;
}

// one batch of a .cache file (hash_multimap.hpp:1037-1082): keys, bucket sizes, packed {u32 window, target id} values.  The lists keep
// their order and stand before everything sketched later, as in a table that was read from the file and then inserted into.
int mc_build_add_locations(mc_builder* b, const uint32_t* keys, const uint8_t* sizes, const void* values, uint64_t nkeys, uint32_t targetBytes)
{
    if (!b || (nkeys && (!keys || !sizes || !values)) || (targetBytes != 2 && targetBytes != 4)) return MC_ERR_INVALID;
    if (b->finished) { b->err = "builder is finished"; return MC_ERR_STATE; }
    if (!b->targets.empty() && !b->targets.back().existing) { b->err = "existing location lists go in before new targets"; return MC_ERR_STATE; }
    std::vector<uint32_t> hk; std::vector<uint64_t> hv;
    const uint8_t* v = static_cast<const uint8_t*>(values);
    const uint32_t shards = std::max<uint32_t>(b->cfg.key_shard_count, 1);
    for (uint64_t i = 0; i < nkeys; ++i) {
        const bool own = shards == 1 || key_owner(keys[i], shards) == b->cfg.key_shard_index;
        for (uint32_t t = 0; t < sizes[i]; ++t, v += 4 + targetBytes) {
            if (!own) continue;
            uint32_t win, tgt;
            std::memcpy(&win, v, 4);
            if (targetBytes == 2) { uint16_t t16; std::memcpy(&t16, v + 4, 2); tgt = t16; } else std::memcpy(&tgt, v + 4, 4);
            if (tgt >= b->targets.size()) { b->err = "location of a target the builder does not know"; return MC_ERR_INVALID; }
            hk.push_back(keys[i]); hv.push_back(((uint64_t)tgt << 32) | win);
        }
    }
    if (hk.empty()) return MC_OK;
    B_TRY(b, hipSetDevice(b->cfg.device));
    int rc = grow_pairs(b, b->npairs + hk.size());
    if (rc) return rc;
    B_TRY(b, hipMemcpyAsync(b->dkeys + b->npairs, hk.data(), hk.size() * 4, hipMemcpyHostToDevice, b->st));
    B_TRY(b, hipMemcpyAsync(b->dvals + b->npairs, hv.data(), hv.size() * 8, hipMemcpyHostToDevice, b->st));
    B_TRY(b, hipStreamSynchronize(b->st));
    b->npairs += hk.size();
    return MC_OK;
}

// sort + run-length encode + truncate: split off so that every scratch buffer is released on every way out (FlushBufs)
static int finish_sorted(mc_builder* b)
{
    const uint64_t n = b->npairs;
    hipStream_t st = b->st;
    FlushBufs fb;
    uint32_t* k2 = nullptr; uint64_t* v2 = nullptr; void* tmp = nullptr; size_t tmpBytes = 0;
    B_TRY(b, fb.get(&k2, n * 4));
    B_TRY(b, fb.get(&v2, n * 8));
    B_TRY(b, rocprim::radix_sort_pairs(nullptr, tmpBytes, b->dkeys, k2, b->dvals, v2, n, 0, 32, st));
    B_TRY(b, fb.get(&tmp, tmpBytes + 16));
    B_TRY(b, rocprim::radix_sort_pairs(tmp, tmpBytes, b->dkeys, k2, b->dvals, v2, n, 0, 32, st));
    B_TRY(b, hipStreamSynchronize(st));
    fb.drop(tmp); tmp = nullptr;
    (void)big_free(b->dkeys); (void)big_free(b->dvals);
    b->dkeys = nullptr; b->dvals = nullptr; b->cap = 0; b->npairs = 0;
    b->failed = true;                      // from here on the pairs are gone: an error leaves a builder that cannot be finished again
    // runs of equal features, counted first so that the run arrays are as long as the runs (at RefSeq scale a feature has ~35 locations)
    unsigned long long* dcount = nullptr;
    B_TRY(b, fb.get(&dcount, 16));
    B_TRY(b, hipMemsetAsync(dcount, 0, 16, st));
    hipLaunchKernelGGL(count_runs_kernel, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 65536)), dim3(256), 0, st, k2, n, dcount);
    unsigned long long nrunsAll = 0;
    B_TRY(b, hipMemcpyAsync(&nrunsAll, dcount, 8, hipMemcpyDeviceToHost, st));
    B_TRY(b, hipStreamSynchronize(st));
    uint32_t *uniq = nullptr, *counts = nullptr, *dnruns = nullptr;
    B_TRY(b, fb.get(&uniq, (nrunsAll + 1) * 4));
    B_TRY(b, fb.get(&counts, (nrunsAll + 1) * 4));
    B_TRY(b, fb.get(&dnruns, 16));
    B_TRY(b, rocprim::run_length_encode(nullptr, tmpBytes, k2, (unsigned int)n, uniq, counts, dnruns, st));
    B_TRY(b, fb.get(&tmp, tmpBytes + 16));
    B_TRY(b, rocprim::run_length_encode(tmp, tmpBytes, k2, (unsigned int)n, uniq, counts, dnruns, st));
    uint32_t nruns = 0, lastKey = 0;
    B_TRY(b, hipMemcpyAsync(&nruns, dnruns, 4, hipMemcpyDeviceToHost, st));
    B_TRY(b, hipStreamSynchronize(st));
    if (nruns != nrunsAll) { b->err = "mc_build_finish: run count mismatch"; return MC_ERR_STATE; }
    if (nruns) B_TRY(b, hipMemcpy(&lastKey, uniq + nruns - 1, 4, hipMemcpyDeviceToHost));
    if (nruns && lastKey == 0xFFFFFFFFu) --nruns;            // the padding feature can never be stored (hash_dna.hpp:233)
    fb.drop(tmp); fb.drop(k2); fb.drop(dnruns);
    uint32_t* keep32 = nullptr; uint64_t* runOff = nullptr; void* scanTmp = nullptr;
    B_TRY(b, big_malloc((void**)&b->rS, (size_t)nruns + 16));
    B_TRY(b, fb.get(&keep32, ((size_t)nruns + 1) * 4));
    B_TRY(b, fb.get(&runOff, ((size_t)nruns + 2) * 8));
    B_TRY(b, big_malloc((void**)&b->rVoff, ((size_t)nruns + 2) * 8));
    B_TRY(b, fb.get(&scanTmp, scan_tmp_bytes(nruns + 1)));
    if (nruns) hipLaunchKernelGGL(keep_sizes_kernel, dim3((nruns + 255) / 256), dim3(256), 0, st, counts, nruns, b->maxLocs, b->rS, keep32);
    launch_scan_u32(counts, 1, nruns, nullptr, runOff, scanTmp, st);
    launch_scan_u32(keep32, 1, nruns, nullptr, b->rVoff, scanTmp, st);
    uint64_t nvals = 0;
    B_TRY(b, hipMemcpyAsync(&nvals, b->rVoff + nruns, 8, hipMemcpyDeviceToHost, st));
    B_TRY(b, hipStreamSynchronize(st));
    B_TRY(b, big_malloc((void**)&b->rV, (nvals + 2) * 8));
    if (nruns) hipLaunchKernelGGL(compact_values_kernel, dim3((nruns + 255) / 256), dim3(256), 0, st, v2, runOff, b->rVoff, b->rS, nruns, b->rV);
    B_TRY(b, hipGetLastError());
    B_TRY(b, hipStreamSynchronize(st));
    B_TRY(b, big_malloc((void**)&b->rK, ((size_t)nruns + 1) * 4));
    B_TRY(b, hipMemcpy(b->rK, uniq, (size_t)nruns * 4, hipMemcpyDeviceToDevice));
    b->nkeys = nruns; b->nvals = nvals;
    b->failed = false;
    return MC_OK;
}

int mc_build_finish(mc_builder* b, mc_ctx** outCtx)
{
    if (!b) return MC_ERR_INVALID;
    if (b->failed) { b->err = "builder failed earlier (" + b->err + "): start a new one"; return MC_ERR_STATE; }
    B_TRY(b, hipSetDevice(b->cfg.device));
    if (!b->finished) {
        int rc = flush(b);
        if (rc) return rc;
        if (b->npairs >= 0xFFFFFFF0ull) { b->err = "more than 2^32 (feature, location) pairs in one builder"; return MC_ERR_UNSUPPORTED; }
        if (b->npairs && (rc = finish_sorted(b))) return rc;
        b->finished = true;
    }
    if (outCtx) {
        mc_builder* one[1] = {b};
        return mc_build_finish_shards(one, 1, outCtx);
    }
    return MC_OK;
}

// what a finished builder's lists take in the location store with every list on lines of its own (kernels.h list_alloc): one number per block
__global__ __launch_bounds__(256) void padded_store_kernel(const uint8_t* __restrict__ sizes, uint64_t n, uint32_t rmOver, unsigned long long* __restrict__ out)
{
    unsigned long long sum = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        uint32_t e = sizes[i];
        if (rmOver && e > rmOver) e = 0;
        sum += list_alloc(e, kListAlign);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(out, sum);
}
static uint64_t padded_store_of(mc_builder* b, uint32_t rmOver)
{
    if (!b->finished || !b->rS || b->nkeys == 0) return 0;
    unsigned long long* d = nullptr; unsigned long long h = 0;
    if (hipMalloc((void**)&d, 8) != hipSuccess) return 0;
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(padded_store_kernel, dim3(1024), dim3(256), 0, 0, b->rS, b->nkeys, rmOver, d);
    if (hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) h = 0;
    (void)hipFree(d);
    return h;
}

// A query table filled from finished builders.  mc_build_table_begin creates the context (table sized for the expected totals),
// mc_build_table_add inserts one finished builder (a whole one or one key shard) straight from its device arrays,
// mc_build_table_end closes the load.  mc_build_finish_shards is the three in a row for builders that all fit next to the table;
// the streaming form lets the caller free every shard's builder before the next one is sketched.
static int build_table_begin(mc_builder* b, uint64_t expectKeys, uint64_t expectValues, uint64_t paddedTotalHint, mc_ctx** outCtx);
int mc_build_table_begin(mc_builder* b, uint64_t expectKeys, uint64_t expectValues, mc_ctx** outCtx) { return build_table_begin(b, expectKeys, expectValues, 0, outCtx); }
// paddedTotalHint: the exact padded store of all builders that will be added (mc_build_finish_shards knows them all), 0: estimated from `b`
static int build_table_begin(mc_builder* b, uint64_t expectKeys, uint64_t expectValues, uint64_t paddedTotalHint, mc_ctx** outCtx)
{
    if (!b || !outCtx) return MC_ERR_INVALID;
    *outCtx = nullptr;
    if (expectKeys == 0 || expectValues == 0) {
        // not given: the first shard times the shard count (keys are dealt out by a hash: shards differ by fractions of a percent)
        if (!b->finished) { b->err = "mc_build_table_begin: expected totals missing and the builder is not finished"; return MC_ERR_STATE; }
        const uint64_t n = std::max<uint32_t>(b->cfg.key_shard_count, 1);
        if (!expectKeys) expectKeys = n == 1 ? b->nkeys : b->nkeys * n + b->nkeys * n / 64 + (1u << 16);
        if (!expectValues) expectValues = n == 1 ? b->nvals : b->nvals * n + b->nvals * n / 32 + (1u << 20);
    }
    B_TRY(b, hipSetDevice(b->cfg.device));
    mc_config qc = b->cfg;
    qc.max_locations_per_feature = 0; qc.num_parts = 1;
    qc.remove_overpopulated = b->rmOver ? b->maxLocs - 1 : 0;   // applied while the table is filled (table_build.hip)
    qc.target_id_bytes = 4;                                    // values are handed over as {u32 win, u32 tgt}
    qc.key_shard_index = 0; qc.key_shard_count = 1;            // the table holds all shards
    mc_ctx* ctx = nullptr;
    int rc = mc_create(&qc, &ctx);
    if (rc) { b->err = mc_last_error(nullptr); return rc; }
    ctx->targetCount = b->targets.size();
    ctx->maxLocs = b->maxLocs;
    {
        // every location is (target < targets, window < that target's windows): the table may store them as global window numbers,
        // 4 bytes each (DeviceTable::values32)
        std::vector<uint32_t> windows;
        windows.reserve(b->targets.size());
        bool fits = !b->targets.empty();
        for (const auto& t : b->targets) { fits = fits && t.windows <= 0xFFFFFFF0ull; windows.push_back((uint32_t)t.windows); }
        if (fits) (void)mc_load_target_windows(ctx, windows.data(), windows.size());
    }
    rc = mc_load_begin(ctx, 0, expectKeys, expectValues);
    if (rc) { b->err = mc_last_error(ctx); mc_destroy(ctx); return rc; }
    {
        // list alignment: this builder's lists padded, scaled to the expected total (the shards' lists are alike: keys are dealt out by a hash)
        const uint64_t mine = padded_store_of(b, qc.remove_overpopulated);
        if (mine && b->nvals) mcamd::announce_store(ctx, paddedTotalHint ? paddedTotalHint : (uint64_t)((double)mine / (double)b->nvals * (double)expectValues * 1.02) + (1u << 20));
    }
    *outCtx = ctx;
    big_cache_hold(+1);                                            // until mc_build_table_end (or mc_destroy): the builders freed meanwhile leave their large buffers to the next ones
    ctx->buildHold = true;
    return MC_OK;
}

int mc_build_table_add(mc_ctx* ctx, mc_builder* b)
{
    if (!ctx || !b) return MC_ERR_INVALID;
    if (!b->finished) { int rc = mc_build_finish(b, nullptr); if (rc) return rc; }
    if (b->targets.size() != ctx->targetCount) { b->err = "mc_build_table_add: builder holds other targets than the table"; return MC_ERR_INVALID; }
    B_TRY(b, hipSetDevice(b->cfg.device));
    // device arrays go straight into the table builder, in chunks whose value count stays below 2^32
    const uint64_t chunk = 1ull << 22;
    uint64_t vbeg = 0;
    for (uint64_t i = 0; i < b->nkeys; i += chunk) {
        const uint64_t nb = std::min<uint64_t>(chunk, b->nkeys - i);
        uint64_t vend = 0;
        B_TRY(b, hipMemcpy(&vend, b->rVoff + i + nb, 8, hipMemcpyDeviceToHost));
        const int rc = load_chunk_device(ctx, b->rK + i, b->rS + i, reinterpret_cast<const uint8_t*>(b->rV + vbeg), (uint32_t)nb, vend - vbeg);
        if (rc) { b->err = mc_last_error(ctx); return rc; }
        vbeg = vend;
    }
    return MC_OK;
}

int mc_build_table_end(mc_ctx* ctx)
{
    if (!ctx) return MC_ERR_INVALID;
    if (ctx->buildHold) { big_cache_hold(-1); ctx->buildHold = false; }   // (the shards' scratch, kept since mc_build_table_begin, goes back to the device)
    return mc_load_end(ctx, 0);
}

// One query table from the results of several builders that sketched the SAME targets with different key shards
// (cfg.key_shard_index = 0 .. n-1 of key_shard_count = n): a table beyond the 2^32 (feature, location) pairs one sort can take is
// built shard after shard.  With n = 1 this is the loading half of mc_build_finish.
int mc_build_finish_shards(mc_builder** bs, uint32_t n, mc_ctx** outCtx)
{
    if (!bs || !n || !outCtx || !bs[0]) return MC_ERR_INVALID;
    mc_builder* b = bs[0];
    *outCtx = nullptr;
    uint64_t nkeys = 0, nvals = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (!bs[i]) return MC_ERR_INVALID;
        if (!bs[i]->finished) { int rc = mc_build_finish(bs[i], nullptr); if (rc) { b->err = bs[i]->err; return rc; } }
        if (bs[i]->targets.size() != b->targets.size() || bs[i]->cfg.key_shard_count != (n > 1 ? n : bs[i]->cfg.key_shard_count) ||
            (n > 1 && bs[i]->cfg.key_shard_index != i)) { b->err = "mc_build_finish_shards: builders do not form one key-sharded set"; return MC_ERR_INVALID; }
        nkeys += bs[i]->nkeys; nvals += bs[i]->nvals;
    }
    uint64_t paddedAll = 0;
    for (uint32_t i = 0; i < n; ++i) paddedAll += padded_store_of(bs[i], b->rmOver ? b->maxLocs - 1 : 0);
    mc_ctx* ctx = nullptr;
    int rc = build_table_begin(b, std::max<uint64_t>(nkeys, 1), std::max<uint64_t>(nvals, 1), paddedAll + 64, &ctx);
    if (rc) return rc;
    for (uint32_t s = 0; !rc && s < n; ++s) { rc = mc_build_table_add(ctx, bs[s]); if (rc) b->err = bs[s]->err; }
    if (!rc) { rc = mc_build_table_end(ctx); if (rc) b->err = mc_last_error(ctx); }
    if (rc) { mc_destroy(ctx); return rc; }
    *outCtx = ctx;
    return MC_OK;
}

int mc_build_counts(const mc_builder* b, uint64_t* keys, uint64_t* values)
{
    if (!b || !b->finished) return MC_ERR_STATE;
    if (keys) *keys = b->nkeys;
    if (values) *values = b->nvals;
    return MC_OK;
}

int mc_build_remove_ambiguous(mc_builder* b, const uint32_t* ancestorOfTarget, uint64_t numTargets, uint32_t maxAmbig, uint64_t* removed)
{
    if (!b || !ancestorOfTarget) return MC_ERR_INVALID;
    if (!b->finished) { b->err = "mc_build_remove_ambiguous: call mc_build_finish first"; return MC_ERR_STATE; }
    if (numTargets != b->targets.size()) { b->err = "mc_build_remove_ambiguous: one ancestor per target expected"; return MC_ERR_INVALID; }
    if (removed) *removed = 0;
    if (maxAmbig == 0) maxAmbig = 1;                                           // host_hashmap.hpp:505
    const uint32_t nk = (uint32_t)b->nkeys;
    if (!nk) return MC_OK;
    B_TRY(b, hipSetDevice(b->cfg.device));
    hipStream_t st = b->st;
    uint32_t *danc = nullptr, *keep = nullptr, *keepSize = nullptr, *oK = nullptr; uint8_t* oS = nullptr;
    uint64_t *kpos = nullptr, *nvoff = nullptr, *oV = nullptr; void* scanTmp = nullptr;
    B_TRY(b, hipMalloc((void**)&danc, (numTargets + 1) * 4));
    B_TRY(b, hipMemcpyAsync(danc, ancestorOfTarget, numTargets * 4, hipMemcpyHostToDevice, st));
    B_TRY(b, hipMalloc((void**)&keep, ((size_t)nk + 1) * 4));
    B_TRY(b, hipMalloc((void**)&keepSize, ((size_t)nk + 1) * 4));
    B_TRY(b, hipMalloc((void**)&kpos, ((size_t)nk + 2) * 8));
    B_TRY(b, hipMalloc((void**)&nvoff, ((size_t)nk + 2) * 8));
    B_TRY(b, hipMalloc(&scanTmp, scan_tmp_bytes(nk + 1)));
    hipLaunchKernelGGL(ambig_flags_kernel, dim3((nk + 255) / 256), dim3(256), 0, st, b->rV, b->rVoff, b->rS, nk, danc, maxAmbig, keep, keepSize);
    launch_scan_u32(keep, 1, nk, nullptr, kpos, scanTmp, st);
    launch_scan_u32(keepSize, 1, nk, nullptr, nvoff, scanTmp, st);
    uint64_t nkNew = 0, nvNew = 0;
    B_TRY(b, hipMemcpyAsync(&nkNew, kpos + nk, 8, hipMemcpyDeviceToHost, st));
    B_TRY(b, hipMemcpyAsync(&nvNew, nvoff + nk, 8, hipMemcpyDeviceToHost, st));
    B_TRY(b, hipStreamSynchronize(st));
    if (nkNew != nk) {
        B_TRY(b, hipMalloc((void**)&oK, (nkNew + 1) * 4));
        B_TRY(b, hipMalloc((void**)&oS, nkNew + 16));
        B_TRY(b, hipMalloc((void**)&oV, (nvNew + 2) * 8));
        hipLaunchKernelGGL(ambig_compact_kernel, dim3((nk + 255) / 256), dim3(256), 0, st, b->rK, b->rS, b->rV, b->rVoff, nk, keep, kpos, nvoff, oK, oS, oV);
        B_TRY(b, hipGetLastError());
        B_TRY(b, hipStreamSynchronize(st));
        (void)big_free(b->rK); (void)big_free(b->rS); (void)big_free(b->rV);
        b->rK = oK; b->rS = oS; b->rV = oV;
        // rVoff[j] for the kept features = nvoff at their old places, moved to the front
        uint64_t* oOff = nullptr;
        B_TRY(b, hipMalloc((void**)&oOff, (nkNew + 2) * 8));
        hipLaunchKernelGGL(ambig_offsets_kernel, dim3((nk + 255) / 256), dim3(256), 0, st, keep, kpos, nvoff, nk, nvNew, nkNew, oOff);
        B_TRY(b, hipGetLastError());
        B_TRY(b, hipStreamSynchronize(st));
        (void)big_free(b->rVoff);
        b->rVoff = oOff;
        if (removed) *removed = nk - nkNew;
        b->nkeys = nkNew; b->nvals = nvNew;
    }
    for (void* p : {(void*)danc, (void*)keep, (void*)keepSize, (void*)kpos, (void*)nvoff, scanTmp}) (void)hipFree(p);
    return MC_OK;
}

int mc_build_set_query_config(mc_builder* b, const mc_config* q)
{
    if (!b || !q) return MC_ERR_INVALID;
    b->cfg.max_candidates = q->max_candidates; b->cfg.max_load_factor = q->max_load_factor;
    b->cfg.num_slots = q->num_slots; b->cfg.slot_max_queries = q->slot_max_queries; b->cfg.slot_max_chars = q->slot_max_chars;
    b->cfg.copy_allhits = q->copy_allhits;
    return MC_OK;
}

int mc_build_write(mc_builder* b, const char* name, const mc_taxon_rec* taxa, uint64_t ntaxa)
{
    if (!b) return MC_ERR_INVALID;
    if (b->cfg.key_shard_count > 1) { b->err = "mc_build_write: a key-sharded builder holds only part of the database (mc_build_write_shards)"; return MC_ERR_UNSUPPORTED; }
    mc_builder* one[1] = {b};
    return mc_build_write_shards(one, 1, name, taxa, ntaxa);
}

// <name>.meta + <name>.cache0, written shard by shard: the shards' features follow each other in the file's batch stream (any order of
// keys is a valid file: the reader inserts key by key, hash_multimap.hpp:970-1030), so a builder can be written and freed before the next
// one is sketched -- a RefSeq-scale database (2 x 10^10 locations) never has to exist as a whole next to its files.
struct mc_db_writer {
    std::string name, err;
    OutFile* f = nullptr;
    uint32_t tb = 4, maxLocs = 254, shardCount = 1, nextShard = 0;
    size_t targets = 0;
    uint64_t hdr[3] = {0, 0, 1ull << 20};
    std::vector<uint32_t> outK; std::vector<uint8_t> outS, packed;
    bool failed = false;
    ~mc_db_writer() { delete f; }
    void flush_batch()
    {
        if (outK.empty()) return;
        hdr[0] += outK.size();
        wr(*f, outK.data(), outK.size() * 4);
        wr(*f, outS.data(), outS.size());
        wr(*f, packed.data(), packed.size());
        outK.clear(); outS.clear(); packed.clear();
    }
    int give_up(mc_builder* b, const std::string& why)           // no half-written database stays behind
    {
        failed = true;
        if (f) { f->close(); std::remove(f->name.c_str()); }
        std::remove((name + ".meta").c_str());
        err = why;
        if (b) b->err = why;
        return MC_ERR_IO;
    }
};

int mc_build_write_begin(mc_builder* b, const char* name, const mc_taxon_rec* taxa, uint64_t ntaxa, mc_db_writer** out)
{
    if (!b || !name || !out) return MC_ERR_INVALID;
    *out = nullptr;
    if (!b->finished) { b->err = "mc_build_write: call mc_build_finish first"; return MC_ERR_STATE; }
    const uint32_t tb = b->cfg.target_id_bytes;
    {   // .meta  (database.cpp:247-290)
        OutFile f(std::string(name) + ".meta");
        if (!f.f) { b->err = "cannot write .meta"; return MC_ERR_IO; }
        const uint64_t ver = 20200820ull;
        wr(f, &ver, 8);
        const uint8_t wd[7] = {4, (uint8_t)tb, 4, 1, 4, 8, MC_NUM_RANKS};
        wr(f, wd, 7);
        const uint64_t sk[4] = {b->sp.k, b->sp.s, b->sp.w, b->sp.stride};
        wr(f, sk, 32); wr(f, sk, 32);
        const uint64_t ml = b->maxLocs;
        wr(f, &ml, 8);
        if (tb == 2) { uint16_t t = (uint16_t)b->targets.size(); wr(f, &t, 2); } else { uint32_t t = (uint32_t)b->targets.size(); wr(f, &t, 4); }
        const uint32_t parts = 1;
        wr(f, &parts, 4);
        const uint64_t nt = ntaxa + b->targets.size();
        wr(f, &nt, 8);
        for (uint64_t i = 0; i < ntaxa; ++i) {
            wr(f, &taxa[i].id, 8); wr(f, &taxa[i].parent, 8);
            const uint8_t rk = (uint8_t)taxa[i].rank; wr(f, &rk, 1);
            wr_str(f, taxa[i].name ? taxa[i].name : ""); wr_str(f, "");
            const uint64_t z = 0; wr(f, &z, 8); wr(f, &z, 8);
        }
        for (uint64_t t = 0; t < b->targets.size(); ++t) {
            const int64_t id = -(int64_t)t - 1;
            wr(f, &id, 8); wr(f, &b->targets[t].parent, 8);
            const uint8_t rk = 0; wr(f, &rk, 1);                       // rank::Sequence (taxonomy.hpp:453)
            wr_str(f, b->targets[t].name); wr_str(f, b->targets[t].filename);
            wr(f, &b->targets[t].fileIndex, 8); wr(f, &b->targets[t].windows, 8);
        }
        if (!f.close()) { std::remove(f.name.c_str()); b->err = "write error on " + f.name + " (disk full?)"; return MC_ERR_IO; }
    }
    auto* w = new mc_db_writer;
    w->name = name; w->tb = tb; w->maxLocs = b->maxLocs; w->targets = b->targets.size();
    w->shardCount = std::max<uint32_t>(b->cfg.key_shard_count, 1);
    // .cache0  (hash_multimap.hpp:1037-1082): only non-empty buckets are written; the key / value totals of the header follow at the end
    w->f = new OutFile(std::string(name) + ".cache0");
    if (!w->f->f) { std::remove((w->name + ".meta").c_str()); b->err = "cannot write .cache0"; delete w; return MC_ERR_IO; }
    wr(*w->f, w->hdr, 24);
    *out = w;
    return MC_OK;
}

int mc_build_write_add(mc_db_writer* w, mc_builder* c)
{
    if (!w || !c) return MC_ERR_INVALID;
    if (w->failed) return MC_ERR_STATE;
    if (!c->finished) { c->err = "mc_build_write: call mc_build_finish first"; return MC_ERR_STATE; }
    if (c->targets.size() != w->targets || c->cfg.target_id_bytes != w->tb || std::max<uint32_t>(c->cfg.key_shard_count, 1) != w->shardCount ||
        (w->shardCount > 1 && c->cfg.key_shard_index != w->nextShard)) {
        c->err = "mc_build_write: builders do not form one key-sharded set (shards 0 .. n-1 in order, same targets)"; return MC_ERR_INVALID; }
    ++w->nextShard;
    const uint32_t tb = w->tb;
    const uint64_t batch = w->hdr[2];
    std::vector<uint32_t> keys; std::vector<uint8_t> sizes; std::vector<uint64_t> values;
    // one shard at a time on the host, in slices of 2^24 keys; with -remove-overpopulated-features the buckets that reached the limit
    // are gone (remove_features_with_more_locations_than(maxLocs - 1), building.cpp:516-534)
    const uint64_t slice = 1ull << 24;
    uint64_t vbeg = 0;
    for (uint64_t k0 = 0; k0 < c->nkeys; k0 += slice) {
        const uint64_t nk = std::min<uint64_t>(slice, c->nkeys - k0);
        uint64_t vend = 0;
        B_TRY(c, hipSetDevice(c->cfg.device));
        B_TRY(c, hipMemcpy(&vend, c->rVoff + k0 + nk, 8, hipMemcpyDeviceToHost));
        keys.resize(nk); sizes.resize(nk); values.resize(vend - vbeg);
        B_TRY(c, hipMemcpy(keys.data(), c->rK + k0, nk * 4, hipMemcpyDeviceToHost));
        B_TRY(c, hipMemcpy(sizes.data(), c->rS + k0, nk, hipMemcpyDeviceToHost));
        if (vend > vbeg) B_TRY(c, hipMemcpy(values.data(), c->rV + vbeg, (vend - vbeg) * 8, hipMemcpyDeviceToHost));
        vbeg = vend;
        auto kept = [&](uint64_t i) { return !(c->rmOver && sizes[i] > c->maxLocs - 1); };
        // the reader takes batches of exactly 'batch' keys (the last one may be shorter): a batch runs on across slices and shards
        uint64_t voff = 0;
        for (uint64_t i = 0; i < nk; ++i) {
            const uint32_t sz = sizes[i];
            if (kept(i)) {
                w->outK.push_back(keys[i]); w->outS.push_back((uint8_t)sz);
                const size_t at0 = w->packed.size();
                w->packed.resize(at0 + (size_t)sz * (4 + tb));
                for (uint32_t t = 0; t < sz; ++t) {
                    const uint64_t v = values[voff + t];
                    const uint32_t win = (uint32_t)v, tgt = (uint32_t)(v >> 32);
                    uint8_t* at = &w->packed[at0 + (size_t)t * (4 + tb)];
                    std::memcpy(at, &win, 4);
                    if (tb == 2) { uint16_t t16 = (uint16_t)tgt; std::memcpy(at + 4, &t16, 2); }
                    else std::memcpy(at + 4, &tgt, 4);
                }
                w->hdr[1] += sz;
                if (w->outK.size() == batch) { w->flush_batch(); if (w->f->bad) return w->give_up(c, "write error on " + w->f->name + " (disk full?)"); }
            }
            voff += sz;
        }
    }
    return MC_OK;
}

int mc_build_write_end(mc_db_writer* w)
{
    if (!w) return MC_ERR_INVALID;
    int rc = MC_OK;
    if (w->failed) rc = MC_ERR_IO;
    else if (w->shardCount > 1 && w->nextShard != w->shardCount) rc = w->give_up(nullptr, "mc_build_write_end: not every key shard was written");
    else {
        w->flush_batch();
        if (std::fseek(w->f->f, 0, SEEK_SET) != 0) w->f->bad = true;
        wr(*w->f, w->hdr, 24);
        if (!w->f->close()) rc = w->give_up(nullptr, "write error on " + w->f->name + " (disk full?)");
    }
    delete w;
    return rc;
}

int mc_build_write_shards(mc_builder** bs, uint32_t n, const char* name, const mc_taxon_rec* taxa, uint64_t ntaxa)
{
    if (!bs || !n || !bs[0] || !name) return MC_ERR_INVALID;
    mc_builder* b = bs[0];
    for (uint32_t s = 0; s < n; ++s) {
        if (!bs[s]) return MC_ERR_INVALID;
        if (!bs[s]->finished) { b->err = "mc_build_write: call mc_build_finish first"; return MC_ERR_STATE; }
        if (bs[s]->targets.size() != b->targets.size() || (n > 1 && (bs[s]->cfg.key_shard_count != n || bs[s]->cfg.key_shard_index != s))) {
            b->err = "mc_build_write_shards: builders do not form one key-sharded set"; return MC_ERR_INVALID; }
    }
    mc_db_writer* w = nullptr;
    int rc = mc_build_write_begin(b, name, taxa, ntaxa, &w);
    if (rc) return rc;
    for (uint32_t s = 0; s < n && !rc; ++s) { rc = mc_build_write_add(w, bs[s]); if (rc) b->err = bs[s]->err; }
    const int rc2 = mc_build_write_end(w);
    return rc ? rc : rc2;
}

void mc_build_free(mc_builder* b)
{
    if (!b) return;
    (void)hipSetDevice(b->cfg.device);
    if (b->dkeys) (void)big_free(b->dkeys);
    if (b->dvals) (void)big_free(b->dvals);
    for (void* p : {(void*)b->rK, (void*)b->rS, (void*)b->rV, (void*)b->rVoff}) if (p) (void)big_free(p);
    if (b->st) (void)hipStreamDestroy(b->st);
    delete b;
}

const char* mc_build_last_error(const mc_builder* b) { return b ? b->err.c_str() : mc_last_error(nullptr); }

}  // extern "C"
