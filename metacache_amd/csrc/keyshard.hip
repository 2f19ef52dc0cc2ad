// metacache_amd/csrc/keyshard.hip -- Mode K (ONE database part whose features are key-sharded over the GPUs) with 4-byte
// locations on the wire.  What crosses the GPUs is the compact store's own form of a location: the 32-bit GLOBAL WINDOW NUMBER
// gw = gwBase[target] + window (kernels.h DeviceTable) -- every shard numbers the windows of ALL targets the same way, because the
// numbering comes from the targets' window counts (metadata), not from the shard's features.
//
//   shard side   mask_foreign_features : a shard looks up only the features it owns (key_owner): the lookup work of a batch divides
//                                        by the number of shards instead of being repeated on every one of them
//                pack_numbers          : the partial lists mc_query_device(MC_WANT_PARTIAL_HITS) left -> numbers, back to back in read
//                                        order (the piece for the owner of reads [lo, hi) is ONE contiguous range) + per-read counts
//   owner side   owner_entries         : NO union copy -- the receive buffer stands in for the table's location store, the piece of
//                                        every source is one ENTRY of the read (what a found feature's bucket list is in the
//                                        replicated mode); reads the filtered path takes become records of its work list (list 6)
//                decode_union          : the rest (short lists, what the filtered path hands back): pieces -> (target, window) lists
//                                        for the sort of cands_from_hits_kernel
// Replaces the reference's per-part forwarding chain between its GPUs (query_batch.cu:464-527, :638-652, gpu_hashmap.cu:1255-1290).
#include "device_common.h"

#include <algorithm>

namespace mcamd {

__global__ __launch_bounds__(256) void mask_foreign_features_kernel(uint32_t* __restrict__ features, const uint32_t* __restrict__ totalWindows, uint32_t s,
                                                                    uint32_t shardIdx, uint32_t shardCnt)
{
    // four features per lane (the buffer is 16-byte aligned; a tail is taken one by one)
    const uint64_t nfeat = (uint64_t)totalWindows[0] * s;
    const uint64_t stride = (uint64_t)gridDim.x * 256, n4 = nfeat / 4;
    uint4* f4 = reinterpret_cast<uint4*>(features);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        uint4 f = f4[i];
        const uint4 o = f;
        if (f.x != 0xFFFFFFFFu && key_owner(f.x, shardCnt) != shardIdx) f.x = 0xFFFFFFFFu;
        if (f.y != 0xFFFFFFFFu && key_owner(f.y, shardCnt) != shardIdx) f.y = 0xFFFFFFFFu;
        if (f.z != 0xFFFFFFFFu && key_owner(f.z, shardCnt) != shardIdx) f.z = 0xFFFFFFFFu;
        if (f.w != 0xFFFFFFFFu && key_owner(f.w, shardCnt) != shardIdx) f.w = 0xFFFFFFFFu;
        if (f.x != o.x || f.y != o.y || f.z != o.z || f.w != o.w) f4[i] = f;
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nfeat; i += stride) {
        const uint32_t f = features[i];
        if (f != 0xFFFFFFFFu && key_owner(f, shardCnt) != shardIdx) features[i] = 0xFFFFFFFFu;
    }
}
// totalWindows: device word holding the batch's window count (winOff[n]); maxFeat: the buffer's capacity (sizes the grid)
void launch_mask_foreign_features(uint32_t* features, const uint32_t* totalWindows, uint32_t s, uint64_t maxFeat, uint32_t shardIdx, uint32_t shardCnt, hipStream_t st)
{
    if (maxFeat == 0 || shardCnt <= 1) return;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((maxFeat / 4 + 255) / 256 + 1, 256 * 16);
    hipLaunchKernelGGL(mask_foreign_features_kernel, dim3(grid), dim3(256), 0, st, features, totalWindows, s, shardIdx, shardCnt);
}

__global__ __launch_bounds__(256) void pack_numbers_kernel(const uint64_t* __restrict__ hits, uint64_t total, DeviceTable tab, uint32_t* __restrict__ out)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) out[i] = tab.gw_of(hits[i]);
}
__global__ __launch_bounds__(256) void partial_counts_kernel(const uint64_t* __restrict__ hitOff, uint32_t n, uint32_t* __restrict__ counts)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q < n) counts[q] = (uint32_t)(hitOff[q + 1] - hitOff[q]);
}
void launch_pack_numbers(const uint64_t* hits, const uint64_t* hitOff, uint64_t total, uint32_t n, const DeviceTable& tab, uint32_t* numbers, uint32_t* counts,
                         hipStream_t st)
{
    if (n) hipLaunchKernelGGL(partial_counts_kernel, dim3((n + 255) / 256), dim3(256), 0, st, hitOff, n, counts);
    if (total) hipLaunchKernelGGL(pack_numbers_kernel, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 256 * 32)), dim3(256), 0, st, hits, total, tab, numbers);
}

// MC_WANT_PARTIAL_NUMBERS: the reads the wave kernels took (their lists lie in ws.hits as (target, window)) -> numbers; the reads that wait
// for gather_lists_kernel<true> get theirs straight from the table.  Runs after sort_candidates_kernel and before the gather.
__global__ __launch_bounds__(256) void pack_other_reads_kernel(BatchView b, DeviceTable tab, Workspace ws, uint32_t* __restrict__ numbers)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nWaves = gridDim.x * 4, waveId = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (uint32_t base = waveId * 64; base < b.n; base += nWaves * 64) {
        const uint32_t qq = base + lane;
        const uint32_t fl = qq < b.n ? ws.qflag[qq] : kFlagGather;
        uint64_t todo = __ballot(qq < b.n && fl != kFlagGather && fl != kFlagGatherAll && ws.hitOff[qq + 1] != ws.hitOff[qq]);
        while (todo) {
            const uint32_t j = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint64_t at = ws.hitOff[base + j], c = ws.hitOff[base + j + 1] - at;
            for (uint64_t t = lane; t < c; t += 64) numbers[at + t] = tab.gw_of(ws.hits[at + t]);
        }
    }
}
void launch_pack_other_reads(const BatchView& b, const DeviceTable& tab, const Workspace& ws, uint32_t* numbers, uint32_t* counts, hipStream_t st)
{
    if (b.n == 0) return;
    hipLaunchKernelGGL(partial_counts_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, ws.hitOff, b.n, counts);
    hipLaunchKernelGGL(pack_other_reads_kernel, dim3(std::min<uint32_t>((b.n + 255) / 256, 256 * 8)), dim3(256), 0, st, b, tab, ws, numbers);
}

// One lane per read.  counts[s * m + q] numbers of read q came from source s, at srcStart[s * (m + 1) + q] inside that source's block,
// which begins at bases.b[s] of tab.values32 (= the receive buffer).  Entry s of read q: slot q * S + s.
__global__ __launch_bounds__(256) void owner_entries_kernel(BatchView b, DeviceTable tab, Workspace ws, const uint32_t* __restrict__ counts,
                                                            const uint64_t* __restrict__ srcStart, KeyshardBases bases, uint32_t S)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t m = b.n;
    bool take = false, wide = false;
    uint32_t H = 0, mw = 0;
    if (q < m) {
        bool ok = true;
        for (uint32_t s = 0; s < S; ++s) {
            const uint32_t c = counts[(size_t)s * m + q];
            const uint64_t at = bases.b[s] + srcStart[(size_t)s * (m + 1) + q];
            H += c; ok = ok && c <= 0xFFFFu;
            ws.psize[(size_t)q * S + s] = c;
            // (a single location is its own payload in the 8-byte form, as in the table's buckets)
            ws.ppay[(size_t)q * S + s] = c == 1 ? tab.gw_widen(tab.values32[at]) : at;
        }
        mw = b.maxWin ? b.maxWin[q] : b.maxWinUniform;
        take = ok && H > 64u && H <= kMaxHitsPerQuery && mw <= tab.gwGap;
        wide = take && H > kGwSmallH;
        QueryStat qs; qs.hits = H; qs.nfeat = 0; qs.nfound = S; qs.nsteps = 0;
        ws.qstat[q] = qs;
        ws.qflag[q] = take ? kFlagMid : kFlagCands;
        ws.hitScan[q] = take ? 0u : (H <= kMaxHitsPerQuery ? H : 0u);
    }
    const uint64_t mask = __ballot(take);
    if (take) {
        const uint32_t leader = __ffsll((unsigned long long)mask) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&ws.midCount[9], (uint32_t)__popcll(mask));
        base = __shfl(base, leader);
        reinterpret_cast<uint4*>(ws.midList)[(size_t)6 * m + base + __popcll(mask & ((1ull << lane) - 1ull))] = make_uint4(q, q * S, S | (H << 12), mw);
    }
    const uint64_t wmask = __ballot(wide);
    if (wmask && lane == (uint32_t)__ffsll((unsigned long long)wmask) - 1) atomicAdd(&ws.midCount[10], (uint32_t)__popcll(wmask));
}
void launch_owner_entries(const BatchView& b, const DeviceTable& tab, const Workspace& ws, const uint32_t* counts, const uint64_t* srcStart,
                          const KeyshardBases& bases, uint32_t S, hipStream_t st)
{
    if (b.n) hipLaunchKernelGGL(owner_entries_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b, tab, ws, counts, srcStart, bases, S);
}

// One wave per read that is left for the sort (qflag == kFlagCands): its pieces, source after source, as (target << 32 | window)
__global__ __launch_bounds__(256) void decode_union_kernel(BatchView b, DeviceTable tab, Workspace ws, const uint32_t* __restrict__ counts,
                                                           const uint64_t* __restrict__ srcStart, KeyshardBases bases, uint32_t S)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nWaves = gridDim.x * 4, waveId = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t m = b.n;
    for (uint32_t base = waveId * 64; base < m; base += nWaves * 64) {
        const uint32_t qq = base + lane;
        uint64_t todo = __ballot(qq < m && ws.qflag[qq] == kFlagCands && ws.hitScan[qq] != 0u);
        while (todo) {
            const uint32_t j = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t q = base + j;
            uint64_t* dst = ws.hits + ws.hitOff[q];
            for (uint32_t s = 0; s < S; ++s) {
                const uint32_t c = counts[(size_t)s * m + q];
                const uint32_t* src = tab.values32 + bases.b[s] + srcStart[(size_t)s * (m + 1) + q];
                for (uint32_t t = lane; t < c; t += 64) dst[t] = tab.gw_widen(src[t]);
                dst += c;
            }
        }
    }
}
void launch_decode_union(const BatchView& b, const DeviceTable& tab, const Workspace& ws, const uint32_t* counts, const uint64_t* srcStart,
                         const KeyshardBases& bases, uint32_t S, hipStream_t st)
{
    if (b.n) hipLaunchKernelGGL(decode_union_kernel, dim3(std::min<uint32_t>((b.n + 255) / 256, 256 * 8)), dim3(256), 0, st, b, tab, ws, counts, srcStart, bases, S);
}

}  // namespace mcamd
