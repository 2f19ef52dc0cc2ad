// metacache_amd/csrc/table_build.hip -- building the query table ON the GPU from the reference's batch stream
// (keys u32 | sizes u8 | packed values), the MI355X counterpart of gpu_hashmap::deserialize
// (gpu_hashmap.cu:1319-1362, batch layout hash_multimap.hpp:1037-1082).  The host only forwards the file's
// batches; slot claiming, payload placement and the location-list copy run as three small kernels per batch:
//   table_prep    : per key, the size after the load-time modifiers (host_hashmap.hpp:454-495)
//   table_insert  : one lane per key walks the key's probe sequence (kernels.h next_bucket) and claims the first
//                   free slot with a 64-bit CAS on the bucket's four u16 sizes; key and payload are written after
//   table_values  : one lane per FILE value: binary search for its key, decode {win, tgt} -> (tgt << 32) | win, or -> the global
//                   window number gwBase[tgt] + win of the compact store (kernels.h DeviceTable)
// Slot placement depends on the claim order, lookup results do not: a key always sits in the first bucket of its
// probe sequence that had a free slot when it arrived, and nothing is ever removed.
#include "kernels.h"

namespace mcamd {

namespace {

__device__ __forceinline__ uint32_t effective_size(uint32_t key, uint32_t fileSize, const LoadFilter& lf)
{
    uint32_t size = fileSize;
    if (lf.rmOver && size > lf.rmOver) size = 0;           // bucket emptied (host_hashmap.hpp:480-495)
    if (lf.maxLocs && size > lf.maxLocs) size = lf.maxLocs; // keep the FIRST n values (:454-466)
    if (lf.shardCnt > 1 && key_owner(key, lf.shardCnt) != lf.shardIdx) size = 0;   // Mode K: another GPU's key
    return size;
}

__device__ __forceinline__ uint64_t decode_value(const uint8_t* p, uint32_t tb)
{
    // packed {window_id win (u32); target_id tgt (u16|u32)}, unaligned
    uint32_t win = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    uint32_t tgt = (uint32_t)p[4] | ((uint32_t)p[5] << 8);
    if (tb == 4) tgt |= ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
    return ((uint64_t)tgt << 32) | win;
}

__device__ __forceinline__ bool in_gw_range(const GwLayout& gw, uint32_t tgt, uint32_t win)
{
    return tgt < gw.targets && win < gw.base[tgt + 1] - gw.base[tgt] - gw.gap;
}

__global__ __launch_bounds__(256) void table_prep_kernel(const uint32_t* __restrict__ keys, const uint8_t* __restrict__ sizes, uint32_t n, LoadFilter lf,
                                                         uint32_t* __restrict__ fileSz, uint32_t* __restrict__ storeSz,
                                                         unsigned long long* __restrict__ counters)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t eff = 0;
    if (i < n) {
        const uint32_t fs = sizes[i];
        eff = effective_size(keys[i], fs, lf);
        fileSz[i] = fs;
        storeSz[i] = list_alloc(eff, lf.align);              // (what the list takes in the store: its length, or that rounded up to whole lines)
    }
    // counters[0] = keys stored, [1] = locations kept: one atomic per wave
    uint32_t nk = eff > 0 ? 1u : 0u, locs = eff;
    for (int off = 32; off > 0; off >>= 1) { nk += __shfl_down(nk, off); locs += __shfl_down(locs, off); }
    if ((threadIdx.x & 63) == 0 && nk) { atomicAdd(&counters[0], (unsigned long long)nk); atomicAdd(&counters[1], (unsigned long long)locs); }
}

__global__ __launch_bounds__(256) void table_insert_kernel(const uint32_t* __restrict__ keys, const uint8_t* __restrict__ sizes, uint32_t n, LoadFilter lf,
                                                           const uint32_t* __restrict__ fileOff, const uint32_t* __restrict__ storeOff,
                                                           const uint8_t* __restrict__ vals, uint32_t tb, uint64_t storeBase,
                                                           TableBucket* __restrict__ buckets, uint32_t nbuckets,
                                                           unsigned int* __restrict__ maxProbe, unsigned int* __restrict__ full,
                                                           GwLayout gw, unsigned int* __restrict__ rangeErr)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = keys[i];
    const uint32_t eff = effective_size(key, sizes[i], lf);
    if (eff == 0) return;
    const uint32_t home = (uint32_t)(((uint64_t)mix32(key) * nbuckets) >> 32);
    uint32_t cur = home, probe = 1, slot = kSlotsPerBucket;
    TableBucket* b = nullptr;
    for (;; ++probe) {
        b = buckets + cur;
        unsigned long long* sz = reinterpret_cast<unsigned long long*>(&b->size[0]);
        unsigned long long old = __atomic_load_n(sz, __ATOMIC_RELAXED);
        for (;;) {
            uint32_t j = 0;
            while (j < kSlotsPerBucket && ((old >> (16 * j)) & 0xFFFFull)) ++j;
            if (j == kSlotsPerBucket) break;                                   // bucket full: next one of the probe sequence
            const unsigned long long want = old | ((unsigned long long)eff << (16 * j));
            const unsigned long long prev = atomicCAS(sz, old, want);
            if (prev == old) { slot = j; break; }
            old = prev;
        }
        if (slot < kSlotsPerBucket) break;
        if (probe > nbuckets) { atomicExch(full, 1u); return; }
        cur = next_bucket(home, cur, probe, nbuckets);
    }
    b->key[slot] = key;
    uint64_t pay = storeBase + storeOff[i];
    if (eff == 1) {
        // a single location stays in the bucket in its 8-byte form; with the compact store the kernels turn it into a global window
        // number (DeviceTable::gw_of), so it must lie inside its target's windows like every location of the lists
        pay = decode_value(vals + (size_t)fileOff[i] * (4 + tb), tb);
        if (gw.base && !in_gw_range(gw, (uint32_t)(pay >> 32), (uint32_t)pay)) atomicExch(rangeErr, 1u);
    }
    b->payload[slot] = pay;
    if (probe > 1) atomicMax(maxProbe, probe);
}

template <bool COMPACT>
__global__ __launch_bounds__(256) void table_values_kernel(const uint32_t* __restrict__ keys, const uint8_t* __restrict__ sizes, uint32_t n, LoadFilter lf,
                                                           const uint32_t* __restrict__ fileOff, const uint32_t* __restrict__ storeOff,
                                                           const uint8_t* __restrict__ vals, uint32_t tb, uint64_t totalFileVals,
                                                           uint64_t* __restrict__ dst, uint32_t* __restrict__ dst32, GwLayout gw,
                                                           unsigned int* __restrict__ rangeErr)
{
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= totalFileVals) return;
    // key of value v = last i with fileOff[i] <= v   (fileOff has n + 1 entries, fileOff[n] = total)
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (fileOff[mid] <= v) lo = mid; else hi = mid;
    }
    const uint32_t eff = effective_size(keys[lo], sizes[lo], lf);
    const uint32_t t = (uint32_t)(v - fileOff[lo]);
    if (eff > 1 && t < eff) {
        const uint64_t loc = decode_value(vals + v * (4 + tb), tb);
        if constexpr (COMPACT) {
            const uint32_t tgt = (uint32_t)(loc >> 32), win = (uint32_t)loc;
            if (!in_gw_range(gw, tgt, win)) atomicExch(rangeErr, 1u);
            else dst32[storeOff[lo] + t] = gw.base[tgt] + win;
        } else {
            dst[storeOff[lo] + t] = loc;
        }
    }
}

}  // namespace

// direct-address index: one thread per bucket slot, a scattered 8-byte store per stored key
__global__ __launch_bounds__(256) void direct_index_kernel(const TableBucket* __restrict__ buckets, uint32_t nbuckets, uint64_t* __restrict__ direct, unsigned int* flag)
{
    const uint64_t total = (uint64_t)nbuckets * kSlotsPerBucket;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const TableBucket& B = buckets[i >> 2];
        const uint32_t slot = (uint32_t)i & 3u, size = B.size[slot];
        if (size == 0) continue;
        const uint64_t pay = B.payload[slot];
        const bool fits = size == 1 ? ((pay >> 32) < (1u << 24) && (uint32_t)pay < (1u << 24)) : (pay >> 48) == 0;
        if (!fits) { atomicOr(flag, 1u); continue; }
        direct[B.key[slot]] = direct_pack(size, pay);
    }
}
void launch_direct_index(const TableBucket* buckets, uint32_t nbuckets, uint64_t* direct, unsigned int* flag, hipStream_t st)
{
    if (nbuckets) hipLaunchKernelGGL(direct_index_kernel, dim3(256 * 32), dim3(256), 0, st, buckets, nbuckets, direct, flag);
}

void launch_table_prep(const uint32_t* keys, const uint8_t* sizes, uint32_t n, LoadFilter lf, uint32_t* fileSz, uint32_t* storeSz,
                       unsigned long long* counters, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(table_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, st, keys, sizes, n, lf, fileSz, storeSz, counters);
}

void launch_table_insert(const uint32_t* keys, const uint8_t* sizes, uint32_t n, LoadFilter lf, const uint32_t* fileOff,
                         const uint32_t* storeOff, const uint8_t* vals, uint32_t tb, uint64_t storeBase, TableBucket* buckets,
                         uint32_t nbuckets, unsigned int* maxProbe, unsigned int* full, hipStream_t st, GwLayout gw, unsigned int* rangeErr)
{
    if (n) hipLaunchKernelGGL(table_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, st, keys, sizes, n, lf, fileOff, storeOff,
                              vals, tb, storeBase, buckets, nbuckets, maxProbe, full, gw, rangeErr);
}

void launch_table_values(const uint32_t* keys, const uint8_t* sizes, uint32_t n, LoadFilter lf, const uint32_t* fileOff, const uint32_t* storeOff,
                         const uint8_t* vals, uint32_t tb, uint64_t totalFileVals, uint64_t* dst, hipStream_t st)
{
    if (n && totalFileVals)
        hipLaunchKernelGGL(table_values_kernel<false>, dim3((uint32_t)((totalFileVals + 255) / 256)), dim3(256), 0, st, keys, sizes, n, lf,
                           fileOff, storeOff, vals, tb, totalFileVals, dst, nullptr, GwLayout{}, nullptr);
}

void launch_table_values_compact(const uint32_t* keys, const uint8_t* sizes, uint32_t n, LoadFilter lf, const uint32_t* fileOff, const uint32_t* storeOff,
                                 const uint8_t* vals, uint32_t tb, uint64_t totalFileVals, uint32_t* dst32, GwLayout gw, unsigned int* rangeErr, hipStream_t st)
{
    if (n && totalFileVals)
        hipLaunchKernelGGL(table_values_kernel<true>, dim3((uint32_t)((totalFileVals + 255) / 256)), dim3(256), 0, st, keys, sizes, n, lf,
                           fileOff, storeOff, vals, tb, totalFileVals, nullptr, dst32, gw, rangeErr);
}

}  // namespace mcamd
