// metacache_amd/csrc/kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the MetaCache
// query hot path.  Semantics follow the reference's CPU classifier (file:line cite muellan/metacache
// src/); the structure is our own: one 64-lane wavefront per query, windows staged 2-bit packed
// through LDS, ballot/DPP based min-hash selection, 8-lane cooperative probing of 128-byte bucket
// groups, in-LDS (or in-HBM for huge lists) bitonic sorting and a segmented-scan candidate search.
//
// No MFMA anywhere: this is integer hashing and gathering; the bound is HBM/L2 random access.
#include "kernels.h"
#include "device_common.h"
#ifndef MC_LANE_HITS
#define MC_LANE_HITS 24
#endif
#ifndef MC_CHUNK_WINS
#define MC_CHUNK_WINS 1
#endif

#include <algorithm>
#include <cstdlib>

namespace mcamd {

// ================================================================================================
// rows 3-4: canonical k-mer + hash   (dna_encoding.hpp:168-177, :215-226; hash_int.hpp:41-48)
// ================================================================================================
__device__ __forceinline__ uint32_t tm_hash(uint32_t x)
{
    x = ((x >> 16) ^ x) * 0x45d9f3bu;
    x = ((x >> 16) ^ x) * 0x45d9f3bu;
    x = ((x >> 16) ^ x);
    return x;
}
// kmer holds the k-mer in its low 2k bits
__device__ __forceinline__ uint32_t canonical_hash(uint32_t kmer, uint32_t k)
{
    uint32_t r = __brev(kmer);                                     // reverses bit pairs AND the bits inside a pair
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);       // undo the swap inside each pair
    r = (~r) >> (32u - 2u * k);                                    // complement, keep low 2k bits
    return tm_hash(kmer < r ? kmer : r);
}
// ================================================================================================
// row 1: window arithmetic (hash_dna.hpp:54-75 + :222)
// ================================================================================================
__global__ __launch_bounds__(256) void plan_kernel(BatchView b, SketchParams sp, uint32_t* __restrict__ winCount)
{
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= b.n) return;
    uint4 qi = reinterpret_cast<const uint4*>(b.qinfo)[q];
    bool noTail = qi.w == kNoTail;
    uint32_t c = windows_of(qi.y, sp, noTail);
    if (!noTail) c += windows_of(qi.w, sp, false);
    winCount[q] = c;
}

void launch_plan(const BatchView& b, const SketchParams& sp, uint32_t* winCount, hipStream_t st)
{
    if (b.n == 0) return;
    hipLaunchKernelGGL(plan_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b, sp, winCount);
}

// ================================================================================================
// exclusive scan of u32 counts (element i at in[i*stride]) -> out32[n+1] and/or out64[n+1]
// three small kernels: block sums, scan of block sums (one block), block scan + offset
// ================================================================================================
constexpr uint32_t kScanBlock = 256;
constexpr uint32_t kScanItems = 8;
constexpr uint32_t kScanTile = kScanBlock * kScanItems;   // 2048 elements per block

__device__ __forceinline__ uint64_t block_reduce_u64(uint64_t v, uint64_t* sh)
{
    uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    uint64_t t = 0;
    for (uint32_t i = 0; i < blockDim.x / 64; ++i) t += sh[i];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(kScanBlock) void scan_block_sums(const uint32_t* __restrict__ in, uint32_t stride, uint32_t n,
                                                              uint64_t* __restrict__ blockSums)
{
    __shared__ uint64_t sh[kScanBlock / 64];
    uint32_t base = blockIdx.x * kScanTile;
    uint64_t s = 0;
    for (uint32_t i = threadIdx.x; i < kScanTile; i += kScanBlock) {
        uint32_t idx = base + i;
        if (idx < n) s += in[(size_t)idx * stride];
    }
    s = block_reduce_u64(s, sh);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = s;
}

__global__ __launch_bounds__(kScanBlock) void scan_of_sums(uint64_t* __restrict__ blockSums, uint32_t nblocks)
{
    // single block, sequential over tiles of 256 block sums with a running carry
    __shared__ uint64_t sh[kScanBlock];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += kScanBlock) {
        uint32_t i = base + threadIdx.x;
        uint64_t v = i < nblocks ? blockSums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < kScanBlock; d <<= 1) {
            uint64_t o = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += o;
            __syncthreads();
        }
        uint64_t incl = sh[threadIdx.x];
        uint64_t c = carry;
        if (i < nblocks) blockSums[i] = c + incl - v;   // exclusive
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) blockSums[nblocks] = carry;    // grand total
}

__global__ __launch_bounds__(kScanBlock) void scan_apply(const uint32_t* __restrict__ in, uint32_t stride, uint32_t n,
                                                         const uint64_t* __restrict__ blockSums, uint32_t nblocks,
                                                         uint32_t* __restrict__ out32, uint64_t* __restrict__ out64)
{
    __shared__ uint64_t sh[kScanBlock];
    uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint64_t local = 0;
#pragma unroll
    for (uint32_t j = 0; j < kScanItems; ++j) {
        uint32_t idx = base + j;
        v[j] = idx < n ? in[(size_t)idx * stride] : 0;
        local += v[j];
    }
    sh[threadIdx.x] = local;
    __syncthreads();
    for (uint32_t d = 1; d < kScanBlock; d <<= 1) {
        uint64_t o = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += o;
        __syncthreads();
    }
    uint64_t run = blockSums[blockIdx.x] + sh[threadIdx.x] - local;
#pragma unroll
    for (uint32_t j = 0; j < kScanItems; ++j) {
        uint32_t idx = base + j;
        if (idx < n) {
            if (out32) out32[idx] = (uint32_t)run;
            if (out64) out64[idx] = run;
        }
        run += v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t total = blockSums[nblocks];
        if (out32) out32[n] = (uint32_t)total;
        if (out64) out64[n] = total;
    }
}

size_t scan_tmp_bytes(uint32_t n) { return ((size_t)(n + kScanTile - 1) / kScanTile + 2) * sizeof(uint64_t); }

// Small inputs (the batches of the host slots: 4 096 reads each, a few of them united): ONE block of 1 024 threads walks the input in
// tiles of 4 096 with a running carry -- one launch instead of three.  A launch costs the host ~5 us and the stream a dependent
// dispatch; at 30 launches for 0.1 ms of device work per batch the slot path is bound by them (docs/LAB_NOTEBOOK_r06.md section 5).
constexpr uint32_t kSmallScan = 16384, kSmallPlan = 32768, kSmallScanBlock = 256, kSmallScanItems = 16;   // (a tile of 4 096: plain scans up to four tiles -- beyond that three parallel kernels are shorter --, the fused plan, which stands for five steps, up to eight)
// (256 threads: a block of 1 024 has to find sixteen free wave slots on ONE CU, which on a device busy with other pipes' kernels took
// longer than its work; a tile of 4 096 elements costs one barrier: the waves' sums are double-buffered and every thread adds up the ones before its wave)
template <typename Value>
__device__ __forceinline__ void scan_one_block(const uint32_t n, Value value, uint32_t* __restrict__ out32, uint64_t* __restrict__ out64, uint64_t* hostTotal = nullptr)
{
    constexpr uint32_t kWaves = kSmallScanBlock / 64;
    __shared__ uint64_t waveSum[2][kWaves];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t carry = 0;
    uint32_t buf = 0;
    for (uint32_t base = 0; base < n; base += kSmallScanBlock * kSmallScanItems, buf ^= 1u) {
        const uint32_t i0 = base + threadIdx.x * kSmallScanItems;
        uint32_t v[kSmallScanItems];
        uint64_t local = 0;
#pragma unroll
        for (uint32_t j = 0; j < kSmallScanItems; ++j) { v[j] = i0 + j < n ? value(i0 + j) : 0u; local += v[j]; }
        uint64_t incl = local;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) { const uint64_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63) waveSum[buf][wave] = incl;
        __syncthreads();
        uint64_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) { const uint64_t t = waveSum[buf][w]; all += t; if (w < wave) before += t; }
        uint64_t run = carry + before + incl - local;
        carry += all;
#pragma unroll
        for (uint32_t j = 0; j < kSmallScanItems; ++j) {
            if (i0 + j < n) {
                if (out32) out32[i0 + j] = (uint32_t)run;
                if (out64) out64[i0 + j] = run;
            }
            run += v[j];
        }
    }
    if (threadIdx.x == 0) {
        if (out32) out32[n] = (uint32_t)carry;
        if (out64) out64[n] = carry;
        if (hostTotal) { *hostTotal = carry; __threadfence_system(); }   // (pinned host memory: the host sizes its segments by it)
    }
}
__global__ __launch_bounds__(kSmallScanBlock) void scan_small_kernel(const uint32_t* __restrict__ in, uint32_t stride, uint32_t n,
                                                                     uint32_t* __restrict__ out32, uint64_t* __restrict__ out64, uint64_t* hostTotal)
{
    scan_one_block(n, [&](uint32_t i) { return in[(size_t)i * stride]; }, out32, out64, hostTotal);
}
// ... and the window arithmetic of plan_kernel in the same block: winCount, its scan, and the lane path's work-list counters cleared
// (`zero32`: 32 words, or null) -- one launch for what were five
__global__ __launch_bounds__(kSmallScanBlock) void plan_scan_small_kernel(BatchView b, SketchParams sp, uint32_t* __restrict__ winCount, uint32_t* __restrict__ winOff,
                                                                          uint32_t* __restrict__ zero32)
{
    if (zero32 && threadIdx.x < 32) zero32[threadIdx.x] = 0u;
    scan_one_block(b.n, [&](uint32_t q) {
        const uint4 qi = reinterpret_cast<const uint4*>(b.qinfo)[q];
        const bool noTail = qi.w == kNoTail;
        uint32_t c = windows_of(qi.y, sp, noTail);
        if (!noTail) c += windows_of(qi.w, sp, false);
        winCount[q] = c;
        return c;
    }, winOff, nullptr);
}
bool launch_plan_scan_small(const BatchView& b, const SketchParams& sp, uint32_t* winCount, uint32_t* winOff, uint32_t* zero32, hipStream_t st)
{
    if (b.n == 0 || b.n > kSmallPlan) return false;                // (the caller takes the three-kernel way)
    hipLaunchKernelGGL(plan_scan_small_kernel, dim3(1), dim3(kSmallScanBlock), 0, st, b, sp, winCount, winOff, zero32);
    return true;
}

void launch_words_to_host(uint32_t* hostDst, const uint32_t* src, uint32_t nwords, hipStream_t st);
void launch_scan_u32(const uint32_t* in, uint32_t stride, uint32_t n, uint32_t* out32, uint64_t* out64, void* tmp, hipStream_t st, uint64_t* hostTotal)
{
    if (n <= kSmallScan) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(kSmallScanBlock), 0, st, in, stride, n, out32, out64, hostTotal);
        return;
    }
    struct TotalOut { uint64_t* host; uint64_t* out64; uint32_t n; hipStream_t st; ~TotalOut() { if (host && out64) launch_words_to_host(reinterpret_cast<uint32_t*>(host), reinterpret_cast<const uint32_t*>(out64 + n), 2, st); } } totalOut{hostTotal, out64, n, st};
    uint32_t nblocks = (n + kScanTile - 1) / kScanTile;
    if (nblocks == 0) nblocks = 1;
    uint64_t* sums = (uint64_t*)tmp;
    hipLaunchKernelGGL(scan_block_sums, dim3(nblocks), dim3(kScanBlock), 0, st, in, stride, n, sums);
    hipLaunchKernelGGL(scan_of_sums, dim3(1), dim3(kScanBlock), 0, st, sums, nblocks);
    hipLaunchKernelGGL(scan_apply, dim3(nblocks), dim3(kScanBlock), 0, st, in, stride, n, sums, nblocks, out32, out64);
}

// ================================================================================================
// sketch_probe: one wave per query.
//   rows 1-5 (hash_dna.hpp:54-75, :208-255; dna_encoding.hpp:270-316) then row 6-7 lookups
//   (host_hashmap.hpp:646-651) against OUR table layout.
// ================================================================================================
constexpr uint32_t kCodeWords = kMaxWinLen / 16 + 2;   // 2 bits / base, MSB first inside a word
constexpr uint32_t kAmbWords  = kMaxWinLen / 32 + 2;   // 1 bit / base, LSB first

// A/a=0 C/c=1 G/g=2 T/t/U/u=3 (dna_encoding.hpp:297-304); returns code | (ambiguous << 2)
__device__ __forceinline__ uint32_t encode_base(uint32_t c)
{
    uint32_t x = (c >> 1) & 3u;
    uint32_t code = x ^ (x >> 1);                 // A0 C1 G3 T2 -> 0 1 2 3 ; U (0x55) -> 3 as well
    uint32_t u = (c & 0xDFu) - 0x41u;             // upper-cased letter index, 'A' = 0
    // valid letters: A(0) C(2) G(6) T(19) U(20)
    bool ok = u < 32u && ((0x00180045u >> u) & 1u);
    return ok ? code : 4u;
}

__device__ __forceinline__ uint32_t mbcnt(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ void probe_start(const DeviceTable& tab, uint32_t f, uint32_t& home, BucketRegs& r)
{
    home = home_group(f, tab.nbuckets);
    r.k = r.sz = r.p0 = r.p1 = make_uint4(0, 0, 0, 0);
    if (f != 0xFFFFFFFFu) r = load_bucket(tab, home);
}
// f == ~0 (no feature) yields size 0.  'steps' counts the buckets read.
__device__ __forceinline__ void probe_finish(const DeviceTable& tab, uint32_t f, uint32_t home, BucketRegs r, uint32_t& size, uint64_t& pay, uint32_t& steps)
{
    size = 0; pay = 0;
    if (f == 0xFFFFFFFFu) return;
    uint32_t cur = home;
    for (uint32_t step = 1;; ++step) {
        ++steps;
        const uint32_t keys[4] = {r.k.x, r.k.y, r.k.z, r.k.w};
        const uint32_t s01 = r.sz.x, s23 = r.sz.y;
        const uint32_t sz[4] = {s01 & 0xFFFFu, s01 >> 16, s23 & 0xFFFFu, s23 >> 16};
        const uint64_t pl[4] = {((uint64_t)r.p0.y << 32) | r.p0.x, ((uint64_t)r.p0.w << 32) | r.p0.z,
                                ((uint64_t)r.p1.y << 32) | r.p1.x, ((uint64_t)r.p1.w << 32) | r.p1.z};
        bool anyFree = false;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            anyFree = anyFree || sz[i] == 0;
            if (sz[i] != 0 && keys[i] == f) { size = sz[i]; pay = pl[i]; }
        }
        // found, or a bucket with a free slot ends the chain (insertion fills the first bucket with room)
        if (size != 0 || anyFree || step >= tab.maxProbe) return;
        cur = next_bucket(home, cur, step, tab.nbuckets);
        r = load_bucket(tab, cur);
    }
}

// ================================================================================================
// sort_candidates: one wave per query.
//   row 7-8: gather the location lists and sort them by (tgt, win)   (host_hashmap.hpp:646-651,
//            query_handler.hpp:75-101 -- for a single part the merge result == full sort)
//   row 9  : best contiguous window range per target                   (candidate_generation.hpp:47-108)
//   row 10 : top-K list, ties keep arrival order, optional taxon merge (candidate_generation.hpp:172-231)
// Lists of up to kLdsCap locations live in LDS; longer ones are processed in place in HBM.
// ================================================================================================

template <bool LDS>
__device__ __forceinline__ void wsync() { if (LDS) wave_lds_sync(); else wave_mem_sync(); }

__device__ __forceinline__ void cmpxchg(uint64_t* buf, uint32_t i, uint32_t p)
{
    uint64_t a = buf[i], c = buf[p];
    if (c < a) { buf[i] = c; buf[p] = a; }
}

// ascending bitonic sort of buf[0..n) by one wave; indices >= n behave as +infinity
template <bool LDS>
__device__ __forceinline__ void wave_sort_u64(uint64_t* buf, uint32_t n, uint32_t lane)
{
    if (n < 2) return;
    uint32_t npow = 2;
    while (npow < n) npow <<= 1;
    const uint32_t half = npow >> 1;
    for (uint32_t k = 2; k <= npow; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = lane; t < half; t += 64) {       // flip: i <-> mirror inside blocks of k
            const uint32_t blk = t / hk, r = t & (hk - 1);
            const uint32_t i = blk * k + r, p = blk * k + (k - 1 - r);
            if (p < n) cmpxchg(buf, i, p);
        }
        wsync<LDS>();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < half; t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                if (p < n) cmpxchg(buf, i, p);
            }
            wsync<LDS>();
        }
    }
}

// n <= 64: sort in registers with lane shuffles, one key per lane (padding = ~0)
__device__ __forceinline__ uint64_t wave_sort64_reg(uint64_t key, uint32_t lane)
{
#pragma unroll
    for (uint32_t k = 2; k <= 64; k <<= 1) {
        {
            uint64_t o = __shfl(key, lane ^ (k - 1));
            bool lower = (lane & (k >> 1)) == 0;
            key = lower ? (o < key ? o : key) : (o > key ? o : key);
        }
#pragma unroll
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            uint64_t o = __shfl(key, lane ^ j);
            bool lower = (lane & j) == 0;
            key = lower ? (o < key ? o : key) : (o > key ? o : key);
        }
    }
    return key;
}

template <bool LDS>
__device__ __forceinline__ void sort_list(uint64_t* buf, uint32_t n, uint32_t lane)
{
    if (n <= 64) {
        uint64_t key = lane < n ? buf[lane] : ~0ull;
        key = wave_sort64_reg(key, lane);
        if (lane < n) buf[lane] = key;
        wsync<LDS>();
    } else {
        wave_sort_u64<LDS>(buf, n, lane);
    }
}

// first index in [0, hi] whose key >= lb (buf[hi] >= lb is guaranteed by the caller)
__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t* buf, uint32_t hi, uint64_t lb)
{
    uint32_t lo = 0;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (buf[mid] < lb) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// left end of the sliding window range that ends at list position i (candidate_generation.hpp:79-85)
__device__ __forceinline__ uint32_t range_first(const uint64_t* buf, uint32_t i, uint64_t key, uint32_t maxWin)
{
    const uint32_t win = (uint32_t)key;
    const uint32_t lowWin = win >= maxWin - 1 ? win - (maxWin - 1) : 0u;
    return lower_bound_u64(buf, i, (key & 0xFFFFFFFF00000000ull) | lowWin);
}

// packed candidate: hits(20) | ~ord(20) | position of the range end in the sorted list(20)
// -> unsigned descending order == (hits desc, arrival order asc)  (candidate_generation.hpp:189-201)
__device__ __forceinline__ uint64_t pack_cand(uint32_t hits, uint32_t ord, uint32_t besti)
{
    return ((uint64_t)hits << 40) | ((uint64_t)((~ord) & 0xFFFFFu) << 20) | besti;
}

__device__ __forceinline__ void emit_empty(mc_candidate_dev* out, uint32_t from, uint32_t K, uint32_t lane)
{
    for (uint32_t r = from + lane; r < K; r += 64) {
        mc_candidate_dev e; e.tgt = 0xFFFFFFFFu; e.hits = 0; e.beg = 0; e.end = 0;
        out[r] = e;
    }
}

// buf[0..H) sorted.  Builds the per-target candidates in C (C2 = scratch for taxon merging) and
// writes the top K to out.
template <bool LDS>
__device__ __forceinline__ void candidates_from_sorted(
    const uint64_t* buf, uint64_t* C, uint64_t* C2, const uint32_t H, const uint32_t maxWin, const uint32_t K,
    const uint32_t* __restrict__ taxkey, const uint32_t tgtMask, mc_candidate_dev* out, uint32_t lane)
{
    // ---- row 9: one candidate per maximal run of equal tgt
    uint32_t ncand = 0;
    uint64_t carryVal = 0;
    uint32_t carryTgt = 0;
    for (uint32_t c0 = 0; c0 < H; c0 += 64) {
        const uint32_t i = c0 + lane;
        const bool valid = i < H;
        const uint64_t key = valid ? buf[i] : 0;
        const uint32_t tgt = (uint32_t)(key >> 32);
        uint32_t prevTgt = __shfl_up(tgt, 1);
        if (lane == 0) prevTgt = carryTgt;
        const bool head = valid && (i == 0 || tgt != prevTgt);
        uint32_t nextTgt = __shfl_down(tgt, 1);
        bool nextValid = i + 1 < H;
        if (lane == 63 && nextValid) nextTgt = (uint32_t)(buf[i + 1] >> 32);
        const bool tail = valid && (!nextValid || nextTgt != tgt);

        uint64_t val = 0;
        if (valid) {
            const uint32_t fst = range_first(buf, i, key, maxWin);
            val = ((uint64_t)(i - fst + 1) << 32) | (0xFFFFFFFFu - i);   // max => most hits, then earliest end
        }
        // segmented inclusive max-scan over the runs inside this chunk
        bool f = head;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t o = __shfl_up(val, d);
            int of = __shfl_up((int)f, d);
            if (lane >= (uint32_t)d) {
                if (!f && o > val) val = o;
                f = f || of;
            }
        }
        if (valid && !f && carryVal > val) val = carryVal;              // run continues from the previous chunk
        const uint64_t tailMask = __ballot(tail);
        if (tail) {
            const uint32_t ord = ncand + __popcll(tailMask & ((1ull << lane) - 1ull));
            C[ord] = pack_cand((uint32_t)(val >> 32), ord, 0xFFFFFFFFu - (uint32_t)val);
        }
        ncand += __popcll(tailMask);
        carryVal = rdlane64(val, 63);
        carryTgt = rdlane(tgt, 63);
    }
    wsync<LDS>();

    // ---- row 10: order candidates
    uint64_t* S = C;       // list that ends up sorted ascending by ~packed
    if (taxkey) {
        // at most one entry per taxon: keep the one with most hits, earliest arrival (:205-216)
        for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
            const uint32_t j = c0 + lane;
            if (j < ncand) {
                const uint64_t ck = C[j];
                const uint32_t besti = (uint32_t)(ck & 0xFFFFFu);
                const uint32_t tgt = (uint32_t)(buf[besti] >> 32);
                const uint32_t tax = taxkey[tgt & tgtMask];               // 0 = no taxon -> dropped (:187)
                const uint32_t hits = (uint32_t)(ck >> 40);
                C2[j] = tax ? (((uint64_t)tax << 40) | ((uint64_t)((~hits) & 0xFFFFFu) << 20) | j) : ~0ull;
            }
        }
        wsync<LDS>();
        sort_list<LDS>(C2, ncand, lane);
        uint32_t carryTax = 0;
        for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
            const uint32_t j = c0 + lane;
            const uint64_t k1 = j < ncand ? C2[j] : ~0ull;
            const uint32_t tax = (uint32_t)(k1 >> 40);
            uint32_t prevTax = __shfl_up(tax, 1);
            if (lane == 0) prevTax = carryTax;
            const bool winner = k1 != ~0ull && (j == 0 || tax != prevTax);
            uint64_t nv = ~0ull;
            if (winner) nv = ~C[(uint32_t)(k1 & 0xFFFFFu)];
            carryTax = rdlane(tax, 63);
            wsync<LDS>();
            if (j < ncand) C2[j] = nv;
        }
        wsync<LDS>();
        S = C2;
    } else {
        for (uint32_t j = lane; j < ncand; j += 64) C[j] = ~C[j];
        wsync<LDS>();
    }
    sort_list<LDS>(S, ncand, lane);

    // ---- emit the first K
    const uint32_t nout = min(K, ncand);
    uint32_t written = 0;
    for (uint32_t r0 = 0; r0 < nout; r0 += 64) {
        const uint32_t r = r0 + lane;
        const uint64_t x = r < nout ? ~S[r] : 0;
        const bool ok = x != 0;                                           // 0 = non-winner / dropped
        if (ok) {
            const uint32_t besti = (uint32_t)(x & 0xFFFFFu);
            const uint64_t key = buf[besti];
            const uint32_t fst = range_first(buf, besti, key, maxWin);
            mc_candidate_dev e;
            e.tgt = (uint32_t)(key >> 32) & tgtMask;
            e.hits = (uint32_t)(x >> 40);
            e.beg = (uint32_t)buf[fst];
            e.end = (uint32_t)key;
            out[r] = e;
        }
        written += __popcll(__ballot(ok));                                // valid entries are a prefix
    }
    emit_empty(out, written, K, lane);
}

// row 7 + 8: gather the location lists of this query's features (window / feature order; singletons
// are inline in the payload) into buf and sort them
template <bool LDS>
__device__ __forceinline__ void gather_and_sort(uint64_t* buf, const Workspace& ws, const DeviceTable& tab,
                                                uint32_t fbeg, uint32_t nf, uint32_t H, uint32_t lane)
{
    uint32_t base = 0;
    for (uint32_t e0 = 0; e0 < nf; e0 += 64) {
        const uint32_t e = e0 + lane;
        const uint32_t sz = e < nf ? (ws.psize[fbeg + e] & 0xFFFFu) : 0u;
        const uint64_t pay = e < nf ? ws.ppay[fbeg + e] : 0ull;
        const uint32_t incl = wave_incl_scan_u32(sz, lane);
        const uint32_t dst = base + incl - sz;
        if (sz == 1) buf[dst] = pay;
        uint64_t multi = __ballot(sz > 1);
        while (multi) {                                                   // wave-uniform loop over big buckets
            const uint32_t j = __ffsll((unsigned long long)multi) - 1;
            multi &= multi - 1;
            const uint32_t nj = rdlane(sz, j), dj = rdlane(dst, j);
            const uint64_t src = rdlane64(pay, j);
            for (uint32_t t = lane; t < nj; t += 64) buf[dj + t] = tab.loc(src + t);   // coalesced copy
        }
        base += rdlane(incl, 63);
    }
    wsync<LDS>();
    sort_list<LDS>(buf, H, lane);
}

struct SortLds {
    uint64_t buf[kLdsCap];
    uint64_t c[kLdsCap];
    uint64_t c2[kLdsCap];
};

__device__ __forceinline__ void sort_candidates_one(
    const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, const uint32_t* __restrict__ taxkey,
    uint32_t K, int wantAllhits, mc_candidate_dev* __restrict__ cands, SortLds& L, uint32_t q, uint32_t lane)
{
    mc_candidate_dev* out = cands + (size_t)q * K;
    const uint32_t H = ws.qstat[q].hits;
    if (H == 0 || H > kMaxHitsPerQuery) { emit_empty(out, 0, K, lane); return; }

    const uint32_t maxWin = b.maxWin ? b.maxWin[q] : b.maxWinUniform;
    const uint64_t hoff = ws.hitOff[q];
    const uint32_t fbeg = ws.winOff[q] * sp.s;
    const uint32_t nf = (ws.winOff[q + 1] - ws.winOff[q]) * sp.s;
    const uint64_t m64 = ((uint64_t)tab.tgtMask << 32) | 0xFFFFFFFFull;   // multi-part tables: strip the part number from targets
    // two instantiations so that the LDS flavour compiles to ds_* instructions
    if (H <= kLdsCap) {
        gather_and_sort<true>(L.buf, ws, tab, fbeg, nf, H, lane);
        if (wantAllhits)
            for (uint32_t i = lane; i < H; i += 64) ws.hits[hoff + i] = L.buf[i] & m64;
        candidates_from_sorted<true>(L.buf, L.c, L.c2, H, maxWin, K, taxkey, tab.tgtMask, out, lane);
    } else {
        gather_and_sort<false>(ws.hits + hoff, ws, tab, fbeg, nf, H, lane);
        candidates_from_sorted<false>(ws.hits + hoff, ws.cscr + hoff, ws.cscr2 + hoff, H, maxWin, K, taxkey, tab.tgtMask, out, lane);
        if (wantAllhits && tab.tgtMask != 0xFFFFFFFFu)
            for (uint32_t i = lane; i < H; i += 64) ws.hits[hoff + i] &= m64;
    }
}

__global__ __launch_bounds__(256) void sort_candidates_kernel(
    BatchView b, SketchParams sp, DeviceTable tab, Workspace ws, const uint32_t* __restrict__ taxkey,
    uint32_t K, int wantAllhits, mc_candidate_dev* __restrict__ cands)
{
    __shared__ SortLds lds[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nWaves = gridDim.x * 4, waveId = blockIdx.x * 4 + wave;
    for (uint32_t base = waveId * 64; base < b.n; base += nWaves * 64) {
        const uint32_t qq = base + lane;
        const uint32_t fl = qq < b.n ? ws.qflag[qq] : kFlagDone;
        uint64_t m = __ballot(fl == kFlagCands);
        while (m) {
            const uint32_t j = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            sort_candidates_one(b, sp, tab, ws, taxkey, K, wantAllhits, cands, lds[wave], base + j, lane);
            wave_lds_sync();
        }
    }
}

// rows 8-10 on lists that are already gathered (Mode K: the union of the key shards' partial lists): ws.hits[hitOff[q] ..
// hitOff[q+1]) in any order; sorted in place (LDS copy for short lists), the sorted list is written back.
__global__ __launch_bounds__(256) void cands_from_hits_kernel(BatchView b, DeviceTable tab, Workspace ws, const uint32_t* __restrict__ taxkey,
                                                              uint32_t K, mc_candidate_dev* __restrict__ cands)
{
    __shared__ SortLds lds[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (q >= b.n) return;
    if (ws.qflag && ws.qflag[q] != kFlagCands) return;           // (owner side of Mode K: the long lists took the filtered path)
    SortLds& L = lds[wave];
    mc_candidate_dev* out = cands + (size_t)q * K;
    const uint64_t hoff = ws.hitOff[q];
    const uint64_t H64 = ws.hitOff[q + 1] - hoff;
    if (lane == 0) { QueryStat qs; qs.hits = (uint32_t)min(H64, (uint64_t)0xFFFFFFFFu); qs.nfeat = 0; qs.nfound = 0; qs.nsteps = 0; ws.qstat[q] = qs; }
    if (H64 == 0 || H64 > kMaxHitsPerQuery) { emit_empty(out, 0, K, lane); return; }
    const uint32_t H = (uint32_t)H64;
    const uint32_t maxWin = b.maxWin ? b.maxWin[q] : b.maxWinUniform;
    if (H <= kLdsCap) {
        for (uint32_t i = lane; i < H; i += 64) L.buf[i] = ws.hits[hoff + i];
        wave_lds_sync();
        sort_list<true>(L.buf, H, lane);
        for (uint32_t i = lane; i < H; i += 64) ws.hits[hoff + i] = L.buf[i];
        candidates_from_sorted<true>(L.buf, L.c, L.c2, H, maxWin, K, taxkey, tab.tgtMask, out, lane);
    } else {
        sort_list<false>(ws.hits + hoff, H, lane);
        candidates_from_sorted<false>(ws.hits + hoff, ws.cscr + hoff, ws.cscr2 + hoff, H, maxWin, K, taxkey, tab.tgtMask, out, lane);
    }
}

void launch_cands_from_hits(const BatchView& b, const DeviceTable& tab, const Workspace& ws, const uint32_t* taxkey, uint32_t maxCand,
                            void* cands, hipStream_t st)
{
    if (b.n == 0) return;
    hipLaunchKernelGGL(cands_from_hits_kernel, dim3((b.n + 3) / 4), dim3(256), 0, st, b, tab, ws, taxkey, maxCand, (mc_candidate_dev*)cands);
}

// Mode K, owner side: the partial location lists of `sources` key shards (counts[s * m + i] locations of read i from source s; the
// locations of one source back to back in read order, the sources' blocks back to back in `hits`) -> one list per read.
//   union_totals : locations per read over all sources                      (-> scan -> ws.hitOff)
//   union_copy   : one wave per read copies its pieces, source after source; srcStart[s * (m + 1) + i] = first location of read i
//                  inside source s' block (exclusive scans of the counts), the block of source s begins where the blocks before end
__global__ __launch_bounds__(256) void union_totals_kernel(const uint32_t* __restrict__ counts, uint32_t sources, uint32_t m, uint32_t* __restrict__ tot)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    uint32_t t = 0;
    for (uint32_t s = 0; s < sources; ++s) t += counts[(size_t)s * m + i];
    tot[i] = t;
}
__global__ __launch_bounds__(256) void union_copy_kernel(const uint32_t* __restrict__ counts, const uint64_t* __restrict__ srcStart, uint32_t sources, uint32_t m,
                                                         const uint64_t* __restrict__ hits, const uint64_t* __restrict__ hitOff, uint64_t* __restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    uint64_t dst = hitOff[i], base = 0;
    for (uint32_t s = 0; s < sources; ++s) {
        const uint32_t c = counts[(size_t)s * m + i];
        const uint64_t* src = hits + base + srcStart[(size_t)s * (m + 1) + i];
        for (uint32_t t = lane; t < c; t += 64) out[dst + t] = src[t];
        dst += c;
        base += srcStart[(size_t)s * (m + 1) + m];
    }
}
// Mode K, owner side: which united lists take the filtered path (big_filter_kernel / big_count_kernel with the union buffer standing in
// for the table's location store: ONE entry per read = its whole list) and which the sort.  Also the identity 'window offsets' the
// counting kernel's step D finds a read's entry through.
__global__ __launch_bounds__(256) void owner_classify_kernel(BatchView b, Workspace ws, uint32_t minLen)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    bool big = false;
    uint32_t H = 0, mw = 0;
    if (q < b.n) {
        const uint64_t H64 = ws.hitOff[q + 1] - ws.hitOff[q];
        mw = b.maxWin ? b.maxWin[q] : b.maxWinUniform;
        big = H64 > minLen && H64 <= 0xFFFFull && mw <= 8u;          // (8 = kHashWin, the widest window range the counting kernel takes)
        H = (uint32_t)min(H64, (uint64_t)0xFFFFFFFFu);
        ws.winOff[q] = q;
        if (q + 1 == b.n) ws.winOff[b.n] = b.n;
        ws.qflag[q] = big ? kFlagMid : kFlagCands;
        if (big) {
            ws.psize[q] = H; ws.ppay[q] = ws.hitOff[q];
            QueryStat qs; qs.hits = H; qs.nfeat = 0; qs.nfound = 1; qs.nsteps = 0;
            ws.qstat[q] = qs;
        }
    }
    const uint64_t mask = __ballot(big);
    if (big) {
        const uint32_t leader = __ffsll((unsigned long long)mask) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&ws.midCount[9], (uint32_t)__popcll(mask));
        base = __shfl(base, leader);
        reinterpret_cast<uint4*>(ws.midList)[(size_t)6 * b.n + base + __popcll(mask & ((1ull << lane) - 1ull))] = make_uint4(q, q, 1u | (H << 12), mw);
    }
}
void launch_owner_classify(const BatchView& b, const Workspace& ws, uint32_t minLen, hipStream_t st)
{
    if (b.n) hipLaunchKernelGGL(owner_classify_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b, ws, minLen);
}

void launch_union_partial(const uint32_t* counts, uint32_t sources, uint32_t m, const uint64_t* hits, uint32_t* tot, uint64_t* srcStart, uint64_t* hitOff,
                          uint64_t* out, void* scanTmp, hipStream_t st)
{
    if (m == 0) return;
    hipLaunchKernelGGL(union_totals_kernel, dim3((m + 255) / 256), dim3(256), 0, st, counts, sources, m, tot);
    launch_scan_u32(tot, 1, m, nullptr, hitOff, scanTmp, st);
    for (uint32_t s = 0; s < sources; ++s) launch_scan_u32(counts + (size_t)s * m, 1, m, nullptr, srcStart + (size_t)s * (m + 1), scanTmp, st);
    hipLaunchKernelGGL(union_copy_kernel, dim3((m + 3) / 4), dim3(256), 0, st, counts, srcStart, sources, m, hits, hitOff, out);
}

void launch_sort_candidates(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws,
                            const uint32_t* taxkey, uint32_t maxCand, bool wantAllhits, void* cands, hipStream_t st)
{
    if (b.n == 0) return;
    hipLaunchKernelGGL(sort_candidates_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b, sp, tab, ws, taxkey, maxCand,
                       wantAllhits ? 1 : 0, (mc_candidate_dev*)cands);
}

// ================================================================================================
// query_kernel: the fused fast path.  One wave per query:
//   stage the read ONCE as 2-bit codes (all windows of a <= 1024-character segment share it),
//   sketch every window (threshold filter + ballot compaction + rank sort + unique),
//   probe up to 32 features with all loads in flight,
//   and -- when the query has a single probe group, at most kFuseCap location hits, no taxon merging
//   and no -allhits -- finish rows 7-10 in registers with an all-pairs readlane pass (no sort, no
//   LDS lists, no second kernel).  Everything else is handed to sort_candidates_kernel exactly like
//   sketch_probe_kernel does.
// ================================================================================================
constexpr uint32_t kFuseCap = 32;
constexpr uint32_t kGroupSlots = 64;                    // features probed together by one wave (one lane each)

struct FusedLds {
    uint32_t code[kCodeWords];
    uint32_t amb[kAmbWords];
    uint32_t cand[128];      // [0,64) compaction target, [64,128) dump slots for masked-off lanes
    uint32_t srt[128];
    uint32_t sk[128];
    uint32_t fbuf[kGroupSlots + 64];   // features of the current probe group (+ dump slots)
    uint64_t sbuf[kFuseCap + 32];
};

// Stage 'n' characters starting at seq[start] (n <= kMaxWinLen); branch-free per lane.
template <class LDS>
__device__ __forceinline__ void stage_segment(const uint8_t* __restrict__ seq, uint64_t start, uint32_t n, LDS& L, uint32_t lane)
{
    uint8_t* codeB = reinterpret_cast<uint8_t*>(L.code);
    uint8_t* ambB = reinterpret_cast<uint8_t*>(L.amb);
    for (uint32_t c0 = 0; c0 < n; c0 += 256) {            // 64 lanes x 4 characters per pass
        const uint32_t c = c0 + lane * 4;
        // clamp the address so that every lane may load (8 slack bytes follow the buffer)
        const uint64_t a = start + (c < n ? c : 0u);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(seq + (a & ~(uint64_t)3));
        const uint32_t chars = __builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)(a & 3));
        uint32_t code = 0, amb = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t e = encode_base((chars >> (8 * j)) & 0xFFu);
            const bool bad = (e & 4u) || (c + j >= n);
            code = (code << 2) | (bad ? 0u : (e & 3u));
            amb |= (bad ? 1u : 0u) << j;
        }
        const uint32_t ci = c >> 2;
        codeB[ci ^ 3u] = (uint8_t)code;                     // big-endian inside each 32-bit word
        const uint32_t other = dpp_mov<0xB1>(amb);          // neighbour lane's nibble
        ambB[(lane & 1u) ? (kAmbWords * 4 - 1) : (ci >> 1)] = (uint8_t)(amb | (other << 4));   // odd lanes: dump byte
    }
}

// hash of the k-mer at window position p (window starts at segment offset wo); ~0 if invalid
template <class LDS>
__device__ __forceinline__ uint32_t seg_hash_at(const LDS& L, uint32_t wo, uint32_t p, uint32_t nk, uint32_t k, uint32_t kbits)
{
    const bool ok = p < nk;
    const uint32_t a = wo + (ok ? p : 0u);
    const uint32_t wq = a >> 4, sh = (a & 15u) * 2u;
    const uint32_t kmer = __funnelshift_l(L.code[wq + 1], L.code[wq], sh) >> (32u - 2u * k);
    const uint32_t aw = a >> 5;
    const uint32_t am = __funnelshift_r(L.amb[aw], L.amb[aw + 1], a & 31u) & kbits;
    const uint32_t h = canonical_hash(kmer, k);
    return (ok && am == 0) ? h : 0xFFFFFFFFu;
}

// min-hash sketch of window [wo, wo+n) of the staged segment -> L.fbuf[slotBase .. slotBase+s)
// (ascending, ~0 padded); returns the number of valid features.
__device__ __forceinline__ uint32_t sketch_window_v3(FusedLds& L, uint32_t wo, uint32_t n, uint32_t k, uint32_t s, uint32_t slotBase, uint32_t lane)
{
    const uint32_t kbits = 0xFFFFu >> (16u - k);
    const uint32_t nk = n - k + 1;
    const uint32_t sl = min(s, nk);
    const uint64_t t64 = (((uint64_t)7 * sl) << 32) / ((uint64_t)4 * nk);
    uint32_t T = t64 >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t64;
    uint32_t cnt;
    for (;;) {
        cnt = 0;
        for (uint32_t base = 0; base < nk; base += 128) {
            const uint32_t h0 = seg_hash_at(L, wo, base + lane, nk, k, kbits);
            const uint32_t h1 = seg_hash_at(L, wo, base + 64 + lane, nk, k, kbits);
            const bool b0 = h0 < T, b1 = h1 < T;
            const uint64_t m0 = __ballot(b0), m1 = __ballot(b1);
            const uint32_t n0 = __popcll(m0), n1 = __popcll(m1);
            const uint32_t nc = cnt + n0 + n1;
            if (nc == cnt) continue;
            if (nc <= 64) {
                if (cnt) {                                   // carried sketch of earlier rounds (windows > 143 chars)
                    const uint32_t old = L.sk[lane];
                    wave_lds_sync();
                    L.cand[lane < cnt ? lane : 64 + lane] = old;
                }
                L.cand[b0 ? cnt + mbcnt(m0) : 64 + lane] = h0;
                L.cand[b1 ? cnt + n0 + mbcnt(m1) : 64 + lane] = h1;
                wave_lds_sync();
                uint32_t v = L.cand[lane];
                v = lane < nc ? v : 0xFFFFFFFFu;
                // rank sort: position = number of smaller (value, lane) pairs
                const uint64_t key = ((uint64_t)v << 32) | lane;
                uint32_t rank = 0;
#pragma unroll 4
                for (uint32_t j = 0; j < nc; ++j) {
                    const uint64_t kj = ((uint64_t)rdlane(v, j) << 32) | j;
                    rank += kj < key ? 1u : 0u;
                }
                L.srt[lane < nc ? rank : 64 + lane] = v;
                wave_lds_sync();
                const uint32_t x = L.srt[lane], prev = L.srt[(lane + 63u) & 63u];
                const bool uniq = lane < nc && (lane == 0 || x != prev);
                const uint64_t um = __ballot(uniq);
                const uint32_t upos = mbcnt(um);
                L.sk[(uniq && upos < sl) ? upos : 64 + lane] = x;
                cnt = min(sl, (uint32_t)__popcll(um));
                wave_lds_sync();
            } else {
                // degenerate input (> 64 candidates): serial minimum extraction, still exact
                const uint32_t a0 = b0 ? h0 : 0xFFFFFFFFu, a1 = b1 ? h1 : 0xFFFFFFFFu;
                uint32_t skv = L.sk[lane];
                skv = lane < cnt ? skv : 0xFFFFFFFFu;
                uint32_t nsk = 0xFFFFFFFFu, lb = 0, c = 0;
                for (uint32_t t = 0; t < sl; ++t) {
                    const uint32_t cand = min(min(a0 >= lb ? a0 : 0xFFFFFFFFu, a1 >= lb ? a1 : 0xFFFFFFFFu), skv >= lb ? skv : 0xFFFFFFFFu);
                    const uint32_t m = wave_min_u32(cand);
                    if (m == 0xFFFFFFFFu) break;
                    if (lane == t) nsk = m;
                    lb = m + 1; ++c;
                }
                wave_lds_sync();
                L.sk[lane] = nsk;
                cnt = c;
                wave_lds_sync();
            }
        }
        if (cnt >= sl || T == 0xFFFFFFFFu) break;
        T = T >= 0x40000000u ? 0xFFFFFFFFu : T << 2;
    }
    const uint32_t fin = L.sk[lane];
    L.fbuf[lane < s ? slotBase + lane : kGroupSlots + lane] = lane < cnt ? fin : 0xFFFFFFFFu;
    return cnt;
}

// rows 7-10 for at most kFuseCap location hits, entirely in registers (sequence-level candidates).
// Every lane looked ONE feature of the group up: (size, pay) is its result (size 0 = nothing).
__device__ __forceinline__ void fused_candidates(FusedLds& L, uint32_t size, uint64_t pay, const DeviceTable& tab, uint32_t H, uint32_t maxWin,
                                                 uint32_t K, mc_candidate_dev* out, uint32_t lane)
{
    // compact the hit locations into lanes 0..H-1 (their order is irrelevant: ranks are computed below)
    uint32_t base = 0;
    {
        const bool single = size == 1;
        const uint64_t m = __ballot(single);
        if (single) L.sbuf[mbcnt(m)] = pay;
        base = __popcll(m);
    }
    uint64_t mm = __ballot(size > 1);
    while (mm) {
        const uint32_t j = __ffsll((unsigned long long)mm) - 1;
        mm &= mm - 1;
        const uint32_t nj = rdlane(size, j);
        const uint64_t src = rdlane64(pay, j);
        if (lane < nj) L.sbuf[base + lane] = tab.loc(src + lane);    // nj <= H <= 32 < 64
        base += nj;
    }
    wave_lds_sync();
    const uint64_t key = L.sbuf[lane < H ? lane : 0];
    const uint32_t tgt = (uint32_t)(key >> 32), win = (uint32_t)key;
    uint32_t rank = 0, hits = 0, beg = win;
    for (uint32_t j = 0; j < H; ++j) {
        const uint32_t tj = rdlane(tgt, j), wj = rdlane(win, j);
        const uint64_t kj = ((uint64_t)tj << 32) | wj;
        const bool lt = kj < key || (kj == key && j < lane);      // position of j before this lane's element in sorted order
        const bool le = lt || j == lane;
        rank += lt ? 1u : 0u;
        const bool inr = le && tj == tgt && (win - wj) < maxWin;   // j lies in the window range that ends here (:79-85)
        hits += inr ? 1u : 0u;
        beg = inr ? min(beg, wj) : beg;
    }
    // top-K: most hits first, ties by list position (= target order, then window order) (:189-201, :87-91)
    uint64_t ckey = lane < H ? (((uint64_t)hits << 32) | (0xFFFFFFFFu - rank)) : 0ull;
    uint32_t nout = 0;
    for (; nout < K; ++nout) {
        const uint32_t mh = wave_max_u32((uint32_t)(ckey >> 32));
        if (mh == 0) break;
        const uint32_t ml = wave_max_u32((uint32_t)(ckey >> 32) == mh ? (uint32_t)ckey : 0u);
        const bool winner = ckey == (((uint64_t)mh << 32) | ml);
        const uint32_t wl = __ffsll((unsigned long long)__ballot(winner)) - 1;
        const uint32_t wt = rdlane(tgt, wl);
        if (winner) {
            mc_candidate_dev e; e.tgt = tgt & tab.tgtMask; e.hits = hits; e.beg = beg; e.end = win;
            out[nout] = e;
        }
        ckey = tgt == wt ? 0ull : ckey;                          // one candidate per target
    }
    emit_empty(out, nout, K, lane);
}

template <bool FUSE>
__device__ __forceinline__ void query_one(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, int wantAllhits,
                                          uint32_t K, mc_candidate_dev* __restrict__ cands, FusedLds& L, uint32_t q, uint32_t lane)
{
    const uint4 qi = reinterpret_cast<const uint4*>(b.qinfo)[q];
    const bool noTail = qi.w == kNoTail;
    const uint32_t widx0 = ws.winOff[q];
    const uint32_t nwinTotal = ws.winOff[q + 1] - widx0;
    const uint32_t k = sp.k, s = sp.s;
    const uint32_t winsPerGroup = kGroupSlots / s;
    const bool single = nwinTotal <= winsPerGroup;          // the whole query is one probe group

    uint32_t widx = widx0;
    uint32_t myHits = 0, nfeat = 0, nfound = 0, nsteps = 0;  // nfeat is wave-uniform, the others per lane
    uint32_t gslot = 0, gfirst = widx;
    uint32_t rsize = 0; uint64_t rpay = 0;                   // this lane's lookup result of the current group

    // one lane per feature of the group: all lookups of the group are in flight together
    auto probe_group = [&]() {
        const uint32_t nslots = gslot * s;
        const uint32_t fv = L.fbuf[lane];
        const uint32_t f = lane < nslots ? fv : 0xFFFFFFFFu;
        uint32_t g; BucketRegs h;
        probe_start(tab, f, g, h);
        probe_finish(tab, f, g, h, rsize, rpay, nsteps);
        myHits += rsize;
        nfound += rsize ? 1u : 0u;
        if (!(FUSE && single) && lane < nslots) { ws.psize[gfirst * s + lane] = rsize; ws.ppay[gfirst * s + lane] = rpay; }
    };

    for (uint32_t mate = 0; mate < 2; ++mate) {
        const uint32_t off = mate ? qi.z : qi.x;
        const uint32_t len = mate ? (noTail ? 0u : qi.w) : qi.y;
        const uint32_t nwin = windows_of(len, sp, mate == 0 && noTail);
        uint32_t wi = 0;
        while (wi < nwin) {
            const uint32_t segFirst = (len <= sp.w) ? 0u : wi * sp.stride;
            const uint32_t segChars = min(len - segFirst, kMaxWinLen);
            stage_segment(b.seq, (uint64_t)off + segFirst, segChars, L, lane);
            wave_lds_sync();
            for (; wi < nwin; ++wi, ++widx) {
                const uint32_t first = (len <= sp.w) ? 0u : wi * sp.stride;
                const uint32_t n = min(sp.w, len - first);
                const uint32_t wo = first - segFirst;
                if (wo + n > segChars) break;                // continues in the next segment
                nfeat += sketch_window_v3(L, wo, n, k, s, gslot * s, lane);
                wave_lds_sync();
                if (ws.features && lane < s) ws.features[widx * s + lane] = L.fbuf[gslot * s + lane];
                ++gslot;
                if (gslot == winsPerGroup) {
                    if (ws.psize) probe_group();
                    if (!single) { gslot = 0; gfirst = widx + 1; }
                }
            }
        }
    }
    const bool pending = single ? (gslot > 0 && gslot < winsPerGroup) : gslot > 0;
    if (pending && ws.psize) probe_group();
    if (!ws.psize) return;                                   // sketch-only use (database builder)

    const uint32_t H = wave_sum_u32(myHits);
    const uint32_t Fo = wave_sum_u32(nfound);
    const uint32_t St = wave_sum_u32(nsteps);
    bool done = false;
    if (FUSE && single) {
        if (H <= kFuseCap) {
            const uint32_t maxWin = b.maxWin ? b.maxWin[q] : b.maxWinUniform;
            fused_candidates(L, rsize, rpay, tab, H, maxWin, K, cands + (size_t)q * K, lane);
            done = true;
        } else if (lane < gslot * s) {                       // too many hits for the register path: hand over
            ws.psize[gfirst * s + lane] = rsize; ws.ppay[gfirst * s + lane] = rpay;
        }
    }
    if (lane == 0) {
        QueryStat qs; qs.hits = H; qs.nfeat = nfeat; qs.nfound = Fo; qs.nsteps = St;
        ws.qstat[q] = qs;
        ws.hitScan[q] = (!done && H <= kMaxHitsPerQuery && (wantAllhits || H > kLdsCap)) ? H : 0u;
        ws.qflag[q] = done ? kFlagDone : kFlagCands;
    }
}

// Every wave scans 64 query flags at a time and processes the queries that still need sketching.
template <bool FUSE>
__global__ __launch_bounds__(256) void query_kernel(BatchView b, SketchParams sp, DeviceTable tab, Workspace ws, int wantAllhits,
                                                    uint32_t K, mc_candidate_dev* __restrict__ cands)
{
    __shared__ FusedLds lds[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nWaves = gridDim.x * 4, waveId = blockIdx.x * 4 + wave;
    for (uint32_t base = waveId * 64; base < b.n; base += nWaves * 64) {
        const uint32_t qq = base + lane;
        const uint32_t fl = qq < b.n ? ws.qflag[qq] : kFlagDone;
        uint64_t m = __ballot(fl == kFlagSketch);
        while (m) {
            const uint32_t j = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            query_one<FUSE>(b, sp, tab, ws, wantAllhits, K, cands, lds[wave], base + j, lane);
            wave_lds_sync();
        }
    }
}

// sketch-only use of the wave path (database builder): windows -> features, nothing else
template <bool FLAGGED>
__global__ __launch_bounds__(256) void sketch_only_kernel(BatchView b, SketchParams sp, Workspace ws)
{
    __shared__ FusedLds lds[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (q >= b.n) return;
    if (FLAGGED && ws.qflag[q] != kFlagSketch) return;          // only the records the lane sketcher could not finish
    DeviceTable none{};
    Workspace w = ws;
    w.psize = nullptr;
    query_one<false>(b, sp, none, w, 0, 0, nullptr, lds[wave], q, lane);
}

void launch_sketch_only(const BatchView& b, const SketchParams& sp, const Workspace& ws, hipStream_t st)
{
    if (b.n == 0) return;
    hipLaunchKernelGGL(sketch_only_kernel<false>, dim3((b.n + 3) / 4), dim3(256), 0, st, b, sp, ws);
}

void launch_query(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, bool fuse, bool wantAllhits,
                  const Workspace& ws, uint32_t maxCand, void* cands, hipStream_t st)
{
    if (b.n == 0) return;
    dim3 grid((b.n + 255) / 256), block(256);               // one wave per 64 query flags
    if (fuse) hipLaunchKernelGGL(query_kernel<true>, grid, block, 0, st, b, sp, tab, ws, wantAllhits ? 1 : 0, maxCand, (mc_candidate_dev*)cands);
    else      hipLaunchKernelGGL(query_kernel<false>, grid, block, 0, st, b, sp, tab, ws, wantAllhits ? 1 : 0, maxCand, (mc_candidate_dev*)cands);
}

// ================================================================================================
// Lane-parallel fast path for short reads (the common case: 100-300 bp reads, small hit lists).
//
// Wave-per-query kernels spend most of their instructions on cross-lane bookkeeping that serves ONE
// query.  Short reads are better served by giving every lane its own query for the irregular parts and
// keeping only the memory-bound part cooperative:
//   sketch_lane_kernel     one lane per query : rolling canonical k-mers straight from the characters,
//                                                16-entry min/max insertion chain = the sketch
//   probe_flat_kernel      8 lanes per feature: one 128-byte bucket group per load, several in flight
//   candidates_lane_kernel one lane per query : the CPU algorithm verbatim on <= 32 locations
// Queries that do not qualify (long reads, windows that do not partition the k-mers, duplicate
// hashes inside a sketch, long hit lists, taxon merging, -allhits, K > 4) keep qflag != 0 and are
// finished by the wave-per-query kernels (query_kernel / sort_candidates_kernel), which skip the rest.
// ================================================================================================
constexpr uint32_t kLaneMaxLen = 512;     // longest mate handled by one lane
constexpr uint32_t kLaneS = 16;           // sketch entries held in registers
constexpr uint32_t kLaneHits = MC_LANE_HITS;  // longest location list handled by one lane.  The row length sets the occupancy of
                                              // probe_cands_kernel (LDS): 32 -> 8 waves/CU, 24 -> 12, 20 -> 14; measured on configs[1]
                                              // (mean list 15.8): 44.3 / 48.6 / 49.5 / 41.1 G reads/min for 32 / 24 / 20 / 16
#ifndef MC_LANE_U
#define MC_LANE_U 2
#endif
constexpr uint32_t kLaneU = MC_LANE_U;    // lookups in flight per lane

__device__ __forceinline__ void lane_encode4(uint32_t w, uint32_t& codes, uint32_t& ambs)
{
    // 4 characters -> 4 x 2-bit codes (byte j of 'codes' holds base j) and 4 ambiguity bits
    const uint32_t x = (w >> 1) & 0x03030303u;
    codes = x ^ ((x >> 1) & 0x01010101u);
    ambs = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t u = (((w >> (8 * j)) & 0xDFu) - 0x41u);
        const bool ok = u < 32u && ((0x00180045u >> u) & 1u);
        ambs |= ok ? 0u : (1u << j);
    }
}

// rows 1-5 for `len` characters at seq + off by ONE lane: rolling canonical k-mers, hash, 16-entry insertion chain; a window is complete
// after `stride` k-mers or at the end of the span (the k-mers of a sequence partition into its windows because stride = w - k + 1).
// Window sketches go to out0 + wcount * s (wcount advances); dup = a window held the same hash twice (the wave path sorts that out).
__device__ __forceinline__ void lane_sketch_span(const uint8_t* __restrict__ seq, const uint64_t off, const uint32_t len, const uint32_t k,
                                                 const uint32_t s, const uint32_t stride, uint32_t* out0, uint32_t& wcount, bool& dup)
{
    const uint32_t kmask = 0xFFFFFFFFu >> (32u - 2u * k);
    const uint32_t rcshift = 2u * k - 2u;
    uint32_t sk[kLaneS];
#pragma unroll
    for (uint32_t i = 0; i < kLaneS; ++i) sk[i] = 0xFFFFFFFFu;
    uint32_t fwd = 0, rc = 0, since = 0, wpos = 0;
    const uint4* src = reinterpret_cast<const uint4*>(seq + off);       // sequences start 4-byte aligned
    uint4 nxt = src[0];
    for (uint32_t j0 = 0; j0 < len; j0 += 16) {
        const uint4 cur = nxt;
        if (j0 + 16 < len) nxt = src[(j0 >> 4) + 1];
        const uint32_t wd[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
            uint32_t codes, ambs;
            lane_encode4(wd[d], codes, ambs);
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) {
                const uint32_t j = j0 + d * 4 + t;
                if (j < len) {
                    const uint32_t c = (codes >> (8 * t)) & 3u;
                    const bool a = (ambs >> t) & 1u;
                    fwd = ((fwd << 2) | c) & kmask;
                    rc = (rc >> 2) | ((3u - c) << rcshift);
                    since = a ? 0u : since + 1u;
                    if (j + 1 >= k) {
                        uint32_t h = 0xFFFFFFFFu;
                        if (since >= k) h = tm_hash(fwd < rc ? fwd : rc);
                        // insertion chain: sk stays sorted ascending, the largest value falls out
#pragma unroll
                        for (uint32_t i = 0; i < kLaneS; ++i) {
                            const uint32_t lo = min(sk[i], h);
                            h = max(sk[i], h);
                            sk[i] = lo;
                        }
                        ++wpos;
                        if (wpos == stride || j + 1 == len) {              // window complete (row 1: k-mers partition)
                            uint32_t* out = out0 + (size_t)wcount * s;
#pragma unroll
                            for (uint32_t i = 0; i < kLaneS; ++i) {
                                if (i < s) out[i] = sk[i];
                                if (i + 1 < s) dup = dup || (sk[i] == sk[i + 1] && sk[i] != 0xFFFFFFFFu);
                                sk[i] = 0xFFFFFFFFu;
                            }
                            ++wcount; wpos = 0;
                        }
                    }
                }
            }
        }
    }
}

// k = 16 and a stride that is a multiple of 16 (the defaults: 112): the same multiset of the s smallest hashes per window, selected in
// batches.  The insertion chain above spends 32 min/max per k-mer; sorting 16 hashes with Batcher's odd-even merge network (63
// compare-exchanges, checked on all 2^16 0/1 inputs) and merging the sorted batch into the sorted sketch (16 min + a 32-exchange bitonic
// merge) costs 13 per k-mer.  Character t of a 16-character step completes k-mer (step * 16 + t - 15): slot (t + 1) % 16 of a batch, so
// every register index is a compile-time constant; windows end on batch boundaries.
__device__ __forceinline__ void sort16(uint32_t (&x)[16])
{
    auto ce = [&](const uint32_t i, const uint32_t j) { const uint32_t lo = min(x[i], x[j]); x[j] = max(x[i], x[j]); x[i] = lo; };
    ce(0, 1); ce(2, 3); ce(0, 2); ce(1, 3); ce(1, 2); ce(4, 5); ce(6, 7); ce(4, 6); ce(5, 7); ce(5, 6); ce(0, 4); ce(2, 6); ce(2,
    4); ce(1, 5); ce(3, 7); ce(3, 5); ce(1, 2); ce(3, 4); ce(5, 6); ce(8, 9); ce(10, 11); ce(8, 10); ce(9, 11); ce(9, 10); ce(12,
    13); ce(14, 15); ce(12, 14); ce(13, 15); ce(13, 14); ce(8, 12); ce(10, 14); ce(10, 12); ce(9, 13); ce(11, 15); ce(11, 13);
    ce(9, 10); ce(11, 12); ce(13, 14); ce(0, 8); ce(4, 12); ce(4, 8); ce(2, 10); ce(6, 14); ce(6, 10); ce(2, 4); ce(6, 8); ce(10,
    12); ce(1, 9); ce(5, 13); ce(5, 9); ce(3, 11); ce(7, 15); ce(7, 11); ce(3, 5); ce(7, 9); ce(11, 13); ce(1, 2); ce(3, 4); ce(5,
    6); ce(7, 8); ce(9, 10); ce(11, 12); ce(13, 14);
}
// sk (ascending) <- the 16 smallest of sk and x (ascending), ascending
__device__ __forceinline__ void merge16(uint32_t (&sk)[16], const uint32_t (&x)[16])
{
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) sk[i] = min(sk[i], x[15 - i]);               // bitonic: falls, then rises
#pragma unroll
    for (uint32_t dist = 8; dist > 0; dist >>= 1)
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i)
            if ((i & dist) == 0) { const uint32_t lo = min(sk[i], sk[i + dist]); sk[i + dist] = max(sk[i], sk[i + dist]); sk[i] = lo; }
}

__device__ __forceinline__ void lane_sketch_span16(const uint8_t* __restrict__ seq, const uint64_t off, const uint32_t len, const uint32_t s,
                                                   const uint32_t stride, uint32_t* out0, uint32_t& wcount, bool& dup)
{
    static_assert(kLaneS == 16, "batch = sketch registers");
    uint32_t sk[16], hb[16];
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) { sk[i] = 0xFFFFFFFFu; hb[i] = 0xFFFFFFFFu; }
    uint32_t fwd = 0, rc = 0, since = 0, wpos = 0;
    bool pending = false;                                               // hb holds hashes that are not in sk yet
    auto flush = [&]() {
        sort16(hb);
        merge16(sk, hb);
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) hb[i] = 0xFFFFFFFFu;
        pending = false;
    };
    auto emit = [&]() {
        uint32_t* out = out0 + (size_t)wcount * s;
        if (s == 16) {                                                  // a window's sketch = one 64-byte line: four 16-byte stores (every
            uint4* o4 = reinterpret_cast<uint4*>(out);                  // lane writes its own line, so 4 instead of 16 requests per line)
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) o4[i] = make_uint4(sk[4 * i], sk[4 * i + 1], sk[4 * i + 2], sk[4 * i + 3]);
        }
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) {
            if (s != 16 && i < s) out[i] = sk[i];
            if (i + 1 < s) dup = dup || (sk[i] == sk[i + 1] && sk[i] != 0xFFFFFFFFu);
            sk[i] = 0xFFFFFFFFu;
        }
        ++wcount; wpos = 0;
    };
    const uint4* src = reinterpret_cast<const uint4*>(seq + off);       // sequences start 4-byte aligned
    uint4 nxt = src[0];
    // one step = 16 characters.  GUARD = false: every lane of the wave is inside its sequence and past its first 15 characters, so
    // the per-character tests (and their exec-mask bookkeeping) are gone; first and last steps take the guarded form.
    auto step = [&](const uint32_t j0, const uint4 cur, auto guard) {
        constexpr bool GUARD = decltype(guard)::value;
        const uint32_t wd[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
            uint32_t codes, ambs;
            lane_encode4(wd[d], codes, ambs);
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) {
                const uint32_t dt = d * 4 + t, j = j0 + dt;
                if (!GUARD || j < len) {
                    const uint32_t c = (codes >> (8 * t)) & 3u;
                    const bool a = (ambs >> t) & 1u;
                    fwd = (fwd << 2) | c;
                    rc = (rc >> 2) | ((3u - c) << 30);
                    since = a ? 0u : since + 1u;
                    if (!GUARD || j >= 15) {
                        hb[(dt + 1) & 15] = since >= 16 ? tm_hash(fwd < rc ? fwd : rc) : 0xFFFFFFFFu;
                        pending = true;
                        ++wpos;
                    }
                }
                if (dt == 14) {                                         // the batch that began with the previous step's last character is complete
                    if (!GUARD || pending) flush();
                    if (wpos == stride) emit();                         // window complete (row 1: the k-mers partition)
                }
            }
        }
    };
    for (uint32_t j0 = 0; j0 < len; j0 += 16) {
        const uint4 cur = nxt;
        if (j0 + 16 < len) nxt = src[(j0 >> 4) + 1];
        const bool inside = j0 >= 16 && j0 + 16 <= len;
        if (__all(inside)) step(j0, cur, std::false_type{});
        else step(j0, cur, std::true_type{});
    }
    if (wpos > 0) { if (pending) flush(); emit(); }                     // tail window
}

__device__ __forceinline__ uint32_t sketch_lane_one(const BatchView& b, const SketchParams& sp, const uint32_t* __restrict__ winOff,
                                                    uint32_t* features, const uint32_t q)
{
    const uint4 qi = reinterpret_cast<const uint4*>(b.qinfo)[q];
    const uint32_t k = sp.k, s = sp.s, stride = sp.stride;
    const bool noTail = qi.w == kNoTail;
    const uint32_t widx0 = winOff[q];
    if (noTail || qi.y > kLaneMaxLen || qi.w > kLaneMaxLen) {
        const uint32_t nw = winOff[q + 1] - widx0;
        for (uint32_t i = 0; i < nw * s; ++i) features[(size_t)widx0 * s + i] = 0xFFFFFFFFu;   // nothing to probe here
        return kFlagSketch;
    }
    bool dup = false;
    uint32_t wcount = 0;
    for (uint32_t mate = 0; mate < 2; ++mate) {
        const uint32_t off = mate ? qi.z : qi.x;
        const uint32_t len = mate ? qi.w : qi.y;
        if (len < k) continue;
        if (k == 16 && (stride & 15u) == 0) lane_sketch_span16(b.seq, off, len, s, stride, features + (size_t)widx0 * s, wcount, dup);
        else lane_sketch_span(b.seq, off, len, k, s, stride, features + (size_t)widx0 * s, wcount, dup);
    }
    return dup ? kFlagSketch : kFlagProbe;
}

constexpr uint32_t kChunkWins = MC_CHUNK_WINS;   // windows per chunk lane of a long read

// Long single reads (> kLaneMaxLen) are cut into chunks of kChunkWins windows; every chunk is sketched and probed by its own
// lane (chunk_sketch_kernel / chunk_probe_kernel), so a 19 000 bp read keeps 43 lanes busy instead of one wave for 170 windows.
// Here: the read's lane appends its chunk records {query, chunk} to the work list (one atomic per wave).
// (the wave's lanes with long reads append their chunk records to the work list; returns this lane's number of chunks -- 0: a read of its own)
__device__ __forceinline__ uint32_t lane_chunk_records(const BatchView& b, const SketchParams& sp, const Workspace& ws, const uint32_t q, const uint32_t lane)
{
    const bool chunkable = (sp.stride & 3u) == 0 && ws.chunkList != nullptr;    // chunk starts stay 4-byte aligned
    uint32_t nch = 0;
    if (q < b.n && chunkable) {
        const uint4 qi = reinterpret_cast<const uint4*>(b.qinfo)[q];
        if (qi.w == 0 && qi.y > kLaneMaxLen) nch = (ws.winOff[q + 1] - ws.winOff[q] + kChunkWins - 1) / kChunkWins;
    }
    const uint32_t incl = wave_incl_scan_u32(nch, lane);
    const uint32_t total = rdlane(incl, 63);
    if (total) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ws.midCount[5], total);
        base = rdlane(base, 0);
        // the wave writes its records together (64 consecutive ones per round); record r belongs to the first lane with incl > r
        for (uint32_t r0 = 0; r0 < total; r0 += 64) {
            const uint32_t r = r0 + lane;
            uint32_t lo = 0, hi = 63;
#pragma unroll
            for (uint32_t it = 0; it < 6; ++it) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t v = __shfl(incl, mid);
                if (v > r) hi = mid; else lo = mid + 1;
            }
            const uint32_t owner = min(lo, 63u);
            const uint32_t first = __shfl(incl, owner) - __shfl(nch, owner);
            const uint32_t oq = __shfl(q, owner);
            if (r < total) ws.chunkList[base + r] = make_uint2(oq, r - first);
        }
    }
    return nch;
}
__global__ __launch_bounds__(256) void sketch_lane_kernel(BatchView b, SketchParams sp, Workspace ws)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nch = lane_chunk_records(b, sp, ws, q, lane);
    if (q >= b.n) return;
    if (nch) {
        ws.qflag[q] = kFlagChunks;
        return;
    }
    ws.qflag[q] = sketch_lane_one(b, sp, ws.winOff, ws.features, q);
}

// one lane per chunk: the window sketches of its <= kChunkWins windows
__global__ __launch_bounds__(128) void chunk_sketch_kernel(BatchView b, SketchParams sp, Workspace ws)
{
    const uint32_t total = ws.midCount[5];
    for (uint32_t id = blockIdx.x * 128 + threadIdx.x; id < total; id += gridDim.x * 128) {
        const uint2 rec = ws.chunkList[id];
        const uint32_t q = rec.x, c = rec.y;
        const uint4 qi = reinterpret_cast<const uint4*>(b.qinfo)[q];
        const uint32_t p0 = c * kChunkWins * sp.stride;                            // first k-mer of the chunk
        const uint32_t len = min(qi.y - p0, kChunkWins * sp.stride + sp.k - 1);
        uint32_t wcount = 0; bool dup = false;
        const uint32_t w0 = ws.winOff[q] + c * kChunkWins, w1 = min(ws.winOff[q + 1], w0 + kChunkWins);
        if ((sp.s & 3u) == 0) {                                                    // chunk_probe_kernel writes the found features only
            uint4* z = reinterpret_cast<uint4*>(ws.psize + (size_t)w0 * sp.s);
            for (uint32_t i = 0; i < (w1 - w0) * sp.s / 4; ++i) z[i] = make_uint4(0, 0, 0, 0);
        } else for (uint32_t i = w0 * sp.s; i < w1 * sp.s; ++i) ws.psize[i] = 0u;
        if (sp.k == 16 && (sp.stride & 15u) == 0) lane_sketch_span16(b.seq, (uint64_t)qi.x + p0, len, sp.s, sp.stride, ws.features + (size_t)w0 * sp.s, wcount, dup);
        else lane_sketch_span(b.seq, (uint64_t)qi.x + p0, len, sp.k, sp.s, sp.stride, ws.features + (size_t)w0 * sp.s, wcount, dup);
        if (dup) ws.qflag[q] = kFlagSketch;                                         // the wave kernel redoes the whole read
    }
}

// Database builder: the chunk records of target sequences (<= kBuildRecWins windows each, window-aligned, tail only in a target's last
// record) sketched by lanes: ONE WAVE per record, lane l takes the windows [l * kBuildLaneWins, (l + 1) * kBuildLaneWins) -- the k-mers
// of a sequence partition into its windows (stride = w - k + 1), so a lane only needs its own characters.  A window that holds the
// same hash twice flags its record: sketch_only_kernel<true> redoes those records with the exact wave path.
constexpr uint32_t kBuildLaneWins = 4, kBuildRecWins = 64 * kBuildLaneWins;
__global__ __launch_bounds__(256) void build_sketch_lanes_kernel(BatchView b, SketchParams sp, Workspace ws)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= b.n) return;
    const uint4 qi = reinterpret_cast<const uint4*>(b.qinfo)[q];
    const uint32_t w0 = ws.winOff[q], nw = ws.winOff[q + 1] - w0;
    const uint32_t first = lane * kBuildLaneWins;
    bool dup = false;
    if (first < nw) {
        const uint32_t p0 = first * sp.stride;
        // (a target's last record holds up to kBuildRecWins full windows PLUS the tail window: the last lane takes that one too)
        const uint32_t len = lane == 63u ? qi.y - p0 : min(qi.y - p0, kBuildLaneWins * sp.stride + sp.k - 1);
        uint32_t wcount = 0;
        uint32_t* out = ws.features + (size_t)(w0 + first) * sp.s;
        if (sp.k == 16 && (sp.stride & 15u) == 0) lane_sketch_span16(b.seq, (uint64_t)qi.x + p0, len, sp.s, sp.stride, out, wcount, dup);
        else lane_sketch_span(b.seq, (uint64_t)qi.x + p0, len, sp.k, sp.s, sp.stride, out, wcount, dup);
    }
    const bool any = __ballot(dup) != 0;
    if (lane == 0) ws.qflag[q] = any ? kFlagSketch : kFlagDone;
}

template <bool FLAGGED>
__global__ __launch_bounds__(256) void sketch_only_kernel(BatchView b, SketchParams sp, Workspace ws);

void launch_build_sketch(const BatchView& b, const SketchParams& sp, const Workspace& ws, hipStream_t st)
{
    if (b.n == 0) return;
    if (lane_path_supported(sp) && (sp.stride & 3u) == 0 && ws.qflag) {
        hipLaunchKernelGGL(build_sketch_lanes_kernel, dim3((b.n + 3) / 4), dim3(256), 0, st, b, sp, ws);
        hipLaunchKernelGGL(sketch_only_kernel<true>, dim3((b.n + 3) / 4), dim3(256), 0, st, b, sp, ws);
    } else launch_sketch_only(b, sp, ws, st);
}
uint32_t build_record_windows() { return kBuildRecWins; }

// one lane per chunk: lookups of its <= kChunkWins * s features (kLaneU in flight), (size, payload) of the found features for the wave
// kernel.  QUAD as in probe_cands_one.
template <bool QUAD>
__global__ __launch_bounds__(128) void chunk_probe_kernel(BatchView b, uint32_t s, DeviceTable tab, Workspace ws)
{
    const uint32_t total = ws.midCount[5];
    for (uint32_t base = blockIdx.x * 128; base < total; base += gridDim.x * 128) {   // block-uniform: quads stay together
        const uint32_t id = base + threadIdx.x;
        uint2 rec = make_uint2(0, 0);
        if (id < total) rec = ws.chunkList[id];
        const uint32_t q = rec.x, c = rec.y;
        const bool valid = id < total && ws.qflag[q] == kFlagChunks;                 // not: a chunk of the read saw duplicate hashes
        const uint32_t w0 = ws.winOff[q] + c * kChunkWins, w1 = min(ws.winOff[q + 1], w0 + kChunkWins);
        const uint32_t fbase = w0 * s, nf = valid ? (w1 - w0) * s : 0u;
        const uint32_t* feats = ws.features + fbase;
        uint32_t e = 0;
        uint4 fcache = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        const bool wide = (s & 3u) == 0;
        uint32_t f[kLaneU], home[kLaneU], cur[kLaneU], step[kLaneU], slot[kLaneU];
        QuadRaw raw[kLaneU];
        bool busy[kLaneU];
#pragma unroll
        for (uint32_t u = 0; u < kLaneU; ++u) busy[u] = false;
        for (bool first = true;; first = false) {
#pragma unroll
            for (uint32_t u = 0; u < kLaneU; ++u) {
                if (first || !__ballot(busy[u])) continue;
                BucketRegs r;
                if constexpr (QUAD) r = quad_collect(raw[u]);
                else { r.k = raw[u].v[0]; r.sz = raw[u].v[1]; r.p0 = raw[u].v[2]; r.p1 = raw[u].v[3]; }
                if (busy[u]) {
                    const uint32_t keys[4] = {r.k.x, r.k.y, r.k.z, r.k.w};
                    const uint32_t s01 = r.sz.x, s23 = r.sz.y;
                    const uint32_t sz[4] = {s01 & 0xFFFFu, s01 >> 16, s23 & 0xFFFFu, s23 >> 16};
                    const uint64_t pl[4] = {((uint64_t)r.p0.y << 32) | r.p0.x, ((uint64_t)r.p0.w << 32) | r.p0.z,
                                            ((uint64_t)r.p1.y << 32) | r.p1.x, ((uint64_t)r.p1.w << 32) | r.p1.z};
                    uint32_t size = 0; uint64_t pay = 0;
                    bool anyFree = false;
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) {
                        anyFree = anyFree || sz[i] == 0;
                        if (sz[i] != 0 && keys[i] == f[u]) { size = sz[i]; pay = pl[i]; }
                    }
                    if (size || anyFree || step[u] >= tab.maxProbe) {
                        if (size) { ws.psize[fbase + slot[u]] = size; ws.ppay[fbase + slot[u]] = pay; }
                        busy[u] = false;
                    } else {
                        cur[u] = next_bucket(home[u], cur[u], step[u], tab.nbuckets);
                        ++step[u];
                    }
                }
            }
            bool any = false;
#pragma unroll
            for (uint32_t u = 0; u < kLaneU; ++u) {
                while (!busy[u] && e < nf) {
                    slot[u] = e;
                    if (wide) {                                   // four features per request
                        if ((e & 3u) == 0) fcache = reinterpret_cast<const uint4*>(feats)[e >> 2];
                        const uint32_t i4 = e++ & 3u;
                        f[u] = i4 == 0 ? fcache.x : i4 == 1 ? fcache.y : i4 == 2 ? fcache.z : fcache.w;
                    } else f[u] = feats[e++];
                    if (f[u] != 0xFFFFFFFFu) {
                        home[u] = home_group(f[u], tab.nbuckets);
                        cur[u] = home[u]; step[u] = 1;
                        busy[u] = true;
                    }
                }
                any = any || busy[u];
            }
            if (!__ballot(any)) break;
#pragma unroll
            for (uint32_t u = 0; u < kLaneU; ++u) {
                if (!__ballot(busy[u])) continue;
                if constexpr (QUAD) quad_issue(tab, busy[u] ? cur[u] : kNoBucket, raw[u]);
                else if (busy[u]) { const BucketRegs r = load_bucket(tab, cur[u]); raw[u].v[0] = r.k; raw[u].v[1] = r.sz; raw[u].v[2] = r.p0; raw[u].v[3] = r.p1; }
            }
        }
    }
}

// one wave per long read (found through its first chunk record): totals over the read's feature slots, hand-over to the wave kernel.
// (Collecting the totals with atomics in chunk_probe_kernel cost more than this pass; a per-lane device-scope fence for a
// "last chunk finishes the read" scheme cost far more: the L2s of the 8 XCDs are written back and invalidated every time.
// Compacting short lists here for mid_cands_kernel was measured too: what the wave kernel saves, the compaction costs.)
// Tables with the compact location store: the read goes on the filtered path's work list (gw_filter_stream_kernel reads its feature
// slots, found or not, as the entries) instead of the wave kernel's sort of everything.
__global__ __launch_bounds__(256) void chunk_finish_kernel(uint32_t s, Workspace ws, BatchView b, DeviceTable tab)
{
    const uint32_t total = ws.midCount[5];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nWaves = gridDim.x * 4, waveId = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (uint32_t base = waveId * 64; base < total; base += nWaves * 64) {
        const uint32_t id = base + lane;
        uint2 rec = make_uint2(0, 1);
        if (id < total) rec = ws.chunkList[id];
        const bool mineRead = rec.y == 0 && ws.qflag[rec.x] == kFlagChunks;       // first chunk records = one per long read
        uint64_t m = __ballot(mineRead);
        uint32_t myH = 0, myFound = 0, myFeat = 0;                                  // lane j: the sums of ITS read
        while (m) {
            const uint32_t j = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const uint32_t q = rdlane(rec.x, j);
            uint32_t H = 0, nfound = 0, nfeat = 0;
            for (uint32_t i = ws.winOff[q] * s + lane; i < ws.winOff[q + 1] * s; i += 64) {   // the whole wave sums the read's slots
                const uint32_t sz = ws.psize[i] & 0xFFFFu;
                H += sz; nfound += sz ? 1u : 0u; nfeat += ws.features[i] != 0xFFFFFFFFu ? 1u : 0u;
            }
            H = wave_sum_u32(H); nfound = wave_sum_u32(nfound); nfeat = wave_sum_u32(nfeat);
            if (lane == j) { myH = H; myFound = nfound; myFeat = nfeat; }
        }
        // every lane finishes its own read; the work list's places are reserved with ONE atomic per wave and step (one per read --
        // 10^5 on one counter in a batch of long reads -- took 2.5 ms)
        bool filtered = false, wide = false;
        const uint32_t q = rec.x;
        uint32_t slots = 0, mw = 0;
        if (mineRead) {
            const uint32_t H = myH;
            QueryStat qs; qs.hits = H; qs.nfeat = myFeat; qs.nfound = myFound; qs.nsteps = myFeat;   // probe steps are not counted on this path
            ws.qstat[q] = qs;
            slots = (ws.winOff[q + 1] - ws.winOff[q]) * s; mw = b.maxWin ? b.maxWin[q] : b.maxWinUniform;
            filtered = tab.values32 && !ws.partialLists && H > ws.bigMin && H > 64u && H <= kMaxHitsPerQuery && slots <= 0xFFFu && mw <= tab.gwGap;
            wide = filtered && H > kGwSmallH;
            // (a key shard's partial lists are wanted as they are: gather_lists_kernel copies them -- no sort of tens of thousands of locations)
            ws.hitScan[q] = filtered ? 0u : ((H <= kMaxHitsPerQuery && (H > kLdsCap || ws.partialLists)) ? H : 0u);   // (lists wanted: every query gets its segment)
            ws.qflag[q] = filtered ? kFlagMid : ws.partialLists ? kFlagGatherAll : kFlagCands;
        }
        const uint64_t fm = __ballot(filtered);
        if (fm) {
            const uint32_t leader = __ffsll((unsigned long long)fm) - 1;
            uint32_t at = 0;
            if (lane == leader) at = atomicAdd(&ws.midCount[9], (uint32_t)__popcll(fm));
            at = __shfl(at, leader);
            if (filtered)
                reinterpret_cast<uint4*>(ws.midList)[(size_t)6 * b.n + at + __popcll(fm & ((1ull << lane) - 1ull))] = make_uint4(q, ws.winOff[q] * s, slots | (myH << 12), mw);
            const uint64_t wm = __ballot(wide);
            if (wm && lane == leader) atomicAdd(&ws.midCount[10], (uint32_t)__popcll(wm));
        }
    }
}

// probe_cands_kernel: ONE LANE PER QUERY for rows 6-10.
//   rows 6-7: the lane walks through its features (16-byte loads of its own 128-byte feature block), kLaneU
//             lookups in flight (probe_start / probe_finish: three 16-byte loads of the bucket group's line,
//             8-byte payload on a hit); hits go to the lane's private LDS row (<= kLaneHits locations);
//   row 8   : insertion sort of the row;
//   rows 9-10: the CPU's sequential window-range scan and a shifting top-K insert -- verbatim, so the tie
//             behaviour is identical by construction.
// Why not 8 lanes per bucket group (the first version of this kernel): lane-private lookups reach the same
// random-line rate (tools/gather_bench3.hip: 57 G lines/s vs 55 G lines/s cooperative) with a tenth of the
// instructions -- no ballots, no cross-lane compaction, no per-query serial section.
// Lists longer than kLaneHits: second pass that stores (size, payload) per feature for sort_candidates_kernel.
constexpr uint32_t kLaneRow = kLaneHits + 1;                  // odd stride (in u64): conflict-free lane-private rows
constexpr uint32_t kLaneBlock = 128;

// QUAD: quad-cooperative bucket fetches (tables beyond the reach of the infinity cache / TLBs), else lane-private 4 x 16-byte loads
// DIRECT: the lookups go to the direct-address index (DeviceTable::direct): one 8-byte load per feature, eight in flight per lane
template <bool QUAD, bool DIRECT = false>
__device__ __forceinline__ void probe_cands_one(const BatchView& b, const uint32_t s, const DeviceTable& tab, const Workspace& ws, const uint32_t K,
                                                const uint32_t* __restrict__ taxkey, mc_candidate_dev* __restrict__ cands, const uint32_t q,
                                                uint64_t* L, const bool valid)
{
    // lanes without a query of their own (valid == false) stay in the wave: they help with the cooperative hand-over below
    const uint32_t fbase = valid ? ws.winOff[q] * s : 0u, nf = valid ? (ws.winOff[q + 1] - ws.winOff[q]) * s : 0u;
    const uint32_t* feats = ws.features + fbase;

    // kLaneU lookup slots per lane, each a little state machine: a slot takes the lane's next feature and
    // issues the load of its home bucket; when the bucket arrives the lookup either ends (found / free slot
    // seen) or issues the load of the next bucket of its chain and stays pending.  One memory round trip per
    // loop iteration, no inner waits: a long chain delays only its own slot, not the wave.
    // (Measured alternatives at the same 1.2 ms: 8 lanes per 128-byte group with ballots; cooperative 4-lane
    // fetch of the bucket handed to its owner through LDS.)
    uint32_t H = 0, n = 0, m = 0, nfeat = 0, nfound = 0, nsteps = 0;
    bool over = false;
    uint32_t gnent = 0, goff = 0;                                // hand-over area: entries written, locations they stand for
    // hand-over entry j = (size | start offset in the list << 16, payload); found features only, singletons first
    // entries written by the lane itself go out two at a time (8 + 16 bytes): every store of a lane is a request of its own
    uint32_t heldSize = 0; uint64_t heldPay = 0;                 // the even entry of the pair being formed
    const bool pairs = (fbase & 1u) == 0;
    auto put_entry = [&](uint32_t ps, uint64_t pp) {
        if (!pairs) { ws.psize[fbase + gnent] = ps; ws.ppay[fbase + gnent] = pp; }
        else if ((gnent & 1u) == 0) { heldSize = ps; heldPay = pp; }
        else {
            *reinterpret_cast<uint2*>(ws.psize + fbase + gnent - 1) = make_uint2(heldSize, ps);
            *reinterpret_cast<uint4*>(ws.ppay + fbase + gnent - 1) = make_uint4((uint32_t)heldPay, (uint32_t)(heldPay >> 32), (uint32_t)pp, (uint32_t)(pp >> 32));
        }
        ++gnent;
    };
    auto flush_entries = [&]() { if (pairs && (gnent & 1u)) { ws.psize[fbase + gnent - 1] = heldSize; ws.ppay[fbase + gnent - 1] = heldPay; } };
    auto dump_row = [&]() {
        for (uint32_t j = 0; j < n; ++j) put_entry(1u | (j << 16), L[j]);
        goff = n;
        for (uint32_t i = 0; i < m; ++i) {
            const uint64_t d = L[kLaneHits - i];
            const uint32_t size = (uint32_t)(d >> 48);
            put_entry(size | (goff << 16), d & 0xFFFFFFFFFFFFull);
            goff += size;
        }
    };
    // the lane's features arrive four at a time (s % 4 == 0: sketches are 16-byte aligned): every lane reads its own line, so what
    // counts is the number of requests, not their width
    uint32_t e = 0;                                                  // next feature of this lane
    uint4 fcache = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    const bool wide = (s & 3u) == 0;
    auto next_feature = [&]() -> uint32_t {
        if (!wide) return feats[e++];
        if ((e & 3u) == 0) fcache = reinterpret_cast<const uint4*>(feats)[e >> 2];
        const uint32_t i = e++ & 3u;
        return i == 0 ? fcache.x : i == 1 ? fcache.y : i == 2 ? fcache.z : fcache.w;
    };
    if constexpr (DIRECT) {
        constexpr uint32_t kDirectU = 8;
        while (e < nf) {
            uint32_t f[kDirectU]; uint64_t ent[kDirectU];
    #pragma unroll
            for (uint32_t u = 0; u < kDirectU; ++u) {
                f[u] = 0xFFFFFFFFu; ent[u] = 0;
                if (e < nf) f[u] = next_feature();
                if (f[u] != 0xFFFFFFFFu) { ++nfeat; ent[u] = tab.direct[f[u]]; }
            }
    #pragma unroll
            for (uint32_t u = 0; u < kDirectU; ++u) {
                if (f[u] == 0xFFFFFFFFu) continue;
                ++nsteps;
                const uint32_t size = (uint32_t)ent[u] & 0xFFFFu;
                if (!size) continue;
                const uint64_t pay = direct_payload(ent[u]);
                ++nfound;
                H += size;
                if (!over && n + m >= kLaneHits) { dump_row(); over = true; }      // (as below: the row moves to the hand-over area)
                if (over) { put_entry(size | (goff << 16), pay); goff += size; }
                else if (size == 1) L[n++] = pay;
                else { L[kLaneHits - m] = pay | ((uint64_t)size << 48); ++m; }
            }
        }
    } else if constexpr (QUAD) {
        uint32_t f[kLaneU], home[kLaneU], cur[kLaneU], step[kLaneU];
        QuadRaw raw[kLaneU];
        bool busy[kLaneU];
    #pragma unroll
        for (uint32_t u = 0; u < kLaneU; ++u) busy[u] = false;
        // Every round: (1) resolve the lookups whose buckets were requested in the previous round, (2) hand idle slots their next
        // feature, (3) request the bucket every busy slot needs now.  The requests are quad-cooperative (quad_issue), so all branches
        // around them are wave-uniform; a lane that has run out of features idles along until the wave is done.
        for (bool first = true;; first = false) {
    #pragma unroll
            for (uint32_t u = 0; u < kLaneU; ++u) {
                if (first || !__ballot(busy[u])) continue;
                const BucketRegs r = quad_collect(raw[u]);
                if (busy[u]) {
                    ++nsteps;
                    const uint32_t keys[4] = {r.k.x, r.k.y, r.k.z, r.k.w};
                    const uint32_t s01 = r.sz.x, s23 = r.sz.y;
                    const uint32_t sz[4] = {s01 & 0xFFFFu, s01 >> 16, s23 & 0xFFFFu, s23 >> 16};
                    const uint64_t pl[4] = {((uint64_t)r.p0.y << 32) | r.p0.x, ((uint64_t)r.p0.w << 32) | r.p0.z,
                                            ((uint64_t)r.p1.y << 32) | r.p1.x, ((uint64_t)r.p1.w << 32) | r.p1.z};
                    uint32_t size = 0; uint64_t pay = 0;
                    bool anyFree = false;
    #pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) {
                        anyFree = anyFree || sz[i] == 0;
                        if (sz[i] != 0 && keys[i] == f[u]) { size = sz[i]; pay = pl[i]; }
                    }
                    if (size) {
                        ++nfound;
                        H += size;
                        // singletons are the location itself; longer lists are only noted here (descriptor = first index |
                        // size << 48, kept at the END of the row, growing downwards) and fetched after the lookups
                        if (!over && n + m >= kLaneHits) { dump_row(); over = true; }   // row full (pairs, rich tables): it moves to
                        if (over) {                                                      // the hand-over area, later entries go there directly
                            put_entry(size | (goff << 16), pay);
                            goff += size;
                        }
                        else if (size == 1) L[n++] = pay;
                        else { L[kLaneHits - m] = pay | ((uint64_t)size << 48); ++m; }
                        busy[u] = false;
                    } else if (anyFree || step[u] >= tab.maxProbe) {
                        busy[u] = false;                               // a bucket with a free slot ends the chain
                    } else {
                        cur[u] = next_bucket(home[u], cur[u], step[u], tab.nbuckets);
                        ++step[u];                                     // stays busy: its next bucket is requested below
                    }
                }
            }
            bool any = false;
    #pragma unroll
            for (uint32_t u = 0; u < kLaneU; ++u) {
                while (!busy[u] && e < nf) {
                    f[u] = next_feature();
                    if (f[u] != 0xFFFFFFFFu) {
                        ++nfeat;
                        home[u] = home_group(f[u], tab.nbuckets);
                        cur[u] = home[u]; step[u] = 1;
                        busy[u] = true;
                    }
                }
                any = any || busy[u];
            }
            if (!__ballot(any)) break;
    #pragma unroll
            for (uint32_t u = 0; u < kLaneU; ++u)
                if (__ballot(busy[u])) quad_issue(tab, busy[u] ? cur[u] : kNoBucket, raw[u]);
        }
    } else {
        uint32_t f[kLaneU], home[kLaneU], cur[kLaneU], step[kLaneU];
        BucketRegs r[kLaneU];
        bool busy[kLaneU];
    #pragma unroll
        for (uint32_t u = 0; u < kLaneU; ++u) busy[u] = false;
        for (;;) {
            bool any = false;
    #pragma unroll
            for (uint32_t u = 0; u < kLaneU; ++u) {
                if (!busy[u] && e < nf) {
                    f[u] = next_feature();
                    if (f[u] != 0xFFFFFFFFu) {
                        ++nfeat;
                        home[u] = home_group(f[u], tab.nbuckets);
                        cur[u] = home[u]; step[u] = 1;
                        r[u] = load_bucket(tab, cur[u]);
                        busy[u] = true;
                    }
                }
                any = any || busy[u];
            }
            if (!any && e >= nf) break;
    #pragma unroll
            for (uint32_t u = 0; u < kLaneU; ++u) {
                if (busy[u]) {
                    ++nsteps;
                    const uint32_t keys[4] = {r[u].k.x, r[u].k.y, r[u].k.z, r[u].k.w};
                    const uint32_t s01 = r[u].sz.x, s23 = r[u].sz.y;
                    const uint32_t sz[4] = {s01 & 0xFFFFu, s01 >> 16, s23 & 0xFFFFu, s23 >> 16};
                    const uint64_t pl[4] = {((uint64_t)r[u].p0.y << 32) | r[u].p0.x, ((uint64_t)r[u].p0.w << 32) | r[u].p0.z,
                                            ((uint64_t)r[u].p1.y << 32) | r[u].p1.x, ((uint64_t)r[u].p1.w << 32) | r[u].p1.z};
                    uint32_t size = 0; uint64_t pay = 0;
                    bool anyFree = false;
    #pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) {
                        anyFree = anyFree || sz[i] == 0;
                        if (sz[i] != 0 && keys[i] == f[u]) { size = sz[i]; pay = pl[i]; }
                    }
                    if (size) {
                        ++nfound;
                        H += size;
                        // singletons are the location itself; longer lists are only noted here (descriptor = first index |
                        // size << 48, kept at the END of the row, growing downwards) and fetched after the lookups
                        if (!over && n + m >= kLaneHits) { dump_row(); over = true; }   // row full (pairs, rich tables): it moves to
                        if (over) {                                                      // the hand-over area, later entries go there directly
                            put_entry(size | (goff << 16), pay);
                            goff += size;
                        }
                        else if (size == 1) L[n++] = pay;
                        else { L[kLaneHits - m] = pay | ((uint64_t)size << 48); ++m; }
                        busy[u] = false;
                    } else if (anyFree || step[u] >= tab.maxProbe) {
                        busy[u] = false;                               // a bucket with a free slot ends the chain
                    } else {
                        cur[u] = next_bucket(home[u], cur[u], step[u], tab.nbuckets);
                        ++step[u];
                        r[u] = load_bucket(tab, cur[u]);               // stays pending; resolved in the next iteration
                    }
                }
            }
        }
    }
    if (over) flush_entries();
    if (valid) { QueryStat qs; qs.hits = H; qs.nfeat = nfeat; qs.nfound = nfound; qs.nsteps = nsteps; ws.qstat[q] = qs; }
    // Lists too long for a lane go to the mid / wave kernels as (size | list offset << 16, payload) per found feature, straight from
    // the row -- no second round of lookups.  Rows that are still in LDS are written out by the WAVE: row after row, one lane per
    // entry, so that a row becomes two contiguous stores instead of 2 x entries scattered ones (the per-lane version of this
    // hand-over cost as much as a third of the lookups on strain-rich tables).
    // (ws.partialLists: the caller wants every query's location list as it is -- a key shard's partial lists -- so every row is handed over)
    const bool hand = valid && (H > kLaneHits || over || ws.partialLists);
    const bool coop = hand && !over;
    const uint32_t lane = threadIdx.x & 63u;
    if (coop) {
        uint32_t off = n;                                         // descriptors get their list offset: index(36) | size(16) | offset(12)
                                                                  // (offsets matter up to kHashMax only; 2^36 locations = 512 GB)
        for (uint32_t i = 0; i < m; ++i) {
            const uint64_t d = L[kLaneHits - i];
            const uint64_t size = d >> 48;
            L[kLaneHits - i] = (d & 0xFFFFFFFFFull) | (size << 36) | ((uint64_t)(off & 0xFFFu) << 52);
            off += (uint32_t)size;
        }
        gnent = n + m;
    }
    uint64_t rows = __ballot(coop);
    if (rows) {
        wave_lds_sync();
        const uint64_t* wrows = L - (size_t)lane * kLaneRow;     // row of wave lane r = wrows + r * kLaneRow
        while (rows) {
            const uint32_t r = __ffsll((unsigned long long)rows) - 1;
            rows &= rows - 1;
            const uint32_t rn = rdlane(n, r), rm = rdlane(m, r), rfb = rdlane(fbase, r);
            if (lane < rn + rm) {
                const uint64_t* row = wrows + (size_t)r * kLaneRow;
                uint32_t ps; uint64_t pp;
                if (lane < rn) { ps = 1u | (lane << 16); pp = row[lane]; }
                else {
                    const uint64_t d = row[kLaneHits - (lane - rn)];
                    ps = (uint32_t)((d >> 36) & 0xFFFFu) | ((uint32_t)(d >> 52) << 16); pp = d & 0xFFFFFFFFFull;
                }
                ws.psize[rfb + lane] = ps; ws.ppay[rfb + lane] = pp;
            }
        }
    }
    if (!valid) return;
    if (hand && ws.partialLists) {
        // gather_lists_kernel copies the lists to their place in ws.hits (hitOff = scan of hitScan); entries = the found features
        ws.hitScan[q] = H <= kMaxHitsPerQuery ? H : 0u;
        ws.qflag[q] = kFlagGather;
        return;
    }
    if (hand) {
        const uint32_t nent = gnent;
        // the wave kernel reads all nf slots -- it gets the lists beyond the lane kernels' reach and whatever the filtered path hands back
        if (H > kMidMax || H > ws.bigMin) for (uint32_t j = nent; j < nf; ++j) ws.psize[fbase + j] = 0u;
        ws.hitScan[q] = (H <= kMaxHitsPerQuery && H > kLdsCap) ? H : 0u;
        // lists of up to 256 locations: work lists of mid_cands_kernel (4 / 8 / 16 lanes per query); one atomic per wave and class
        // ... and of hash_cands_kernel (257 .. 1024 locations, one wave per query, no sort); longer ones, wide window ranges -> wave kernel
        const uint32_t mw = b.maxWin ? b.maxWin[q] : b.maxWinUniform;
        // work list slots: 0 / 1 / 2 mid_cands (64 / 128 / 256), 3 / 4 / 5 hash_cands (512 / 1024 / 256), 6 = wave kernel.  From 129 locations
        // on counting beats sorting (measured per list: 3.3 vs 4.7 ns at 129..256); below, the register sort wins (1.6 vs 2 ns)
        const bool hashOK = nent <= kHashEnt && mw <= kHashWin;
        // 7 = big_cands_kernel (filter first): lists beyond ws.bigMin locations from at most kBigEnt found features
        // (compact store: any number of entries the record can name, any window range the gap between two targets covers; filtered lists
        // the counting kernels do not take are sorted, gw_sorted_cands_kernel)
        const bool bigOK = H > ws.bigMin && H > 64u &&              // (lists up to 64 are sorted in registers, mid_cands_kernel)
                           (tab.values32 ? (nent <= 0xFFFu && mw <= tab.gwGap && H <= kMaxHitsPerQuery) : (nent <= kBigEnt * kBigEPL && mw <= kHashWin));
        const uint32_t cls = bigOK ? 7u : H <= 64 ? 0u : H <= 128 ? (hashOK ? 5u : 1u) : H <= kMidMax ? (hashOK ? 5u : 2u) : (H <= kHashMax && hashOK) ? (H <= kHashMax / 2 ? 3u : 4u) : 6u;
        ws.qflag[q] = cls != 6 ? kFlagMid : kFlagCands;
        if (cls >= 3 && cls != 6) ws.hitScan[q] = 0u;                            // no segment in HBM
        const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            if (c == 6) continue;
            const uint64_t mask = __ballot(cls == c);
            if (cls == c) {
                const uint32_t leader = __ffsll((unsigned long long)mask) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&ws.midCount[c < 5 ? c : c == 5 ? 8u : 9u], (uint32_t)__popcll(mask));
                base = __shfl(base, leader);
                reinterpret_cast<uint4*>(ws.midList)[(size_t)(c == 7 ? 6u : c) * b.n + base + __popcll(mask & ((1ull << lane) - 1ull))] = make_uint4(q, fbase, nent | (H << 12), b.maxWin ? b.maxWin[q] : b.maxWinUniform);
            }
        }
        {   // [10]: how many of the filter's queries have more than kBigEnt entries (its second instance runs only for those)
            // (compact store: how many have more than kGwSmallH locations -- gw_filter_kernel's second instance)
            const uint64_t wide = __ballot(cls == 7 && (tab.values32 ? H > kGwSmallH : nent > kBigEnt));
            if (wide && lane == (uint32_t)__ffsll((unsigned long long)wide) - 1) atomicAdd(&ws.midCount[10], (uint32_t)__popcll(wide));
        }
        return;
    }
    // fetch the noted lists, lowest row position first: the slot of a consumed descriptor is free before the hits reach it
    // (every descriptor still waiting stands for >= 2 of the <= 32 hits)
    for (uint32_t i = 0; i < m; ++i) {
        const uint64_t d = L[kLaneHits - m + 1 + i];
        const uint32_t size = (uint32_t)(d >> 48);
        const uint64_t src = d & 0xFFFFFFFFFFFFull;
        for (uint32_t t = 0; t < size; ++t) L[n++] = tab.loc(src + t);
    }
    ws.hitScan[q] = 0;
    for (uint32_t t = 1; t < n; ++t) {                            // row 8: insertion sort by (tgt, win)
        const uint64_t x = L[t];
        uint32_t j = t;
        while (j > 0 && L[j - 1] > x) { L[j] = L[j - 1]; --j; }
        L[j] = x;
    }
    // rows 9-10, sequentially as on the CPU (candidate_generation.hpp:47-108, :172-201)
    LaneCand top[kLaneK];
    uint32_t toptax[kLaneK];                                      // taxon of each entry (taxon merging only)
#pragma unroll
    for (uint32_t i = 0; i < kLaneK; ++i) { top[i].tgt = 0xFFFFFFFFu; top[i].hits = 0; top[i].beg = 0; top[i].end = 0; toptax[i] = 0; }
    const uint32_t maxWin = b.maxWin ? b.maxWin[q] : b.maxWinUniform;
    auto insert = [&](LaneCand c) { top_insert(top, toptax, c, K, taxkey, tab.tgtMask); };
    if (n > 0) {
        uint32_t fst = 0, hits = 1;
        LaneCand best; best.tgt = (uint32_t)(L[0] >> 32); best.hits = 1; best.beg = best.end = (uint32_t)L[0];
        for (uint32_t i = 1; i < n; ++i) {
            const uint64_t key = L[i];
            const uint32_t tgt = (uint32_t)(key >> 32), win = (uint32_t)key;
            if (tgt == best.tgt) {
                ++hits;
                while (fst != i && (win - (uint32_t)L[fst]) >= maxWin) { --hits; ++fst; }
                if (hits > best.hits) { best.hits = hits; best.beg = (uint32_t)L[fst]; best.end = win; }
            } else {
                insert(best);
                fst = i; hits = 1;
                best.tgt = tgt; best.hits = 1; best.beg = best.end = win;
            }
        }
        insert(best);
    }
    mc_candidate_dev* out = cands + (size_t)q * K;
#pragma unroll
    for (uint32_t i = 0; i < kLaneK; ++i)
        if (i < K) {
            mc_candidate_dev e; e.tgt = top[i].hits ? (top[i].tgt & tab.tgtMask) : 0xFFFFFFFFu; e.hits = top[i].hits; e.beg = top[i].beg; e.end = top[i].end;
            out[i] = e;
        }
    ws.qflag[q] = kFlagDone;
}

template <bool QUAD, bool DIRECT = false>
__global__ __launch_bounds__(kLaneBlock) void probe_cands_kernel(BatchView b, uint32_t s, DeviceTable tab, Workspace ws, uint32_t K,
                                                                 const uint32_t* __restrict__ taxkey, mc_candidate_dev* __restrict__ cands)
{
    __shared__ uint64_t lst[kLaneBlock * kLaneRow];
    const uint32_t q = blockIdx.x * kLaneBlock + threadIdx.x;
    const bool valid = q < b.n && ws.qflag[q] == kFlagProbe;
    probe_cands_one<QUAD, DIRECT>(b, s, tab, ws, K, taxkey, cands, q, lst + threadIdx.x * kLaneRow, valid);
}

// Both halves in one kernel: sketching is ALU work (rolling k-mers, hash, 16-entry insertion chain), probing is waiting for random
// HBM lines; with waves of one CU in different phases the two overlap instead of running one after the other.  The window
// sketches still go through ws.features (a lane reads back what it wrote itself; they are also the MC_WANT_FEATURES output).
template <bool QUAD, bool DIRECT = false>
__global__ __launch_bounds__(kLaneBlock) void sketch_probe_lane_kernel(BatchView b, SketchParams sp, DeviceTable tab, Workspace ws, uint32_t K,
                                                                       const uint32_t* __restrict__ taxkey, mc_candidate_dev* __restrict__ cands)
{
    __shared__ uint64_t lst[kLaneBlock * kLaneRow];
    const uint32_t q = blockIdx.x * kLaneBlock + threadIdx.x;
    uint32_t flag = kFlagDone;
    const uint32_t nch = lane_chunk_records(b, sp, ws, q, threadIdx.x & 63u);   // (long reads: the chunk lanes' kernels follow this one)
    if (q < b.n) {
        flag = nch ? kFlagChunks : sketch_lane_one(b, sp, ws.winOff, ws.features, q);
        if (flag != kFlagProbe) ws.qflag[q] = flag;
    }
    __threadfence_block();                                        // own feature stores before own feature loads
    probe_cands_one<QUAD, DIRECT>(b, sp.s, tab, ws, K, taxkey, cands, q, lst + threadIdx.x * kLaneRow, q < b.n && flag == kFlagProbe);
}

uint32_t lane_max_len() { return kLaneMaxLen; }

void launch_sketch_probe_lane(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                              const uint32_t* taxkey, void* cands, int quadMode, hipStream_t st)
{
    if (b.n == 0) return;
    const bool quad = quadMode >= 0 ? quadMode != 0 : (uint64_t)tab.nbuckets * sizeof(TableBucket) > kQuadTableBytes;
    if (tab.direct) hipLaunchKernelGGL((sketch_probe_lane_kernel<false, true>), dim3((b.n + kLaneBlock - 1) / kLaneBlock), dim3(kLaneBlock), 0, st, b, sp, tab, ws, maxCand,
                                       taxkey, (mc_candidate_dev*)cands);
    else if (quad) hipLaunchKernelGGL(sketch_probe_lane_kernel<true>, dim3((b.n + kLaneBlock - 1) / kLaneBlock), dim3(kLaneBlock), 0, st, b, sp, tab, ws, maxCand,
                                 taxkey, (mc_candidate_dev*)cands);
    else      hipLaunchKernelGGL(sketch_probe_lane_kernel<false>, dim3((b.n + kLaneBlock - 1) / kLaneBlock), dim3(kLaneBlock), 0, st, b, sp, tab, ws, maxCand,
                                 taxkey, (mc_candidate_dev*)cands);
}

void launch_sketch_lane(const BatchView& b, const SketchParams& sp, const Workspace& ws, hipStream_t st)
{
    if (b.n == 0) return;
    hipLaunchKernelGGL(sketch_lane_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b, sp, ws);
}
void launch_chunk_lanes(int stage, const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, int quadMode, hipStream_t st)
{
    if (b.n == 0 || !ws.chunkList) return;
    // persistent grids over the chunk work list (usually empty: its length stays on the device)
    if (stage == 0) hipLaunchKernelGGL(chunk_sketch_kernel, dim3(2048), dim3(128), 0, st, b, sp, ws);
    else {
        const bool quad = quadMode >= 0 ? quadMode != 0 : (uint64_t)tab.nbuckets * sizeof(TableBucket) > kQuadTableBytes;
        if (quad) hipLaunchKernelGGL(chunk_probe_kernel<true>, dim3(2048), dim3(128), 0, st, b, sp.s, tab, ws);
        else      hipLaunchKernelGGL(chunk_probe_kernel<false>, dim3(2048), dim3(128), 0, st, b, sp.s, tab, ws);
        hipLaunchKernelGGL(chunk_finish_kernel, dim3(1024), dim3(256), 0, st, sp.s, ws, b, tab);
    }
}
// Mode K, shard side: the location lists of the lane path's queries as they are (any order inside a list; the owner rank sorts the
// union), copied from the table to ws.hits + hitOff[q].  One wave per 64 queries' flags, then one query at a time: its found
// features (entry table of probe_cands: nfound entries from fbase on; long reads of the chunk lanes: one entry per feature slot),
// bucket after bucket, coalesced.  NUM: the lists as they are STORED -- 4-byte global window numbers, to numbers + hitOff[q] -- no
// decoding to (target, window) and back (mc_partial_numbers; 5.5 -> 0.6 ms per 10^6 reads of 195 locations).
template <bool NUM>
__global__ __launch_bounds__(256) void gather_lists_kernel(BatchView b, uint32_t s, DeviceTable tab, Workspace ws, uint32_t* __restrict__ numbers)
{
    // Flat copy: the 64 entries' sizes are scanned, then every lane takes one OUTPUT place and finds its list by a binary search over
    // the scan in LDS (six reads) -- all places of a read are independent, two groups of 64 are in flight at a time.  (One list after the
    // other -- 26 lists of 7 locations per read at 15 Gbp -- was a chain of 26 dependent load/store pairs: 2.5 ms per 10^6 reads.)
    __shared__ uint32_t inclS[4][64];
    __shared__ uint64_t payS[4][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* sIncl = inclS[wave];
    uint64_t* sPay = payS[wave];
    const uint32_t nWaves = gridDim.x * 4, waveId = blockIdx.x * 4 + wave;
    for (uint32_t base = waveId * 64; base < b.n; base += nWaves * 64) {
        const uint32_t qq = base + lane;
        const uint32_t fl = qq < b.n ? ws.qflag[qq] : kFlagDone;
        uint64_t m = __ballot(fl == kFlagGather || fl == kFlagGatherAll);
        const uint64_t mall = __ballot(fl == kFlagGatherAll);
        while (m) {
            const uint32_t j = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const uint32_t q = base + j;
            const uint32_t fbase = ws.winOff[q] * s, nent = ((mall >> j) & 1ull) ? (ws.winOff[q + 1] - ws.winOff[q]) * s : ws.qstat[q].nfound;
            const uint64_t at = ws.hitOff[q];
            if (ws.hitOff[q + 1] != at) {                                           // (a read beyond kMaxHitsPerQuery has no segment)
                uint64_t* dst = ws.hits + at;
                uint32_t* dst32 = numbers + at;
                for (uint32_t e0 = 0; e0 < nent; e0 += 64) {
                    const uint32_t e = e0 + lane;
                    const uint32_t sz = e < nent ? (ws.psize[fbase + e] & 0xFFFFu) : 0u;
                    const uint64_t pay = e < nent ? ws.ppay[fbase + e] : 0ull;
                    const uint32_t incl = wave_incl_scan_u32(sz, lane);
                    if (sz == 1) {                                                  // a single location is its own payload
                        if constexpr (NUM) dst32[incl - 1] = tab.gw_of(pay); else dst[incl - 1] = pay;
                    }
                    sIncl[lane] = incl; sPay[lane] = pay;
                    wave_lds_sync();
                    const uint32_t tot = rdlane(incl, 63);
                    auto place = [&](uint32_t p, uint64_t& src) -> bool {         // the list holding output place p (lists of one location are done)
                        uint32_t lo = 0;
#pragma unroll
                        for (uint32_t step = 32; step >= 1; step >>= 1) if (sIncl[lo + step - 1] <= p) lo += step;
                        const uint32_t end = sIncl[lo], beg = lo ? sIncl[lo - 1] : 0u;
                        src = sPay[lo] + (p - beg);
                        return end - beg > 1u;
                    };
                    for (uint32_t p0 = 0; p0 < tot; p0 += 128) {
                        const uint32_t pa = p0 + lane, pb = p0 + 64 + lane;
                        uint64_t sa = 0, sb = 0;
                        const bool da = pa < tot && place(pa, sa), db = pb < tot && place(pb, sb);
                        if constexpr (NUM) {
                            const uint32_t va = da ? tab.values32[sa] : 0u, vb = db ? tab.values32[sb] : 0u;
                            if (da) dst32[pa] = va;
                            if (db) dst32[pb] = vb;
                        } else {
                            const uint64_t va = da ? tab.loc(sa) : 0ull, vb = db ? tab.loc(sb) : 0ull;
                            if (da) dst[pa] = va;
                            if (db) dst[pb] = vb;
                        }
                    }
                    wave_lds_sync();
                    dst += tot; dst32 += tot;
                }
            }
            if (lane == 0) ws.qflag[q] = kFlagDone;
        }
    }
}
// numbers != nullptr (compact store only): the lists as 4-byte numbers to numbers + hitOff[q] instead of ws.hits
void launch_gather_lists(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t* numbers, hipStream_t st)
{
    if (b.n == 0) return;
    const dim3 grid(std::min<uint32_t>((b.n + 255) / 256, 256 * 8));
    if (numbers && tab.values32) hipLaunchKernelGGL(gather_lists_kernel<true>, grid, dim3(256), 0, st, b, sp.s, tab, ws, numbers);
    else hipLaunchKernelGGL(gather_lists_kernel<false>, grid, dim3(256), 0, st, b, sp.s, tab, ws, (uint32_t*)nullptr);
}

void launch_probe_cands(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                        const uint32_t* taxkey, void* cands, int quadMode, hipStream_t st)
{
    if (b.n == 0) return;
    // tables that reach beyond the infinity cache and the TLBs: quad-cooperative bucket fetches (see quad_issue)
    const bool quad = quadMode >= 0 ? quadMode != 0 : (uint64_t)tab.nbuckets * sizeof(TableBucket) > kQuadTableBytes;
    if (tab.direct) hipLaunchKernelGGL((probe_cands_kernel<false, true>), dim3((b.n + kLaneBlock - 1) / kLaneBlock), dim3(kLaneBlock), 0, st, b, sp.s, tab, ws, maxCand,
                                       taxkey, (mc_candidate_dev*)cands);
    else if (quad) hipLaunchKernelGGL(probe_cands_kernel<true>, dim3((b.n + kLaneBlock - 1) / kLaneBlock), dim3(kLaneBlock), 0, st, b, sp.s, tab, ws, maxCand,
                                 taxkey, (mc_candidate_dev*)cands);
    else      hipLaunchKernelGGL(probe_cands_kernel<false>, dim3((b.n + kLaneBlock - 1) / kLaneBlock), dim3(kLaneBlock), 0, st, b, sp.s, tab, ws, maxCand,
                                 taxkey, (mc_candidate_dev*)cands);
}
// ================================================================================================
// mid_cands_kernel<G>: location lists of 33 .. 16*G entries (G = 4, 8, 16 lanes per query; 64/G queries per wave).
// A whole wave per query (sort_candidates_kernel) spends ~10 k issue cycles on such a list, most of it waiting for
// LDS round trips of a 64-lane bitonic sort.  Here every lane keeps 16 keys in REGISTERS (blocked layout: element
// i of a query lives in lane i/16, register i%16), so 22 of the 28 compare-exchange stages of a 128-key sort are
// register-to-register; only the stages with distance >= 16 cross lanes (one 64-bit shuffle per key).
//   row 7    : gather through a small LDS list (row stride 17 u64: conflict-free for lane-private rows)
//   row 8    : bitonic sort in registers, written back to LDS
//   row 9    : every lane runs the CPU's sliding-window scan (candidate_generation.hpp:47-108) over its 16 keys; the
//              left end of the first range comes from one binary search; per target run it emits (best end, hits)
//   row 10   : lane 0 of the group walks the emitted segments in order, joins the pieces of runs that span lanes
//              (strictly more hits wins = earliest best kept) and feeds top_insert -- the same code as the lane path
// ================================================================================================
constexpr uint32_t kMidR = 16;                                // keys per lane
__device__ __forceinline__ uint32_t mid_ix(uint32_t i) { return i + (i >> 4); }   // list index -> padded LDS index

// 64-bit keys live in two 32-bit registers each.  Ascending compare-exchange with ONE comparison: the borrow of the 64-bit
// subtraction b - a is (b < a); four selects on it.  (Written as umin/umax the compiler emits two v_cmp_*_u64 per exchange.)
__device__ __forceinline__ void ce64(uint32_t& a0, uint32_t& a1, uint32_t& b0, uint32_t& b1)
{
    uint32_t l0, l1, h0, h1, t;
    asm("v_sub_co_u32 %4, vcc, %7, %5\n\t"
        "v_subb_co_u32 %4, vcc, %8, %6, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %0, %5, %7, vcc\n\t"
        "v_cndmask_b32 %1, %6, %8, vcc\n\t"
        "v_cndmask_b32 %2, %7, %5, vcc\n\t"
        "v_cndmask_b32 %3, %8, %6, vcc"
        : "=&v"(l0), "=&v"(l1), "=&v"(h0), "=&v"(h1), "=&v"(t)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "vcc");
    a0 = l0; a1 = l1; b0 = h0; b1 = h1;
}
// cross-lane step: keep min(a, o) in the lanes of 'flipMask' == 0, max(a, o) in the others: select o where (o < a) != flip
__device__ __forceinline__ void sel64(uint32_t& a0, uint32_t& a1, uint32_t o0, uint32_t o1, uint64_t flipMask)
{
    uint32_t r0, r1, t;
    asm("v_sub_co_u32 %2, vcc, %5, %3\n\t"
        "v_subb_co_u32 %2, vcc, %6, %4, vcc\n\t"
        "s_nop 1\n\t"
        "s_xor_b64 vcc, vcc, %7\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %0, %3, %5, vcc\n\t"
        "v_cndmask_b32 %1, %4, %6, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(t)
        : "v"(a0), "v"(a1), "v"(o0), "v"(o1), "s"(flipMask)
        : "vcc");
    a0 = r0; a1 = r1;
}

template <uint32_t G, bool TAX>
__global__ __launch_bounds__(256) void mid_cands_kernel(BatchView b, DeviceTable tab, Workspace ws, uint32_t K,
                                                        const uint32_t* __restrict__ taxkey, mc_candidate_dev* __restrict__ cands, uint32_t cls)
{
    constexpr uint32_t QPW = 64 / G, N = G * kMidR, ROW = kMidR + 1, kRounds = 4;
    __shared__ uint64_t listS[4][64 * ROW];
    __shared__ uint32_t segS[4][64 * ROW];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t lg = lane % G, qi = lane / G;
    uint64_t* buf = listS[wave] + qi * (G * ROW);            // this query's list, padded: element i at mid_ix(i)
    uint32_t* sg = segS[wave] + qi * (G * ROW);              // entry offsets while gathering, then the query's segments

    const uint32_t total = ws.midCount[cls];
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)cls * b.n;
    const uint32_t nWaves = gridDim.x * 4;
    // two-deep software pipeline over the work list: the record of iteration t+2 and the first kRounds x G entries of iteration
    // t+1 are requested while iteration t is processed (three dependent HBM round trips per query otherwise)
    auto load_rec = [&](uint32_t w) -> uint4 {
        const uint32_t slot = w * QPW + qi;
        return (w * QPW < total && slot < total) ? work[slot] : make_uint4(0, 0, 0, 0);      // .z (entries | locations << 12) == 0: idle group
    };
    uint32_t esz[kRounds]; uint64_t epay[kRounds];
    auto load_entries = [&](const uint4& rec) {
#pragma unroll
        for (uint32_t u = 0; u < kRounds; ++u) {
            const uint32_t e = u * G + lg;
            esz[u] = e < (rec.z & 0xFFFu) ? ws.psize[rec.y + e] : 0u;
            epay[u] = e < (rec.z & 0xFFFu) ? ws.ppay[rec.y + e] : 0ull;
        }
    };
    const uint32_t w0 = blockIdx.x * 4 + wave;
    uint4 rec = load_rec(w0), recNext = load_rec(w0 + nWaves);
    load_entries(rec);
    for (uint32_t w = w0; w * QPW < total; w += nWaves) {
        const bool act = rec.z != 0;
        const uint32_t q = rec.x, fbase = rec.y, nent = rec.z & 0xFFFu, H = rec.z >> 12, maxWin = rec.w;
        uint32_t klo[kMidR], khi[kMidR];
        if (act) {
            // ---- row 7a: entry table in LDS: payload in the list area, start offset in the segment area (at most one entry per
            //      location, so it always fits)
#pragma unroll
            for (uint32_t u = 0; u < kRounds; ++u) {
                const uint32_t e = u * G + lg;
                if (e < nent) { buf[e] = epay[u]; sg[e] = esz[u] >> 16; }
            }
            for (uint32_t e = kRounds * G + lg; e < nent; e += G) { buf[e] = ws.ppay[fbase + e]; sg[e] = ws.psize[fbase + e] >> 16; }
            if (lg == 0) { sg[nent] = H; sg[nent + 1] = 0xFFFFFFFFu; }     // end of the last entry; stopper of the walk
        }
        rec = recNext;
        recNext = load_rec(w + 2 * nWaves);
        load_entries(rec);
        wave_lds_sync();
        if (act) {
            // ---- row 7b: element i of the list goes to lane i/16, register i%16: the lane finds the entry that holds its first
            //      element, then walks; the 16 loads are independent of each other
            const uint32_t i0 = lg * kMidR;
            uint32_t lo = 0, hi = nent;                          // last entry with offset <= i0
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sg[mid] <= i0) lo = mid; else hi = mid; }
            uint32_t e = lo;
#pragma unroll
            for (uint32_t r = 0; r < kMidR; ++r) {
                const uint32_t i = i0 + r;
                while (sg[e + 1] <= i) ++e;                      // i >= H ends on the stopper entry
                const uint64_t pay = buf[e];
                const uint32_t first = sg[e];
                const bool single = sg[e + 1] - first == 1;
                const uint64_t kv = i >= H ? ~0ull : single ? pay : tab.loc(pay + (i - first));
                klo[r] = (uint32_t)kv; khi[r] = (uint32_t)(kv >> 32);
            }
        }
        wave_lds_sync();
        if (act) {
            // ---- row 8: bitonic sort, "flip" formulation (every merge ascending: the lower index keeps the minimum)
#pragma unroll
            for (uint32_t k = 2; k <= kMidR; k <<= 1) {
#pragma unroll
                for (uint32_t r = 0; r < kMidR; ++r) { const uint32_t p = r ^ (k - 1); if (p > r) ce64(klo[r], khi[r], klo[p], khi[p]); }
#pragma unroll
                for (uint32_t j = k >> 2; j > 0; j >>= 1)
#pragma unroll
                    for (uint32_t r = 0; r < kMidR; ++r) { const uint32_t p = r ^ j; if (p > r) ce64(klo[r], khi[r], klo[p], khi[p]); }
            }
#pragma unroll
            for (uint32_t k = 2 * kMidR; k <= N; k <<= 1) {
                {   // flip: partner element i ^ (k-1) = lane ^ (k/16 - 1), register 15 - r
                    const uint32_t lm = k / kMidR - 1;
                    const uint64_t upper = __ballot((lg & (k / (2 * kMidR))) != 0);
#pragma unroll
                    for (uint32_t r = 0; r < kMidR / 2; ++r) {
                        const uint32_t p = kMidR - 1 - r;
                        const uint32_t x0 = __shfl_xor(klo[p], lm), x1 = __shfl_xor(khi[p], lm);
                        const uint32_t y0 = __shfl_xor(klo[r], lm), y1 = __shfl_xor(khi[r], lm);
                        sel64(klo[r], khi[r], x0, x1, upper);
                        sel64(klo[p], khi[p], y0, y1, upper);
                    }
                }
#pragma unroll
                for (uint32_t j = k >> 2; j >= kMidR; j >>= 1) {
                    const uint64_t upper = __ballot((lg & (j / kMidR)) != 0);
#pragma unroll
                    for (uint32_t r = 0; r < kMidR; ++r) {
                        const uint32_t o0 = __shfl_xor(klo[r], j / kMidR), o1 = __shfl_xor(khi[r], j / kMidR);
                        sel64(klo[r], khi[r], o0, o1, upper);
                    }
                }
#pragma unroll
                for (uint32_t j = kMidR / 2; j > 0; j >>= 1)
#pragma unroll
                    for (uint32_t r = 0; r < kMidR; ++r) { const uint32_t p = r ^ j; if (p > r) ce64(klo[r], khi[r], klo[p], khi[p]); }
            }
#pragma unroll
            for (uint32_t r = 0; r < kMidR; ++r) buf[lg * ROW + r] = ((uint64_t)khi[r] << 32) | klo[r];
        }
        wave_lds_sync();
        uint32_t nsegTotal = 0;
        if (act) {
            // ---- row 9: the CPU's scan over this lane's keys (list positions < 256, hits <= 256).  Segments = (best end position |
            //      hits << 16) per target run, written to ONE list per query: their number per lane is counted first.
            const uint32_t i0 = lg * kMidR;
            uint32_t cnt = 0;
            if (i0 < H) {
                cnt = 1;
#pragma unroll
                for (uint32_t r = 1; r < kMidR; ++r)
                    if (i0 + r < H && khi[r] != khi[r - 1]) ++cnt;
            }
            uint32_t incl = cnt;
#pragma unroll
            for (uint32_t d = 1; d < G; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d, G);
                if (lg >= d) incl += o;
            }
            nsegTotal = __shfl(incl, G - 1, G);
            uint32_t nseg = incl - cnt;
            if (i0 < H) {
                uint32_t curTgt = khi[0], fst = i0, hits = 1;
                if (lg > 0 && (uint32_t)(buf[mid_ix(i0 - 1)] >> 32) == curTgt) {
                    // the run began in an earlier lane: left end of the range that ends here (candidate_generation.hpp:79-85)
                    const uint32_t win = klo[0];
                    const uint32_t lowWin = win >= maxWin - 1 ? win - (maxWin - 1) : 0u;
                    const uint64_t lb = ((uint64_t)curTgt << 32) | lowWin;
                    uint32_t lo = 0, hi = i0;
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (buf[mid_ix(mid)] < lb) lo = mid + 1; else hi = mid; }
                    fst = lo; hits = i0 - fst + 1;
                }
                uint32_t bestHits = hits, bestPos = i0;
#pragma unroll
                for (uint32_t r = 1; r < kMidR; ++r) {
                    const uint32_t i = i0 + r;
                    if (i < H) {
                        const uint32_t tgt = khi[r], win = klo[r];
                        if (tgt == curTgt) {
                            ++hits;
                            while (fst != i && (win - (uint32_t)buf[mid_ix(fst)]) >= maxWin) { --hits; ++fst; }
                            if (hits > bestHits) { bestHits = hits; bestPos = i; }
                        } else {
                            sg[nseg++] = bestPos | (bestHits << 16);
                            curTgt = tgt; fst = i; hits = 1; bestHits = 1; bestPos = i;
                        }
                    }
                }
                sg[nseg++] = bestPos | (bestHits << 16);
            }
        }
        wave_lds_sync();
        if (act && lg == 0) {
            // ---- row 10: segments in list order; pieces of one target's run are adjacent.  Four segments (and their targets) are
            //      fetched at a time; a candidate that cannot enter a full list is dropped with one comparison.
            LaneCand top[kLaneK];
            uint32_t toptax[kLaneK];
#pragma unroll
            for (uint32_t i = 0; i < kLaneK; ++i) { top[i].tgt = 0xFFFFFFFFu; top[i].hits = 0; top[i].beg = 0; top[i].end = 0; toptax[i] = 0; }
            uint32_t pTgt = 0, pHits = 0, pPos = 0, pTax = ~0u, lastHits = 0;
            auto flush = [&]() {
                if (pHits > lastHits || lastHits == 0) {         // candidate_generation.hpp:178-181 and :189-201: otherwise no effect
                    LaneCand x; x.tgt = pTgt; x.hits = pHits; x.beg = pPos; x.end = 0;
                    top_insert(top, toptax, x, K, taxkey, tab.tgtMask, pTax);
#pragma unroll
                    for (uint32_t i = 0; i < kLaneK; ++i) if (i + 1 == K) lastHits = top[i].hits;
                }
            };
            for (uint32_t j0 = 0; j0 < nsegTotal; j0 += 4) {
                uint32_t sv[4], tv[4]; uint64_t kk[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) sv[u] = j0 + u < nsegTotal ? sg[j0 + u] : 0u;
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) kk[u] = buf[mid_ix(sv[u] & 0xFFFFu)];
                // taxon merging: the four taxa are requested together (one lane walks: a dependent global load per candidate would
                // cost more than everything else in this kernel)
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) tv[u] = (TAX && j0 + u < nsegTotal) ? taxkey[(uint32_t)(kk[u] >> 32) & tab.tgtMask] : ~0u;
#pragma unroll
                for (uint32_t u = 0; u < 4; ++u) {
                    if (j0 + u < nsegTotal) {
                        const uint32_t pos = sv[u] & 0xFFFFu, h = sv[u] >> 16, tgt = (uint32_t)(kk[u] >> 32);
                        if (pHits && tgt == pTgt) { if (h > pHits) { pHits = h; pPos = pos; } }
                        else { if (pHits) flush(); pTgt = tgt; pHits = h; pPos = pos; pTax = tv[u]; }
                    }
                }
            }
            if (pHits) flush();
            mc_candidate_dev* out = cands + (size_t)q * K;
#pragma unroll
            for (uint32_t i = 0; i < kLaneK; ++i)
                if (i < K) {
                    mc_candidate_dev e; e.tgt = 0xFFFFFFFFu; e.hits = 0; e.beg = 0; e.end = 0;
                    if (top[i].hits) {
                        const uint32_t pos = top[i].beg;
                        e.tgt = top[i].tgt & tab.tgtMask; e.hits = top[i].hits;
                        e.end = (uint32_t)buf[mid_ix(pos)]; e.beg = (uint32_t)buf[mid_ix(pos + 1 - top[i].hits)];
                    }
                    out[i] = e;
                }
            ws.qflag[q] = kFlagDone;
        }
        wave_lds_sync();
    }
}

template <uint32_t G>
static void launch_mid_g(uint32_t grid, uint32_t cls, const BatchView& b, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                         const uint32_t* taxkey, void* cands, hipStream_t st)
{
    if (taxkey) hipLaunchKernelGGL((mid_cands_kernel<G, true>), dim3(grid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, (mc_candidate_dev*)cands, cls);
    else        hipLaunchKernelGGL((mid_cands_kernel<G, false>), dim3(grid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, (mc_candidate_dev*)cands, cls);
}

void launch_mid_cands(uint32_t cls, const BatchView& b, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                      const uint32_t* taxkey, void* cands, hipStream_t st)
{
    if (b.n == 0) return;
    // persistent grids: the work lists are usually short (their lengths stay on the device); 3 blocks fit a CU (53 KB of LDS each)
    const uint32_t blocks = 256 * 3;
    if (cls == 0)      launch_mid_g<4>(std::min<uint32_t>(blocks, (b.n + 63) / 64), 0u, b, tab, ws, maxCand, taxkey, cands, st);
    else if (cls == 1) launch_mid_g<8>(std::min<uint32_t>(blocks, (b.n + 31) / 32), 1u, b, tab, ws, maxCand, taxkey, cands, st);
    else               launch_mid_g<16>(std::min<uint32_t>(blocks, (b.n + 15) / 16), 2u, b, tab, ws, maxCand, taxkey, cands, st);
}
// ================================================================================================
// hash_cands_kernel: location lists of 257 .. 1024 entries (RefSeq-scale tables: 32-bit features collide, a 150 bp read collects
// ~300 locations, most of them single hits on unrelated targets), one WAVE per query, NO sort.  What rows 8-10 deliver is, per
// target, the window range with the most list entries (the earliest one among equals), and of those the K best by (hits
// descending, target ascending) -- at most one per taxon when merging.  None of that needs the list in order:
//   1. every location is counted in an LDS hash table keyed by (target, window) (64-bit CAS claims a slot, packed 16-bit counters);
//   2. the lane that claimed a slot adds the counts of the windows w-1 .. w-(maxWindowsInRange-1) of its target: hits of the range
//      that ENDS in w, begin = smallest window present (candidate_generation.hpp:47-108 evaluates exactly these ranges; the first
//      one that reaches the maximum is the one with the smallest end window);
//   3. every lane keeps its K best (range per target / taxon) under the total order (hits desc, target asc, window asc); K rounds of
//      a wave-wide maximum pick the result, a picked target / taxon is struck from all lanes' lists.  The winner of a target under
//      that order is its earliest best range, and the order among winners is the CPU's insertion order (ties keep arrival order =
//      ascending target), so the result equals the sequential top-K insert (candidate_generation.hpp:172-231).
// ================================================================================================
template <uint32_t LOG2S>
__device__ __forceinline__ uint32_t hash_slot(uint64_t v)
{
    return ((uint32_t)v * 0x9E3779B1u + (uint32_t)(v >> 32) * 0x85EBCA77u) >> (32 - LOG2S);   // multiplicative: the high bits
}

// Steps 1 (counting) to 3 (the K rounds) on the PER elements v[] every lane holds (kEmptyLoc = none), with the wave's (target, window)
// table in LDS (keys: 2^LOG2S slots, all empty; cnts: packed 16-bit counters, all zero).  Writes the K candidates of the query to out;
// returns the number of them that have >= 2 hits (the ones that cannot be displaced by a single-hit target, see big_cands_kernel).
constexpr uint64_t kEmptyLoc = ~0ull;
template <uint32_t LOG2S>
__device__ __forceinline__ uint32_t hash_slot(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - LOG2S); }
// KT = uint64_t: keys are locations (tgt << 32) | win.  (Tables with the compact location store count global window numbers instead:
// gw_count_kernel, gw_kernels.hip.)
template <uint32_t LOG2S, uint32_t PER, bool TAX, class KT>
__device__ __forceinline__ uint32_t count_and_pick(const KT (&v)[PER], KT* keys, uint32_t* cnts, const uint32_t lane, const uint32_t maxWin,
                                                   const uint32_t K, const uint32_t* __restrict__ taxkey, const DeviceTable& tab,
                                                   mc_candidate_dev* __restrict__ out, uint32_t (&picked)[kLaneK])
{
    constexpr uint32_t kMask = (1u << LOG2S) - 1;
    constexpr KT kEmpty = (KT)~(KT)0;
    constexpr bool kWide = sizeof(KT) == 8;
    using cas_t = std::conditional_t<kWide, unsigned long long, unsigned int>;
    static_assert(kWide, "keys are 8-byte locations (the compact store has its own kernels: gw_kernels.hip)");
    auto tgt_of = [&](KT x) -> uint32_t { return (uint32_t)(x >> 32); };
    auto win_of = [&](KT x) -> uint32_t { return (uint32_t)x; };
    auto count_of = [&](uint32_t slot) -> uint32_t { return reinterpret_cast<const uint16_t*>(cnts)[slot]; };   // ds_read_u16
    uint32_t slot[PER];                                       // slot | claimed << 31
    {
        KT old[PER];
        bool coll = false;
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            slot[r] = hash_slot<LOG2S>(v[r]);
            old[r] = v[r] != kEmpty ? (KT)atomicCAS(reinterpret_cast<cas_t*>(&keys[slot[r]]), (cas_t)kEmpty, (cas_t)v[r]) : v[r];
            coll = coll || (old[r] != kEmpty && old[r] != v[r]);
        }
        if (__ballot(coll)) {                                  // somebody else's key in the home slot: next slots, one at a time
#pragma unroll
            for (uint32_t r = 0; r < PER; ++r) {
                if (old[r] != kEmpty && old[r] != v[r]) {
                    uint32_t sl = slot[r];
                    for (;;) {
                        sl = (sl + 1) & kMask;
                        old[r] = (KT)atomicCAS(reinterpret_cast<cas_t*>(&keys[sl]), (cas_t)kEmpty, (cas_t)v[r]);
                        if (old[r] == kEmpty || old[r] == v[r]) break;
                    }
                    slot[r] = sl;
                }
            }
        }
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            if (v[r] != kEmpty) atomicAdd(&cnts[slot[r] >> 1], 1u << (16u * (slot[r] & 1u)));
            slot[r] |= (v[r] != kEmpty && old[r] == kEmpty) ? 0x80000000u : 0u;
        }
    }
    wave_lds_sync();
    // ---- 2. ranges that end in the windows this lane claimed: hits | (end - begin) << 16
    uint32_t ptax[PER];
    if constexpr (TAX) {
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) ptax[r] = (slot[r] >> 31) ? taxkey[tgt_of(v[r]) & tab.tgtMask] : 0u;
    }
    uint32_t T[PER];
#pragma unroll
    for (uint32_t r = 0; r < PER; ++r) T[r] = (slot[r] >> 31) ? count_of(slot[r] & kMask) : 0u;
    for (uint32_t d = 1; d < maxWin; ++d) {
        KT k[PER]; uint32_t sl[PER];
        bool chain = false;
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            sl[r] = hash_slot<LOG2S>((KT)(v[r] - d));
            k[r] = keys[sl[r]];
            const bool live = (slot[r] >> 31) && win_of(v[r]) >= d;
            if (!live) { k[r] = kEmpty; sl[r] = 0xFFFFFFFFu; }   // (target 0, window < d: v - d would equal the empty key)
            chain = chain || (live && k[r] != v[r] - d && k[r] != kEmpty);
        }
        if (__ballot(chain)) {
#pragma unroll
            for (uint32_t r = 0; r < PER; ++r)
                while (sl[r] != 0xFFFFFFFFu && k[r] != v[r] - d && k[r] != kEmpty) { sl[r] = (sl[r] + 1) & kMask; k[r] = keys[sl[r]]; }
        }
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            const uint32_t c = count_of(sl[r] & kMask);
            if (sl[r] != 0xFFFFFFFFu && k[r] == v[r] - d) T[r] = ((T[r] & 0xFFFFu) + c) | (d << 16);
        }
    }
    // ---- 3. K rounds: every lane offers the best of its ranges whose target / taxon has not been picked yet, the wave takes the
    //      maximum under (hits desc, target asc, window asc) and strikes that target / taxon everywhere
    uint32_t live = 0;                                         // bit r: this lane's range r is still in the race
#pragma unroll
    for (uint32_t r = 0; r < PER; ++r) {
        bool ok = (slot[r] >> 31) != 0;
        if constexpr (TAX) ok = ok && ptax[r] != 0;            // no taxon at that rank: skipped (candidate_generation.hpp:185)
        live |= ok ? (1u << r) : 0u;
    }
    uint32_t strong = 0;
    for (uint32_t rnd = 0; rnd < K; ++rnd) {
        uint64_t hk = 0; uint32_t hw = 0xFFFFFFFFu, hg = 0, hd = 0;
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            const uint32_t t = tgt_of(v[r]), win = win_of(v[r]);
            const uint64_t ck = ((uint64_t)(T[r] & 0xFFFFu) << 32) | (uint32_t)~t;
            const bool take = ((live >> r) & 1u) && (ck > hk || (ck == hk && win < hw));
            if (take) { hk = ck; hw = win; hd = T[r] >> 16; if constexpr (TAX) hg = ptax[r]; else hg = t; }
        }
        // wave-wide maximum of (hits, ~target), then the smallest window among its holders: three 32-bit DPP reductions (VALU speed)
        // instead of 18 ds_bpermute round trips per round
        const uint32_t khi = (uint32_t)(hk >> 32), klo = (uint32_t)hk;
        const uint32_t mhi = wave_max_u32(khi);
        const uint32_t mlo = wave_max_u32(khi == mhi ? klo : 0u);
        const uint64_t m = ((uint64_t)mhi << 32) | mlo;
        mc_candidate_dev e; e.tgt = 0xFFFFFFFFu; e.hits = 0; e.beg = 0; e.end = 0;
        if (mhi != 0) {
            const uint32_t wm = wave_min_u32(hk == m ? hw : 0xFFFFFFFFu);
            const uint32_t winner = __ffsll((unsigned long long)__ballot(hk == m && hw == wm)) - 1;
            const uint32_t g = rdlane(hg, winner), d = rdlane(hd, winner);
#pragma unroll
            for (uint32_t r = 0; r < PER; ++r) {
                uint32_t gr;
                if constexpr (TAX) gr = ptax[r]; else gr = tgt_of(v[r]);
                if (gr == g) live &= ~(1u << r);
            }
            e.tgt = ~(uint32_t)m & tab.tgtMask; e.hits = (uint32_t)(m >> 32); e.end = wm; e.beg = wm - d;
            strong += e.hits >= 2 ? 1u : 0u;
#pragma unroll
            for (uint32_t i = 0; i < kLaneK; ++i) if (i == rnd) picked[i] = ~(uint32_t)m;      // the stored target id (part folded in)
        }
        if (lane == 0) out[rnd] = e;
    }
    return strong;
}

// LOG2S: log2 of the table slots; lists of up to 2^(LOG2S-1) locations (class 3: 512 in 1024 slots, 4 waves per block; class 4: 1024
// in 2048 slots, 2 waves per block).  All LDS traffic of a phase is issued for the lane's elements together (the operations of
// different elements are independent); only collisions fall back to a serial walk.
template <uint32_t LOG2S, uint32_t WAVES, bool TAX>
__global__ __launch_bounds__(WAVES * 64) void hash_cands_kernel(BatchView b, DeviceTable tab, Workspace ws, uint32_t K,
                                                                const uint32_t* __restrict__ taxkey, mc_candidate_dev* __restrict__ cands, uint32_t cls)
{
    constexpr uint32_t kSlots = 1u << LOG2S, kPer = kSlots / 2 / 64, kRounds = kHashEnt / 64;
    constexpr uint64_t kEmpty = kEmptyLoc;
    __shared__ uint64_t keyS[WAVES][kSlots];
    __shared__ uint32_t cntS[WAVES][kSlots / 2];
    __shared__ uint64_t entPayS[WAVES][kHashEnt];
    __shared__ uint32_t entOffS[WAVES][kHashEnt + 2];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t* keys = keyS[wave];
    uint32_t* cnts = cntS[wave];
    uint64_t* entPay = entPayS[wave];
    uint32_t* entOff = entOffS[wave];
    const uint32_t total = ws.midCount[cls < 5 ? cls : 8u];
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)cls * b.n;
    const uint32_t nWaves = gridDim.x * WAVES;
    auto load_rec = [&](uint32_t w) -> uint4 { return w < total ? work[w] : make_uint4(0, 0, 0, 0); };
    uint32_t esz[kRounds]; uint64_t epay[kRounds];
    auto load_entries = [&](const uint4& rec) {
#pragma unroll
        for (uint32_t u = 0; u < kRounds; ++u) {
            const uint32_t e = u * 64 + lane;
            esz[u] = e < (rec.z & 0xFFFu) ? ws.psize[rec.y + e] : 0u;
            epay[u] = e < (rec.z & 0xFFFu) ? ws.ppay[rec.y + e] : 0ull;
        }
    };
    const uint32_t w0 = blockIdx.x * WAVES + wave;
    uint4 rec = load_rec(w0), recNext = load_rec(w0 + nWaves);
    load_entries(rec);
    for (uint32_t w = w0; w < total; w += nWaves) {
        const uint32_t q = rec.x, nent = rec.z & 0xFFFu, H = rec.z >> 12, maxWin = rec.w;
        // ---- empty table; entry table (payload, first list index) in LDS
        {
            uint4* k4 = reinterpret_cast<uint4*>(keys);
            uint4* c4 = reinterpret_cast<uint4*>(cnts);
#pragma unroll
            for (uint32_t i = 0; i < kSlots / 2 / 64; ++i) k4[i * 64 + lane] = make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
            for (uint32_t i = 0; i < kSlots / 8 / 64; ++i) c4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t u = 0; u < kRounds; ++u) {
            const uint32_t e = u * 64 + lane;
            if (e < nent) { entPay[e] = epay[u]; entOff[e] = esz[u] >> 16; }
        }
        if (lane == 0) { entOff[nent] = H; entOff[nent + 1] = 0xFFFFFFFFu; }
        rec = recNext;                                             // the next query's record and entries are on their way meanwhile
        recNext = load_rec(w + 2 * nWaves);
        load_entries(rec);
        wave_lds_sync();
        // the lane's share of the list: per elements.  The rest of the query is instantiated for a few values of PER >= per, so that the
        // unrolled per-element code is not run for slots no lane fills (289 locations = 5 per lane; at PER = 8 the wave spent 40 % of
        // its instructions on empty slots)
        const uint32_t per = (H + 63u) / 64u;
        auto body = [&](auto perc) {
            constexpr uint32_t PER = decltype(perc)::value;
            // ---- 1. gather (lane: per consecutive list elements, one search, then a walk -- as mid_cands_kernel; handing element i to
            //      lane i % 64 makes a wave's loads coalesce, but costs a search per element and was not faster)
            const uint32_t i0 = lane * per;
            uint64_t v[PER];
            {
                uint32_t lo = 0, hi = nent;
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (entOff[mid] <= i0) lo = mid; else hi = mid; }
                uint32_t e = lo;
    #pragma unroll
                for (uint32_t r = 0; r < PER; ++r) {
                    const uint32_t i = i0 + r;
                    v[r] = kEmpty;
                    if (r < per && i < H) {
                        while (entOff[e + 1] <= i) ++e;
                        const uint64_t pay = entPay[e];
                        const uint32_t first = entOff[e];
                        v[r] = entOff[e + 1] - first == 1 ? pay : tab.loc(pay + (i - first));
                    }
                }
            }
            uint32_t picked[kLaneK];
            count_and_pick<LOG2S, PER, TAX>(v, keys, cnts, lane, maxWin, K, taxkey, tab, cands + (size_t)q * K, picked);
        };
        if constexpr (kPer == 4) {
            if (per <= 2) body(std::integral_constant<uint32_t, 2>{}); else if (per <= 3) body(std::integral_constant<uint32_t, 3>{}); else body(std::integral_constant<uint32_t, 4>{});
        } else if constexpr (kPer == 8) {
            switch (per) {
                case 5: body(std::integral_constant<uint32_t, 5>{}); break;
                case 6: body(std::integral_constant<uint32_t, 6>{}); break;
                case 7: body(std::integral_constant<uint32_t, 7>{}); break;
                default: body(std::integral_constant<uint32_t, 8>{}); break;
            }
        } else {
            if (per <= kPer * 5 / 8) body(std::integral_constant<uint32_t, kPer * 5 / 8>{});
            else if (per <= kPer * 6 / 8) body(std::integral_constant<uint32_t, kPer * 6 / 8>{});
            else if (per <= kPer * 7 / 8) body(std::integral_constant<uint32_t, kPer * 7 / 8>{});
            else body(std::integral_constant<uint32_t, kPer>{});
        }
        if (lane == 0) ws.qflag[q] = kFlagDone;
        wave_lds_sync();
    }
}

// ================================================================================================
// big_filter_kernel + big_count_kernel: location lists beyond hash_cands_kernel's reach (RefSeq scale: 2 x 10^10 locations behind
// 32-bit features, a 150 bp read collects 1000 .. 8000 locations), one WAVE per query.  Nearly all of those locations are single
// hits on unrelated targets; what rows 8-10 keep of them is at most "the smallest target ids with one hit" when fewer than K targets
// reach two.  So the list is FILTERED before anything is counted per (target, window):
//   big_filter_kernel
//   A. every location enters its TARGET into a Bloom filter in LDS ("seen"); a target found there already goes into a second one
//      ("twice") -- 3 KB per wave, one returning ds_or per location, one more for repeats;
//   B. second sweep over the list: locations whose target is in "twice" are compacted (ballot) into the wave's slice of a pool in
//      HBM -- all locations of every target with >= 2 hits and about 1 % of the rest, typically a fifth of the list.
//   big_count_kernel
//   C. count_and_pick (hash_cands_kernel's steps 1-3) on the filtered list: exact hits per window range, K rounds;
//   D. only if fewer than K picked candidates have >= 2 hits: the open places go to the smallest targets among ALL other
//      locations, each with its smallest window (hits = 1 candidates are ordered by target id alone = arrival order in the sorted
//      list, candidate_generation.hpp:172-201) -- a sweep over the original lists.  With taxon merging this case goes to the wave kernel.
// Two kernels because both halves wait most of their time and are held back by different resources: the sweeps need 66 registers and
// 5.6 KB of LDS per wave (28 waves per CU), the counting 80+ registers and a 6 - 10 KB hash table (24 / 16 waves).  Fused they
// ran at 8 waves per CU: 22 us per list, nearly all of it memory latency of the sweeps (R / kBigU dependent round trips each).
// The sweeps read the bucket lists straight from the table: one coalesced wave load per 64 locations of a bucket ("round"),
// kBigU rounds in flight; sweep B re-reads what sweep A brought into the L2 / infinity cache.
// Filtered lists beyond 512 take the counting instance with the larger hash table, beyond 1024 (or when the pool is full) the
// wave kernel.
// ================================================================================================
#ifndef MC_BIG_U
#define MC_BIG_U 8
#endif
constexpr uint32_t kBigU = MC_BIG_U;      // wave loads in flight per wave.  8, 12 or 16 make no difference at equal waves per CU (18.4 / - / 18.0 ms
                                          // per 5 x 10^6 reads at 16 waves); 8 keeps the kernel at 62 registers and 7.3 KB of LDS per wave,
                                          // which lets a fifth block onto each CU: 15.9 ms at 20 waves
// Target states of the filter: two blocked Bloom filters in LDS.  "seen": 2 bits of ONE 32-bit word per target (word and bits from a
// multiplicative hash), set with a single returning ds_or -- both bits found set = the target was seen before (or, 1 % of the time, two
// other targets set them); "twice": the same for the targets found seen, read in sweep B.  16 384 + 8 192 bits = 3 KB per wave; the
// first form (16 384 two-bit states, 4 KB) let 7.6 % of the single-hit locations through, this one about 1 %.
#ifndef MC_BIG_T1
#define MC_BIG_T1 14
#endif
#ifndef MC_BIG_T2
#define MC_BIG_T2 13
#endif
constexpr uint32_t kBigT1Log2 = MC_BIG_T1, kBigT2Log2 = MC_BIG_T2;     // bits of the two filters
#ifndef MC_BIG_POS_T1
#define MC_BIG_POS_T1 17                  // ... of the second instance (twice the keys of 3.5 x the locations)
#endif
#ifndef MC_BIG_POS_T2
#define MC_BIG_POS_T2 14
#endif
#ifndef MC_BIG_MIN_SHIFT
#define MC_BIG_MIN_SHIFT 3
#endif
constexpr uint32_t kBigMinShift = MC_BIG_MIN_SHIFT;   // smallest round: 1 << shift lanes
constexpr uint32_t kBigMaxFiltered = 1024;
constexpr uint32_t kBigMaxRounds = kBigEnt * 4;

uint32_t big_filter_grid(uint32_t n, bool compact, int bpcOverride);
static uint32_t big_count_bpc(bool compact)
{
    static const uint32_t env = [] { const char* e = std::getenv("MC_BIG_COUNT_BPC"); return e ? (uint32_t)std::max(1, std::atoi(e)) : 0u; }();
    return env ? env : compact ? 6u : 4u;
}
// A ROUND = up to G consecutive locations of one bucket, read by G neighbouring lanes; 64 / G rounds share one wave load.  G = 16
// unless the query's buckets would need more than kBigMaxRounds such rounds (then G = 64): a bucket of 16 locations fills a quarter
// of a 64-lane round but a whole 16-lane one, one of 49 takes 4 x 16 either way -- at 430 locations per read the sweeps issue a
// third of the wave loads (26 buckets of 17), at 1 270 (26 buckets of 49) 60 %.
struct BigTables {                        // per wave: the round table of one query (the entries themselves stay in the lanes' registers:
    uint64_t rounds[kBigMaxRounds + kBigU * 8];   // 768 bytes less per wave is what lets a sixth block of big_filter_kernel onto a CU)
};
static_assert(kBigU * 8 % 64 == 0, "whole waves of padding entries");
struct BigShape { uint32_t rounds, shift; };   // rounds of 1 << shift lanes

// entries -> LDS tables (EPL entries per lane: entry e * 64 + lane); rounds > kBigMaxRounds even at 64 lanes per round: merged buckets
// of a partitioned database, not handled here
template <uint32_t EPL>
__device__ __forceinline__ BigShape big_setup(BigTables& T, const uint32_t lane, const uint32_t nent, const uint32_t (&mySz)[EPL], const uint64_t (&myPay)[EPL],
                                              const uint32_t minShift = kBigMinShift)
{
    uint32_t m8 = 0, m16 = 0;
#pragma unroll
    for (uint32_t e = 0; e < EPL; ++e) {
        const bool list = e * 64 + lane < nent && mySz[e] > 1;
        m8 += list ? (mySz[e] + 7u) / 8u : 0u; m16 += list ? (mySz[e] + 15u) / 16u : 0u;
    }
    const uint32_t r8 = wave_sum_u32(m8), r16 = wave_sum_u32(m16);
    const uint32_t shift = (minShift <= 3 && r8 <= kBigMaxRounds) ? 3u : r16 <= kBigMaxRounds ? 4u : 6u, G = 1u << shift;
    uint32_t myRounds = 0;
#pragma unroll
    for (uint32_t e = 0; e < EPL; ++e) myRounds += (e * 64 + lane < nent && mySz[e] > 1) ? (mySz[e] + G - 1u) >> shift : 0u;
    const uint32_t incl = wave_incl_scan_u32(myRounds, lane);
    const uint32_t R = rdlane(incl, 63);
    if (R <= kBigMaxRounds) {
        uint32_t at = incl - myRounds;
#pragma unroll
        for (uint32_t e = 0; e < EPL; ++e) {
            const uint32_t n = (e * 64 + lane < nent && mySz[e] > 1) ? (mySz[e] + G - 1u) >> shift : 0u;
            for (uint32_t j = 0; j < n; ++j)
                T.rounds[at + j] = (myPay[e] + (uint64_t)G * j) | ((uint64_t)min(G, mySz[e] - G * j) << 40);
            at += n;
        }
#pragma unroll
        for (uint32_t i = 0; i < kBigU * 8; i += 64) T.rounds[R + i + lane] = 0ull;   // the last batch of a sweep reads up to kBigU * 8 entries past R
    }
    return BigShape{R, shift};
}
// one sweep over a query's locations: f(v) for the lane's element of every wave load (kEmptyLoc = none), kBigU loads in flight
template <class F>
__device__ __forceinline__ void big_sweep(const BigTables& T, const DeviceTable& tab, const uint32_t lane, const BigShape sh, F&& f)
{
    const uint32_t perLoad = 64u >> sh.shift, grp = lane >> sh.shift, sub = lane & ((1u << sh.shift) - 1u);
    for (uint32_t g0 = 0; g0 < sh.rounds; g0 += kBigU * perLoad) {
        uint64_t rv[kBigU];
#pragma unroll
        for (uint32_t u = 0; u < kBigU; ++u) {
            const uint64_t rd = T.rounds[g0 + u * perLoad + grp];
            rv[u] = sub < (uint32_t)(rd >> 40) ? tab.values[(rd & 0xFFFFFFFFFFull) + sub] : kEmptyLoc;
        }
#pragma unroll
        for (uint32_t u = 0; u < kBigU; ++u) f(rv[u]);
    }
}

// No atomics on global memory: a wave appends its filtered lists to ITS OWN slice of the pool (a million waves bumping one cursor
// cost more than the sweeps: 43 ms instead of 14), and the record for big_count_kernel goes to the place of the query's own work
// record (list 7 runs parallel to list 6; n2 = 0xFFFF marks lists that went to the wave kernel instead).
#ifdef MC_BIG_WPE
#define MC_BIG_WPE_ATTR __attribute__((amdgpu_waves_per_eu(MC_BIG_WPE, MC_BIG_WPE)))
#else
#define MC_BIG_WPE_ATTR
#endif
// EPL = 1: the queries with up to 64 found features (one entry per lane).  EPL > 1: the instance for the others (up to 64 * EPL: reads
// and pairs of 5 .. 10 windows); it runs after the first one on the same grid and goes on in the same pool slices (ws.sliceFill).
// POS (the second instance): the filters are keyed on (target, block of 16 windows) in two grids half a block apart instead of the
// target alone -- two locations that can share a window range (maxWindowsInRange <= kHashWin = 8) share a block in one of the grids.
// A read of 500 bp collects 4 500 locations at RefSeq scale and hits 250 targets TWICE BY CHANCE, at unrelated places: keyed on targets
// the filtered list outgrows the counting kernels (1024), keyed on places only true neighbours stay.  Twice the keys of 3.5 x the
// locations: T1LOG2 / T2LOG2 = 17 / 14 (18 KB per wave, two waves per block).
template <uint32_t WAVES, uint32_t EPL, bool POS, uint32_t T1LOG2, uint32_t T2LOG2>
__global__ __launch_bounds__(WAVES * 64) MC_BIG_WPE_ATTR void big_filter_kernel(BatchView b, DeviceTable tab, Workspace ws)
{
    static_assert(!POS || kHashWin <= 8, "two grids of 16 windows, 8 apart, cover window ranges up to 8");
    using pool_t = uint64_t;
    constexpr uint32_t kW1 = (1u << T1LOG2) / 32, kW2 = (1u << T2LOG2) / 32, kBitWords = kW1 + kW2;
    static_assert(kBitWords % 256 == 0, "cleared with one uint4 per lane and step");
    __shared__ uint32_t bitS[WAVES][kBitWords];
    __shared__ BigTables tabS[WAVES];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* bits = bitS[wave];
    BigTables& T = tabS[wave];
    const uint32_t total = ws.midCount[9];
    if (EPL > 1 && ws.midCount[10] == 0) return;                  // no query with more than kBigEnt entries in this batch
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)6 * b.n;
    uint4* __restrict__ outRec = reinterpret_cast<uint4*>(ws.midList) + (size_t)7 * b.n;
    const uint32_t nWaves = gridDim.x * WAVES;
    auto load_rec = [&](uint32_t w) -> uint4 { return w < total ? work[w] : make_uint4(0, 0, 0, 0); };
    auto mine = [](uint32_t nent) { return EPL == 1 ? nent <= kBigEnt : nent > kBigEnt; };
    const uint32_t w0 = blockIdx.x * WAVES + wave;
    // this wave's slice of the pool
    const uint64_t sliceCap = ws.bigPoolCap / nWaves;
    pool_t* const slice = reinterpret_cast<pool_t*>(ws.bigPool) + (uint64_t)w0 * sliceCap;
    uint64_t sliceUsed = (EPL > 1 && ws.sliceFill) ? ws.sliceFill[w0] : 0u;
    uint4 rec = load_rec(w0), recNext = load_rec(w0 + nWaves);
    uint32_t esz[EPL]; uint64_t epay[EPL];
    auto load_entries = [&](const uint4& r) {
        const uint32_t ne = mine(r.z & 0xFFFu) ? (r.z & 0xFFFu) : 0u;
#pragma unroll
        for (uint32_t e = 0; e < EPL; ++e) {
            esz[e] = e * 64 + lane < ne ? ws.psize[r.y + e * 64 + lane] : 0u;
            epay[e] = e * 64 + lane < ne ? ws.ppay[r.y + e * 64 + lane] : 0ull;
        }
    };
    load_entries(rec);
    // key g of a location: 0 = its target (POS: and its block of 16 windows), 1 (POS only) = the block of the grid shifted by 8 windows
    auto hash_of = [&](uint64_t v, uint32_t g) -> uint32_t {
        const uint32_t t = (uint32_t)(v >> 32) * 0x9E3779B1u;
        if constexpr (POS) return (t + ((((uint32_t)v + 8u * g) >> 4) * 2u + g) * 0x85EBCA77u) * 0xC2B2AE3Du;
        else return t;
    };
    auto seen_of = [&](uint32_t h, uint32_t& word, uint32_t& mask) {
        word = h >> (32 - (T1LOG2 - 5)); mask = (1u << ((h >> 12) & 31u)) | (1u << ((h >> 7) & 31u));
    };
    auto twice_of = [&](uint32_t h, uint32_t& word, uint32_t& mask) {            // the same hash: other bits for the word, the same two bits in it
        word = kW1 + ((h >> 17) & (kW2 - 1u)); mask = (1u << ((h >> 12) & 31u)) | (1u << ((h >> 7) & 31u));
    };
    constexpr uint32_t kKeys = POS ? 2 : 1;
    for (uint32_t w = w0; w < total; w += nWaves) {
        const uint32_t q = rec.x, nent = rec.z & 0xFFFu, H = rec.z >> 12, maxWin = rec.w;
        if (!mine(nent)) {                                         // the other instance's query
            rec = recNext; recNext = load_rec(w + 2 * nWaves);
            load_entries(rec);
            continue;
        }
        {
            uint4* z4 = reinterpret_cast<uint4*>(bits);
#pragma unroll
            for (uint32_t i = 0; i < kBitWords / 4 / 64; ++i) z4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
        uint32_t mySz[EPL]; uint64_t myPay[EPL], single[EPL];
#pragma unroll
        for (uint32_t e = 0; e < EPL; ++e) {
            mySz[e] = esz[e] & 0xFFFFu; myPay[e] = epay[e];
            single[e] = (e * 64 + lane < nent && mySz[e] == 1) ? myPay[e] : kEmptyLoc;     // inline locations of buckets of one
        }
        const BigShape sh = big_setup<EPL>(T, lane, nent, mySz, myPay, kBigMinShift);
        rec = recNext;                                             // the next query's record and entries are on their way meanwhile
        recNext = load_rec(w + 2 * nWaves);
        load_entries(rec);
        wave_lds_sync();
        bool fallback = sh.rounds > kBigMaxRounds;
        uint32_t n2 = 0;
        if (!fallback) {
            // ---- A. target states.  (Holding the list in LDS for sweep B was measured: at 1536 / 1024 / 2048 staged locations the kernel
            //      took 31.6 / 37.2 / 57.5 ms instead of 26.5 per 5 x 10^6 reads -- the LDS costs more waves than the re-read costs.)
            auto mark = [&](uint64_t v) {
                if (v != kEmptyLoc) {
#pragma unroll
                    for (uint32_t g = 0; g < kKeys; ++g) {
                        const uint32_t h = hash_of(v, g);
                        uint32_t word, mask;
                        seen_of(h, word, mask);
                        const uint32_t old = atomicOr(&bits[word], mask);
                        if ((old & mask) == mask) { uint32_t w2, m2; twice_of(h, w2, m2); atomicOr(&bits[w2], m2); }
                    }
                }
            };
#pragma unroll
            for (uint32_t e = 0; e < EPL; ++e) mark(single[e]);
            big_sweep(T, tab, lane, sh, mark);
            wave_lds_sync();
            // ---- B. locations of targets seen twice or more -> this wave's pool slice (as long as they fit), counted
            pool_t* dst = slice + sliceUsed;
            const uint32_t room = (uint32_t)min((uint64_t)kBigMaxFiltered, sliceCap - sliceUsed);
            auto take = [&](uint64_t v) {
                bool keep = false;
                if (v != kEmptyLoc) {
#pragma unroll
                    for (uint32_t g = 0; g < kKeys; ++g) {
                        uint32_t word, mask;
                        twice_of(hash_of(v, g), word, mask);
                        keep = keep || (bits[word] & mask) == mask;
                    }
                }
                const uint64_t m = __ballot(keep);
                if (keep) {
                    const uint32_t at = n2 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (at < room) {
                        dst[at] = v;
                    }
                }
                n2 += (uint32_t)__popcll(m);
            };
#pragma unroll
            for (uint32_t e = 0; e < EPL; ++e) take(single[e]);
            big_sweep(T, tab, lane, sh, take);
            fallback = n2 > room;                                  // too long for big_count_kernel, or the slice is full
        }
        if (lane == 0) {
            if (fallback) { ws.hitScan[q] = H; ws.qflag[q] = kFlagCands; outRec[w] = make_uint4(q, 0u, 0xFFFFu, maxWin); }
            else outRec[w] = make_uint4(q, (uint32_t)((uint64_t)w0 * sliceCap + sliceUsed), n2 | (nent << 16), maxWin);
        }
        if (!fallback) sliceUsed += n2;
        wave_lds_sync();
    }
    if (EPL == 1 && ws.sliceFill && lane == 0) ws.sliceFill[w0] = (uint32_t)sliceUsed;
}

#ifndef MC_BIG_COUNT_PREFETCH
#define MC_BIG_COUNT_PREFETCH 0     // fetching the next query's filtered list during this one: measured, no gain (7.35 against 7.28 ms; spills at 80 registers)
#endif
#ifndef MC_BIG_COUNT2_WPE
#define MC_BIG_COUNT2_WPE 3     // second instance, compact keys: 168 registers, six blocks of two waves per CU (pairs at full scale: 9.4 -> 7.2 ms)
#endif
#ifndef MC_BIG_COUNT_WPE
#define MC_BIG_COUNT_WPE 6     // compact keys: 6 KB of LDS per wave; at 80 registers six blocks fit a CU (8.0 / 7.3 / 6.9 ms at 16 / 20 / 24 waves)
#endif
template <uint32_t LOG2S, uint32_t WAVES, bool TAX>
__global__ __launch_bounds__(WAVES * 64) void big_count_kernel(BatchView b, uint32_t s, DeviceTable tab, Workspace ws, uint32_t K,
                                                               const uint32_t* __restrict__ taxkey, mc_candidate_dev* __restrict__ cands,
                                                               uint32_t minN2)
{
    constexpr uint32_t kSlots = 1u << LOG2S, kList = kSlots / 2;
    using pool_t = uint64_t;
    static_assert(sizeof(BigTables) <= kSlots * sizeof(pool_t), "step D's tables live in the key table, which is done with by then");
    __shared__ __attribute__((aligned(16))) pool_t keyS[WAVES][kSlots];
    __shared__ uint32_t cntS[WAVES][kSlots / 2];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    pool_t* keys = keyS[wave];
    uint32_t* cnts = cntS[wave];
    BigTables& T = *reinterpret_cast<BigTables*>(keyS[wave]);      // (10 KB instead of 13 KB of LDS per wave: 16 waves per CU instead of 12)
    // the records big_filter_kernel left (list 7, one per query of work list 6); this instance takes the filtered lists that fit its
    // table: n2 in (minN2, kList]
    const uint32_t total = ws.midCount[9];
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * b.n;
    const uint32_t nWaves = gridDim.x * WAVES;
    auto load_rec = [&](uint32_t w) -> uint4 { return w < total ? work[w] : make_uint4(0, 0, 0xFFFFu, 0); };
    const uint32_t w0 = blockIdx.x * WAVES + wave;
    // (MC_BIG_COUNT_PREFETCH: the first instance fetches the NEXT query's filtered list while it works on this one)
    constexpr bool kPrefetch = MC_BIG_COUNT_PREFETCH && LOG2S == 10;
    constexpr uint32_t kPre = kPrefetch ? kList / 64 : 1;
    constexpr pool_t kNone = (pool_t)~(pool_t)0;
    uint4 rec = load_rec(w0), recN = load_rec(w0 + nWaves);
    pool_t pre[kPre];
    auto fetch = [&](const uint4& r) {
        if constexpr (kPrefetch) {
            uint32_t n = r.z & 0xFFFFu;
            if (n > kList || (minN2 != 0 && n <= minN2)) n = 0;
            const pool_t* __restrict__ p = reinterpret_cast<const pool_t*>(ws.bigPool) + r.y;
#pragma unroll
            for (uint32_t i = 0; i < kPre; ++i) pre[i] = i * 64 + lane < n ? p[i * 64 + lane] : kNone;
        }
    };
    fetch(rec);
    for (uint32_t w = w0; w < total; w += nWaves) {
        const uint32_t q = rec.x, n2 = rec.z & 0xFFFFu, nent = rec.z >> 16, maxWin = rec.w;
        const pool_t* __restrict__ src = reinterpret_cast<const pool_t*>(ws.bigPool) + rec.y;
        pool_t cur[kPre];
#pragma unroll
        for (uint32_t i = 0; i < kPre; ++i) cur[i] = pre[i];
        rec = recN; recN = load_rec(w + 2 * nWaves);
        fetch(rec);
        if (n2 > kList || (minN2 != 0 && n2 <= minN2)) continue;   // (an EMPTY filtered list -- no target seen twice -- is the first instance's: step D fills the places)
        {
            uint4* k4 = reinterpret_cast<uint4*>(keys);
            uint4* c4 = reinterpret_cast<uint4*>(cnts);
#pragma unroll
            for (uint32_t i = 0; i < kSlots * sizeof(pool_t) / 16 / 64; ++i) k4[i * 64 + lane] = make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
            for (uint32_t i = 0; i < (kSlots / 8 + 63) / 64; ++i) if (i * 64 + lane < kSlots / 8) c4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
        wave_lds_sync();
        uint32_t picked[kLaneK];
#pragma unroll
        for (uint32_t i = 0; i < kLaneK; ++i) picked[i] = 0xFFFFFFFFu;
        uint32_t strong = 0;
        mc_candidate_dev* out = cands + (size_t)q * K;
        auto body = [&](auto perc) {
            constexpr uint32_t PER = decltype(perc)::value;
            pool_t v[PER];
#pragma unroll
            for (uint32_t r = 0; r < PER; ++r) {
                if constexpr (kPrefetch) v[r] = cur[r < kPre ? r : 0];
                else v[r] = r * 64 + lane < n2 ? src[r * 64 + lane] : kNone;
            }
            strong = count_and_pick<LOG2S, PER, TAX>(v, keys, cnts, lane, maxWin, K, taxkey, tab, out, picked);
        };
        const uint32_t per = (n2 + 63u) / 64u;
        if constexpr (kList / 64 <= 8) {
            if (per <= 1) body(std::integral_constant<uint32_t, 1>{});
            else if (per <= 2) body(std::integral_constant<uint32_t, 2>{});
            else if (per <= 3) body(std::integral_constant<uint32_t, 3>{});
            else if (per <= 4) body(std::integral_constant<uint32_t, 4>{});
            else if (per <= 6) body(std::integral_constant<uint32_t, 6>{});
            else body(std::integral_constant<uint32_t, 8>{});
        } else {
            if (per <= 10) body(std::integral_constant<uint32_t, 10>{});
            else if (per <= 12) body(std::integral_constant<uint32_t, 12>{});
            else body(std::integral_constant<uint32_t, kList / 64>{});
        }
        strong = __builtin_amdgcn_readfirstlane(strong);
        bool done = true;
        if (strong < K) {
            if (TAX || nent > kBigEnt) {
                // places left for single-hit taxa: the order among those depends on every target's taxon -> the exact wave kernel
                // (so do the few queries of the filter's second instance that come here: step D below reads one entry per lane)
                if (lane == 0) { ws.hitScan[q] = ws.qstat[q].hits; ws.qflag[q] = kFlagCands; }
                done = false;
            } else if constexpr (!TAX) {
                // ---- D. the open places: smallest targets (with their smallest window) among the locations of all targets that were
                //      not picked with >= 2 hits -- every such target's best range is a single location
                const uint32_t fbase = ws.winOff[q] * s;
                const uint32_t mySz1[1] = {lane < nent ? (ws.psize[fbase + lane] & 0xFFFFu) : 0u};
                const uint64_t myPay1[1] = {lane < nent ? ws.ppay[fbase + lane] : 0ull};
                const uint32_t mySz = mySz1[0]; const uint64_t myPay = myPay1[0];
                const BigShape sh = big_setup<1>(T, lane, nent, mySz1, myPay1, kBigMinShift);
                wave_lds_sync();
                uint64_t best[kLaneK];
#pragma unroll
                for (uint32_t i = 0; i < kLaneK; ++i) best[i] = kEmptyLoc;
                auto visit = [&](uint64_t v) {
                    if (v == kEmptyLoc) return;
                    const uint32_t t = (uint32_t)(v >> 32);
                    bool skip = false;
#pragma unroll
                    for (uint32_t i = 0; i < kLaneK; ++i) skip = skip || (i < strong && picked[i] == t);
                    if (skip) return;
                    bool same = false;
#pragma unroll
                    for (uint32_t i = 0; i < kLaneK; ++i)
                        if (best[i] != kEmptyLoc && (uint32_t)(best[i] >> 32) == t) { same = true; best[i] = min(best[i], v); }
                    if (same) return;
                    uint64_t c = v;                                 // sorted insert, the largest falls out
#pragma unroll
                    for (uint32_t i = 0; i < kLaneK; ++i) { const uint64_t lo = min(best[i], c); c = max(best[i], c); best[i] = lo; }
                };
                visit((lane < nent && mySz == 1) ? myPay : kEmptyLoc);
                big_sweep(T, tab, lane, sh, visit);
                for (uint32_t rnd = strong; rnd < K; ++rnd) {
                    uint64_t m = best[0];
#pragma unroll
                    for (uint32_t off = 32; off > 0; off >>= 1) {
                        const uint64_t o = ((uint64_t)__shfl_xor((uint32_t)(m >> 32), off) << 32) | __shfl_xor((uint32_t)m, off);
                        m = o < m ? o : m;
                    }
                    mc_candidate_dev e; e.tgt = 0xFFFFFFFFu; e.hits = 0; e.beg = 0; e.end = 0;
                    if (m != kEmptyLoc) {
                        const uint32_t t = (uint32_t)(m >> 32);
                        e.tgt = t & tab.tgtMask; e.hits = 1; e.beg = e.end = (uint32_t)m;
#pragma unroll
                        for (uint32_t i = 0; i < kLaneK; ++i) {      // that target leaves every lane's list
                            if (best[i] != kEmptyLoc && (uint32_t)(best[i] >> 32) == t) {
#pragma unroll
                                for (uint32_t j = i; j + 1 < kLaneK; ++j) best[j] = best[j + 1];
                                best[kLaneK - 1] = kEmptyLoc;
                            }
                        }
                    }
                    if (lane == 0) out[rnd] = e;
                }
            }
        }
        if (done && lane == 0) ws.qflag[q] = kFlagDone;
        wave_lds_sync();
    }
}

void launch_big_cands(uint32_t stage, const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                      const uint32_t* taxkey, void* cands, hipStream_t st)
{
    if (b.n == 0) return;
    // tables with the compact location store (global window numbers): their own kernels, gw_kernels.hip
    if (tab.values32) { launch_gw_cands(stage, b, sp, tab, ws, maxCand, taxkey, cands, st); return; }
    mc_candidate_dev* c = (mc_candidate_dev*)cands;
    // persistent grids.  stage 0: the filter; 1: counting of filtered lists up to 512; 2: 513 .. 1024
    auto count = [&](auto log2s, auto waves, uint32_t grid, uint32_t minN2) {
        constexpr uint32_t L = decltype(log2s)::value, W = decltype(waves)::value;
        if (taxkey) hipLaunchKernelGGL((big_count_kernel<L, W, true>), dim3(grid), dim3(W * 64), 0, st, b, sp.s, tab, ws, maxCand, taxkey, c, minN2);
        else        hipLaunchKernelGGL((big_count_kernel<L, W, false>), dim3(grid), dim3(W * 64), 0, st, b, sp.s, tab, ws, maxCand, taxkey, c, minN2);
    };
    if (stage == 0) {
        hipLaunchKernelGGL((big_filter_kernel<4, 1, false, kBigT1Log2, kBigT2Log2>), dim3(big_filter_grid(b.n, false, ws.filterBpc)), dim3(256), 0, st, b, tab, ws);
    } else if (stage == 3) {                                   // the filter's second instance: queries with 65 .. 192 found features
        // two waves per block, twice the blocks: the same number of waves -- and so the same pool slices -- as the first instance
        hipLaunchKernelGGL((big_filter_kernel<2, kBigEPL, true, MC_BIG_POS_T1, MC_BIG_POS_T2>), dim3(2 * big_filter_grid(b.n, false, ws.filterBpc)), dim3(128), 0, st, b, tab, ws);
    } else if (stage == 1) {
        // blocks per CU by LDS: 40 KB per block
        count(std::integral_constant<uint32_t, 10>{}, std::integral_constant<uint32_t, 4>{}, std::min<uint32_t>(256 * big_count_bpc(false), (b.n + 3) / 4), 0u);
    } else if (stage == 2) {
        static const uint32_t env2 = [] { const char* e = std::getenv("MC_BIG_COUNT2_BPC"); return e ? (uint32_t)std::max(1, std::atoi(e)) : 0u; }();
        const uint32_t bpc2 = env2 ? env2 : 4u;
        count(std::integral_constant<uint32_t, 11>{}, std::integral_constant<uint32_t, 2>{}, std::min<uint32_t>(256 * bpc2, (b.n + 1) / 2), 512u);
    }
}
// Persistent grids of 4-wave blocks; what matters is how many of them a CU HOLDS at a time.  big_filter_kernel (8-byte store): 22.5 KB of
// LDS per block, 66 registers: seven per CU (4 / 5 / 6 blocks with the 4 KB state table: 18.4 / 15.9 / 15.0 ms).  gw_filter_kernel
// (compact store): 96 registers = five waves per SIMD = five blocks per CU; a grid of seven left two blocks per CU to start when the
// first five had done their whole share (11.3 ms; six: 12.3).  Whole rounds of five: 5 -> 10.4 ms, 10 -> 10.2, 20 -> 9.96, 25 -> 9.92 ms per 5 x 10^6 reads (finer shares end closer together).
uint32_t big_filter_grid(uint32_t n, bool compact, int bpcOverride)
{
    if (bpcOverride > 0) return std::min<uint32_t>(256u * (uint32_t)bpcOverride, (n + 3) / 4);
    static const uint32_t env = [] { const char* e = std::getenv("MC_BIG_FILTER_BPC"); return e ? (uint32_t)std::max(1, std::atoi(e)) : 0u; }();
    const uint32_t bpc = env ? env : compact ? 48u : 7u;   // (compact: 20 -> 40 -> 80 blocks per CU with two batches in flight: 19.05 / 18.56 / 18.55 ms per step -- finer shares let the other pipe's kernels in sooner;
                                                           // round 6: 48 = whole rounds of the six blocks a CU holds of the fused kernel, and of the four of the pair filter: 36 / 40 / 48 / 60: 16.88 / 16.95 / 16.67 / 16.71 ms)
    return std::min<uint32_t>(256 * bpc, (n + 3) / 4);
}

void launch_hash_cands(uint32_t cls, const BatchView& b, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand, const uint32_t* taxkey, void* cands,
                       hipStream_t st)
{
    if (b.n == 0) return;
    // persistent grids; LDS per block: 52 KB (512), 50 KB (1024), 40 KB (256)
    if (cls == 3) {
        const uint32_t grid = std::min<uint32_t>(256 * 3, (b.n + 3) / 4);
        if (taxkey) hipLaunchKernelGGL((hash_cands_kernel<10, 4, true>), dim3(grid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, (mc_candidate_dev*)cands, cls);
        else        hipLaunchKernelGGL((hash_cands_kernel<10, 4, false>), dim3(grid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, (mc_candidate_dev*)cands, cls);
    } else if (cls == 4) {
        const uint32_t grid = std::min<uint32_t>(256 * 3, (b.n + 1) / 2);
        if (taxkey) hipLaunchKernelGGL((hash_cands_kernel<11, 2, true>), dim3(grid), dim3(128), 0, st, b, tab, ws, maxCand, taxkey, (mc_candidate_dev*)cands, cls);
        else        hipLaunchKernelGGL((hash_cands_kernel<11, 2, false>), dim3(grid), dim3(128), 0, st, b, tab, ws, maxCand, taxkey, (mc_candidate_dev*)cands, cls);
    } else {
        const uint32_t grid = std::min<uint32_t>(256 * 4, (b.n + 3) / 4);
        if (taxkey) hipLaunchKernelGGL((hash_cands_kernel<9, 4, true>), dim3(grid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, (mc_candidate_dev*)cands, cls);
        else        hipLaunchKernelGGL((hash_cands_kernel<9, 4, false>), dim3(grid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, (mc_candidate_dev*)cands, cls);
    }
}
// Compact store: reads the wave kernel sketched and probed (equal hashes inside a sketch, odd sketching parameters, pairs with a long
// mate) join the filtered path's work list when it can take them -- their feature slots, found or not, are the entries -- instead of
// having their whole location lists sorted by sort_candidates_kernel (a 10 kbp read: 3 x 10^4 locations, 0.2 ms of ONE wave).
__global__ __launch_bounds__(256) void wave_rejoin_kernel(BatchView b, uint32_t s, DeviceTable tab, Workspace ws)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u;
    bool join = false;
    uint32_t H = 0, slots = 0, mw = 0;
    if (q < b.n && ws.qflag[q] == kFlagCands) {
        H = ws.qstat[q].hits; slots = (ws.winOff[q + 1] - ws.winOff[q]) * s; mw = b.maxWin ? b.maxWin[q] : b.maxWinUniform;
        join = H > ws.bigMin && H > 64u && H <= kMaxHitsPerQuery && slots <= 0xFFFu && mw <= tab.gwGap;
    }
    const uint64_t m = __ballot(join);
    if (!m) return;
    const uint32_t leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&ws.midCount[9], (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    const uint64_t large = __ballot(join && H > kGwSmallH);
    if (large && lane == leader) atomicAdd(&ws.midCount[10], (uint32_t)__popcll(large));
    if (join) {
        reinterpret_cast<uint4*>(ws.midList)[(size_t)6 * b.n + base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint4(q, ws.winOff[q] * s, slots | (H << 12), mw);
        ws.hitScan[q] = 0u; ws.qflag[q] = kFlagMid;
    }
}
void launch_wave_rejoin(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, hipStream_t st)
{
    if (b.n && tab.values32) hipLaunchKernelGGL(wave_rejoin_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b, sp.s, tab, ws);
}

// Mode P / part groups: the per-part top lists of a read, fed IN PART ORDER through the CPU's list insert (candidate_generation.hpp:
// 172-231) -- what querying the parts one after the other with one candidate list gives (host_hashmap.hpp:695-723 concatenates the
// parts' sorted lists; a target belongs to one part, so per-part lists are what the scan of the concatenation yields).  One lane per read.
struct PartLists { const mc_candidate_dev* p[16]; };
__global__ __launch_bounds__(256) void merge_parts_kernel(PartLists L, uint32_t nlists, uint32_t n, uint32_t K, const uint32_t* __restrict__ taxkey,
                                                          mc_candidate_dev* out)   // (no __restrict__: rounds of more than 16 lists pass the output as list 0 -- a lane reads its own rows before it writes them)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    LaneCand top[kLaneK];
    uint32_t toptax[kLaneK];
#pragma unroll
    for (uint32_t i = 0; i < kLaneK; ++i) { top[i].tgt = 0xFFFFFFFFu; top[i].hits = 0; top[i].beg = 0; top[i].end = 0; toptax[i] = 0; }
    for (uint32_t l = 0; l < nlists; ++l) {
        const mc_candidate_dev* c = L.p[l] + (size_t)q * K;
        for (uint32_t j = 0; j < K; ++j) {
            const mc_candidate_dev x = c[j];
            if (x.hits == 0) break;
            LaneCand lc; lc.tgt = x.tgt; lc.hits = x.hits; lc.beg = x.beg; lc.end = x.end;
            top_insert(top, toptax, lc, K, taxkey, 0xFFFFFFFFu);
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < kLaneK; ++i)
        if (i < K) {
            mc_candidate_dev e; e.tgt = top[i].hits ? top[i].tgt : 0xFFFFFFFFu; e.hits = top[i].hits; e.beg = top[i].beg; e.end = top[i].end;
            out[(size_t)q * K + i] = e;
        }
}
int launch_merge_parts(const void* const* lists, uint32_t nlists, uint32_t n, uint32_t K, const uint32_t* taxkey, void* out, hipStream_t st)
{
    if (nlists == 0 || nlists > 16 || K > kLaneK) return -1;
    if (n == 0) return 0;
    PartLists L{};
    for (uint32_t i = 0; i < nlists; ++i) L.p[i] = (const mc_candidate_dev*)lists[i];
    hipLaunchKernelGGL(merge_parts_kernel, dim3((n + 255) / 256), dim3(256), 0, st, L, nlists, n, K, taxkey, (mc_candidate_dev*)out);
    return 0;
}

// after the lane kernels: how many queries are left for the wave kernels (midCount[6]: to be sketched, [7]: candidates from a list in
// HBM).  The host reads the eight counters once and launches only the kernels that have work -- a batch of 65 536 short reads spent
// a fifth of its device time on launches of kernels with nothing to do.
__global__ __launch_bounds__(256) void flag_count_kernel(const uint32_t* __restrict__ qflag, uint32_t n, uint32_t* __restrict__ counts)
{
    uint32_t a = 0, c = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t f = qflag[i];
        a += f == kFlagSketch; c += f == kFlagCands;
    }
    const uint64_t ma = __ballot(a != 0), mc = __ballot(c != 0);
    if (ma) { for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off); if ((threadIdx.x & 63) == 0) atomicAdd(&counts[6], a); }
    if (mc) { for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off); if ((threadIdx.x & 63) == 0) atomicAdd(&counts[7], c); }
}

void launch_flag_count(const Workspace& ws, uint32_t n, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(flag_count_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 1024u)), dim3(256), 0, st, ws.qflag, n, ws.midCount);
}
// small batches: ONE block counts, and hands the sixteen work-list counters to the host (pinned memory) itself
__global__ __launch_bounds__(256) void flag_count_small_kernel(const uint32_t* __restrict__ qflag, uint32_t n, uint32_t* __restrict__ counts, uint32_t* __restrict__ hostCounts)
{
    uint32_t a = 0, c = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t f = qflag[i];
        a += f == kFlagSketch; c += f == kFlagCands;
    }
    const uint64_t ma = __ballot(a != 0), mc = __ballot(c != 0);
    if (ma) { for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off); if ((threadIdx.x & 63) == 0) atomicAdd(&counts[6], a); }
    if (mc) { for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off); if ((threadIdx.x & 63) == 0) atomicAdd(&counts[7], c); }
    __threadfence();
    __syncthreads();
    if (threadIdx.x < 16) { hostCounts[threadIdx.x] = __hip_atomic_load(&counts[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __threadfence_system(); }
}
void launch_flag_count_host(const Workspace& ws, uint32_t n, uint32_t* hostCounts, hipStream_t st)
{
    if (n && n <= kSmallPlan) { hipLaunchKernelGGL(flag_count_small_kernel, dim3(1), dim3(256), 0, st, ws.qflag, n, ws.midCount, hostCounts); return; }
    launch_flag_count(ws, n, st);
    launch_words_to_host(hostCounts, ws.midCount, 16, st);
}

// A few words of device memory to PINNED HOST memory by a one-wave kernel on the batch's own stream: what the host looks at inside a
// batch (work-list counters, segment totals).  hipMemcpyAsync would take them through a copy engine -- a hop to another queue and back
// for 64 bytes, which is most of a small batch's latency (docs/LAB_NOTEBOOK_r06.md section 5).
__global__ __launch_bounds__(64) void words_to_host_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ hostDst, uint32_t nwords)
{
    for (uint32_t i = threadIdx.x; i < nwords; i += 64) hostDst[i] = src[i];
    __threadfence_system();
}
void launch_words_to_host(uint32_t* hostDst, const uint32_t* src, uint32_t nwords, hipStream_t st)
{
    hipLaunchKernelGGL(words_to_host_kernel, dim3(1), dim3(64), 0, st, src, hostDst, nwords);
}

__global__ __launch_bounds__(256) void deliver_kernel(DeliverTable t, const uint4* __restrict__ cands, const uint4* __restrict__ qstat, uint32_t K)
{
    const DeliverEntry e = t.e[blockIdx.y];
    uint4* __restrict__ dc = reinterpret_cast<uint4*>(e.cands);
    uint4* __restrict__ dq = reinterpret_cast<uint4*>(e.qstat);
    const uint32_t nc = e.count * K;
    const uint4* __restrict__ sc = cands + (size_t)e.first * K;
    const uint4* __restrict__ sq = qstat + e.first;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nc; i += gridDim.x * 256) dc[i] = sc[i];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < e.count; i += gridDim.x * 256) dq[i] = sq[i];
    __threadfence_system();
}
void launch_deliver(const DeliverTable& t, const void* cands, const void* qstat, uint32_t K, hipStream_t st)
{
    static_assert(sizeof(mc_candidate_dev) == 16 && sizeof(QueryStat) == 16, "whole uint4 records");
    if (!t.n) return;
    uint32_t most = 0;
    for (uint32_t i = 0; i < t.n; ++i) most = std::max(most, t.e[i].count * K);
    hipLaunchKernelGGL(deliver_kernel, dim3(std::max<uint32_t>(1, std::min<uint32_t>(64, (most + 1023) / 1024)), t.n), dim3(256), 0, st, t, (const uint4*)cands, (const uint4*)qstat, K);
}

bool lane_path_supported(const SketchParams& sp) { return sp.s <= kLaneS && sp.stride == sp.w - sp.k + 1 && sp.k <= 16; }
bool lane_candidates_supported(uint32_t maxCand) { return maxCand <= kLaneK; }

// ================================================================================================
// batch statistics (on demand, not on the timed path)
// ================================================================================================
__global__ __launch_bounds__(256) void batch_stats_kernel(const QueryStat* __restrict__ qs, const uint32_t* __restrict__ winOff,
                                                          uint32_t n, uint64_t* __restrict__ stats)
{
    __shared__ uint64_t sh[4];
    uint64_t h = 0, f = 0, fo = 0, st = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        QueryStat s = qs[i];
        h += s.hits; f += s.nfeat; fo += s.nfound; st += s.nsteps;
    }
    h = block_reduce_u64(h, sh); f = block_reduce_u64(f, sh); fo = block_reduce_u64(fo, sh); st = block_reduce_u64(st, sh);
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long*)&stats[1], (unsigned long long)f);
        atomicAdd((unsigned long long*)&stats[2], (unsigned long long)h);
        atomicAdd((unsigned long long*)&stats[3], (unsigned long long)fo);
        atomicAdd((unsigned long long*)&stats[4], (unsigned long long)st);
        if (blockIdx.x == 0) stats[0] = winOff[n];
    }
}

// the filtered path's records of the batch (list 7): [5] = locations kept by the filter, [6] = reads that took the filtered path |
// those with more than 512 kept << 32, [7] = reads the first filter kernel left to the second (compact store) | handed to the wave kernel << 32
__global__ __launch_bounds__(256) void big_stats_kernel(const uint32_t* __restrict__ midCount, const uint4* __restrict__ list7, uint64_t* __restrict__ stats, uint32_t overMin)
{
    const uint32_t total = midCount[9];
    unsigned long long kept = 0; uint32_t over = 0, fb = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint32_t n2 = list7[i].z;
        if (n2 >= 0xFFFFFFFEu) ++fb;
        else { n2 &= 0x7FFFFFFFu; kept += n2; over += n2 > overMin ? 1u : 0u; }   // (bit 31: counted inside the filter kernel, gw_filter_count_kernel)
    }
    atomicAdd((unsigned long long*)&stats[5], kept);
    atomicAdd((unsigned long long*)&stats[6], (unsigned long long)over << 32);
    atomicAdd((unsigned long long*)&stats[7], (unsigned long long)fb << 32);
    if (blockIdx.x == 0 && threadIdx.x == 0) { atomicAdd((unsigned long long*)&stats[6], (unsigned long long)total); atomicAdd((unsigned long long*)&stats[7], (unsigned long long)midCount[10]); }
}

void launch_batch_stats(const Workspace& ws, uint32_t n, hipStream_t st)
{
    (void)hipMemsetAsync(ws.stats, 0, 8 * sizeof(uint64_t), st);
    if (n == 0) return;
    uint32_t blocks = min((n + 255u) / 256u, 1024u);
    hipLaunchKernelGGL(batch_stats_kernel, dim3(blocks), dim3(256), 0, st, ws.qstat, ws.winOff, n, ws.stats);
    // ("more than 512 kept"; MC_STATS_OVER=n: another threshold, for looking at the distribution of the filtered lists' lengths)
    static const uint32_t overMin = [] { const char* e = std::getenv("MC_STATS_OVER"); return e ? (uint32_t)std::max(0, std::atoi(e)) : 512u; }();
    if (ws.midCount) hipLaunchKernelGGL(big_stats_kernel, dim3(blocks), dim3(256), 0, st, ws.midCount, reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n, ws.stats, overMin);
}

}  // namespace mcamd
