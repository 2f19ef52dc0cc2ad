// metacache_amd/csrc/gw_sort.hip -- the filtered lists the counting kernels do not take (more than 1024 numbers, or window ranges wider
// than 8: long reads, pairs with a large insert size) are SORTED -- 32-bit global window numbers, whose order is the order of
// (target, window) (query_handler.hpp:75-101 sorts locations by exactly that; the reference's GPU path sorts segments inside the query
// path too, query_batch.cu:27-61, :563-564).  gw_sorted_cands_kernel (gw_kernels.hip) reads the result.
//
// Round 5: hand-written -- a segmented MERGE sort (rounds 3 and 4 called rocPRIM's segmented radix sort here, its largest single kernel
// on configs[4]'s reads; a bitonic block sort of round 4 had lost to it).  Why merging and not radix passes: a block radix sort needs a
// stable rank per key and pass -- on wave64 that is either one ballot per digit bit and key (~50 VALU instructions per key and 8-bit
// pass) or per-thread digit counters (4-bit digits, eight passes, a block-wide scan of packed counters each) -- while a merge round is
// 10 instructions per key with no stability question at all:
//   gw_sort_small / _chunk   ONE block sorts one list of up to 16 x THREADS numbers in LDS: every thread sorts 16 numbers in registers
//                          (Batcher's network, 63 exchanges), then log2(threads) merge rounds -- per round a thread finds where its 16
//                          outputs begin in the two runs (merge path: a binary search on its diagonal), loads 16 candidates of either
//                          run and keeps the 16 smallest of the 32 (a bitonic half-cleaner + a 16-key merge network: 80 min / max).
//                          One read of the list from HBM, one write.  Instances by list length: 128 / 256 / 512 threads (whole lists up to
//                          2 048 / 4 096 / 8 192), 1 024 (up to 16 384, and the 16 384-number chunks of longer lists).
//   gw_merge_pass_kernel   lists beyond 16 384 numbers (reads of 5 kbp and more: 4 400 of a batch's 76 800 sorted lists, 90 of its
//                          235 x 10^6 numbers): their sorted chunks are merged pairwise through HBM, ceil(log2(chunks)) passes (three
//                          for a 19 kbp read's 10^5 numbers, where an LSD radix sort takes four per 8 bits) of streaming reads and
//                          writes; a block merges a TILE of 4 096 outputs: its ends found by a wave-wide 64-ary search on the tile's
//                          two diagonals, the two pieces loaded into LDS, then as in the block kernel.
// Work lists on the device (the host knows how many lists, not their lengths): gw_sort_plan_kernel writes chunks and tiles per list,
// two scans (launch_scan_u32) turn them into offsets, blocks find their list by a binary search.
// launch_gw_order (longest first, for the kernels that take one list per wave or block): a counting sort over 4 096 length classes --
// the order inside a class is whatever the atomics give, which changes the schedule, never a result.
#include "device_common.h"

#include <algorithm>
#include <cstdlib>

namespace mcamd {

namespace {

constexpr uint32_t kInf = 0xFFFFFFFFu;                    // never a stored number: padding sorts to the end
constexpr uint32_t kWholeMax = 8192, kChunk = 16384, kTile = 4096;   // whole lists by length class (blocks of 128 / 256 / 512 threads) up to kWholeMax; chunks of 16 384
constexpr uint32_t kMaxPasses = 6;                        // 16 384 << 6 = 2^20 > kGwMaxKept

// ascending, in place: Batcher's odd-even merge sort of 16 (63 compare-exchanges)
__device__ __forceinline__ void sort16(uint32_t (&x)[16])
{
    auto ce = [&](const uint32_t i, const uint32_t j) { const uint32_t lo = min(x[i], x[j]); x[j] = max(x[i], x[j]); x[i] = lo; };
    ce(0, 1); ce(2, 3); ce(0, 2); ce(1, 3); ce(1, 2); ce(4, 5); ce(6, 7); ce(4, 6); ce(5, 7); ce(5, 6); ce(0, 4); ce(2, 6); ce(2, 4);
    ce(1, 5); ce(3, 7); ce(3, 5); ce(1, 2); ce(3, 4); ce(5, 6); ce(8, 9); ce(10, 11); ce(8, 10); ce(9, 11); ce(9, 10); ce(12, 13);
    ce(14, 15); ce(12, 14); ce(13, 15); ce(13, 14); ce(8, 12); ce(10, 14); ce(10, 12); ce(9, 13); ce(11, 15); ce(11, 13); ce(9, 10);
    ce(11, 12); ce(13, 14); ce(0, 8); ce(4, 12); ce(4, 8); ce(2, 10); ce(6, 14); ce(6, 10); ce(2, 4); ce(6, 8); ce(10, 12); ce(1, 9);
    ce(5, 13); ce(5, 9); ce(3, 11); ce(7, 15); ce(7, 11); ce(3, 5); ce(7, 9); ce(11, 13); ce(1, 2); ce(3, 4); ce(5, 6); ce(7, 8);
    ce(9, 10); ce(11, 12); ce(13, 14);
}
// a <- the 16 smallest of a and b (both ascending), ascending: min(a[i], b[15 - i]) is the lower half of a bitonic sequence, four
// merge stages sort it
__device__ __forceinline__ void low16(uint32_t (&a)[16], const uint32_t (&b)[16])
{
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) a[i] = min(a[i], b[15 - i]);
#pragma unroll
    for (uint32_t dist = 8; dist > 0; dist >>= 1)
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i)
            if ((i & dist) == 0) { const uint32_t lo = min(a[i], a[i + dist]); a[i + dist] = max(a[i], a[i + dist]); a[i] = lo; }
}
// LDS layout of the blocks' arrays: one word of padding behind every 16 -- thread t's 16 numbers begin at word 17 t, so the lanes of a
// wave hit 32 different banks whichever of their 16 numbers they touch together (unpadded, 16 t + u is two banks for all 64 lanes).
__device__ __forceinline__ uint32_t phys(uint32_t x) { return x + (x >> 4); }
constexpr uint32_t padded(uint32_t n) { return n + n / 16; }
// merge path: how many of the first d outputs of merge(A, B) come from A (ties: A first).  A = s[a0 ..), B = s[b0 ..) (logical places of
// the padded array), ascending, lengths la, lb, d <= la + lb.
__device__ __forceinline__ uint32_t merge_path(const uint32_t* s, uint32_t a0, uint32_t la, uint32_t b0, uint32_t lb, uint32_t d)
{
    uint32_t lo = d > lb ? d - lb : 0u, hi = min(d, la);
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s[phys(a0 + mid)] <= s[phys(b0 + d - 1u - mid)]) lo = mid + 1u; else hi = mid;
    }
    return lo;
}
// the 16 outputs from diagonal d on of merge(A, B) -> r (ascending; kInf past the end)
__device__ __forceinline__ void merge16_at(const uint32_t* s, uint32_t a0, uint32_t la, uint32_t b0, uint32_t lb, uint32_t d, uint32_t (&r)[16])
{
    const uint32_t i = merge_path(s, a0, la, b0, lb, d), j = d - i;
    uint32_t b[16];
#pragma unroll
    for (uint32_t u = 0; u < 16; ++u) { r[u] = i + u < la ? s[phys(a0 + i + u)] : kInf; b[u] = j + u < lb ? s[phys(b0 + j + u)] : kInf; }
    low16(r, b);
}

// ---- one block, one list (or chunk) of len <= 16 x THREADS numbers: in -> out, ascending.  s: LDS, padded(16 x THREADS) words.
// (The next list's numbers fetched into registers while this one is sorted -- no memory latency left on the block's path -- was measured:
// 16 registers more, 89 .. 124 in all, fewer blocks per CU: 2.45 ms for the four instances against 2.11.)
template <uint32_t THREADS>
__device__ __forceinline__ void block_sort(uint32_t* s, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t len)
{
    const uint32_t t = threadIdx.x;
    uint32_t nT = 64;                                              // threads at work: a power of two, 16 numbers each
    while (nT * 16u < len) nT <<= 1;
    const uint32_t nPad = nT * 16u;
    for (uint32_t j = t; j < nPad; j += THREADS) s[phys(j)] = j < len ? in[j] : kInf;
    __syncthreads();
    uint32_t k[16];
    uint32_t* const mine = s + t * 17u;                            // = phys(16 t)
    if (t < nT) {
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) k[u] = mine[u];
        sort16(k);
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) mine[u] = k[u];
    }
    for (uint32_t L = 16; L < nPad; L <<= 1) {                     // runs of L -> runs of 2 L
        __syncthreads();
        if (t < nT) {
            const uint32_t o = t * 16u, pair = o / (2u * L), d = o - pair * 2u * L, a0 = pair * 2u * L;
            merge16_at(s, a0, L, a0 + L, L, d, k);
        }
        __syncthreads();                                           // every thread has read its candidates
        if (t < nT) {
#pragma unroll
            for (uint32_t u = 0; u < 16; ++u) mine[u] = k[u];
        }
    }
    __syncthreads();
    for (uint32_t j = t; j < len; j += THREADS) out[j] = s[phys(j)];
    __syncthreads();                                               // (the next list takes the same LDS)
}

// passes a list of len numbers needs behind its chunks' block sorts
__device__ __forceinline__ uint32_t merge_passes(uint32_t len)
{
    uint32_t p = 0;
    for (uint32_t run = kChunk; run < len; run <<= 1) ++p;
    return p;
}
// where a long list's data lies after `done` of its `passes` merge passes: the last pass writes `out`, the buffers alternate backwards
__device__ __forceinline__ uint32_t* pass_buffer(uint32_t passes, uint32_t done, uint32_t* out, uint32_t* tmp) { return ((passes - done) & 1u) ? tmp : out; }

// per list of the sorted class: chunks of the 1 024-thread instance (0: the 256-thread instance's list) and tiles of the merge passes
__global__ __launch_bounds__(256) void gw_sort_plan_kernel(Workspace ws, uint32_t n, uint32_t nseg, uint32_t* __restrict__ items, uint32_t* __restrict__ tiles,
                                                            uint32_t* __restrict__ classCount)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t len = 0;
    if (i < nseg && i < ws.midCount[13]) len = (reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n)[ws.sideList[(size_t)3 * n + i]].z;
    if (i < nseg) { items[i] = len <= kWholeMax ? 0u : (len + kChunk - 1u) / kChunk; tiles[i] = len <= kChunk ? 0u : (len + kTile - 1u) / kTile; }
    // how many lists every instance takes: the lists come longest first (launch_gw_order: classes of 256 lengths whose borders are the
    // instances'), so instance c's lists are the places [sum of the counts before, + count) -- nobody walks over the others' lists
    const uint32_t cls = len > kWholeMax ? 0u : len > 4096u ? 1u : len > 2048u ? 2u : len ? 3u : 4u;
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        const uint64_t m = __ballot(cls == c);
        if (m && (threadIdx.x & 63u) == (uint32_t)__ffsll((unsigned long long)m) - 1u) atomicAdd(&classCount[c], (uint32_t)__popcll(m));
    }
}

// whole lists by length class: a block of THREADS threads takes the lists of 8 x THREADS + 1 .. 16 x THREADS numbers (the smallest
// instance all shorter ones as well)
template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void gw_sort_lists_kernel(Workspace ws, uint32_t n, const uint32_t* __restrict__ classCount, uint32_t cls,
                                                                const uint32_t* __restrict__ in, uint32_t* __restrict__ out)
{
    __shared__ uint32_t s[padded(THREADS * 16)];
    const uint4* __restrict__ rec = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n;
    const uint32_t* __restrict__ side = ws.sideList + (size_t)3 * n;
    uint32_t first = 0;
    for (uint32_t c = 0; c < cls; ++c) first += classCount[c];
    const uint32_t end = first + classCount[cls];
    for (uint32_t i = first + blockIdx.x; i < end; i += gridDim.x) {
        const uint4 r = rec[side[i]];
        if (r.z == 0u || r.z > THREADS * 16u) continue;            // (cannot happen: the order's classes are the instances')
        block_sort<THREADS>(s, in + r.y, out + r.y, r.z);
    }
}

// the list that holds item / tile `x`: the last i with off[i] <= x (off: exclusive scan, off[nseg] = total)
__device__ __forceinline__ uint32_t list_of(const uint32_t* __restrict__ off, uint32_t nseg, uint32_t x)
{
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}

// lists of 8 193 .. 16 384 numbers and the 16 384-number chunks of longer ones: a block of 1 024 threads each
__global__ __launch_bounds__(1024) void gw_sort_chunk_kernel(Workspace ws, uint32_t n, uint32_t nseg, const uint32_t* __restrict__ itemOff,
                                                              const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ tmp)
{
    __shared__ uint32_t s[padded(kChunk)];
    __shared__ uint32_t listS;
    const uint4* __restrict__ rec = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n;
    const uint32_t* __restrict__ side = ws.sideList + (size_t)3 * n;
    const uint32_t total = itemOff[nseg];
    for (uint32_t x = blockIdx.x; x < total; x += gridDim.x) {
        if (threadIdx.x == 0) listS = list_of(itemOff, nseg, x);
        __syncthreads();
        const uint32_t i = listS;
        const uint4 r = rec[side[i]];
        const uint32_t c = x - itemOff[i], at = c * kChunk, len = min(kChunk, r.z - at);
        uint32_t* dst = pass_buffer(merge_passes(r.z), 0u, out, tmp);
        block_sort<1024>(s, in + r.y + at, dst + r.y + at, len);
    }
}

// wave-wide merge path on global memory: 64 places of the diagonal's range are tried per step (three or four steps of two loads for
// runs of 10^5 numbers, where a binary search by one lane takes seventeen).  All 64 lanes of the wave call it; the result is uniform.
__device__ __forceinline__ uint32_t merge_path_wave(const uint32_t* __restrict__ A, uint32_t la, const uint32_t* __restrict__ B, uint32_t lb, uint32_t d, uint32_t lane)
{
    uint32_t lo = d > lb ? d - lb : 0u, hi = min(d, la);
    while (lo < hi) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t mid = lo + lane * step;                     // lane 0 tries lo itself
        const bool in = mid < hi;
        const bool take = in && A[mid] <= B[d - 1u - mid];         // monotone over the lanes: true ... true false ... false
        const uint32_t cnt = (uint32_t)__popcll(__ballot(take));
        // the answer lies behind the last place that takes (lane cnt - 1) and not behind the first that does not (lane cnt)
        const uint32_t nlo = cnt ? lo + (cnt - 1u) * step + 1u : lo;
        const uint32_t nhi = min(hi, lo + cnt * step);
        lo = nlo; hi = max(nlo, nhi);
    }
    return lo;
}

// merge pass `pass` (1 ..) of the lists beyond 16 384 numbers: sorted runs of 16 384 << (pass - 1) numbers are merged pairwise, a tile of
// 4 096 outputs per block and step
__global__ __launch_bounds__(256) void gw_merge_pass_kernel(Workspace ws, uint32_t n, uint32_t nseg, const uint32_t* __restrict__ tileOff, uint32_t pass,
                                                             uint32_t* __restrict__ out, uint32_t* __restrict__ tmp)
{
    __shared__ uint32_t s[padded(kTile) + 32];
    __shared__ uint32_t cutS[4];
    const uint4* __restrict__ rec = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n;
    const uint32_t* __restrict__ side = ws.sideList + (size_t)3 * n;
    const uint32_t total = tileOff[nseg];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    for (uint32_t x = blockIdx.x; x < total; x += gridDim.x) {
        if (t == 0) cutS[2] = list_of(tileOff, nseg, x);
        __syncthreads();
        const uint32_t i = cutS[2];
        const uint4 r = rec[side[i]];
        const uint32_t len = r.z, passes = merge_passes(len);
        if (pass > passes) { __syncthreads(); continue; }          // (block-uniform) this list is through
        const uint32_t* src = pass_buffer(passes, pass - 1u, out, tmp) + r.y;
        uint32_t* dst = pass_buffer(passes, pass, out, tmp) + r.y;
        const uint32_t R = kChunk << (pass - 1u);
        const uint32_t o = (x - tileOff[i]) * kTile, pair = o / (2u * R), base = pair * 2u * R;
        const uint32_t lenPair = min(2u * R, len - base), la = min(R, lenPair), lb = lenPair - la;
        const uint32_t d0 = o - base, d1 = min(d0 + kTile, lenPair);
        const uint32_t* A = src + base;
        const uint32_t* B = A + la;
        // the tile's ends in both runs: waves 0 and 1 search one diagonal each
        if (wave < 2) { const uint32_t cut = merge_path_wave(A, la, B, lb, wave ? d1 : d0, lane); if (lane == 0) cutS[wave] = cut; }
        __syncthreads();
        const uint32_t i0 = cutS[0], i1 = cutS[1], j0 = d0 - i0, j1 = d1 - i1, na = i1 - i0, nb = j1 - j0;
        for (uint32_t j = t; j < na; j += 256) s[phys(j)] = A[i0 + j];
        for (uint32_t j = t; j < nb; j += 256) s[phys(na + j)] = B[j0 + j];
        __syncthreads();
        const uint32_t dl = t * 16u;
        if (dl < na + nb) {
            uint32_t k[16];
            merge16_at(s, 0u, na, na, nb, dl, k);
#pragma unroll
            for (uint32_t u = 0; u < 16; ++u) if (dl + u < na + nb) dst[base + d0 + dl + u] = k[u];
        }
        __syncthreads();
    }
}

// ---- longest first: a counting sort over 4 096 length classes (length / 256, capped)
constexpr uint32_t kOrderClasses = 4096, kOrderSlots = 256 * 17;   // + one class behind all others for the entries past the list's end (key 0), padded to 17 per thread
// (the classes' borders are the sort instances' borders: lengths 2 048 / 2 049, 4 096 / 4 097, 8 192 / 8 193 fall into different classes)
__device__ __forceinline__ uint32_t order_class(uint32_t key) { return key ? kOrderClasses - 1u - min((key - 1u) >> 8, kOrderClasses - 1u) : kOrderClasses; }   // descending
__device__ __forceinline__ uint32_t order_key(const Workspace& ws, uint32_t n, uint32_t list, uint32_t i)
{
    // list 3 (sorted class): the numbers its filtered list holds; list 0 (reads of gw_filter_stream_kernel): the read's locations.
    // Entries beyond the list's length (device-side count) get key 0 and end up last.
    const uint32_t len = ws.midCount[list == 3 ? 13 : 12];
    if (i >= len) return 0u;
    const uint32_t w = ws.sideList[(size_t)list * n + i];
    return list == 3 ? (reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n)[w].z : (reinterpret_cast<const uint4*>(ws.midList) + (size_t)6 * n)[w].z >> 12;
}
__global__ __launch_bounds__(256) void order_hist_kernel(Workspace ws, uint32_t n, uint32_t list, uint32_t count, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[kOrderSlots];
    for (uint32_t j = threadIdx.x; j < kOrderSlots; j += 256) h[j] = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
        const uint32_t key = order_key(ws, n, list, i);
        if (key) atomicAdd(&h[order_class(key)], 1u);              // (entries past the list's end: nobody reads them -- 250 000 atomics on ONE counter took a millisecond)
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < kOrderSlots; j += 256) if (h[j]) atomicAdd(&hist[j], h[j]);
}
__global__ __launch_bounds__(256) void order_scan_kernel(uint32_t* __restrict__ hist)   // exclusive scan of the class counts, in place
{
    __shared__ uint32_t part[256];
    constexpr uint32_t kPer = kOrderSlots / 256;
    uint32_t v[kPer], sum = 0;
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) { v[u] = hist[threadIdx.x * kPer + u]; sum += v[u]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t j = 0; j < 256; ++j) { const uint32_t x = part[j]; part[j] = run; run += x; } }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) { hist[threadIdx.x * kPer + u] = run; run += v[u]; }
}
__global__ __launch_bounds__(256) void order_scatter_kernel(Workspace ws, uint32_t n, uint32_t list, uint32_t count, uint32_t* __restrict__ cursor, uint32_t* __restrict__ outv)
{
    // a block ranks its 256 entries per class in LDS and reserves the classes' places with one global atomic per class it holds
    // (76 800 lists sit in a few dozen classes: one atomic per entry on those few counters took 0.26 ms)
    __shared__ uint32_t cnt[kOrderSlots];
    for (uint32_t j = threadIdx.x; j < kOrderSlots; j += 256) cnt[j] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t key = i < count ? order_key(ws, n, list, i) : 0u;
    const uint32_t c = order_class(key);
    uint32_t local = 0;
    if (key) local = atomicAdd(&cnt[c], 1u);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < kOrderClasses; j += 256) { const uint32_t v = cnt[j]; if (v) cnt[j] = atomicAdd(&cursor[j], v); }
    __syncthreads();
    if (key) outv[cnt[c] + local] = ws.sideList[(size_t)list * n + i];
}
// back to the side list: the list's own entries only (what lies behind its end stays as it is)
__global__ __launch_bounds__(256) void order_copy_kernel(Workspace ws, uint32_t n, uint32_t list, uint32_t count, const uint32_t* __restrict__ outv)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < count && i < ws.midCount[list == 3 ? 13 : 12]) ws.sideList[(size_t)list * n + i] = outv[i];
}

}  // namespace

// Longest first.  The kernels that take ONE read per wave (or block) over a strided work list -- the stream filter, the sort's blocks,
// the scan of the sorted lists -- finished when the wave that happened to hold several 19 kbp reads did: with the records in descending
// order of their work a stride hands every wave the same mix.  scratch: count words + tempBytes (size query: scratch == nullptr).
int launch_gw_order(uint32_t list, const Workspace& ws, uint32_t n, uint32_t count, uint32_t* scratch, size_t& tempBytes, hipStream_t st)
{
    if (!scratch) { tempBytes = (size_t)kOrderSlots * 4 + 256; return 0; }
    if (count == 0) return 0;
    uint32_t* side = ws.sideList + (size_t)list * n;
    uint32_t* outv = scratch;
    uint32_t* hist = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(scratch + count) + 255u) & ~(uintptr_t)255u);
    if (hipMemsetAsync(hist, 0, (size_t)kOrderSlots * 4, st) != hipSuccess) return 1;
    hipLaunchKernelGGL(order_hist_kernel, dim3(std::min<uint32_t>((count + 255) / 256, 1024u)), dim3(256), 0, st, ws, n, list, count, hist);
    hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(256), 0, st, hist);
    hipLaunchKernelGGL(order_scatter_kernel, dim3((count + 255) / 256), dim3(256), 0, st, ws, n, list, count, hist, outv);
    hipLaunchKernelGGL(order_copy_kernel, dim3((count + 255) / 256), dim3(256), 0, st, ws, n, list, count, outv);
    return (int)hipGetLastError();
}

// segments = the first nseg records of the sorted class (ws.sideList[3]); temp: caller's buffer (size query with temp == nullptr)
int launch_gw_segsort(void* temp, size_t& tempBytes, const uint32_t* in, uint32_t* out, uint64_t poolCap, const Workspace& ws, uint32_t n, uint32_t nseg,
                      uint32_t endBit, hipStream_t st, const GwSortSide* side2)
{
    (void)endBit;                                                  // (a comparison sort: the numbers' width does not matter)
    const size_t planWords = 4 * ((size_t)nseg + 4) + 16;          // items, tiles, itemOff, tileOff, the instances' list counts
    const size_t scanBytes = scan_tmp_bytes(nseg + 1);
    const size_t planBytes = (planWords * 4 + 255) & ~(size_t)255, scanAt = planBytes, tmpAt = (scanAt + scanBytes + 255) & ~(size_t)255;
    if (!temp) { tempBytes = tmpAt + (size_t)poolCap * 4 + 256; return 0; }
    if (nseg == 0) return 0;
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(temp) + 255u) & ~(uintptr_t)255u);
    uint32_t* items = reinterpret_cast<uint32_t*>(base);
    uint32_t* tiles = items + nseg + 4, *itemOff = tiles + nseg + 4, *tileOff = itemOff + nseg + 4;
    uint32_t* classCount = tileOff + nseg + 4;
    uint32_t* tmp = reinterpret_cast<uint32_t*>(base + tmpAt);
    if (hipMemsetAsync(classCount, 0, 64, st) != hipSuccess) return 1;
    hipLaunchKernelGGL(gw_sort_plan_kernel, dim3((nseg + 255) / 256), dim3(256), 0, st, ws, n, nseg, items, tiles, classCount);
    launch_scan_u32(items, 1, nseg, itemOff, nullptr, base + scanAt, st);
    launch_scan_u32(tiles, 1, nseg, tileOff, nullptr, base + scanAt, st);
    // The instances' lists are disjoint: the shorter lists' instances run on a second stream beside the chunks and their merge passes (every
    // persistent grid ends with a tail of half-empty CUs; side by side the tails fill each other: fork / join with two events).
    hipStream_t s2 = st;
    if (side2 && side2->stream && hipEventRecord(side2->fork, st) == hipSuccess && hipStreamWaitEvent(side2->stream, side2->fork, 0) == hipSuccess) s2 = side2->stream;
    hipLaunchKernelGGL(gw_sort_chunk_kernel, dim3(256 * 2), dim3(1024), 0, st, ws, n, nseg, itemOff, in, out, tmp);
    hipLaunchKernelGGL(gw_sort_lists_kernel<128>, dim3(std::min<uint32_t>(nseg, 256u * 16u)), dim3(128), 0, s2, ws, n, classCount, 3u, in, out);
    hipLaunchKernelGGL(gw_sort_lists_kernel<256>, dim3(std::min<uint32_t>(nseg, 256u * 8u)), dim3(256), 0, s2, ws, n, classCount, 2u, in, out);
    hipLaunchKernelGGL(gw_sort_lists_kernel<512>, dim3(std::min<uint32_t>(nseg, 256u * 4u)), dim3(512), 0, s2, ws, n, classCount, 1u, in, out);
    for (uint32_t pass = 1; pass <= kMaxPasses; ++pass)
        hipLaunchKernelGGL(gw_merge_pass_kernel, dim3(256 * 8), dim3(256), 0, st, ws, n, nseg, tileOff, pass, out, tmp);
    if (s2 != st) { if (hipEventRecord(side2->join, s2) != hipSuccess || hipStreamWaitEvent(st, side2->join, 0) != hipSuccess) return 1; }
    return (int)hipGetLastError();
}

}  // namespace mcamd
