// metacache_amd/csrc/gw_sort.hip -- the filtered lists the counting kernels do not take (more than 1024 numbers, or window ranges wider
// than 8: long reads, pairs with a large insert size) are SORTED: one segmented radix sort (rocPRIM) over the pool of filtered lists,
// every such list a segment, 32-bit global window numbers as keys -- the order of the numbers is the order of (target, window)
// (query_handler.hpp:75-101 sorts locations by exactly that).  gw_sorted_cands_kernel (gw_kernels.hip) reads the result.
// A library sort, like the builder's: the hand-written kernels around it are where the path's time goes (a 10 kbp read keeps
// ~10^4 of its 3 x 10^4 locations; 2 Gbases/s are 2 x 10^9 keys per second, a tenth of what the sort delivers).
#include "device_common.h"

#include <algorithm>
#include <cstdlib>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace mcamd {

namespace {

// segment i = the list of the i-th record of the sorted class (ws.sideList[3], longest list first: launch_gw_order)
constexpr uint32_t kGwLdsSortMax = 16384;                 // longest list gw_lds_sort_kernel takes (64 KB of LDS)
struct SegOffset {
    const uint4* rec; const uint32_t* side; const uint32_t* midCount; uint32_t end, minLen;   // lists of up to minLen numbers are somebody else's: empty segments
    __device__ uint32_t operator()(uint32_t i) const
    {
        if (i >= midCount[13]) return 0u;
        const uint4 r = rec[side[i]];
        return r.y + ((end && r.z > minLen) ? r.z : 0u);
    }
};

// ---- the library's place for everything but the longest lists: ONE block sorts ONE list in LDS -- the list comes in from HBM once and goes
// out once (rocPRIM's segmented sort takes a block of 256 threads per list too, but lists beyond 4 352 numbers go through HBM radix pass
// by radix pass: 5.2 of 17.3 ms per step on configs[4]'s reads in round 3).  A bitonic network on the next power of two (padded with
// 0xFFFFFFFF, never a stored number): n log^2 n / 4 compare-exchanges with nothing but LDS and barriers in between -- for lists of
// 10^3 .. 10^4 numbers about the instruction count of a block radix sort's eight 4-bit passes, without its scans and scatters.
// The last strides of every merge stay inside a thread's own pair of elements' wave: the loop is the textbook one.
template <uint32_t THREADS, uint32_t CAP>
__global__ __launch_bounds__(THREADS) void gw_lds_sort_kernel(Workspace ws, uint32_t n, const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                              uint32_t minLen, uint32_t maxLen)
{
    __shared__ uint32_t s[CAP];
    const uint4* __restrict__ rec = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n;
    const uint32_t* __restrict__ side = ws.sideList + (size_t)3 * n;
    const uint32_t nseg = ws.midCount[13];
    for (uint32_t i = blockIdx.x; i < nseg; i += gridDim.x) {
        const uint4 r = rec[side[i]];
        const uint32_t len = r.z;
        if (len <= minLen || len > maxLen) continue;              // (block-uniform: another instance's list, or the library's)
        uint32_t M = 64;
        while (M < len) M <<= 1;
        for (uint32_t t = threadIdx.x; t < M; t += THREADS) s[t] = t < len ? in[r.y + t] : 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t k = 2; k <= M; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = threadIdx.x; t < M / 2; t += THREADS) {
                    const uint32_t lo = 2u * t - (t & (j - 1u));  // the pair (lo, lo + j): t with a zero bit put in at j's place
                    const uint32_t a = s[lo], b = s[lo + j];
                    const bool up = (lo & k) == 0u;
                    if ((a > b) == up) { s[lo] = b; s[lo + j] = a; }
                }
                __syncthreads();
            }
        }
        for (uint32_t t = threadIdx.x; t < len; t += THREADS) out[r.y + t] = s[t];
        __syncthreads();
    }
}

// keys of the ordering: the work a record stands for -- list 3 (sorted class): the numbers its filtered list holds; list 0 (reads of
// gw_filter_stream_kernel): the read's locations.  Entries beyond the list's length (device-side count) get key 0 and end up last.
__global__ __launch_bounds__(256) void order_keys_kernel(Workspace ws, uint32_t n, uint32_t list, uint32_t count, uint32_t* __restrict__ keys)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t len = ws.midCount[list == 3 ? 13 : 12];
    uint32_t k = 0;
    if (i < len) {
        const uint32_t w = ws.sideList[(size_t)list * n + i];
        k = list == 3 ? (reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n)[w].z : (reinterpret_cast<const uint4*>(ws.midList) + (size_t)6 * n)[w].z >> 12;
    }
    keys[i] = k;
}

}  // namespace

// Longest first.  The kernels that take ONE read per wave (or block) over a strided work list -- the stream filter, the segmented
// sort, the scan of the sorted lists -- finished when the wave that happened to hold several 19 kbp reads did: with the records in
// descending order of their work a stride hands every wave the same mix.  scratch: 3 x count words + tempBytes (size query: scratch == nullptr).
int launch_gw_order(uint32_t list, const Workspace& ws, uint32_t n, uint32_t count, uint32_t* scratch, size_t& tempBytes, hipStream_t st)
{
    uint32_t* side = ws.sideList + (size_t)list * n;
    uint32_t* keysIn = scratch, *keysOut = scratch ? scratch + count : nullptr, *valsOut = scratch ? scratch + 2 * (size_t)count : nullptr;
    // (rocPRIM lays its temporaries out from an aligned base: round up to 256 bytes -- the callers' + 256 bytes of slack are for this)
    void* temp = scratch ? reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(scratch + 3 * (size_t)count) + 255u) & ~(uintptr_t)255u) : nullptr;
    if (!scratch) return (int)rocprim::radix_sort_pairs_desc(nullptr, tempBytes, keysIn, keysOut, side, valsOut, count, 0u, 32u, st);
    if (count == 0) return 0;
    hipLaunchKernelGGL(order_keys_kernel, dim3((count + 255) / 256), dim3(256), 0, st, ws, n, list, count, keysIn);
    const int rc = (int)rocprim::radix_sort_pairs_desc(temp, tempBytes, keysIn, keysOut, side, valsOut, count, 0u, 32u, st);
    if (rc) return rc;
    return (int)hipMemcpyAsync(side, valsOut, (size_t)count * 4, hipMemcpyDeviceToDevice, st);
}

// segments = the first nseg records of the sorted class (ws.sideList[3]); temp: caller's buffer (size query with temp == nullptr)
int launch_gw_segsort(void* temp, size_t& tempBytes, const uint32_t* in, uint32_t* out, uint64_t poolCap, const Workspace& ws, uint32_t n, uint32_t nseg,
                      uint32_t endBit, hipStream_t st)
{
    const uint4* rec = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n;
    const uint32_t* side = ws.sideList + (size_t)3 * n;
    // lists up to kGwLdsSortMax numbers: our own block sort in LDS (two instances by length); the library keeps the longer ones -- its
    // segments of the others are empty
    // MEASURED SLOWER than the library on configs[4]'s reads at full scale (7.99 against 5.26 ms per 250 000 reads, profiles/r04_long_reads_own_sort.json:
    // 330 .. 525 instructions and two LDS round trips per key and stage-pair against the block radix sort's ~400 with far fewer barriers), so it is
    // OFF unless MC_GW_OWN_SORT=1 asks for it; kept as the measured baseline of the next attempt (a wave-ballot radix sort with the keys in registers)
    static const bool own = [] { const char* e = std::getenv("MC_GW_OWN_SORT"); return e && e[0] == '1'; }();
    const uint32_t libMin = own ? kGwLdsSortMax : 0u;
    if (temp && own && nseg) {
        hipLaunchKernelGGL((gw_lds_sort_kernel<256, 2048>), dim3(std::min<uint32_t>(nseg, 256u * 16u)), dim3(256), 0, st, ws, n, in, out, 0u, 2048u);
        hipLaunchKernelGGL((gw_lds_sort_kernel<1024, kGwLdsSortMax>), dim3(std::min<uint32_t>(nseg, 256u * 4u)), dim3(1024), 0, st, ws, n, in, out, 2048u, kGwLdsSortMax);
    }
    auto cnt = rocprim::make_counting_iterator<uint32_t>(0u);
    auto beg = rocprim::make_transform_iterator(cnt, SegOffset{rec, side, ws.midCount, 0u, libMin});
    auto end = rocprim::make_transform_iterator(cnt, SegOffset{rec, side, ws.midCount, 1u, libMin});
    // (the library's default configuration: a block of 256 threads sorts up to 4 352 numbers in registers and LDS, longer lists in passes
    // through HBM.  Larger single-block limits measured worse on configs[4]'s reads at full scale: 1024 x 8: 6.0 ms, 256 x 32: 7.0 ms, default 5.2)
    // This rocPRIM has no configuration tuned for gfx950: its generic default takes 6 bits per pass -- six passes through HBM over the 31 bits of
    // the lists a block cannot hold in LDS (beyond 4 352 numbers).  8 bits (the most a block of 256 threads ranks): four passes, the same block
    // shape: 4.29 -> 3.77 ms per 250 000 long reads (7 bits: 4.00; profiles/r04_exp_v4_long_b*.json).  MC_GW_SORT_BITS=6 / 7: the others.
    static const uint32_t bits = [] { const char* e = std::getenv("MC_GW_SORT_BITS"); return e ? (uint32_t)std::atoi(e) : 8u; }();
    const unsigned int poolN = (unsigned int)std::min<uint64_t>(poolCap, 0xFFFFFFFFull);
    using Warp = rocprim::WarpSortConfig<32, 4, 256, 3000, 32, 4, 256>;
    if (bits == 6) return (int)rocprim::segmented_radix_sort_keys(temp, tempBytes, in, out, poolN, nseg, beg, end, 0u, endBit, st);
    if (bits == 7) return (int)rocprim::segmented_radix_sort_keys<rocprim::segmented_radix_sort_config<7, rocprim::kernel_config<256, 17>, Warp, true>>(temp, tempBytes, in, out, poolN, nseg, beg, end, 0u, endBit, st);
    return (int)rocprim::segmented_radix_sort_keys<rocprim::segmented_radix_sort_config<8, rocprim::kernel_config<256, 17>, Warp, true>>(temp, tempBytes, in, out, poolN, nseg, beg, end, 0u, endBit, st);
}

}  // namespace mcamd
