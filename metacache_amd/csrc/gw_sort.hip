// metacache_amd/csrc/gw_sort.hip -- the filtered lists the counting kernels do not take (more than 1024 numbers, or window ranges wider
// than 8: long reads, pairs with a large insert size) are SORTED: one segmented radix sort (rocPRIM) over the pool of filtered lists,
// every such list a segment, 32-bit global window numbers as keys -- the order of the numbers is the order of (target, window)
// (query_handler.hpp:75-101 sorts locations by exactly that).  gw_sorted_cands_kernel (gw_kernels.hip) reads the result.
// A library sort, like the builder's: the hand-written kernels around it are where the path's time goes (a 10 kbp read keeps
// ~10^4 of its 3 x 10^4 locations; 2 Gbases/s are 2 x 10^9 keys per second, a tenth of what the sort delivers).
#include "device_common.h"

#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace mcamd {

namespace {

struct SegOffset {
    const uint4* rec; const uint32_t* midCount; uint32_t end;
    __device__ uint32_t operator()(uint32_t i) const
    {
        if (i >= midCount[9]) return 0u;
        const uint4 r = rec[i];
        if (!gw_sorted_class(r.z, r.w)) return 0u;
        return r.y + (end ? r.z : 0u);
    }
};

}  // namespace

// segments = the records of list 7 (n of them at most); temp: caller's buffer (size query with temp == nullptr)
int launch_gw_segsort(void* temp, size_t& tempBytes, const uint32_t* in, uint32_t* out, uint64_t poolCap, const Workspace& ws, uint32_t n, uint32_t endBit,
                      hipStream_t st)
{
    const uint4* rec = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n;
    auto cnt = rocprim::make_counting_iterator<uint32_t>(0u);
    auto beg = rocprim::make_transform_iterator(cnt, SegOffset{rec, ws.midCount, 0u});
    auto end = rocprim::make_transform_iterator(cnt, SegOffset{rec, ws.midCount, 1u});
    return (int)rocprim::segmented_radix_sort_keys(temp, tempBytes, in, out, (unsigned int)std::min<uint64_t>(poolCap, 0xFFFFFFFFull), n, beg, end, 0u, endBit, st);
}

}  // namespace mcamd
