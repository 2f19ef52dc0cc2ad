// metacache_amd/csrc/device_common.h -- what the kernel files (kernels.hip, gw_kernels.hip) share on the device side: per-query state
// flags, wave64 primitives, the candidate record and the limits of the work-list classes.  Internal.
#pragma once

#include "kernels.h"

namespace mcamd {

// per-query state: what is still to be done (Workspace::qflag)
constexpr uint32_t kFlagDone = 0, kFlagSketch = 1, kFlagCands = 2, kFlagProbe = 4, kFlagMid = 5, kFlagChunks = 6, kFlagGather = 7,
                   kFlagGatherAll = 8;   // as kFlagGather, the read's entries at their feature slots (long reads of the chunk lanes)

// ================================================================================================
// wave64 primitives
// ================================================================================================
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
// DPP controls: quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E, row_half_mirror=0x141, row_mirror=0x140
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, uint32_t l)
{
    return ((uint64_t)rdlane((uint32_t)(v >> 32), l) << 32) | rdlane((uint32_t)v, l);
}

// all 64 lanes must be active
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, dpp_mov<0xB1>(v));
    v = min(v, dpp_mov<0x4E>(v));
    v = min(v, dpp_mov<0x141>(v));
    v = min(v, dpp_mov<0x140>(v));
    return min(min(rdlane(v, 0), rdlane(v, 16)), min(rdlane(v, 32), rdlane(v, 48)));
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return rdlane(v, 0) + rdlane(v, 16) + rdlane(v, 32) + rdlane(v, 48);
}
// inclusive prefix sum over the wave in DPP moves (row shifts inside the rows of 16, then the row sums broadcast down): no LDS
// round trips (the __shfl_up form is six ds_bpermute).  All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, uint32_t lane)
{
    (void)lane;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}
// the same with max (values >= 0: the lanes a shift leaves empty read as 0)
__device__ __forceinline__ uint32_t wave_incl_scan_max_u32(uint32_t v)
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));   // row_shr:8
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
    return v;
}
// orders this wave's LDS / global accesses (other lanes of the same wave read what this lane wrote).
// Heavy: waits for every outstanding global load AND store of the wave.
__device__ __forceinline__ void wave_mem_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// LDS-only flavour: DS operations of one wave execute in issue order, so all that is needed is that
// the compiler keeps the order and earlier DS results have landed; outstanding global stores are NOT
// drained (an s_waitcnt vmcnt(0) per window exposed the full HBM write latency).
__device__ __forceinline__ void wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, dpp_mov<0xB1>(v));
    v = max(v, dpp_mov<0x4E>(v));
    v = max(v, dpp_mov<0x141>(v));
    v = max(v, dpp_mov<0x140>(v));
    return max(max(rdlane(v, 0), rdlane(v, 16)), max(rdlane(v, 32), rdlane(v, 48)));
}


__device__ __forceinline__ uint32_t home_group(uint32_t key, uint32_t nbuckets)
{
    return (uint32_t)(((uint64_t)mix32(key) * nbuckets) >> 32);
}


// ================================================================================================
// lane-private table lookup: ONE lane looks one feature up with four 16-byte loads of one 64-byte bucket
// (keys, sizes, payloads).  Split in two so that callers can have several lookups in flight per lane.
// ================================================================================================
struct BucketRegs { uint4 k, sz, p0, p1; };      // key[0..3] | size[0..3] (4 x u16) + spare | payload[0..1] | payload[2..3]

__device__ __forceinline__ BucketRegs load_bucket(const DeviceTable& tab, uint32_t bi)
{
    const uint4* p = reinterpret_cast<const uint4*>(tab.buckets + bi);
    BucketRegs r; r.k = p[0]; r.sz = p[1]; r.p0 = p[2]; r.p1 = p[3];
    return r;
}
// Quad-cooperative bucket fetch.  A lane that loads its own 64-byte bucket with four 16-byte loads touches 64 different lines per
// load instruction; on tables far larger than the infinity cache every one of these accesses pays the full price (measured,
// tools/gather_bench3.hip: 52 G buckets/s on a 366 MB table, 16 G/s on 16 GB), whereas four neighbouring lanes reading the four
// quarters of ONE bucket stay at 48 G/s whatever the table size (tools/gather_bench.hip).  So the four lanes of a quad fetch the
// buckets of its members one after the other (lane i of the quad always loads quarter i) and a 4 x 4 register transpose inside the
// quad (DPP quad_perm, no LDS) gives every lane the four quarters of its own bucket.
// Both calls must be made by all four lanes of a quad together; kNoBucket = this lane wants nothing.
constexpr uint32_t kNoBucket = 0xFFFFFFFFu;
constexpr uint64_t kQuadTableBytes = 1ull << 30;               // tables larger than this are probed quad-cooperatively
struct QuadRaw { uint4 v[4]; };                                 // v[t] = my quarter of the bucket wanted by lane t of my quad

__device__ __forceinline__ void quad_issue(const DeviceTable& tab, uint32_t want, QuadRaw& raw)
{
    const uint32_t part = threadIdx.x & 3u;
    const uint32_t w[4] = {dpp_mov<0x00>(want), dpp_mov<0x55>(want), dpp_mov<0xAA>(want), dpp_mov<0xFF>(want)};   // quad broadcasts
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t)
        if (w[t] != kNoBucket) raw.v[t] = reinterpret_cast<const uint4*>(tab.buckets + w[t])[part];
}

template <int CTRL>
__device__ __forceinline__ uint4 dpp_mov4(uint4 v)
{
    return make_uint4(dpp_mov<CTRL>(v.x), dpp_mov<CTRL>(v.y), dpp_mov<CTRL>(v.z), dpp_mov<CTRL>(v.w));
}
__device__ __forceinline__ uint4 sel4(bool c, uint4 a, uint4 b) { return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }

__device__ __forceinline__ BucketRegs quad_collect(const QuadRaw& raw)
{
    // butterfly transpose: after the stage with distance d, register j of lane i holds what register j^d of lane i^d held wherever
    // bit d of i and of j differ
    const bool b0 = (threadIdx.x & 1u) != 0, b1 = (threadIdx.x & 2u) != 0;
    uint4 m0 = raw.v[0], m1 = raw.v[1], m2 = raw.v[2], m3 = raw.v[3];
    {
        const uint4 ra = dpp_mov4<0xB1>(sel4(b0, m0, m1));      // quad_perm [1,0,3,2]
        const uint4 rb = dpp_mov4<0xB1>(sel4(b0, m2, m3));
        m0 = sel4(b0, ra, m0); m1 = sel4(b0, m1, ra);
        m2 = sel4(b0, rb, m2); m3 = sel4(b0, m3, rb);
    }
    {
        const uint4 ra = dpp_mov4<0x4E>(sel4(b1, m0, m2));      // quad_perm [2,3,0,1]
        const uint4 rb = dpp_mov4<0x4E>(sel4(b1, m1, m3));
        m0 = sel4(b1, ra, m0); m2 = sel4(b1, m2, ra);
        m1 = sel4(b1, rb, m1); m3 = sel4(b1, m3, rb);
    }
    BucketRegs r; r.k = m0; r.sz = m1; r.p0 = m2; r.p1 = m3;
    return r;
}


struct mc_candidate_dev { uint32_t tgt, hits, beg, end; };

constexpr uint32_t kLaneK = 4;            // most candidates handled by one lane
constexpr uint32_t kMidMax = 256;         // longest list taken by mid_cands_kernel
constexpr uint32_t kHashMax = 1024, kHashEnt = 256, kHashWin = 8;   // hash_cands_kernel: longest list, entries, maxWindowsInRange
constexpr uint32_t kBigEnt = 64;          // big_filter_kernel: found features per query, one lane each ...
constexpr uint32_t kBigEPL = 3;           // ... or up to three per lane in its second instance (reads and pairs of 5 .. 10 windows: 2 x 250 bp, 500 bp)

struct LaneCand { uint32_t tgt, hits, beg, end; };

// rows 10: one candidate enters the lane's top list exactly as on the CPU (candidate_generation.hpp:172-231)
// pre: the candidate's taxon if the caller has it already (mid_cands_kernel fetches them in parallel), else ~0u = look it up here
__device__ __forceinline__ void top_insert(LaneCand (&top)[kLaneK], uint32_t (&toptax)[kLaneK], LaneCand c, const uint32_t K,
                                           const uint32_t* __restrict__ taxkey, const uint32_t tgtMask, const uint32_t pre = ~0u)
{
    uint32_t ctax = 0;
    bool moving = false;
    if (taxkey) {
        // candidate_generation.hpp:178-216: full list and not better than its last entry -> ignored;
        // no taxon -> skipped; taxon already listed -> replaced only by more hits, then moved up behind
        // the entries with >= hits (std::sort on <= 16 elements = insertion sort)
        uint32_t lastHits = 0;
#pragma unroll
        for (uint32_t i = 0; i < kLaneK; ++i) if (i + 1 == K) lastHits = top[i].hits;
        if (lastHits > 0 && lastHits >= c.hits) return;
        ctax = pre != ~0u ? pre : taxkey[c.tgt & tgtMask];
        if (ctax == 0) return;
        bool found = false;
#pragma unroll
        for (uint32_t i = 0; i < kLaneK; ++i) {
            if (i < K && !found && top[i].hits > 0 && toptax[i] == ctax) {
                found = true;
                if (c.hits > top[i].hits) {
                    top[i] = c;
#pragma unroll
                    for (uint32_t j = kLaneK - 1; j > 0; --j) {       // bubble up while strictly more hits
                        if (j <= i && top[j].hits > top[j - 1].hits) {
                            const LaneCand t = top[j]; top[j] = top[j - 1]; top[j - 1] = t;
                            const uint32_t tt = toptax[j]; toptax[j] = toptax[j - 1]; toptax[j - 1] = tt;
                        }
                    }
                }
            }
        }
        if (found) return;
    }
    // behind every entry with >= hits (ties keep arrival order); displaced entries shift down in order
#pragma unroll
    for (uint32_t i = 0; i < kLaneK; ++i) {
        if (i < K && (moving || c.hits > top[i].hits)) {
            const LaneCand t = top[i]; top[i] = c; c = t;
            const uint32_t tt = toptax[i]; toptax[i] = ctax; ctax = tt;
            moving = true;
        }
    }
}

// big_filter_kernel / gw_filter_kernel -> counting kernels: longest filtered list the counting kernels take
constexpr uint32_t kBigMaxFilteredCount = 1024;
constexpr uint32_t kGwSmallH = 2048;      // compact store: reads with more locations take gw_filter_kernel's instance with the larger filters

constexpr uint32_t kGwMaxKept = kMaxHitsPerQuery;   // longest filtered list handed on (gw_filter kernels)
// a filtered list (n2 numbers, window range maxWin) that is sorted instead of counted
__host__ __device__ inline bool gw_sorted_class(uint32_t n2, uint32_t maxWin) { return n2 <= kGwMaxKept && (n2 > kBigMaxFilteredCount || maxWin > kHashWin); }
// (stage 4 of launch_gw_cands: candidates of the sorted lists; needs ws.bigPool2 filled by launch_gw_segsort, kernels.h)
// gw_kernels.hip: the filtered path of tables with the compact location store (stages as launch_big_cands)
void launch_gw_cands(uint32_t stage, const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                     const uint32_t* taxkey, void* cands, hipStream_t st);

}  // namespace mcamd
