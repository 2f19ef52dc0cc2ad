// metacache_amd/csrc/kernels.h -- internal interface between the host pipeline (context.cpp) and
// the gfx950 kernels (kernels.hip).  Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace mcamd {

constexpr uint32_t kWave = 64;
constexpr uint32_t kMaxSketch = 32;       // MC_MAX_SKETCH
constexpr uint32_t kMaxWinLen = 1024;     // MC_MAX_WINLEN
constexpr uint32_t kNoTail = 0xFFFFFFFFu; // qinfo[3] marker: single sequence, tail window suppressed
constexpr uint32_t kMaxHitsPerQuery = (1u << 20) - 1;  // packed candidate fields are 20 bits wide
constexpr uint32_t kGwGap = 1024;         // compact location store: unused window numbers between two targets (DeviceTable); window ranges up to this width
                                          // (reads up to 114 kbp at the default stride) stay on the kernels that work on the numbers as they are

// Table layout: an array of 64-byte BUCKETS of 4 slots, two per 128-byte line, structure-of-arrays so
// that a single LANE looks a feature up with four 16-byte loads of ONE half line and has key, size and
// payload of the match in registers -- no second, dependent access (a separate payload load re-fetched
// the line from L2 because thousands of lines are in flight per CU and the L1 holds 256):
//   key[4]     the features
//   size[4]    bucket sizes (u16), 0 = free slot
//   payload[4] size == 1: the location itself ((tgt << 32) | win);
//              size  > 1: index of the first location in DeviceTable::values
// Probe sequence of a key: its home bucket, the sibling bucket in the same line, then the following
// buckets linearly.  Insertion uses the first bucket of that sequence with a free slot, so a lookup
// ends at the first bucket that holds the key or has a free slot.
struct __attribute__((aligned(64))) TableBucket {
    uint32_t key[4];
    uint16_t size[4];
    uint32_t spare[2];
    uint64_t payload[4];
};
static_assert(sizeof(TableBucket) == 64, "two buckets per 128-byte line");
constexpr uint32_t kSlotsPerBucket = 4;

__host__ __device__ inline uint32_t next_bucket(uint32_t home, uint32_t cur, uint32_t step, uint32_t nbuckets)
{
    // step = number of buckets already visited (>= 1)
    if (step == 1) return home ^ 1u;                          // sibling half of the same line
    const uint32_t nx = (step == 2 ? (home | 1u) : cur) + 1u;
    return nx >= nbuckets ? 0u : nx;
}

// Mode K: which key shard owns a feature (a different mix than the bucket hash, so a shard's keys spread over its whole table)
__host__ __device__ inline uint32_t key_owner(uint32_t key, uint32_t shards);

// table hash: features are the SMALLEST hash values of a window, i.e. far from uniform in their
// high bits, so they are mixed again (murmur3 fmix32) before the multiply-shift range reduction
// bucket = (mix32(key) * nbuckets) >> 32.
__host__ __device__ inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

__host__ __device__ inline uint32_t key_owner(uint32_t key, uint32_t shards)
{
    return shards <= 1 ? 0u : (uint32_t)(((uint64_t)mix32(key ^ 0x9E3779B9u) * shards) >> 32);
}

struct DeviceTable {
    const TableBucket* buckets; // [nbuckets], nbuckets even
    const uint64_t*  values;  // location lists, (tgt << 32) | win, each bucket sorted ascending
    uint32_t nbuckets;
    uint32_t tgtMask;         // multi-part tables store (part << 24 | target) as target: mask to get the real id
    uint32_t maxProbe;        // longest probe sequence (in groups) needed by any stored key
    // COMPACT location store (mc_load_target_windows): a location is stored as ONE 32-bit GLOBAL WINDOW NUMBER
    //     gw = gwBase[tgt] + win,   gwBase[0] = gwGap,   gwBase[t + 1] = gwBase[t] + windows(t) + gwGap
    // -- same order as (tgt, win), half the bytes per list on the fabric and in HBM, and it fits whatever the targets look like as long
    // as all windows of the database (plus a gap per target) stay below 2^32 (481 Gbp at the default window stride).  The GAP between
    // two targets' numbers is what makes gw arithmetic safe without knowing the target: two locations lie in one window range
    // (candidate_generation.hpp:47-108) iff their numbers differ by less than maxWindowsInRange <= gwGap, and gw - d (d < gwGap) is
    // never another target's window.  values32 != nullptr selects the store (values is unused then); the inline singleton payloads of
    // the buckets keep the 8-byte form.  0xFFFFFFFF is never a stored number.
    const uint32_t* values32 = nullptr;
    // DIRECT-ADDRESS INDEX (round 6): 2^32 entries of 8 bytes, the feature IS the index -- size (16 bits) | payload (48 bits:
    // a list's first place in the store, or a single location as target (24) | window (24)); 0 = the feature is not in the table.
    // One 8-byte load per lookup by ONE lane, no key compare, no chain, no cooperation between lanes (the box serves 47 x 10^9 requests a
    // second whatever their width: tools/gather_width.hip; against the quad-cooperative bucket fetch -- also one request per lookup -- it
    // saves 40 % of the lookup kernel's instructions).  Beside the buckets, which every other kernel keeps using.
    const uint64_t* direct = nullptr;
    const uint32_t* gwBase = nullptr;     // [targets + 1]
    const uint32_t* gwDir = nullptr;      // [(gwBase[targets] >> gwDirShift) + 1]: the target whose numbers (gap included) hold block << gwDirShift
    uint32_t gwDirShift = 0, gwGap = 0, gwTargets = 0;

    // target of a global window number (two dependent loads that hit the L2 / infinity cache: the directory has one entry per 2^gwDirShift windows)
    __device__ __forceinline__ uint32_t gw_target(uint32_t gw) const
    {
        uint32_t t = gwDir[gw >> gwDirShift];
        while (gw >= gwBase[t + 1]) ++t;
        return t;
    }
    // target and its numbers [lo, hi) in TWO dependent loads: the directory entry, then three neighbouring gwBase words at once (a block
    // of 2^gwDirShift numbers rarely holds more than one target boundary; more move on one by one)
    __device__ __forceinline__ void gw_target_bounds(uint32_t gw, uint32_t& t, uint32_t& lo, uint32_t& hi) const
    {
        t = gwDir[gw >> gwDirShift];
        const uint32_t b0 = gwBase[t], b1 = gwBase[t + 1], b2 = gwBase[min(t + 2, gwTargets)];
        lo = b0; hi = b1;
        if (gw >= b1) { ++t; lo = b1; hi = b2; }
        while (gw >= hi) { ++t; lo = hi; hi = gwBase[t + 1]; }
    }
    __device__ __forceinline__ uint64_t gw_widen(uint32_t gw) const { const uint32_t t = gw_target(gw); return ((uint64_t)t << 32) | (gw - gwBase[t]); }
    __device__ __forceinline__ uint32_t gw_of(uint64_t loc) const { return gwBase[(uint32_t)(loc >> 32)] + (uint32_t)loc; }
    __device__ __forceinline__ uint64_t loc(uint64_t i) const { return values32 ? gw_widen(values32[i]) : values[i]; }
};
__host__ __device__ inline uint32_t bits_for(uint32_t maxValue) { uint32_t b = 1; while (b < 32 && (maxValue >> b)) ++b; return b; }

struct SketchParams { uint32_t k, s, w, stride; };

// per-query result of the sketch_probe kernel
struct __attribute__((aligned(16))) QueryStat {
    uint32_t hits;      // sum of bucket sizes found (H)
    uint32_t nfeat;     // valid features probed (F)
    uint32_t nfound;    // features present in the table
    uint32_t nsteps;    // bucket groups read
};

// row 1: number of windows with >= k characters (hash_dna.hpp:54-75 + :222)
__host__ __device__ inline uint32_t windows_of(uint32_t L, const SketchParams& sp, bool noTail)
{
    if (L <= sp.w) return L >= sp.k ? 1u : 0u;
    uint32_t nf = (L - sp.w) / sp.stride + 1;
    uint64_t first = (uint64_t)nf * sp.stride;      // may exceed L for stride > w
    if (!noTail && first < L && L - first >= sp.k) return nf + 1;
    return nf;
}
inline uint32_t windows_of_host(uint32_t L, const SketchParams& sp) { return windows_of(L, sp, false); }

struct BatchView {           // device pointers describing one batch
    const uint8_t*  seq;
    const uint32_t* qinfo;   // [n][4]
    const uint32_t* maxWin;  // [n] or nullptr
    uint32_t maxWinUniform;
    uint32_t n;
};

struct Workspace {           // device buffers sized by the host for this batch
    uint32_t* winCount;      // [n]
    uint32_t* winOff;        // [n+1]      exclusive scan of winCount
    uint32_t* features;      // [W*s]      window sketches (0xFFFFFFFF padded)
    uint32_t* psize;         // [W*s]      bucket size per feature (0 = not found / no feature)
    uint64_t* ppay;          // [W*s]      payload per feature
    QueryStat* qstat;        // [n]
    uint32_t* qflag;         // [n]        0 = done, 1 = needs sketch+probe (wave), 2 = needs candidates (wave),
                             //            4 = sketched by a lane (probe_cands_kernel takes it from there), 5 = on a work list of mid_cands_kernel,
                             //            6 = long read handled by the chunk lane kernels
    uint32_t* midCount;      // [32]; [8] = third work list of hash_cands_kernel (129..256);       lengths of the three work lists of mid_cands_kernel, [3], [4] = of hash_cands_kernel, [5] = chunk records, [6], [7] = queries left for the wave kernels (launch_flag_count) (zeroed per batch)
    uint2*    chunkList;     // [W + n]    {query, chunk}: long single reads, cut into one-window chunks for the chunk lane kernels
    uint32_t  partialLists;  // 1: every lane-path query hands its entry table over and ends there (MC_WANT_PARTIAL_HITS: gather_lists_kernel copies the lists)
    uint32_t  bigMin;        // lists longer than this (and > 256) from <= 64 found features go to big_filter_kernel (midCount[9], list 6), which
                             // hands their filtered parts (bigPool, cursor midCount[11]) to big_count_kernel (midCount[10] / [12], lists 7 / 8)
    uint32_t* sliceFill;     // [waves of big_filter_kernel] entries each wave's pool slice holds after the first instance (nullptr: single instance)
    uint64_t* bigPool;       // [bigPoolCap] filtered locations of a batch
    uint32_t* sideList;      // [5][n] ([4]: the sorted class' lists gw_count_block_kernel takes, midCount[19]) compact store: record numbers (list 6 / 7) of the reads gw_filter_stream_kernel takes ([0], length midCount[12]), of the
                             // filtered lists of 257 .. 512 ([1], midCount[14]) and 513 .. 1024 numbers ([2], midCount[15]) and of the sorted ones ([3], midCount[13])
    uint32_t* bigPool2;      // [bigPoolCap] compact store: the filtered lists that are sorted (gw_sort.hip), at their pool offsets
    uint32_t* orderScratch;  // compact store, batches up to 2^20 reads: scratch of launch_gw_order for the stream filter's list (3 n words + orderTemp bytes); nullptr: no ordering
    size_t    orderTemp;
    uint32_t  bigPoolCap;
    uint32_t  bigOvfCap;     // compact store: entries behind bigPoolCap for filtered lists that may not fit their wave's slice (cursor: midCount[16..17] as u64)
    uint32_t* midList;       // [8][n] x uint4 {query, first entry slot, entries | locations << 8, maxWindowsInRange}: lists of 33..64 / 65..128 / 129..256
    uint32_t* hitScan;       // [n]        hits that need a segment in 'hits' (all, or only lists too long for LDS)
    uint64_t* hitOff;        // [n+1]      exclusive scan of hitScan
    uint64_t* hits;          // [H]        gathered + sorted locations
    uint64_t* cscr;          // [H]        candidate scratch (large queries)
    uint64_t* cscr2;         // [H]        second scratch (taxon merging, large queries)
    void*     scanTmp;       // block sums for the scans
    uint64_t* stats;         // [8]        batch statistics (on demand)
    // host side only: the context's grid tuning switches (mc_set_tuning; 0 = default) -- per context, never process-wide
    uint32_t  gwMidH = 0;         // reads up to this many locations take the stream filter's small-filter instance (2^16 + 2^13 bits: six waves per SIMD; 0: none)
    uint32_t  gwBigH = 32768;     // reads beyond this many locations take the stream filter's fine-block instance (gw_kernels.hip kGwBigH; 0xFFFFFFFF: none)
    int32_t   filterBpc = 0, countBpc = 0, gwFuse = 1;   // gwFuse: gw_filter_count_kernel (1) or gw_filter_kernel + gw_count_kernel (0)
};

// launchers (all asynchronous on 'st')
// Results of a batch of the host slots to the slots' own PINNED host buffers by ONE kernel (uint4 stores over the host link): every
// hipMemcpyAsync D2H costs the stream ~15 us whatever its size, and a united batch has two per slot.  entry i: queries [first, first + count)
// of the batch -> its candidates (count x K x 16 bytes) and statistics (count x 16 bytes).
struct DeliverEntry { void* cands; void* qstat; uint32_t first, count; };
constexpr uint32_t kDeliverMax = 16;
struct DeliverTable { DeliverEntry e[kDeliverMax]; uint32_t n; };
void launch_deliver(const DeliverTable& t, const void* cands, const void* qstat, uint32_t K, hipStream_t st);
void launch_flag_count_host(const Workspace& ws, uint32_t n, uint32_t* hostCounts, hipStream_t st);   // launch_flag_count + the sixteen counters to pinned host memory
void launch_words_to_host(uint32_t* hostDst, const uint32_t* src, uint32_t nwords, hipStream_t st);   // device words -> pinned host memory by a kernel on `st` (no copy engine)
uint32_t lane_max_len();   // longest single read one lane sketches (longer ones are cut into chunk lanes' records)
void launch_plan(const BatchView& b, const SketchParams& sp, uint32_t* winCount, hipStream_t st);
// small batches (up to 32 768 reads): plan + scan of the windows + the lane path's 32 work-list counters cleared (zero32, may be null) in ONE launch; false: too large, nothing launched
bool launch_plan_scan_small(const BatchView& b, const SketchParams& sp, uint32_t* winCount, uint32_t* winOff, uint32_t* zero32, hipStream_t st);
void launch_scan_u32(const uint32_t* in, uint32_t stride, uint32_t n, uint32_t* out32, uint64_t* out64,
                     void* tmp, hipStream_t st, uint64_t* hostTotal = nullptr);   // hostTotal: the grand total also to pinned host memory, by the kernels themselves
size_t scan_tmp_bytes(uint32_t n);
void launch_sketch_only(const BatchView& b, const SketchParams& sp, const Workspace& ws, hipStream_t st);
// database builder: window sketches of window-aligned chunk records (<= build_record_windows() windows each); lanes where the
// sketching parameters allow, the exact wave path for the rest.  Needs ws.winOff, ws.features, ws.qflag.
void launch_build_sketch(const BatchView& b, const SketchParams& sp, const Workspace& ws, hipStream_t st);
uint32_t build_record_windows();
void launch_query(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, bool fuse, bool wantAllhits,
                  const Workspace& ws, uint32_t maxCand, void* cands, hipStream_t st);
void launch_sketch_lane(const BatchView& b, const SketchParams& sp, const Workspace& ws, hipStream_t st);
// quadMode: -1 = by table size, 0 = lane-private bucket loads, 1 = quad-cooperative bucket loads
void launch_probe_cands(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                        const uint32_t* taxkey, void* cands, int quadMode, hipStream_t st);
// table_build.hip: GPU-side table construction from the file's batch stream
struct LoadFilter { uint32_t maxLocs, rmOver, shardIdx, shardCnt, align = 1; };   // load-time modifiers + key shard + list alignment (below)
// LIST ALIGNMENT (round 5, compact store): a list of the location store may begin at a multiple of kListAlign numbers = 128 bytes.  The
// memory system serves random reads line by line (47 x 10^9 requests of 128 bytes per second, whatever part of the line is wanted:
// profiles/r05_fetch_calibration.md): a list of 196 bytes at an arbitrary 4-byte offset touches 2.6 lines, an aligned one 2.0 -- the
// filter kernels of read pairs and long reads run at that request rate.  Space a list takes in the store: list_alloc(size, align).
constexpr uint32_t kListAlign = 32;
__host__ __device__ inline uint32_t list_alloc(uint32_t size, uint32_t align) { return size > 1 ? (size + align - 1) / align * align : 0; }
struct GwLayout { const uint32_t* base = nullptr; uint32_t targets = 0, gap = 0; };   // compact store: gwBase[targets + 1] (DeviceTable)
// the direct-address index of a finished bucket table (DeviceTable::direct): every stored key's entry; *flag != 0: a payload does not fit 48 bits
void launch_direct_index(const TableBucket* buckets, uint32_t nbuckets, uint64_t* direct, unsigned int* flag, hipStream_t st);
constexpr uint64_t kDirectEntries = 1ull << 32;
__host__ __device__ inline uint64_t direct_pack(uint32_t size, uint64_t pay) { return size == 1 ? (uint64_t)size | ((((pay >> 32) << 24) | (pay & 0xFFFFFFu)) << 16) : (uint64_t)size | (pay << 16); }
__host__ __device__ inline uint64_t direct_payload(uint64_t e) { const uint64_t p = e >> 16; return (e & 0xFFFFu) == 1 ? ((p >> 24) << 32) | (p & 0xFFFFFFu) : p; }
void launch_table_prep(const uint32_t* keys, const uint8_t* sizes, uint32_t n, LoadFilter lf, uint32_t* fileSz, uint32_t* storeSz,
                       unsigned long long* counters, hipStream_t st);
void launch_table_insert(const uint32_t* keys, const uint8_t* sizes, uint32_t n, LoadFilter lf, const uint32_t* fileOff,
                         const uint32_t* storeOff, const uint8_t* vals, uint32_t tb, uint64_t storeBase, TableBucket* buckets,
                         uint32_t nbuckets, unsigned int* maxProbe, unsigned int* full, hipStream_t st,
                         GwLayout gw = GwLayout{}, unsigned int* rangeErr = nullptr);   // gw.base: inline single locations are range-checked too
void launch_table_values(const uint32_t* keys, const uint8_t* sizes, uint32_t n, LoadFilter lf, const uint32_t* fileOff, const uint32_t* storeOff,
                         const uint8_t* vals, uint32_t tb, uint64_t totalFileVals, uint64_t* dst, hipStream_t st);
// compact store: dst32[...] = gwBase[tgt] + win; a location outside its target's windows (or of an unknown target) raises *rangeErr instead
void launch_table_values_compact(const uint32_t* keys, const uint8_t* sizes, uint32_t n, LoadFilter lf, const uint32_t* fileOff, const uint32_t* storeOff,
                                 const uint8_t* vals, uint32_t tb, uint64_t totalFileVals, uint32_t* dst32, GwLayout gw, unsigned int* rangeErr, hipStream_t st);
void launch_chunk_lanes(int stage, const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, int quadMode, hipStream_t st);
void launch_sketch_probe_lane(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                              const uint32_t* taxkey, void* cands, int quadMode, hipStream_t st);
// up to 16 candidate lists per read ([n][K] each, device), merged in list order through the CPU's top-list insert; K <= 4.  -1: not supported
int launch_merge_parts(const void* const* lists, uint32_t nlists, uint32_t n, uint32_t K, const uint32_t* taxkey, void* out, hipStream_t st);
void launch_flag_count(const Workspace& ws, uint32_t n, hipStream_t st);
// compact store: reads sketched and probed by the wave kernel join the filtered path's work list (list 6) where it can take them
void launch_wave_rejoin(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, hipStream_t st);
void launch_hash_cands(uint32_t cls, const BatchView& b, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand, const uint32_t* taxkey, void* cands, hipStream_t st);
void launch_big_cands(uint32_t stage, const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                      const uint32_t* taxkey, void* cands, hipStream_t st);
void launch_gather_lists(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t* numbers, hipStream_t st);
// gw_sort.hip: the filtered lists that are sorted instead of counted (the first nseg records of ws.sideList[3]), pool -> out at the same
// offsets; temp == nullptr: size query
struct GwSortSide { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };   // a second stream for the sort's independent instances (the caller's: a Pipe's)
int launch_gw_segsort(void* temp, size_t& tempBytes, const uint32_t* in, uint32_t* out, uint64_t poolCap, const Workspace& ws, uint32_t n, uint32_t nseg,
                      uint32_t endBit, hipStream_t st, const GwSortSide* side2 = nullptr);
// ws.sideList[list] (list 0: the stream filter's reads, 3: the sorted class) in descending order of the records' work; scratch == nullptr: size query
int launch_gw_order(uint32_t list, const Workspace& ws, uint32_t n, uint32_t count, uint32_t* scratch, size_t& tempBytes, hipStream_t st);
uint32_t big_filter_grid(uint32_t n, bool compact, int bpcOverride = 0);     // (bpcOverride: mc_set_tuning "filter_bpc" of the context) blocks of 4 waves the filter kernels run with (compact: the gw kernels): the pool is cut into one slice per wave
void launch_mid_cands(uint32_t cls, const BatchView& b, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                      const uint32_t* taxkey, void* cands, hipStream_t st);
void launch_cands_from_hits(const BatchView& b, const DeviceTable& tab, const Workspace& ws, const uint32_t* taxkey, uint32_t maxCand,
                            void* cands, hipStream_t st);
// Mode K owner side: per-source partial lists -> one list per read (hitOff[m + 1], out); tot[m], srcStart[sources * (m + 1)] scratch
void launch_union_partial(const uint32_t* counts, uint32_t sources, uint32_t m, const uint64_t* hits, uint32_t* tot, uint64_t* srcStart, uint64_t* hitOff,
                          uint64_t* out, void* scanTmp, hipStream_t st);
void launch_owner_classify(const BatchView& b, const Workspace& ws, uint32_t minLen, hipStream_t st);
// keyshard.hip: Mode K with 4-byte locations (global window numbers) on the wire
struct KeyshardBases { uint64_t b[64]; };   // where each source's block begins in the receive buffer (at most 64 key shards)
void launch_mask_foreign_features(uint32_t* features, const uint32_t* totalWindows, uint32_t s, uint64_t maxFeat, uint32_t shardIdx, uint32_t shardCnt, hipStream_t st);
void launch_pack_numbers(const uint64_t* hits, const uint64_t* hitOff, uint64_t total, uint32_t n, const DeviceTable& tab, uint32_t* numbers, uint32_t* counts,
                         hipStream_t st);
// the lists of the reads that are NOT waiting for gather_lists_kernel (the wave kernels left them in ws.hits) as numbers, + all reads' counts
void launch_pack_other_reads(const BatchView& b, const DeviceTable& tab, const Workspace& ws, uint32_t* numbers, uint32_t* counts, hipStream_t st);
void launch_owner_entries(const BatchView& b, const DeviceTable& tab, const Workspace& ws, const uint32_t* counts, const uint64_t* srcStart,
                          const KeyshardBases& bases, uint32_t S, hipStream_t st);
void launch_decode_union(const BatchView& b, const DeviceTable& tab, const Workspace& ws, const uint32_t* counts, const uint64_t* srcStart,
                         const KeyshardBases& bases, uint32_t S, hipStream_t st);
bool lane_path_supported(const SketchParams& sp);
bool lane_candidates_supported(uint32_t maxCand);
constexpr uint32_t kLdsCap = 256;         // location lists up to this length are sorted in LDS
void launch_sort_candidates(const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws,
                            const uint32_t* taxkey, uint32_t maxCand, bool wantAllhits,
                            void* cands, hipStream_t st);
void launch_batch_stats(const Workspace& ws, uint32_t n, hipStream_t st);

}  // namespace mcamd
