// metacache_amd/csrc/gw_kernels.hip -- the filtered candidate path of tables with the COMPACT location store:
// a location is one 32-bit GLOBAL WINDOW NUMBER gw = gwBase[target] + window (kernels.h DeviceTable), with gwGap unused numbers
// between two targets.  That layout is what these kernels are built on:
//   * two locations can lie in one window range (candidate_generation.hpp:47-108: same target, windows less than maxWindowsInRange
//     apart) iff their numbers differ by less than maxWindowsInRange -- no target needed;
//   * gw - d is "the same target, d windows earlier" or no location at all;
//   * the order of the numbers is the order of (target, window) (database.hpp:151-156), so "hits descending, then the smaller
//     number" is the insertion order of the reference's top list (candidate_generation.hpp:172-201).
// One WAVE per read throughout, persistent grids over the work list probe_cands_kernel / chunk_finish_kernel leave (list 6).
//
//   gw_filter_kernel   keeps the locations that have a neighbour (another location of the read less than maxWindowsInRange away):
//                      only those can be part of a window range with two or more hits.  RefSeq scale: 1 271 locations per 150 bp read,
//                      about 250 kept.
//   gw_count_kernel    exact hits per window range of the kept locations in an LDS hash table keyed by the number itself, K rounds
//                      of a wave-wide maximum; the places candidates with ONE hit may take (step D) from a sweep over all lists.
// Both are bound by VALU issue, not by memory (round 2's kernels: 7.5 x 10^9 and 3.1 x 10^9 wave instructions per 5 x 10^6 reads are
// 14 and 6 ms of the SIMDs' time): everything below is written to need few instructions per location -- 16-byte loads of four
// locations per lane, the read's whole list held in registers between the filter's two phases (no second pass over the fabric),
// 24-bit multiplies (full rate; v_mul_lo_u32 is a quarter-rate instruction), one LDS operation per location and phase.
#include "device_common.h"

#include <algorithm>
#include <cstdlib>

namespace mcamd {

namespace {

constexpr uint32_t kGwNone = 0xFFFFFFFFu;                 // never a stored number
constexpr uint32_t kGwRounds = 128;                       // rounds (16 consecutive numbers of one bucket list = 64 bytes) per batch: 8 wave loads
constexpr uint32_t kGwLoads = kGwRounds / 16;             // 16-byte loads per lane and batch: 32 registers hold 2 048 list places

struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };   // 16-byte load from a 4-byte aligned address (global_load_dwordx4)

// ---- the filter's key: the location's BLOCK of 2^A window numbers.  Two locations less than maxWindowsInRange apart share a block
// unless a block boundary lies between them -- and then both are within D = maxWindowsInRange - 1 of that boundary: such locations
// are kept without asking.  With 2^A >= 64 D that is 3 % of the single hits at most; blocks this small hardly ever hold two unrelated
// locations of one read (1 271 numbers in 1.3 x 10^9), which a filter keyed on targets does 40 times per read at RefSeq scale.
__device__ __forceinline__ uint32_t gw_block_shift(uint32_t maxWin)
{
    const uint32_t D = maxWin > 1 ? maxWin - 1 : 1;
    const uint32_t A = 32u - (uint32_t)__builtin_clz(64u * D - 1u);     // 2^A >= 64 D
    return A < 8u ? 8u : A;                                              // (keys below 2^24: 24-bit multiply)
}

// blocked Bloom filters in LDS, as round 2's: "seen" = 2 bits of ONE word per key, set with a returning ds_or; both found set = the
// key was seen before (or two other keys set them) -> the same bits in "twice".  T1 == T2: "twice" is the word at a fixed distance
// (no second address computation).
template <uint32_t T1LOG2, uint32_t T2LOG2>
struct GwBloom {
    static constexpr uint32_t kW1 = (1u << T1LOG2) / 32, kW2 = (1u << T2LOG2) / 32, kWords = kW1 + kW2;
    static constexpr bool kSame = T1LOG2 == T2LOG2;
    static_assert(kWords % 256 == 0, "cleared with one uint4 per lane and step");
    // 24-bit multiply (keys are below 2^24): full rate, where v_mul_lo_u32 takes four issue slots.  (Written as an instruction: the
    // compiler turns __umul24 of a value it knows to be short back into a 32-bit multiply.)
    __device__ __forceinline__ static uint32_t hash(uint32_t key)
    {
        uint32_t h;
        asm("v_mul_u32_u24_e32 %0, 0x9e3779, %1" : "=v"(h) : "v"(key));
        return h;
    }
    // two bits of one word: positions h[4:0] (the shifter takes the low five bits by itself) and h[16:12]
    __device__ __forceinline__ static uint32_t mask_of(uint32_t h)
    {
        uint32_t m1, m;
        asm("v_lshlrev_b32_e64 %0, %1, 1" : "=v"(m1) : "v"(h));
        asm("v_lshl_or_b32 %0, 1, %1, %2" : "=v"(m) : "v"(h >> 12), "v"(m1));
        return m;
    }
    // "twice" holds ONE bit per key, the first of the two (round 5): a key enters it only when it is seen again -- a hundred keys of a
    // read's 1 100 in 2^14 bits: one bit has a false-positive rate of 0.6 % there, and the test is two instructions shorter
    // (only the register-batch kernels' filters, kSame: a long read's stream filter marks thousands of keys twice -- there both bits stay)
    __device__ __forceinline__ static uint32_t mask1_of(uint32_t h)
    {
        if constexpr (!kSame) return mask_of(h);
        uint32_t m1;
        asm("v_lshlrev_b32_e64 %0, %1, 1" : "=v"(m1) : "v"(h));
        return m1;
    }
    __device__ __forceinline__ static bool twice_hit(uint32_t word, uint32_t m) { if constexpr (kSame) return (word & m) != 0u; else return (word & m) == m; }
    // the same test from the hash itself: bit h[4:0] of the word (v_bfe_u32 takes the low five bits of its offset operand by itself) --
    // no mask to build
    __device__ __forceinline__ static bool twice_hit_h(uint32_t word, uint32_t h)
    {
        if constexpr (kSame) { uint32_t t; asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(t) : "v"(word), "v"(h)); return t != 0u; }
        else { const uint32_t m = mask_of(h); return (word & m) == m; }
    }
    __device__ __forceinline__ static uint32_t seen_index(uint32_t h)                                            // the word: the product's top bits
    {
        uint32_t i;                                            // (as an instruction: the compiler would fold the shift into shift + and + add)
        asm("v_lshrrev_b32_e32 %0, %1, %2" : "=v"(i) : "n"(37u - T1LOG2), "v"(h));
        return i;
    }
    __device__ __forceinline__ static uint32_t twice_index(uint32_t seen, uint32_t h)
    {
        if constexpr (kSame) return seen + kW1;
        else return kW1 + ((h >> 17) & (kW2 - 1u));
    }
    // phase A for four keys: the four returning ds_or go out together (one LDS round trip instead of four); "twice" is written by the
    // lanes that found their bits set only (a fifth of them: few lanes, few bank conflicts)
    __device__ __forceinline__ static void mark4(uint32_t* bits, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3)
    {
        const uint32_t k[4] = {k0, k1, k2, k3};
        uint32_t m[4], i1[4], i2[4], old[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t h = hash(k[j]); m[j] = mask_of(h); i1[j] = seen_index(h); i2[j] = twice_index(i1[j], h); }
#pragma unroll
        for (int j = 0; j < 4; ++j) old[j] = atomicOr(&bits[i1[j]], m[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) if ((old[j] & m[j]) == m[j]) atomicOr(&bits[i2[j]], m[j]);   // (both bits: the one bit asked for is among them)
    }
    __device__ __forceinline__ static void mark(uint32_t* bits, uint32_t key)
    {
        const uint32_t h = hash(key), m = mask_of(h), i1 = seen_index(h);
        const uint32_t old = atomicOr(&bits[i1], m);
        if ((old & m) == m) atomicOr(&bits[twice_index(i1, h)], m);
    }
    __device__ __forceinline__ static bool twice(const uint32_t* bits, uint32_t key)
    {
        const uint32_t h = hash(key), m = mask1_of(h);
        return twice_hit(bits[twice_index(seen_index(h), h)], m);
    }
    __device__ __forceinline__ static bool seen(const uint32_t* bits, uint32_t key)     // the key was marked at all (or its bits by others)
    {
        const uint32_t h = hash(key), m = mask_of(h);
        return (bits[seen_index(h)] & m) == m;
    }
};

// What both filter kernels share: the per-read frame (block size of the keys, edge test, the pool slice) and the two phases on a
// batch of kGwRounds rounds held in registers.
struct GwFrame {
    uint32_t A, D, blockMask, inner;                           // keys = number >> A; (number & blockMask) - D >= inner: within D of a block boundary
    uint32_t shv, Dsh, twoDsh;                                 // the same test in two instructions: ((number + D) << (32 - A)) < (2 D << (32 - A))
    // fine = true: blocks of 2^A >= D numbers only (gw_filter_stream_kernel<.., FINE>: the neighbour blocks are asked as well, no edge rule)
    __device__ __forceinline__ explicit GwFrame(uint32_t maxWin, bool fine = false)
    {
        A = gw_block_shift(maxWin); D = maxWin > 1 ? maxWin - 1 : 0u;
        if (fine) { A = 32u - (uint32_t)__builtin_clz((D > 1u ? D : 2u) - 1u); A = A < 8u ? 8u : A; }   // 2^A >= D; keys below 2^24
        blockMask = (1u << A) - 1u; inner = (1u << A) - 2u * D;
        const uint32_t sh = 32u - A;                           // (A <= 16: 2 D < 2^A, nothing is shifted out of 2 D)
        asm volatile("v_mov_b32 %0, %1" : "=v"(shv) : "s"(sh));   // (kept in a vector register: a VOP3 instruction reads one scalar register at most)
        Dsh = D << sh; twoDsh = (2u * D) << sh;
    }
    // (number + D) mod 2^A < 2 D  <=>  the number lies within D of a block boundary; the shift brings the low A bits to the top
    __device__ __forceinline__ bool edge(uint32_t v) const
    {
        uint32_t t;
        asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(t) : "v"(v), "v"(shv), "s"(Dsh));
        return t < twoDsh;
    }
};

// round table of the batch [r0, r0 + kGwRounds) of an entry chunk's rounds: first index | numbers << 40.  start / myR: the lane's entry
// begins at round 'start' of the chunk and has myR rounds.
__device__ __forceinline__ void gw_fill_rounds(uint64_t* T, uint32_t lane, uint32_t r0, uint32_t Rc, uint32_t start, uint32_t myR, uint32_t sz, uint64_t pay)
{
    for (uint32_t i = lane; i < kGwRounds; i += 64) if (r0 + i >= Rc) T[i] = 0ull;
    const uint32_t jlo = r0 > start ? r0 - start : 0u, jhi = min(myR, r0 + kGwRounds > start ? r0 + kGwRounds - start : 0u);
    for (uint32_t j = jlo; j < jhi; ++j) T[start + j - r0] = (pay + 16ull * j) | ((uint64_t)min(16u, sz - 16u * j) << 40);
}
// The same table for a read whose rounds fit ONE batch (r0 = 0, Rc <= kGwRounds), in a fixed number of instructions: gw_fill_rounds has
// every lane write its own rounds one after the other -- as many steps as the longest list has rounds (16 for the lists of 254
// locations every read of a RefSeq-scale table meets): ~130 of the kernel's ~1 380 VALU instructions per read.  Here every lane leaves
// its number at its FIRST round's slot, a max-scan over the slots (lane l: slots l and l + 64) tells every slot whose round it is, and
// the entry's fields come over from that lane (ds_bpermute): ~40 instructions whatever the lists' lengths.  E: 128 words of LDS scratch.
__device__ __forceinline__ void gw_fill_rounds_scan(uint64_t* T, uint32_t* E, uint32_t lane, uint32_t Rc, uint32_t start, uint32_t myR, uint32_t sz, uint64_t pay)
{
    E[lane] = 0u; E[lane + 64u] = 0u;
    wave_lds_sync();
    if (myR && start < kGwRounds) E[start] = lane + 1u;            // (entries' first rounds are distinct slots)
    wave_lds_sync();
    uint32_t lo = wave_incl_scan_max_u32(E[lane]);
    uint32_t hi = max(wave_incl_scan_max_u32(E[lane + 64u]), rdlane(lo, 63));
    const uint32_t payLo = (uint32_t)pay, payHi = (uint32_t)(pay >> 32);
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t slot = lane + 64u * h, who = h ? hi : lo;
        const int src = (int)(who ? who - 1u : 0u);
        const uint32_t eStart = (uint32_t)__shfl((int)start, src), eSz = (uint32_t)__shfl((int)sz, src);
        const uint32_t eLo = (uint32_t)__shfl((int)payLo, src), eHi = (uint32_t)__shfl((int)payHi, src);
        const uint32_t j = slot - eStart;
        const uint64_t ep = ((uint64_t)eHi << 32) | eLo;
        T[slot] = (who && slot < Rc) ? ((ep + 16ull * j) | ((uint64_t)min(16u, eSz - 16u * j) << 40)) : 0ull;
    }
}
// ... and for the batch [r0, r0 + kGwRounds) of a read whose rounds take several batches (r0 = 0: the function above): the entry that
// reaches into the batch from before r0 leaves its number at slot 0
__device__ __forceinline__ void gw_fill_rounds_scan_at(uint64_t* T, uint32_t* E, uint32_t lane, uint32_t r0, uint32_t Rc, uint32_t start, uint32_t myR, uint32_t sz, uint64_t pay)
{
    E[lane] = 0u; E[lane + 64u] = 0u;
    wave_lds_sync();
    if (myR && start < r0 + kGwRounds && start + myR > r0) E[start > r0 ? start - r0 : 0u] = lane + 1u;   // (the entries' rounds are disjoint: distinct slots)
    wave_lds_sync();
    uint32_t lo = wave_incl_scan_max_u32(E[lane]);
    uint32_t hi = max(wave_incl_scan_max_u32(E[lane + 64u]), rdlane(lo, 63));
    const uint32_t payLo = (uint32_t)pay, payHi = (uint32_t)(pay >> 32);
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t slot = lane + 64u * h, who = h ? hi : lo;
        const int src = (int)(who ? who - 1u : 0u);
        const uint32_t eStart = (uint32_t)__shfl((int)start, src), eSz = (uint32_t)__shfl((int)sz, src);
        const uint32_t eLo = (uint32_t)__shfl((int)payLo, src), eHi = (uint32_t)__shfl((int)payHi, src);
        const uint32_t j = r0 + slot - eStart;
        const uint64_t ep = ((uint64_t)eHi << 32) | eLo;
        T[slot] = (who && r0 + slot < Rc) ? ((ep + 16ull * j) | ((uint64_t)min(16u, eSz - 16u * j) << 40)) : 0ull;
    }
}
// the batch's numbers: 16 bytes per lane and load, four lanes per round; places without a round read as kGwNone
__device__ __forceinline__ void gw_load_rounds(const uint64_t* T, const uint32_t* __restrict__ values32, uint32_t grp, uint32_t sub4, uint4 (&x)[kGwLoads],
                                               uint32_t nl = kGwLoads)   // nl: loads of this batch that have rounds at all (wave-uniform: the others are skipped)
{
#pragma unroll
    for (uint32_t u = 0; u < kGwLoads; ++u) {
        x[u] = make_uint4(kGwNone, kGwNone, kGwNone, kGwNone);
        if (u >= nl) continue;
        const uint64_t rd = T[u * 16 + grp];
        if (sub4 < (uint32_t)(rd >> 40)) {
            const U4 t = *reinterpret_cast<const U4*>(values32 + (rd & 0xFFFFFFFFFFull) + sub4);
            x[u] = make_uint4(t.x, t.y, t.z, t.w);
        }
    }
}
// phase A on a batch: every place of the loaded lines marks its block (places past a list's end hold other lists' numbers: more
// marks, never fewer)
template <class Bloom>
__device__ __forceinline__ void gw_mark_rounds(uint32_t* bits, const uint4 (&x)[kGwLoads], uint32_t A, uint32_t nl = kGwLoads)
{
#pragma unroll
    for (uint32_t u = 0; u < kGwLoads; ++u)                    // (lanes without a round hold kGwNone: sixty-four of them on ONE filter word would take turns)
        if (u < nl && x[u].x != kGwNone) Bloom::mark4(bits, x[u].x >> A, x[u].y >> A, x[u].z >> A, x[u].w >> A);
}
// phase B: a number is kept if its block was marked twice or it lies within D of a block boundary; the kept ones are appended to dst
// (ballot compaction), n2 counts them whether they fit or not
// (shared != nullptr: several waves fill one list -- the places are reserved in LDS, one atomic per call, n2 is not carried)
struct GwSink { uint32_t* dst; uint32_t room, n2; uint32_t* shared = nullptr; };
__device__ __forceinline__ uint32_t gw_reserve(uint32_t* shared, uint32_t tot)
{
    uint32_t at = 0;
    if (tot && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) at = atomicAdd(shared, tot);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
}
// FINE: the filter's blocks hold 2^A >= D numbers only -- a neighbour lies in the same block (marked twice) or in one of the two next to
// it (seen): three lookups instead of one and an edge rule, for reads whose tens of thousands of locations leave no block of 64 D
// numbers without a second one (gw_filter_stream_kernel<.., FINE>)
template <class Bloom, bool FINE = false>
__device__ __forceinline__ void gw_take(const uint32_t* bits, const GwFrame& F, GwSink& S, uint32_t v, bool valid)
{
    const uint32_t key = v >> F.A;
    const bool keep = valid & (FINE ? (Bloom::twice(bits, key) | Bloom::seen(bits, key - 1u) | Bloom::seen(bits, key + 1u)) : (Bloom::twice(bits, key) | F.edge(v)));
    const uint64_t m = __ballot(keep);
    if (S.shared) S.n2 = gw_reserve(S.shared, (uint32_t)__popcll(m));
    if (keep) {
        const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, S.n2));
        if (at < S.room) S.dst[at] = v;
    }
    S.n2 += (uint32_t)__popcll(m);
}
// four numbers at a time: the four filter words are read together (one LDS round trip), no branch but the ones around the stores.
// CHECK = false: the caller has made sure that everything this read can keep fits the slice.
template <class Bloom, bool CHECK, bool FINE = false>
__device__ __forceinline__ void gw_take4(const uint32_t* bits, const GwFrame& F, GwSink& S, const uint4 x, const int32_t rem)
{
    const uint32_t v[4] = {x.x, x.y, x.z, x.w};
    uint32_t m[4], wd[4];
    uint32_t mlo[FINE ? 4 : 1], wlo[FINE ? 4 : 1], mhi[FINE ? 4 : 1], whi[FINE ? 4 : 1];   // FINE: the "seen" words of the blocks before and behind
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t key = v[j] >> F.A;
        const uint32_t h = Bloom::hash(key);
        m[j] = h;                                                  // (the "twice" test works on the hash: twice_hit_h)
        wd[j] = bits[Bloom::twice_index(Bloom::seen_index(h), h)];
        if constexpr (FINE) {
            const uint32_t h0 = Bloom::hash(key - 1u), h1 = Bloom::hash(key + 1u);
            mlo[j] = Bloom::mask_of(h0); wlo[j] = bits[Bloom::seen_index(h0)];
            mhi[j] = Bloom::mask_of(h1); whi[j] = bits[Bloom::seen_index(h1)];
        }
    }
    // (ballots of the three compares, combined on the scalar unit: a ballot of the combined condition costs two VALU instructions more)
    uint64_t km[4]; bool kb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool valid = rem > j, hit = Bloom::twice_hit_h(wd[j], m[j]);
        if constexpr (FINE) {
            const bool lo = (wlo[j] & mlo[j]) == mlo[j], hi = (whi[j] & mhi[j]) == mhi[j];
            kb[j] = valid & (hit | lo | hi);
            km[j] = __ballot(valid) & (__ballot(hit) | __ballot(lo) | __ballot(hi));
        } else {
            const bool edge = F.edge(v[j]);
            kb[j] = valid & (hit | edge);
            km[j] = __ballot(valid) & (__ballot(hit) | __ballot(edge));
        }
    }
    if (S.shared) S.n2 = gw_reserve(S.shared, (uint32_t)(__popcll(km[0]) + __popcll(km[1]) + __popcll(km[2]) + __popcll(km[3])));
    uint32_t n2 = S.n2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (kb[j]) {
            // (the running count goes into the UNIFORM base of the store -- scalar arithmetic --, the lane adds its place among the kept ones)
            const uint32_t mine = __builtin_amdgcn_mbcnt_hi((uint32_t)(km[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km[j], 0u));
            uint32_t* const to = S.dst + n2;
            if (!CHECK || n2 + mine < S.room) to[mine] = v[j];
        }
        n2 += (uint32_t)__popcll(km[j]);
    }
    S.n2 = n2;
}
template <class Bloom, bool CHECK, bool FINE = false>
__device__ __forceinline__ void gw_take_rounds(const uint32_t* bits, const uint64_t* T, const GwFrame& F, GwSink& S, uint32_t grp, uint32_t sub4,
                                               const uint4 (&x)[kGwLoads], uint32_t nl = kGwLoads)
{
#pragma unroll
    for (uint32_t u = 0; u < kGwLoads; ++u) {
        if (u >= nl) continue;
        const int32_t rem = (int32_t)(uint32_t)(T[u * 16 + grp] >> 40) - (int32_t)sub4;     // numbers of the round from this lane's first on
        gw_take4<Bloom, CHECK, FINE>(bits, F, S, x[u], rem);
    }
}

// how many of a lane's four places per load hold numbers (0 .. 4), three bits per load: all that phase B needs of the round table -- the
// table's place can go to something else once the loads are issued (gw_filter_count_kernel's seven-wave instance)
__device__ __forceinline__ uint32_t gw_pack_rems(const uint64_t* T, uint32_t grp, uint32_t sub4, uint32_t nl)
{
    uint32_t packed = 0;
#pragma unroll
    for (uint32_t u = 0; u < kGwLoads; ++u) {
        if (u >= nl) continue;
        const uint32_t cnt = (uint32_t)(T[u * 16 + grp] >> 40);
        packed |= (cnt > sub4 ? min(cnt - sub4, 4u) : 0u) << (3u * u);
    }
    return packed;
}
template <class Bloom, bool CHECK>
__device__ __forceinline__ void gw_take_rounds_packed(const uint32_t* bits, const uint32_t packed, const GwFrame& F, GwSink& S, const uint4 (&x)[kGwLoads], uint32_t nl)
{
#pragma unroll
    for (uint32_t u = 0; u < kGwLoads; ++u) if (u < nl) gw_take4<Bloom, CHECK>(bits, F, S, x[u], (int32_t)((packed >> (3u * u)) & 7u));
}

constexpr uint32_t kGwDefer = 0xFFFFFFFEu;                 // record of list 7: left to gw_filter_stream_kernel
constexpr uint32_t kGwFallback = 0xFFFFFFFFu;              // ... handed to the wave kernel

}  // namespace

// The common case in ONE batch: a read with up to 64 found features whose lists are up to kGwRounds rounds (150 bp reads and most pairs
// at RefSeq scale: 26 lists of 49 numbers = 104 rounds).  The numbers are loaded ONCE -- 8 x 16 bytes per lane -- and stay in registers
// for both phases.  Reads that do not fit (and all with more than kGwSmallH locations) are left to gw_filter_stream_kernel.
#ifndef MC_GW_FILTER_WPE
#define MC_GW_FILTER_WPE 4
#endif
template <uint32_t WAVES, uint32_t TLOG2, uint32_t WPE = MC_GW_FILTER_WPE>
__global__ __launch_bounds__(WAVES * 64, WPE) void gw_filter_kernel(BatchView b, DeviceTable tab, Workspace ws)
{
    using Bloom = GwBloom<TLOG2, TLOG2>;
    __shared__ uint32_t bitS[WAVES][Bloom::kWords];
    __shared__ uint64_t roundS[WAVES][kGwRounds];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: the wave's pointers and counters live in scalar registers)
    uint32_t* bits = bitS[wave];
    uint64_t* T = roundS[wave];
    const uint32_t total = ws.midCount[9];
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)6 * b.n;
    uint4* __restrict__ outRec = reinterpret_cast<uint4*>(ws.midList) + (size_t)7 * b.n;
    const uint32_t nWaves = gridDim.x * WAVES;
    const uint32_t w0 = blockIdx.x * WAVES + wave;
    auto load_rec = [&](uint32_t w) -> uint4 { return w < total ? work[w] : make_uint4(0, 0, 0, 0); };
    // this wave's slice of the pool (no atomics on global memory: round 2 measured 43 ms for two shared cursors)
    const uint64_t sliceCap = ws.bigPoolCap / nWaves;
    uint32_t* const slice = reinterpret_cast<uint32_t*>(ws.bigPool) + (uint64_t)w0 * sliceCap;
    uint64_t sliceUsed = 0;
    uint32_t deferred = 0;
    uint4 rec = load_rec(w0), recNext = load_rec(w0 + nWaves);
    uint32_t esz = 0; uint64_t epay = 0;                           // the read's entries, fetched one read ahead
    auto load_entries = [&](const uint4& r) {
        const uint32_t ne = (r.z >> 12) <= kGwSmallH ? min(r.z & 0xFFFu, 64u) : 0u;
        esz = lane < ne ? ws.psize[r.y + lane] : 0u;
        epay = lane < ne ? ws.ppay[r.y + lane] : 0ull;
    };
    load_entries(rec);
    const uint32_t grp = lane >> 2, sub4 = (lane & 3u) * 4u;
    for (uint32_t w = w0; w < total; w += nWaves) {
        const uint32_t q = rec.x, nent = rec.z & 0xFFFu, H = rec.z >> 12, maxWin = rec.w;
        const uint32_t sz = esz & 0xFFFFu; const uint64_t pay = epay;
        rec = recNext;                                             // the next read's record and entries are on their way meanwhile
        recNext = load_rec(w + 2 * nWaves);
        load_entries(rec);
        const uint32_t myR = sz > 1 ? (sz + 15u) >> 4 : 0u;
        const uint32_t incl = wave_incl_scan_u32(myR, lane), Rc = rdlane(incl, 63);
        // (a slice that cannot take everything one batch may keep -- kGwRounds x 16 + 64 numbers -- sends the read on as well: the stores
        // below are not bounds-checked)
        if (H > kGwSmallH || nent > 64u || Rc > kGwRounds || maxWin > tab.gwGap || sliceCap - sliceUsed < kGwRounds * 16u + 64u) {
            if (lane == 0) outRec[w] = make_uint4(q, 0u, kGwDefer, maxWin);
            ++deferred;
            continue;
        }
        {
            uint4* z4 = reinterpret_cast<uint4*>(bits);
#pragma unroll
            for (uint32_t i = 0; i < Bloom::kWords / 4 / 64; ++i) z4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
        gw_fill_rounds(T, lane, 0, Rc, incl - myR, myR, sz, pay);
        wave_lds_sync();
        const GwFrame F(maxWin);
        const uint32_t nl = (Rc + 15u) >> 4;                        // (26 lists of 49 numbers: 104 rounds = 7 of the 8 loads)
        uint4 x[kGwLoads];
        gw_load_rounds(T, tab.values32, grp, sub4, x, nl);
        const uint32_t sv = sz == 1 ? tab.gw_of(pay) : kGwNone;     // single locations live in their buckets in the 8-byte form
        GwSink S{slice + sliceUsed, (uint32_t)min((uint64_t)kGwMaxKept, sliceCap - sliceUsed), 0u};
        // ---- A
        if (sv != kGwNone) Bloom::mark(bits, sv >> F.A);
        gw_mark_rounds<Bloom>(bits, x, F.A, nl);
        wave_lds_sync();
        // ---- B
        gw_take<Bloom>(bits, F, S, sv, sv != kGwNone);
        gw_take_rounds<Bloom, false>(bits, T, F, S, grp, sub4, x, nl);
        const bool fallback = S.n2 > S.room;                       // longer than what is handed on, or the slice is full
        if (lane == 0) {
            if (fallback) { ws.hitScan[q] = H; ws.qflag[q] = kFlagCands; outRec[w] = make_uint4(q, 0u, kGwFallback, maxWin); }
            else {
                outRec[w] = make_uint4(q, (uint32_t)((uint64_t)w0 * sliceCap + sliceUsed), S.n2, maxWin);
            }
        }
        if (!fallback) sliceUsed += S.n2;
        wave_lds_sync();
    }
    if (lane == 0) {
        if (ws.sliceFill) ws.sliceFill[w0] = (uint32_t)sliceUsed;
        if (deferred) atomicAdd(&ws.midCount[10], deferred);       // (one atomic per wave that met such reads at all)
    }
}

// The same for reads of up to 2 x kGwRounds rounds and 2 x kGwSmallH locations (2 x 150 bp pairs at RefSeq scale: 52 lists of 49
// numbers = 208 rounds): TWO batches.  The second one stays in registers for phase B, the first one is read again (from the L2): one
// and a half passes over the fabric instead of the streaming kernel's two, and its occupancy (8 KB of filter bits per wave).
// Takes the records gw_filter_kernel left (kGwDefer), 64 at a time; leaves what does not fit either to gw_filter_stream_kernel.
template <uint32_t WAVES, uint32_t TLOG2, uint32_t T2LOG2 = TLOG2>
__global__ __launch_bounds__(WAVES * 64, T2LOG2 == TLOG2 ? MC_GW_FILTER_WPE : 6) void gw_filter2_kernel(BatchView b, DeviceTable tab, Workspace ws)
{
    using Bloom = GwBloom<TLOG2, T2LOG2>;
    __shared__ uint32_t bitS[WAVES][Bloom::kWords];
    __shared__ uint64_t roundS[WAVES][kGwRounds];
    __shared__ uint32_t scanS[WAVES][kGwRounds];                   // scratch of the round tables' max-scan
    if (ws.midCount[10] == 0) return;                              // nothing was left
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* bits = bitS[wave];
    uint64_t* T = roundS[wave];
    uint32_t* E = scanS[wave];
    const uint32_t total = ws.midCount[9];
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)6 * b.n;
    uint4* outRec = reinterpret_cast<uint4*>(ws.midList) + (size_t)7 * b.n;
    const uint32_t nWaves = gridDim.x * WAVES;
    const uint32_t w0 = blockIdx.x * WAVES + wave;
    const uint64_t sliceCap = ws.bigPoolCap / nWaves;
    uint32_t* const slice = reinterpret_cast<uint32_t*>(ws.bigPool) + (uint64_t)w0 * sliceCap;
    uint64_t sliceUsed = ws.sliceFill ? ws.sliceFill[w0] : 0u;
    const uint32_t grp = lane >> 2, sub4 = (lane & 3u) * 4u;
    constexpr uint32_t kMaxRounds = 2 * kGwRounds, kMaxH = 2 * kGwSmallH;
    uint32_t esz = 0; uint64_t epay = 0;
    auto load_entries = [&](uint32_t fbase, uint32_t ne) {
        esz = lane < ne ? ws.psize[fbase + lane] : 0u;
        epay = lane < ne ? ws.ppay[fbase + lane] : 0ull;
    };
    for (uint32_t chunk = w0 * 64u; chunk < total; chunk += nWaves * 64u) {
      const bool inb = chunk + lane < total;
      const uint4 myRec = inb ? work[chunk + lane] : make_uint4(0, 0, 0, 0);
      const uint32_t myZ = inb ? outRec[chunk + lane].z : 0u;
      uint64_t todo = __ballot(inb && myZ == kGwDefer && (myRec.z >> 12) <= kMaxH && (myRec.z & 0xFFFu) <= 64u && myRec.w <= tab.gwGap);
      if (todo) { const uint32_t j = (uint32_t)__ffsll((unsigned long long)todo) - 1; load_entries(rdlane(myRec.y, j), rdlane(myRec.z, j) & 0xFFFu); }
      while (todo) {
        const uint32_t j = (uint32_t)__ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t w = chunk + j, q = rdlane(myRec.x, j), H = rdlane(myRec.z, j) >> 12, maxWin = rdlane(myRec.w, j);
        const uint32_t sz = esz & 0xFFFFu; const uint64_t pay = epay;
        if (todo) { const uint32_t jn = (uint32_t)__ffsll((unsigned long long)todo) - 1; load_entries(rdlane(myRec.y, jn), rdlane(myRec.z, jn) & 0xFFFu); }
        const uint32_t myR = sz > 1 ? (sz + 15u) >> 4 : 0u;
        const uint32_t incl = wave_incl_scan_u32(myR, lane), Rc = rdlane(incl, 63), start = incl - myR;
        if (Rc > kMaxRounds || sliceCap - sliceUsed < kMaxRounds * 16u + 64u) continue;   // stays deferred: gw_filter_stream_kernel
        {
            uint4* z4 = reinterpret_cast<uint4*>(bits);
#pragma unroll
            for (uint32_t i = 0; i < Bloom::kWords / 4 / 64; ++i) z4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
        const GwFrame F(maxWin);
        const uint32_t sv = sz == 1 ? tab.gw_of(pay) : kGwNone;
        GwSink S{slice + sliceUsed, (uint32_t)min((uint64_t)kGwMaxKept, sliceCap - sliceUsed), 0u};
        uint4 x[kGwLoads];
        // Round 6: the round tables by the max-scan (gw_fill_rounds_scan_at: ~40 instructions whatever the lists' lengths) instead of the
        // lanes' loops (~130 for lists of 254 numbers, three times per pair).  Measured and left (lab notebook r06 section 7): BOTH batches in
        // registers, no second read of the first -- the kernel runs at four waves per SIMD for its LDS anyway, but it needed 143 registers
        // of the 128 there are: 15 spilled, 12.3 against 10.3 ms per 2.5 x 10^6 pairs.
        // ---- A: batch 0, then batch 1 (stays in registers)
        if (sv != kGwNone) Bloom::mark(bits, sv >> F.A);
        gw_fill_rounds_scan_at(T, E, lane, 0, Rc, start, myR, sz, pay);
        wave_lds_sync();
        gw_load_rounds(T, tab.values32, grp, sub4, x);
        gw_mark_rounds<Bloom>(bits, x, F.A);
        const bool two = Rc > kGwRounds;
        if (two) {
            wave_lds_sync();
            gw_fill_rounds_scan_at(T, E, lane, kGwRounds, Rc, start, myR, sz, pay);
            wave_lds_sync();
            gw_load_rounds(T, tab.values32, grp, sub4, x);
            gw_mark_rounds<Bloom>(bits, x, F.A);
        }
        wave_lds_sync();
        // ---- B: the batch in registers, then (two batches) batch 0 again
        gw_take<Bloom>(bits, F, S, sv, sv != kGwNone);
        gw_take_rounds<Bloom, false>(bits, T, F, S, grp, sub4, x);
        if (two) {
            wave_lds_sync();
            gw_fill_rounds_scan_at(T, E, lane, 0, Rc, start, myR, sz, pay);
            wave_lds_sync();
            gw_load_rounds(T, tab.values32, grp, sub4, x);
            gw_take_rounds<Bloom, false>(bits, T, F, S, grp, sub4, x);
        }
        if (lane == 0) outRec[w] = make_uint4(q, (uint32_t)((uint64_t)w0 * sliceCap + sliceUsed), S.n2, maxWin);
        sliceUsed += S.n2;
        wave_lds_sync();
      }
    }
    if (lane == 0 && ws.sliceFill) ws.sliceFill[w0] = (uint32_t)sliceUsed;
}

// The kernels that take FEW of the batch's records get their record numbers as compact lists (ws.sideList), so that they can deal them
// out wave by wave -- scanning 5 x 10^6 records for a few hundred cost a millisecond per kernel, and taking them 64 at a time puts 64
// long reads on one wave (and into one pool slice).  stage 0, after gw_filter_kernel: the reads gw_filter_stream_kernel takes;
// stage 1, after it: filtered lists of 257 .. 512 and 513 .. 1024 numbers (counting instances) and the ones that are sorted.
// One atomic per class and 64 records.
__global__ __launch_bounds__(256) void gw_compact_kernel(Workspace ws, uint32_t n, uint32_t stage)
{
    // a block takes a contiguous stretch of the records: it counts its members of every class first, reserves their places with ONE
    // atomic per class (78 000 atomics on one counter -- one per 64 records -- took 0.4 ms), then writes them in record order
    constexpr uint32_t kC = 4;                                     // classes = side lists; class kC: none
    __shared__ uint32_t cnt[4][kC], base[kC];
    const uint32_t total = ws.midCount[9];
    const uint4* __restrict__ rec7 = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * n;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t per = ((total + gridDim.x - 1) / gridDim.x + 255u) / 256u * 256u;          // records per block, whole 256-record steps
    const uint32_t lo = blockIdx.x * per, hi = min(total, lo + per);
    auto class_of = [&](uint32_t i) -> uint32_t {
        if (i >= hi) return kC;
        const uint4 r = rec7[i];
        if (stage == 0) return r.z == kGwDefer ? 0u : kC;
        if (gw_sorted_class(r.z, r.w)) return 3u;
        if (r.z <= kBigMaxFilteredCount && r.w <= kHashWin) return r.z > 512u ? 2u : r.z > 256u ? 1u : kC;
        return kC;
    };
    uint32_t mine[kC];
#pragma unroll
    for (uint32_t k = 0; k < kC; ++k) mine[k] = 0;
    for (uint32_t i = lo + threadIdx.x; i < lo + per && lo < hi; i += 256) {
        const uint32_t c = class_of(i);
#pragma unroll
        for (uint32_t k = 0; k < kC; ++k) mine[k] += (uint32_t)__popcll(__ballot(c == k));       // (wave-uniform counts)
    }
    if (lane == 0) { for (uint32_t k = 0; k < kC; ++k) cnt[wave][k] = mine[k]; }
    __syncthreads();
    if (threadIdx.x < kC) {
        const uint32_t k = threadIdx.x, tot = cnt[0][k] + cnt[1][k] + cnt[2][k] + cnt[3][k];
        base[k] = tot ? atomicAdd(&ws.midCount[k == 0 ? 12u : k == 1 ? 14u : k == 2 ? 15u : 13u], tot) : 0u;
    }
    __syncthreads();
    // second pass: step s of the block holds records lo + s * 256 .. ; within a step the waves' members follow each other
    uint32_t run[kC];
#pragma unroll
    for (uint32_t k = 0; k < kC; ++k) run[k] = base[k];
    for (uint32_t i0 = lo; i0 < lo + per && lo < hi; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t c = class_of(i);
        uint64_t m[kC];
#pragma unroll
        for (uint32_t k = 0; k < kC; ++k) m[k] = __ballot(c == k);
        __syncthreads();
        if (lane == 0) { for (uint32_t k = 0; k < kC; ++k) cnt[wave][k] = (uint32_t)__popcll(m[k]); }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kC; ++k) {
            uint32_t before = 0, all = 0;
            for (uint32_t w = 0; w < 4; ++w) { before += w < wave ? cnt[w][k] : 0u; all += cnt[w][k]; }
            if (c == k) ws.sideList[(size_t)k * n + run[k] + before + __popcll(m[k] & ((1ull << lane) - 1ull))] = i;
            run[k] += all;
        }
    }
}

// Everything else: reads with more than 64 found features, lists of more than kGwRounds rounds, more than kGwSmallH locations (long
// reads: thousands to tens of thousands).  Entries in chunks of 64, rounds in batches of kGwRounds, two passes over the lists (the
// second one finds them in the L2 / infinity cache), filters sized for tens of thousands of keys.  Same waves, same pool slices
// (ws.sliceFill) as gw_filter_kernel, after which it runs; returns at once when the batch has no such read (midCount[10]).
// Two instances by the read's locations H in (hMin, hMax].  FINE (reads beyond kGwBigH locations -- 4 kbp and more: 1.7 % of configs[4]'s
// reads with an eighth of its locations): with tens of thousands of locations in 1.4 x 10^9 numbers every block of 64 D numbers holds a
// second location and the block filter keeps nearly all of them (a 19 kbp read kept 10^5 of its 1.6 x 10^5: these lists were 90 of the
// 235 x 10^6 numbers a batch sorts and 3.3 of the sort's 5.3 ms).  Their instance asks blocks of 2^A >= D numbers -- the location's own
// ("twice") and the two next to it ("seen") -- in filters of 2^19 + 2^17 bits, one BLOCK of sixteen waves per read and CU.
constexpr uint32_t kGwBigH = 32768;
template <uint32_t WAVES, uint32_t T1LOG2, uint32_t T2LOG2, bool FINE = false>
__global__ __launch_bounds__(WAVES * 64) void gw_filter_stream_kernel(BatchView b, DeviceTable tab, Workspace ws, uint32_t hMin, uint32_t hMax, uint32_t nSlices)
{
    // nSlices: the pool's slices = the waves of gw_filter_kernel's grid (4 x fgrid), whatever this instance's block size: a block owns
    // slices [blockIdx.x * WAVES, min(.. + WAVES, nSlices)) -- the last block of the sixteen-wave instance may own fewer than sixteen
    // (the slicing used to be derived from gridDim.x * WAVES, which is another slicing whenever fgrid is not a multiple of four)
    // One BLOCK per read: its waves share ONE pair of filters (20 KB: with a pair per wave six waves fitted a CU) and take the read's
    // entry chunks in turn -- phase A of all chunks, barrier, phase B; the kept numbers of all waves go to one list, its places
    // reserved with an LDS counter.  The pool: the slices of this block's waves as the filter kernels before left them.
    using Bloom = GwBloom<T1LOG2, T2LOG2>;
    __shared__ uint32_t bits[Bloom::kWords];
    __shared__ uint64_t roundS[WAVES][kGwRounds];
    __shared__ uint32_t scanS[WAVES][kGwRounds];                   // scratch of the round tables' max-scan (round 6: gw_fill_rounds_scan_at)
    __shared__ uint32_t n2S;
    __shared__ unsigned long long ovfS;
    if (ws.midCount[12] == 0) return;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t* T = roundS[wave];
    uint32_t* E = scanS[wave];
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)6 * b.n;
    uint4* outRec = reinterpret_cast<uint4*>(ws.midList) + (size_t)7 * b.n;
    const uint32_t w0 = blockIdx.x * WAVES;                        // the block's first slice
    const uint64_t sliceCap = ws.bigPoolCap / nSlices;
    const uint32_t mySlices = w0 < nSlices ? min(WAVES, nSlices - w0) : 0u;
    if (mySlices == 0) return;
    uint64_t used[WAVES];                                          // block-uniform copies (a slice this block does not own: full)
#pragma unroll
    for (uint32_t k = 0; k < WAVES; ++k) used[k] = k < mySlices ? (ws.sliceFill ? ws.sliceFill[w0 + k] : 0u) : sliceCap;
    const uint32_t grp = lane >> 2, sub4 = (lane & 3u) * 4u;
    const uint32_t mine = ws.midCount[12];
    const uint32_t* __restrict__ side = ws.sideList;
    for (uint32_t i = blockIdx.x; i < mine; i += gridDim.x) {
        const uint32_t w = side[i];
        const uint4 rec = work[w];
        const uint32_t q = rec.x, fbase = rec.y, recZ = rec.z, maxWin = rec.w;
        const uint32_t nent = recZ & 0xFFFu, H = recZ >> 12;
        if (H <= hMin || H > hMax) continue;                       // (block-uniform) the other instance's read
        {
            uint4* z4 = reinterpret_cast<uint4*>(bits);
            for (uint32_t j = threadIdx.x; j < Bloom::kWords / 4; j += WAVES * 64) z4[j] = make_uint4(0, 0, 0, 0);
        }
        if (threadIdx.x == 0) n2S = 0u;
        const GwFrame F(maxWin, FINE);
        // where the list goes: the first of the block's slices that can take all H numbers, else H places of the overflow region
        // (the filter keeps H at most), else what is left of the first slice (a list that outgrows it goes to the wave kernel)
        int k = -1;
#pragma unroll
        for (uint32_t kk = 0; kk < WAVES; ++kk) if (k < 0 && (uint64_t)H <= min((uint64_t)kGwMaxKept, sliceCap - used[kk])) k = (int)kk;
        uint64_t listAt = 0; uint32_t room = 0;
        if (k >= 0) { listAt = (uint64_t)(w0 + (uint32_t)k) * sliceCap + used[k]; room = (uint32_t)min((uint64_t)kGwMaxKept, sliceCap - used[k]); }
        else {
            if (threadIdx.x == 0) ovfS = atomicAdd(reinterpret_cast<unsigned long long*>(ws.midCount + 16), (unsigned long long)H);
            __syncthreads();
            const unsigned long long at = ovfS;
            if (at + H <= ws.bigOvfCap) { listAt = (uint64_t)ws.bigPoolCap + at; room = H; }
            else { k = 0; listAt = (uint64_t)w0 * sliceCap + used[0]; room = (uint32_t)min((uint64_t)kGwMaxKept, sliceCap - used[0]); }
        }
        GwSink S{reinterpret_cast<uint32_t*>(ws.bigPool) + listAt, room, 0u, &n2S};
        __syncthreads();                                           // filters cleared, counter zero
        bool fallback = maxWin > tab.gwGap;                        // (window ranges wider than the gap between two targets: the kernels that know the targets)
        const uint32_t nchunks = (nent + 63u) / 64u;
        uint32_t n2 = 0;
        if (!fallback) {
            for (uint32_t pass = 0; pass < 2; ++pass) {
                for (uint32_t c = wave; c < nchunks; c += WAVES) {
                    const uint32_t e = c * 64u + lane;
                    const uint32_t sz = e < nent ? (ws.psize[fbase + e] & 0xFFFFu) : 0u;
                    const uint64_t pay = e < nent ? ws.ppay[fbase + e] : 0ull;
                    const uint32_t sv = sz == 1 ? tab.gw_of(pay) : kGwNone;
                    if (pass == 0) { if (sv != kGwNone) Bloom::mark(bits, sv >> F.A); } else gw_take<Bloom, FINE>(bits, F, S, sv, sv != kGwNone);
                    const uint32_t myR = sz > 1 ? (sz + 15u) >> 4 : 0u;
                    const uint32_t incl = wave_incl_scan_u32(myR, lane), Rc = rdlane(incl, 63);
                    for (uint32_t r0 = 0; r0 < Rc; r0 += kGwRounds) {
                        gw_fill_rounds_scan_at(T, E, lane, r0, Rc, incl - myR, myR, sz, pay);
                        wave_lds_sync();
                        uint4 x[kGwLoads];
                        gw_load_rounds(T, tab.values32, grp, sub4, x);
                        if (pass == 0) gw_mark_rounds<Bloom>(bits, x, F.A); else gw_take_rounds<Bloom, true, FINE>(bits, T, F, S, grp, sub4, x);
                        wave_lds_sync();                           // the table is rewritten by the next batch
                    }
                }
                __syncthreads();                                   // every wave's marks before anybody's tests; every wave's numbers before the count is read
            }
            n2 = n2S;
            fallback = n2 > room;
        }
        if (threadIdx.x == 0) {
            if (fallback) { ws.hitScan[q] = H; ws.qflag[q] = kFlagCands; outRec[w] = make_uint4(q, 0u, kGwFallback, maxWin); }
            else outRec[w] = make_uint4(q, (uint32_t)listAt, n2, maxWin);
        }
        if (!fallback && k >= 0) used[k] += n2;
        __syncthreads();                                           // (the counter and the filters are reset for the next read)
    }
    if (threadIdx.x == 0 && ws.sliceFill) {                        // (the next instance goes on in the same slices)
#pragma unroll
        for (uint32_t k = 0; k < WAVES; ++k) if (k < mySlices) ws.sliceFill[w0 + k] = (uint32_t)used[k];
    }
}

namespace {

// ================================================================================================
// gw_count_kernel: rows 8-10 on a filtered list, without a sort (round 1's hash_cands_kernel on global window numbers):
//   1. every number is counted in an LDS hash table of {number, count} slots (32-bit compare-and-swap claims a slot, the count sits
//      in the same 8 bytes: one ds_read_b64 per lookup);
//   2. the lane that claimed number g adds the counts of g - 1 .. g - (maxWindowsInRange - 1): the hits of the window range that
//      ENDS in g (candidate_generation.hpp:47-108 evaluates exactly these; the first to reach a target's maximum is the one with the
//      smallest end window); begin = the smallest number present among them;
//   3. K rounds: wave-wide maximum of the hits, the smallest number among its holders, that target (its numbers: one look at the
//      directory per round) or taxon is struck from the race -- the sequential top-K insert of candidate_generation.hpp:172-231;
//   D. when fewer than K picked candidates have two or more hits, the open places go to the smallest numbers of other targets among
//      ALL the read's locations (hits = 1 candidates arrive in (target, window) order and keep it) -- one sweep over the lists.
// ================================================================================================
template <uint32_t LOG2S>
__device__ __forceinline__ uint32_t gw_slot(uint32_t g)                 // byte offset of the home slot
{
    uint32_t h;                                                // (24-bit multiply as an instruction, see GwBloom::hash)
    asm("v_mul_u32_u24_e32 %0, 0x9e3779, %1" : "=v"(h) : "v"(g ^ (g >> 12)));
    return (h >> (29u - LOG2S)) & (((1u << LOG2S) - 1u) << 3);
}

// Phase 1 on the PER numbers every lane holds: counted in the table; the numbers a lane CLAIMED (one lane per distinct number) go,
// with their slots, to the compact list ck -- a filtered list of 195 numbers has about 50 distinct ones, and everything after the
// counting (neighbour windows, the K rounds) is per distinct number: one per lane instead of four.  Returns how many.
template <uint32_t LOG2S, uint32_t PER, uint32_t CKCAP = 0xFFFFFFFFu>
__device__ __forceinline__ uint32_t gw_count_numbers(const uint32_t (&v)[PER], uint2* slots, uint32_t* ck, const uint32_t lane)
{
    constexpr uint32_t kByteMask = ((1u << LOG2S) - 1u) << 3;
    char* base = reinterpret_cast<char*>(slots);
    auto key_at = [&](uint32_t off) -> uint32_t* { return reinterpret_cast<uint32_t*>(base + off); };
    uint32_t off[PER], old[PER];
    bool coll = false;
#pragma unroll
    for (uint32_t r = 0; r < PER; ++r) {
        off[r] = gw_slot<LOG2S>(v[r]);
        old[r] = v[r] != kGwNone ? atomicCAS(key_at(off[r]), kGwNone, v[r]) : v[r];
        coll = coll || (old[r] != kGwNone && old[r] != v[r]);
    }
    if (__ballot(coll)) {                                      // somebody else's number in the home slot: next slots, one at a time
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            if (old[r] != kGwNone && old[r] != v[r]) {
                uint32_t o = off[r];
                for (;;) {
                    o = (o + 8u) & kByteMask;
                    old[r] = atomicCAS(key_at(o), kGwNone, v[r]);
                    if (old[r] == kGwNone || old[r] == v[r]) break;
                }
                off[r] = o;
            }
        }
    }
    uint32_t C = 0;
#pragma unroll
    for (uint32_t r = 0; r < PER; ++r) {
        if (v[r] != kGwNone) atomicAdd(key_at(off[r]) + 1, 1u);
        const bool claimed = v[r] != kGwNone && old[r] == kGwNone;
        const uint64_t m = __ballot(claimed);
        if (claimed) {
            const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, C));
            if (CKCAP == 0xFFFFFFFFu || at < CKCAP) ck[at] = off[r];   // (the slot holds the number; CKCAP: the caller looks at the count returned)
        }
        C += (uint32_t)__popcll(m);
    }
    return C;
}

// Phases 2 and 3 on the distinct numbers (PER of them per lane, from ck): hits of the window range that ends in each, then the K rounds.
template <uint32_t LOG2S, uint32_t PER, bool TAX>
__device__ __forceinline__ uint32_t gw_pick(const uint32_t* ck, const uint32_t C, uint2* slots, const uint32_t lane, const uint32_t maxWin,
                                            const uint32_t K, const uint32_t* __restrict__ taxkey, const DeviceTable& tab,
                                            uint32_t& owv, uint32_t& owh, uint32_t& owd)
{
    constexpr uint32_t kByteMask = ((1u << LOG2S) - 1u) << 3;
    const char* base = reinterpret_cast<const char*>(slots);
    uint32_t v[PER], R[PER];
    // ---- 2. ranges that end in this lane's numbers: hits | (end - begin) << 16
#pragma unroll
    for (uint32_t r = 0; r < PER; ++r) {
        const uint2 c = r * 64 + lane < C ? *reinterpret_cast<const uint2*>(base + ck[r * 64 + lane]) : make_uint2(kGwNone, 0u);   // {number, count}
        v[r] = c.x; R[r] = c.y;
    }
    // (taxon merging: the targets of the numbers -- two dependent loads that hit the L2, started here so that they run behind the
    // neighbour lookups; without it the K rounds strike REGIONS and the targets of the K winners are looked up afterwards)
    uint32_t tgt[TAX ? PER : 1], thi[TAX ? PER : 1];
    if constexpr (TAX) {
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) tgt[r] = v[r] != kGwNone ? tab.gwDir[v[r] >> tab.gwDirShift] : 0u;
    }
    for (uint32_t d = 1; d < maxWin; ++d) {
        uint2 kc[PER]; uint32_t o[PER];
        bool chain = false;
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            o[r] = gw_slot<LOG2S>(v[r] - d);                   // (v - d is never kGwNone: numbers start at gwGap >= maxWin)
            kc[r] = *reinterpret_cast<const uint2*>(base + o[r]);
            if (v[r] == kGwNone) kc[r].x = kGwNone;
            chain = chain || (kc[r].x != v[r] - d && kc[r].x != kGwNone);
        }
        if (__ballot(chain)) {
#pragma unroll
            for (uint32_t r = 0; r < PER; ++r)
                while (kc[r].x != v[r] - d && kc[r].x != kGwNone) { o[r] = (o[r] + 8u) & kByteMask; kc[r] = *reinterpret_cast<const uint2*>(base + o[r]); }
        }
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r)
            if (kc[r].x == v[r] - d) R[r] = ((R[r] & 0xFFFFu) + kc[r].y) | (d << 16);
    }
    uint32_t ptax[TAX ? PER : 1];
    if constexpr (TAX) {
        // (a directory entry names the target of its block's FIRST number: the few numbers behind a target boundary inside a block move on)
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            thi[r] = v[r] != kGwNone ? tab.gwBase[tgt[r] + 1] : 0u;
            while (v[r] != kGwNone && v[r] >= thi[r]) { ++tgt[r]; thi[r] = tab.gwBase[tgt[r] + 1]; }
            ptax[r] = v[r] != kGwNone ? taxkey[tgt[r]] : 0u;
        }
    }
    // ---- 3. K rounds: wave-wide maximum of the hits, the smallest number among its holders.  Struck from the race: the winner's taxon
    //      (taxon merging), else every number within gwGap of the winner -- its REGION, which lies inside its target (that is what the
    //      gap is for).  The targets of the K winners are looked up afterwards, all at once; should two of them be one target (two
    //      regions of it more than 115 kbp apart, both with hits to show) the caller hands the read to the exact wave kernel.
    //      Why that is exact: region striking offers, in every round, a superset of what target striking offers; its winner is either
    //      the same or a number of an earlier winner's target -- which the comparison finds.
    uint32_t live = 0;
#pragma unroll
    for (uint32_t r = 0; r < PER; ++r) {
        bool ok = v[r] != kGwNone;
        if constexpr (TAX) ok = ok && ptax[r] != 0;            // no taxon at that rank: skipped (candidate_generation.hpp:185)
        live |= ok ? (1u << r) : 0u;
    }
    uint32_t strong = 0;
    uint32_t wv = kGwNone, wh = 0, wd = 0;                     // lane i keeps winner i: number, hits, end - begin
    for (uint32_t rnd = 0; rnd < K; ++rnd) {
        uint32_t hh = 0, hv = kGwNone, hd = 0, hg = 0;
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            const uint32_t h = R[r] & 0xFFFFu;
            const bool take = ((live >> r) & 1u) && (h > hh || (h == hh && v[r] < hv));
            if (take) { hh = h; hv = v[r]; hd = R[r] >> 16; if constexpr (TAX) hg = ptax[r]; }
        }
        const uint32_t mh = wave_max_u32(hh);
        if (mh == 0) break;
        const uint32_t mv = wave_min_u32(hh == mh ? hv : kGwNone);
        const uint32_t winner = __ffsll((unsigned long long)__ballot(hh == mh && hv == mv)) - 1;
        const uint32_t d = rdlane(hd, winner);
        if constexpr (TAX) {
            const uint32_t g = rdlane(hg, winner);
#pragma unroll
            for (uint32_t r = 0; r < PER; ++r) if (ptax[r] == g) live &= ~(1u << r);
        } else {
            const uint32_t from = mv - tab.gwGap, span = 2u * tab.gwGap;
#pragma unroll
            for (uint32_t r = 0; r < PER; ++r) if (v[r] - from <= span) live &= ~(1u << r);
        }
        if (lane == rnd) { wv = mv; wh = mh; wd = d; }
        strong += mh >= 2 ? 1u : 0u;
    }
    owv = wv; owh = wh; owd = wd;                              // lane i: winner i (the caller looks their targets up: gw_winners_out)
    return strong;
}

// The K winners of a read (lane i: number, hits, end - begin; target wt with its numbers [wlo, whi) looked up by the caller) -> the
// candidates.  Returns true when two winners are one target (region striking met another region of a picked target): the caller hands
// the read to the exact wave kernel.
template <bool TAX>
__device__ __forceinline__ bool gw_winners_out(const uint32_t lane, const uint32_t K, const uint32_t wv, const uint32_t wh, const uint32_t wd,
                                               const uint32_t wt, const uint32_t wlo, const uint32_t whi, mc_candidate_dev* __restrict__ out,
                                               uint32_t (&pickLo)[kLaneK], uint32_t (&pickHi)[kLaneK])
{
    bool again = false;
    if constexpr (!TAX) {
#pragma unroll
        for (uint32_t i = 0; i + 1 < kLaneK; ++i) {
            const uint32_t ti = rdlane(wt, i);
            again = again || (__ballot(lane > i && lane < K && wt == ti && ti != 0xFFFFFFFFu) != 0);
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < kLaneK; ++i) { pickLo[i] = rdlane(wlo, i); pickHi[i] = rdlane(whi, i); }
    if (lane < K) {
        mc_candidate_dev e; e.tgt = wt; e.hits = wh; e.end = wv - wlo; e.beg = e.end - wd;
        if (wv == kGwNone) { e.tgt = 0xFFFFFFFFu; e.hits = 0; e.beg = 0; e.end = 0; }
        out[lane] = e;
    }
    return again;
}

}  // namespace

namespace {

// The winners' target lookup is two dependent loads from global memory (directory, gwBase) at the very end of a read: it is DEFERRED --
// directory entry requested when the winners are known, gwBase words before the next read's counting, the candidates written after
// it -- so that the loads run behind the next read's LDS phases (a read with places left for single hits finishes at once: step D
// needs the picked targets).  One of these per wave.
struct GwPend {
    bool pend = false;
    uint32_t q = 0, wv = kGwNone, wh = 0, wd = 0, dir = 0, b0 = 0, b1 = 0, b2 = 0;
    __device__ __forceinline__ void bases(const DeviceTable& tab)   // stage 1: the three gwBase words behind the directory entry
    {
        if (pend && wv != kGwNone) { b0 = tab.gwBase[dir]; b1 = tab.gwBase[dir + 1]; b2 = tab.gwBase[min(dir + 2, tab.gwTargets)]; }
    }
    template <bool TAX>
    __device__ __forceinline__ void finish(const uint32_t lane, const uint32_t K, const DeviceTable& tab, const Workspace& ws, mc_candidate_dev* __restrict__ cands)   // stage 2
    {
        if (!pend) return;
        uint32_t t = 0xFFFFFFFFu, lo = 0, hi = 0;
        if (wv != kGwNone) {
            t = dir; lo = b0; hi = b1;
            if (wv >= b1) { ++t; lo = b1; hi = b2; }
            while (wv >= hi) { ++t; lo = hi; hi = tab.gwBase[t + 1]; }
        }
        uint32_t plo[kLaneK], phi[kLaneK];
        const bool again = gw_winners_out<TAX>(lane, K, wv, wh, wd, t, lo, hi, cands + (size_t)q * K, plo, phi);
        if (lane == 0) {
            if (again) { ws.hitScan[q] = ws.qstat[q].hits; ws.qflag[q] = kFlagCands; }
            else ws.qflag[q] = kFlagDone;
        }
        pend = false;
    }
};

// Rows 8-10 for ONE read on its filtered list of n2 <= 2^LOG2S / 2 numbers, window ranges up to kHashWin: getv(r) hands this lane its
// r-th number (position r * 64 + lane; kGwNone past the end).  slots: the wave's table of 2^LOG2S {number, count} slots, ck: room for
// the slots of the distinct numbers, T: a round table (step D; may share memory with slots or ck).  w: the read's record in work list 6.
// DEFER: the winners' target lookup waits in P for the caller's next read (see GwPend; the caller finishes the last one).
// LONG (first instance's table only): the list may hold up to 2^LOG2S numbers as long as no more than half of them are DISTINCT (a filtered
// list of 400 numbers has about 100 distinct ones); a list with more goes to the exact wave kernel.
// entf(): where step D finds the read's entries -- {first entry slot in ws.psize / ws.ppay, entries} (the record of work list 6)
template <uint32_t LOG2S, bool TAX, bool DEFER, bool LONG = false, class GetV, class EntF>
__device__ __forceinline__ bool gw_count_read(const uint32_t q, EntF&& entf, const uint32_t n2, const uint32_t maxWin, GetV&& getv,
                                              uint2* slots, uint32_t* ck, uint64_t* T, const uint32_t lane, const uint32_t grp, const uint32_t sub4,
                                              const uint32_t K, const uint32_t* __restrict__ taxkey, const DeviceTable& tab, const Workspace& ws,
                                              mc_candidate_dev* __restrict__ cands, GwPend& P)
{
    constexpr uint32_t kSlots = 1u << LOG2S, kList = kSlots / 2;
    {
        uint4* k4 = reinterpret_cast<uint4*>(slots);
#pragma unroll
        for (uint32_t i = 0; i < kSlots * 8 / 16 / 64; ++i) k4[i * 64 + lane] = make_uint4(kGwNone, 0u, kGwNone, 0u);
    }
    wave_lds_sync();
    uint32_t pickLo[kLaneK], pickHi[kLaneK];
#pragma unroll
    for (uint32_t i = 0; i < kLaneK; ++i) { pickLo[i] = 0; pickHi[i] = 0; }
    uint32_t strong = 0, wv = kGwNone, wh = 0, wd = 0;
    bool again = false;
    mc_candidate_dev* out = cands + (size_t)q * K;
    if constexpr (DEFER) P.bases(tab);
    uint32_t C = 0;
    auto body = [&](auto perc) {
        constexpr uint32_t PER = decltype(perc)::value;
        uint32_t v[PER];
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) v[r] = getv(r);
        C = gw_count_numbers<LOG2S, PER, LONG ? kList : 0xFFFFFFFFu>(v, slots, ck, lane);
    };
    const uint32_t per = (n2 + 63u) / 64u;
    if constexpr (LOG2S == 9) {
        if (per <= 1) body(std::integral_constant<uint32_t, 1>{});
        else if (per <= 2) body(std::integral_constant<uint32_t, 2>{});
        else if (per <= 3) body(std::integral_constant<uint32_t, 3>{});
        else if (!LONG || per <= 4) body(std::integral_constant<uint32_t, 4>{});
        else if constexpr (LONG) {
            if (per <= 6) body(std::integral_constant<uint32_t, 6>{});
            else body(std::integral_constant<uint32_t, 8>{});
        }
    } else if constexpr (LOG2S == 10) {
        if (per <= 6) body(std::integral_constant<uint32_t, 6>{});
        else body(std::integral_constant<uint32_t, 8>{});
    } else {
        if (per <= 10) body(std::integral_constant<uint32_t, 10>{});
        else if (per <= 12) body(std::integral_constant<uint32_t, 12>{});
        else body(std::integral_constant<uint32_t, kList / 64>{});
    }
    C = __builtin_amdgcn_readfirstlane(C);
    wave_lds_sync();
    if constexpr (LONG) {
        if (C > kList) {                                           // more distinct numbers than the table is made for (the neighbour lookups need free slots):
            if constexpr (DEFER) P.template finish<TAX>(lane, K, tab, ws, cands);   // the caller sends the list on (false)
            wave_lds_sync();
            return false;
        }
    }
    auto pick = [&](auto perc) {
        constexpr uint32_t PER = decltype(perc)::value;
        strong = gw_pick<LOG2S, PER, TAX>(ck, C, slots, lane, maxWin, K, taxkey, tab, wv, wh, wd);
    };
    if (C <= 64) pick(std::integral_constant<uint32_t, 1>{});
    else if (C <= 128) pick(std::integral_constant<uint32_t, 2>{});
    else if (C <= 256) pick(std::integral_constant<uint32_t, 4>{});
    else if constexpr (LOG2S >= 10) {
        if (C <= 512) pick(std::integral_constant<uint32_t, 8>{});
        else if constexpr (LOG2S >= 11) pick(std::integral_constant<uint32_t, 16>{});
    }
    strong = __builtin_amdgcn_readfirstlane(strong);
    if constexpr (DEFER) {
        P.template finish<TAX>(lane, K, tab, ws, cands);           // the previous read's candidates
        if (strong >= K) {                                         // this read's: later
            P.pend = true; P.q = q; P.wv = wv; P.wh = wh; P.wd = wd;
            P.dir = wv != kGwNone ? tab.gwDir[wv >> tab.gwDirShift] : 0u;
            wave_lds_sync();
            return true;
        }
    }
    {
        uint32_t wt = 0xFFFFFFFFu, wlo = 0, whi = 0;
        if (wv != kGwNone) tab.gw_target_bounds(wv, wt, wlo, whi);
        again = gw_winners_out<TAX>(lane, K, wv, wh, wd, wt, wlo, whi, out, pickLo, pickHi);
    }
    bool done = true;
    if (again) {                                                   // two winners of one target: the exact wave kernel
        if (lane == 0) { ws.hitScan[q] = ws.qstat[q].hits; ws.qflag[q] = kFlagCands; }
        done = false;
    } else if (strong < K) {
        const uint2 r6 = entf();
        const uint32_t fbase = r6.x, nent = r6.y;
        if (TAX || nent > kBigEnt) {
            // places left for single-hit taxa: the order among those depends on every target's taxon -> the exact wave kernel
            // (so do reads with more than 64 found features: the sweep below reads one entry per lane)
            if (lane == 0) { ws.hitScan[q] = ws.qstat[q].hits; ws.qflag[q] = kFlagCands; }
            done = false;
        } else if constexpr (!TAX) {
            // ---- D. the smallest numbers of targets that were not picked with >= 2 hits -- every such target's best range is a
            //      single location, and the first of them in (target, window) order are what the CPU's list keeps
            wave_lds_sync();
            const uint32_t sz = lane < nent ? (ws.psize[fbase + lane] & 0xFFFFu) : 0u;
            const uint64_t pay = lane < nent ? ws.ppay[fbase + lane] : 0ull;
            const uint32_t myR = sz > 1 ? (sz + 15u) >> 4 : 0u;
            const uint32_t incl = wave_incl_scan_u32(myR, lane), Rc = rdlane(incl, 63), start = incl - myR;
            // every lane keeps the kLaneK smallest numbers it sees (several may be one target's: the rounds below strike whole
            // targets, and a lane that had to drop numbers and is left with none cannot vouch for its minimum any more)
            uint32_t best[kLaneK];
#pragma unroll
            for (uint32_t i = 0; i < kLaneK; ++i) best[i] = kGwNone;
            uint32_t seen = 0;
            auto visit = [&](uint32_t g, bool valid) {
                if (!valid) return;
                bool skip = false;
#pragma unroll
                for (uint32_t i = 0; i < kLaneK; ++i) skip = skip || (i < strong && g - pickLo[i] < pickHi[i] - pickLo[i]);
                if (skip) return;
                ++seen;
                uint32_t c = g;                                     // sorted insert, the largest falls out
#pragma unroll
                for (uint32_t i = 0; i < kLaneK; ++i) { const uint32_t lo = min(best[i], c); c = max(best[i], c); best[i] = lo; }
            };
            visit(sz == 1 ? tab.gw_of(pay) : kGwNone, sz == 1);
            for (uint32_t r0 = 0; r0 < Rc; r0 += kGwRounds) {
                for (uint32_t i = lane; i < kGwRounds; i += 64) if (r0 + i >= Rc) T[i] = 0ull;
                const uint32_t jlo = r0 > start ? r0 - start : 0u, jhi = min(myR, r0 + kGwRounds > start ? r0 + kGwRounds - start : 0u);
                for (uint32_t j = jlo; j < jhi; ++j) T[start + j - r0] = (pay + 16ull * j) | ((uint64_t)min(16u, sz - 16u * j) << 40);
                wave_lds_sync();
#pragma unroll
                for (uint32_t u = 0; u < kGwLoads; ++u) {
                    const uint64_t rd = T[u * 16 + grp];
                    const int32_t rem = (int32_t)(uint32_t)(rd >> 40) - (int32_t)sub4;
                    if (rem > 0) {
                        const U4 t = *reinterpret_cast<const U4*>(tab.values32 + (rd & 0xFFFFFFFFFFull) + sub4);
                        visit(t.x, true); visit(t.y, rem > 1); visit(t.z, rem > 2); visit(t.w, rem > 3);
                    }
                }
                wave_lds_sync();
            }
            const bool dropped = seen > kLaneK;
            bool unsure = false;
            for (uint32_t rnd = strong; rnd < K; ++rnd) {
                if (__ballot(dropped && best[0] == kGwNone)) { unsure = true; break; }
                const uint32_t m = wave_min_u32(best[0]);
                mc_candidate_dev e; e.tgt = 0xFFFFFFFFu; e.hits = 0; e.beg = 0; e.end = 0;
                if (m != kGwNone) {
                    const uint32_t t = tab.gw_target(m), lo = tab.gwBase[t], hi = tab.gwBase[t + 1];
                    e.tgt = t; e.hits = 1; e.beg = e.end = m - lo;
                    // that target leaves every lane's list (its numbers are neighbours in the sorted list: compact the rest)
                    uint32_t kept[kLaneK];
#pragma unroll
                    for (uint32_t i = 0; i < kLaneK; ++i) kept[i] = kGwNone;
                    uint32_t n = 0;
#pragma unroll
                    for (uint32_t i = 0; i < kLaneK; ++i) {
                        const bool stay = best[i] != kGwNone && !(best[i] - lo < hi - lo);
#pragma unroll
                        for (uint32_t j = 0; j < kLaneK; ++j) if (stay && j == n) kept[j] = best[i];
                        n += stay ? 1u : 0u;
                    }
#pragma unroll
                    for (uint32_t i = 0; i < kLaneK; ++i) best[i] = kept[i];
                }
                if (lane == 0) out[rnd] = e;
            }
            if (unsure) {
                if (lane == 0) { ws.hitScan[q] = ws.qstat[q].hits; ws.qflag[q] = kFlagCands; }
                done = false;
            }
        }
    }
    if (done && lane == 0) ws.qflag[q] = kFlagDone;
    wave_lds_sync();
    return true;
}

}  // namespace

#ifndef MC_GW_COUNT_WPE
#define MC_GW_COUNT_WPE 6
#endif
template <uint32_t LOG2S, uint32_t WAVES, bool TAX>
__global__ __launch_bounds__(WAVES * 64, LOG2S == 9 ? MC_GW_COUNT_WPE : LOG2S == 10 ? 5 : 1) void gw_count_kernel(BatchView b, uint32_t s, DeviceTable tab, Workspace ws, uint32_t K,
                                                                                             const uint32_t* __restrict__ taxkey, mc_candidate_dev* __restrict__ cands,
                                                                                             uint32_t minN2)
{
    constexpr uint32_t kSlots = 1u << LOG2S, kList = kSlots / 2;
    static_assert(kGwRounds * 8 <= kSlots * 8, "step D's round table lives in the slot table, which is done with by then");
    __shared__ __attribute__((aligned(16))) uint2 slotS[WAVES][kSlots];
    __shared__ uint32_t ckS[WAVES][kList];                             // the slots of the list's distinct numbers (gw_count_numbers)
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: the wave's pointers and loop counters live in scalar registers)
    uint2* slots = slotS[wave];
    uint32_t* ck = ckS[wave];
    uint64_t* T = reinterpret_cast<uint64_t*>(slotS[wave]);
    // the records gw_filter_kernel left (list 7, one per read of work list 6); this instance takes the filtered lists that fit its
    // table: n2 in (minN2, kList], window ranges up to kHashWin
    const uint32_t total = ws.midCount[9];
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * b.n;
    const uint4* __restrict__ work6 = reinterpret_cast<const uint4*>(ws.midList) + (size_t)6 * b.n;
    const uint32_t nWaves = gridDim.x * WAVES;
    const uint32_t w0 = blockIdx.x * WAVES + wave;
    const uint32_t grp = lane >> 2, sub4 = (lane & 3u) * 4u;
    const uint32_t* __restrict__ pool = reinterpret_cast<const uint32_t*>(ws.bigPool);
    // The records are taken 64 at a time: one coalesced load, a ballot of the ones that are this instance's, their fields by
    // v_readlane -- an instance that takes a few hundred of 5 x 10^6 records spent a millisecond on one dependent load per record.
    // The smallest instance (most reads) fetches the NEXT list while it works on this one.
    constexpr uint32_t kPre = LOG2S <= 10 ? kList / 64 : 1;
    uint32_t pre[kPre];
    auto fetch = [&](uint32_t off, uint32_t n) {
        if constexpr (LOG2S <= 10) {
#pragma unroll
            for (uint32_t r = 0; r < kPre; ++r) pre[r] = r * 64 + lane < n ? pool[off + r * 64 + lane] : kGwNone;
        }
    };
    // the first instance (most reads) takes ALL records, 64 per step; the others get theirs from the compact lists gw_compact_kernel
    // made (ws.sideList [1] / [2]), 8 per step: with few of them every wave should have some.  The first two instances defer the
    // winners' target lookup behind the next read's counting (GwPend).
    constexpr bool kDefer = LOG2S <= 10;
    GwPend P;
    const uint32_t nmine = LOG2S == 9 ? total : ws.midCount[LOG2S == 10 ? 14 : 15];
    // (few records -- the small batches of the host slots --: fewer per step, so that every wave of the grid has one before any wave has two;
    // eight reads of ~25 us each one after the other on five waves were the longest kernel of a 4 096-read batch)
    const uint32_t kStep = LOG2S == 9 ? 64u : min(8u, max(1u, (nmine + nWaves - 1) / nWaves));
    const uint32_t* __restrict__ side = ws.sideList + (size_t)(LOG2S == 10 ? 1 : 2) * b.n;
    for (uint32_t chunk = w0 * kStep; chunk < nmine; chunk += nWaves * kStep) {
      const bool inb = lane < kStep && chunk + lane < nmine;
      const uint32_t myW = LOG2S == 9 ? chunk + lane : (inb ? side[chunk + lane] : 0u);
      const uint4 myRec = inb ? work[myW] : make_uint4(0, 0, kGwFallback, 0);
      // n2 in (minN2, kList], window ranges up to kHashWin (an EMPTY filtered list is the first instance's: step D fills the places)
      uint64_t todo = __ballot(myRec.z <= kList && (minN2 == 0 || myRec.z > minN2) && myRec.w <= kHashWin);
      if (todo) { const uint32_t j = (uint32_t)__ffsll((unsigned long long)todo) - 1; fetch(rdlane(myRec.y, j), rdlane(myRec.z, j)); }
      while (todo) {
        const uint32_t j = (uint32_t)__ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t w = rdlane(myW, j), q = rdlane(myRec.x, j), n2 = rdlane(myRec.z, j), maxWin = rdlane(myRec.w, j);
        const uint32_t* __restrict__ src = pool + rdlane(myRec.y, j);
        uint32_t cur[kPre];
#pragma unroll
        for (uint32_t r = 0; r < kPre; ++r) cur[r] = pre[r];
        if (todo) { const uint32_t jn = (uint32_t)__ffsll((unsigned long long)todo) - 1; fetch(rdlane(myRec.y, jn), rdlane(myRec.z, jn)); }
        gw_count_read<LOG2S, TAX, kDefer>(q, [&]() -> uint2 { const uint4 r6 = work6[w]; return make_uint2(r6.y, r6.z & 0xFFFu); }, n2, maxWin, [&](uint32_t r) -> uint32_t {
            if constexpr (LOG2S <= 10) return cur[r < kPre ? r : 0];
            else return r * 64 + lane < n2 ? src[r * 64 + lane] : kGwNone;
        }, slots, ck, T, lane, grp, sub4, K, taxkey, tab, ws, cands, P);
      }
    }
    if constexpr (kDefer) { P.bases(tab); P.template finish<TAX>(lane, K, tab, ws, cands); }
}

constexpr uint32_t kGwCounted = 0x80000000u;      // record of list 7: the read was counted inside the filter kernel (| kept numbers)

// FUSED filter + counting (the common case of a 150 bp read at RefSeq scale in ONE kernel): gw_filter_kernel's two phases on the read's
// lists in registers, the kept numbers to LDS instead of the pool (up to 512), gw_count_read on them right there -- no round trip of
// the kept numbers through HBM (4.3 GB written and read back per 5 x 10^6 reads), one kernel's launch and tail less.  The slot table of
// the counting takes the place of the filter bits (4 KB, done with after phase B), the distinct numbers' slots that of the round table.
// Lists that keep more than 512 numbers or have window ranges beyond kHashWin repeat phase B into the pool (the numbers are still in
// registers) and go on to the other kernels as from gw_filter_kernel.  A record whose read was counted here is marked kGwCounted | n2.
// 92 registers: five waves per SIMD -- until round 6: with 384 instead of 512 kept numbers in LDS the block takes 26 KB, six fit a CU, and the
// compiler, which sizes the register budget by the occupancy the LDS allows, brings the kernel to 80 registers (three spilled): SIX waves
// per SIMD, 13.3 -> 12.0 ms per 5 x 10^6 reads (WPE = 6; the kernel waits for its LDS round trips and its lists 45 % of its wave cycles:
// a sixth wave fills them).  (A software-pipelined form -- the next read's loads issued as phase B frees the registers -- needed
// 127 registers = four waves per SIMD and was slower, 14.95 against 13.1 ms per 5 x 10^6 reads; removed in round 5, docs/LAB_NOTEBOOK_r04.md.)
template <uint32_t WAVES, uint32_t TLOG2, bool TAX, uint32_t WPE = MC_GW_FILTER_WPE>
__global__ __launch_bounds__(WAVES * 64, WPE) void gw_filter_count_kernel(BatchView b, DeviceTable tab, Workspace ws, uint32_t K, const uint32_t* __restrict__ taxkey,
                                                                                 mc_candidate_dev* __restrict__ cands)
{
    using Bloom = GwBloom<TLOG2, TLOG2>;
    constexpr uint32_t kKeep = WPE >= 6 ? 384 : 512;               // numbers kept in LDS: the counting takes them when at most 256 are distinct
                                                                   // (WPE >= 6: 384 -- 26 KB of LDS per block, six blocks per CU, 80 registers: six waves per SIMD, "gw_fuse" 6)
    static_assert(kGwRounds * 8 >= 256 * 4, "the distinct numbers' slots take the place of the round table");
    // WPE = 7 (the default): 22 KB per block -- the round table lies in the filter bits' place until the loads are issued (phase B
    // needs three bits per load of it: gw_pack_rems; the bits are cleared behind the loads), the distinct numbers' slots of the counting in the
    // kept numbers' (which are in registers by then): seven blocks per CU
    constexpr bool kSeven = WPE >= 7;
    __shared__ __attribute__((aligned(16))) uint32_t bitS[WAVES][Bloom::kWords];
    __shared__ __attribute__((aligned(16))) uint64_t roundS[kSeven ? 1 : WAVES][kSeven ? 1 : kGwRounds];
    __shared__ __attribute__((aligned(16))) uint32_t keptS[WAVES][kKeep];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* bits = bitS[wave];
    uint64_t* T = kSeven ? reinterpret_cast<uint64_t*>(bits) : roundS[kSeven ? 0 : wave];
    uint32_t* kept = keptS[wave];
    const uint32_t total = ws.midCount[9];
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)6 * b.n;
    uint4* __restrict__ outRec = reinterpret_cast<uint4*>(ws.midList) + (size_t)7 * b.n;
    const uint32_t nWaves = gridDim.x * WAVES;
    const uint32_t w0 = blockIdx.x * WAVES + wave;
    auto load_rec = [&](uint32_t w) -> uint4 { return w < total ? work[w] : make_uint4(0, 0, 0, 0); };
    const uint64_t sliceCap = ws.bigPoolCap / nWaves;
    uint32_t* const slice = reinterpret_cast<uint32_t*>(ws.bigPool) + (uint64_t)w0 * sliceCap;
    uint64_t sliceUsed = 0;
    uint32_t deferred = 0;
    uint4 rec = load_rec(w0), recNext = load_rec(w0 + nWaves);
    uint32_t esz = 0; uint64_t epay = 0;
    auto load_entries = [&](const uint4& r) {
        const uint32_t ne = (r.z >> 12) <= kGwSmallH ? min(r.z & 0xFFFu, 64u) : 0u;
        esz = lane < ne ? ws.psize[r.y + lane] : 0u;
        epay = lane < ne ? ws.ppay[r.y + lane] : 0ull;
    };
    load_entries(rec);
    const uint32_t grp = lane >> 2, sub4 = (lane & 3u) * 4u;
    GwPend P;
    for (uint32_t w = w0; w < total; w += nWaves) {
        const uint32_t q = rec.x, nent = rec.z & 0xFFFu, H = rec.z >> 12, maxWin = rec.w;
        const uint32_t sz = esz & 0xFFFFu; const uint64_t pay = epay;
        rec = recNext;
        recNext = load_rec(w + 2 * nWaves);
        load_entries(rec);
        const uint32_t myR = sz > 1 ? (sz + 15u) >> 4 : 0u;
        const uint32_t incl = wave_incl_scan_u32(myR, lane), Rc = rdlane(incl, 63);
        if (H > kGwSmallH || nent > 64u || Rc > kGwRounds || maxWin > tab.gwGap || sliceCap - sliceUsed < kGwRounds * 16u + 64u) {
            if (lane == 0) outRec[w] = make_uint4(q, 0u, kGwDefer, maxWin);
            ++deferred;
            continue;
        }
        auto clear_bits = [&]() {
            uint4* z4 = reinterpret_cast<uint4*>(bits);
#pragma unroll
            for (uint32_t i = 0; i < Bloom::kWords / 4 / 64; ++i) z4[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        };
        if constexpr (!kSeven) clear_bits();
        gw_fill_rounds_scan(T, kept, lane, Rc, incl - myR, myR, sz, pay);   // (kept: free until phase B)
        wave_lds_sync();
        const GwFrame F(maxWin);
        const uint32_t nl = (Rc + 15u) >> 4;
        uint4 x[kGwLoads];
        gw_load_rounds(T, tab.values32, grp, sub4, x, nl);
        uint32_t rems = 0;
        if constexpr (kSeven) {                                    // the table has been read: its place is the filter's now
            rems = gw_pack_rems(T, grp, sub4, nl);
            wave_lds_sync();
            clear_bits();
            wave_lds_sync();
        }
        auto take_rounds = [&](auto check, GwSink& S) {
            constexpr bool CHECK = decltype(check)::value;
            if constexpr (kSeven) gw_take_rounds_packed<Bloom, CHECK>(bits, rems, F, S, x, nl);
            else gw_take_rounds<Bloom, CHECK>(bits, T, F, S, grp, sub4, x, nl);
        };
        const uint32_t sv = sz == 1 ? tab.gw_of(pay) : kGwNone;
        if (sv != kGwNone) Bloom::mark(bits, sv >> F.A);
        gw_mark_rounds<Bloom>(bits, x, F.A, nl);
        wave_lds_sync();
        const bool here = maxWin <= kHashWin;
        uint32_t n2;
        if (here) {
            GwSink S{kept, kKeep, 0u};
            gw_take<Bloom>(bits, F, S, sv, sv != kGwNone);
            take_rounds(std::true_type{}, S);
            n2 = S.n2;
        } else {
            GwSink S{slice + sliceUsed, (uint32_t)min((uint64_t)kGwMaxKept, sliceCap - sliceUsed), 0u};
            gw_take<Bloom>(bits, F, S, sv, sv != kGwNone);
            take_rounds(std::false_type{}, S);
            n2 = S.n2;
        }
        if (here && n2 <= kKeep) {
            if (lane == 0) outRec[w] = make_uint4(q, 0u, kGwCounted | n2, maxWin);
            wave_lds_sync();
            const bool counted = gw_count_read<9, TAX, true, true>(q, [&]() -> uint2 { const uint4 r6 = work[w]; return make_uint2(r6.y, r6.z & 0xFFFu); }, n2, maxWin,
                                        [&](uint32_t r) -> uint32_t { return r * 64 + lane < n2 ? kept[r * 64 + lane] : kGwNone; },
                                        reinterpret_cast<uint2*>(bits), kSeven ? kept : reinterpret_cast<uint32_t*>(T), kSeven ? reinterpret_cast<uint64_t*>(kept) : T,
                                        lane, grp, sub4, K, taxkey, tab, ws, cands, P);
            if (!counted && kSeven) {
                // (the kept numbers' place went to the counting: the read is filtered again by gw_filter2_kernel, 1 read in 60)
                if (lane == 0) outRec[w] = make_uint4(q, 0u, kGwDefer, maxWin);
                ++deferred;
                wave_lds_sync();
            } else if (!counted) {
                // more than 256 DISTINCT numbers among the kept ones (1 read in 60 at RefSeq scale): the list -- still in LDS -- goes through the
                // pool to gw_count_kernel<10> like the lists of 257 .. 512 numbers round 3's filter left (the exact wave kernel took them: 0.5 ms
                // per 5 x 10^6 reads).  (Room: the slice holds kGwRounds x 16 + 64 numbers more, checked above.)
#pragma unroll
                for (uint32_t r = 0; r < kKeep / 64; ++r) if (r * 64 + lane < n2) slice[sliceUsed + r * 64 + lane] = kept[r * 64 + lane];
                if (lane == 0) outRec[w] = make_uint4(q, (uint32_t)((uint64_t)w0 * sliceCap + sliceUsed), n2, maxWin);
                sliceUsed += n2;
                wave_lds_sync();
            }
            continue;
        }
        if (here) {
            GwSink S{slice + sliceUsed, (uint32_t)min((uint64_t)kGwMaxKept, sliceCap - sliceUsed), 0u};
            gw_take<Bloom>(bits, F, S, sv, sv != kGwNone);
            take_rounds(std::false_type{}, S);
            n2 = S.n2;
        }
        const uint32_t room = (uint32_t)min((uint64_t)kGwMaxKept, sliceCap - sliceUsed);
        const bool fallback = n2 > room;
        if (lane == 0) {
            if (fallback) { ws.hitScan[q] = H; ws.qflag[q] = kFlagCands; outRec[w] = make_uint4(q, 0u, kGwFallback, maxWin); }
            else outRec[w] = make_uint4(q, (uint32_t)((uint64_t)w0 * sliceCap + sliceUsed), n2, maxWin);
        }
        if (!fallback) sliceUsed += n2;
        wave_lds_sync();
    }
    P.bases(tab); P.template finish<TAX>(lane, K, tab, ws, cands);
    if (lane == 0) {
        if (ws.sliceFill) ws.sliceFill[w0] = (uint32_t)sliceUsed;
        if (deferred) atomicAdd(&ws.midCount[10], deferred);
    }
}

// ================================================================================================
// gw_sorted_cands_kernel: rows 9-10 on a SORTED filtered list (gw_sort.hip) -- long reads (thousands of kept locations, window ranges
// of tens to hundreds), pairs with large insert sizes.  One wave per read; the list is taken 64 numbers at a time, one per lane, each
// finding the begin of the CPU's sliding window that ends in it (candidate_generation.hpp:47-108: numbers less than maxWindowsInRange
// apart are one target's, that is what the gap between two targets' numbers is for), one candidate per target through the CPU's
// top-list insert (top_insert: ties, taxon merging) on the lane that holds the target's last number.  Then K rounds over the lanes' lists: the
// best under (hits desc, target asc, end window asc), its target (taxon) struck everywhere -- every lane loses at most one entry per
// round, so its K entries are enough.  Reads with fewer than K candidates of two or more hits go to the exact wave kernel (single
// hits of other targets were filtered away).
// ================================================================================================
constexpr uint32_t kGwBigSorted = 8192;                      // sorted lists longer than this are scanned by a whole block (gw_sorted_cands_kernel<TAX, true>)

// The scan of positions [first, end) of a read's sorted list g (n numbers) by ONE wave: see gw_sorted_cands_kernel.  Leaves the
// candidates in the lanes' top lists.
template <bool TAX>
__device__ __forceinline__ void gw_sorted_scan(const uint32_t* __restrict__ g, const uint32_t first, const uint32_t end, const uint32_t D,
                                               const DeviceTable& tab, const uint32_t* __restrict__ taxkey, const uint32_t K, uint32_t* ring,
                                               const uint32_t lane, LaneCand (&top)[kLaneK], uint32_t (&toptax)[kLaneK])
{
    // The list is taken 64 consecutive numbers at a time, one per lane (coalesced; a lane scanning its own contiguous piece made
    // every load of the wave 64 separate cache-line requests: 300 us of wave time per 2 kbp read).  Element i's window range begins
    // at fst(i) = the first position whose number is >= g[i] - D (numbers less than maxWindowsInRange apart are one target's: the
    // gap); fst is monotone.  The last two chunks lie in an LDS ring: a range of up to 64 elements -- nearly all of them -- is
    // found by a binary search there, wider ones (and what lies before `first`) in the list itself.  hits = i - fst + 1.  Per target
    // the best range = most hits, the first to reach them: a segmented max-scan over the lanes (segments = targets, contiguous in the
    // sorted list) of hits << 6 | (63 - lane); the open target at a chunk's end is carried into the next chunk (its best so far wins
    // ties: it came first; a chunk that lies inside it needs no target lookup).  A finished target's candidate enters the top list of
    // the lane that holds its last element (per-lane lists in target order, merged by the K rounds).
    uint32_t cT = 0xFFFFFFFFu, cHits = 0, cBeg = 0, cEnd = 0, cLo = 0, cHi = 0;   // the open target, its best range so far, its numbers (wave-uniform)
    uint32_t lowFst = 0;
    uint32_t gnext = first + lane < end ? g[first + lane] : 0xFFFFFFFFu;      // (the next chunk's numbers are on their way while this one is scanned,
    uint32_t dnext = gnext != 0xFFFFFFFFu ? tab.gwDir[gnext >> tab.gwDirShift] : 0u;   //  their directory entries behind them)
    for (uint32_t base = first; base < end; base += 64) {
        const uint32_t cnt = min(64u, end - base), ei = base + lane;
        const bool valid = lane < cnt;
        const uint32_t gi = gnext, di = dnext;
        gnext = ei + 64u < end ? g[ei + 64u] : 0xFFFFFFFFu;
        ring[ei & 127u] = gi;
        wave_lds_sync();
        const uint32_t want = gi - D;
        const uint32_t rlo = base >= first + 64u ? base - 64u : first;      // positions from here on are in the ring
        uint32_t lo = 0, hi = 0;
        bool glob = false;
        if (valid) {
            if (lowFst >= rlo) { lo = lowFst; hi = ei; }
            else if (ring[rlo & 127u] < want) { lo = rlo + 1u; hi = ei; }
            else { lo = lowFst; hi = rlo; glob = true; }
        }
        while (__ballot(!glob && lo < hi)) {
            const uint32_t mid = (lo + hi) >> 1;
            if (!glob && lo < hi) { if (ring[mid & 127u] < want) lo = mid + 1u; else hi = mid; }
        }
        while (__ballot(glob && lo < hi)) {
            const uint32_t mid = (lo + hi) >> 1;
            if (glob && lo < hi) { if (g[mid] < want) lo = mid + 1u; else hi = mid; }
        }
        const uint32_t fst = lo;
        const uint32_t gf = !valid ? 0u : fst >= rlo ? ring[fst & 127u] : g[fst];
        lowFst = rdlane(fst, cnt - 1u);
        uint32_t t = 0xFFFFFFFFu, tlo = 0, thi = 0;
        const bool inCur = cT != 0xFFFFFFFFu && (gi - cLo) < (cHi - cLo);
        if (valid) {
            if (inCur) { t = cT; tlo = cLo; thi = cHi; }
            else {                                                 // (DeviceTable::gw_target_bounds with the directory entry at hand)
                t = di;
                const uint32_t b0 = tab.gwBase[t], b1 = tab.gwBase[t + 1], b2 = tab.gwBase[min(t + 2, tab.gwTargets)];
                tlo = b0; thi = b1;
                if (gi >= b1) { ++t; tlo = b1; thi = b2; }
                while (gi >= thi) { ++t; tlo = thi; thi = tab.gwBase[t + 1]; }
            }
        }
        // the open target ended with the previous chunk: its candidate is due (lane 0: before anything of this chunk)
        if (cT != 0xFFFFFFFFu && rdlane(t, 0) != cT && lane == 0) {
            LaneCand c; c.tgt = cT; c.hits = cHits; c.beg = cBeg; c.end = cEnd;
            top_insert(top, toptax, c, K, TAX ? taxkey : nullptr, 0xFFFFFFFFu);
        }
        uint32_t val = valid ? (((ei - fst + 1u) << 6) | (63u - lane)) : 0u;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t ov = (uint32_t)__shfl_up((int)val, d), ot = (uint32_t)__shfl_up((int)t, d);
            if (lane >= d && ot == t) val = max(val, ov);
        }
        const uint32_t tnext = (uint32_t)__shfl_down((int)t, 1);
        const bool tail = valid && (lane + 1u >= cnt || tnext != t);
        const uint32_t wl = 63u - (val & 63u);
        LaneCand c; c.tgt = t; c.hits = val >> 6;
        c.beg = (uint32_t)__shfl((int)gf, (int)wl) - tlo; c.end = (uint32_t)__shfl((int)gi, (int)wl) - tlo;
        if (t == cT && cHits >= c.hits) { c.hits = cHits; c.beg = cBeg; c.end = cEnd; }
        const bool last = base + 64u >= end;
        if (tail && (last || lane + 1u < cnt)) top_insert(top, toptax, c, K, TAX ? taxkey : nullptr, 0xFFFFFFFFu);
        cT = rdlane(c.tgt, cnt - 1u); cHits = rdlane(c.hits, cnt - 1u); cBeg = rdlane(c.beg, cnt - 1u); cEnd = rdlane(c.end, cnt - 1u);
        cLo = rdlane(tlo, cnt - 1u); cHi = rdlane(thi, cnt - 1u);
        dnext = gnext != 0xFFFFFFFFu ? tab.gwDir[gnext >> tab.gwDirShift] : 0u;
    }
    wave_lds_sync();
}

// K rounds over the lanes' lists (each sorted: entry 0 is the lane's best): the best under (hits desc, target asc, end window asc),
// its target (taxon) struck everywhere.  put(round, candidate, its taxon) is called by every lane with the round's wave-uniform pick.
// Returns how many picks had two or more hits.
template <bool TAX, class Put>
__device__ __forceinline__ uint32_t gw_sorted_rounds(LaneCand (&top)[kLaneK], uint32_t (&toptax)[kLaneK], const uint32_t K, Put&& put)
{
    uint32_t strong = 0;
    for (uint32_t rnd = 0; rnd < K; ++rnd) {
        const uint32_t mh = wave_max_u32(top[0].hits);
        mc_candidate_dev ev; ev.tgt = 0xFFFFFFFFu; ev.hits = 0; ev.beg = 0; ev.end = 0;
        uint32_t mtax = 0;
        if (mh != 0) {
            const uint32_t mt = wave_min_u32(top[0].hits == mh ? top[0].tgt : 0xFFFFFFFFu);
            const uint32_t me = wave_min_u32(top[0].hits == mh && top[0].tgt == mt ? top[0].end : 0xFFFFFFFFu);
            const uint32_t winner = __ffsll((unsigned long long)__ballot(top[0].hits == mh && top[0].tgt == mt && top[0].end == me)) - 1;
            ev.tgt = mt; ev.hits = mh; ev.end = me; ev.beg = rdlane(top[0].beg, winner);
            mtax = rdlane(toptax[0], winner);
            strong += mh >= 2 ? 1u : 0u;
            // the picked target (taxon) leaves every lane's list
            LaneCand kept[kLaneK]; uint32_t ktax[kLaneK];
#pragma unroll
            for (uint32_t i = 0; i < kLaneK; ++i) { kept[i].tgt = 0xFFFFFFFFu; kept[i].hits = 0; kept[i].beg = 0; kept[i].end = 0; ktax[i] = 0; }
            uint32_t nk = 0;
#pragma unroll
            for (uint32_t i = 0; i < kLaneK; ++i) {
                const bool stay = top[i].hits != 0 && (TAX ? toptax[i] != mtax : top[i].tgt != mt);
#pragma unroll
                for (uint32_t jj = 0; jj < kLaneK; ++jj) if (stay && jj == nk) { kept[jj] = top[i]; ktax[jj] = toptax[i]; }
                nk += stay ? 1u : 0u;
            }
#pragma unroll
            for (uint32_t i = 0; i < kLaneK; ++i) { top[i] = kept[i]; toptax[i] = ktax[i]; }
        }
        put(rnd, ev, mtax);
    }
    return strong;
}

// BIG = false: one wave per read.  The work list is in descending order of the lists' lengths (launch_gw_order): the longest lists beyond
// kGwBigSorted numbers -- up to kGwFewBig of them: the rare 10-19 kbp reads of a mixed batch keep 10^5 numbers at RefSeq scale, and the one
// wave on such a list kept running long after all others had finished -- are left to the second launch (their count: midCount[18]).
// The rest keeps one wave per read: sixteen waves per list cost more than they gain when every CU has work anyway (20 000 reads of
// 10 kbp: 3.9 ms with one wave each, 6.5 ms with sixteen).
// BIG = true (second launch): one BLOCK of sixteen waves per such read, its list cut into sixteen runs of whole chunks, one per wave.
// A target that spans two runs leaves a candidate in each: the K rounds take the one with more hits (equal: the earlier one, it ends in
// the smaller window) and strike the other -- as they always did between the lanes of one wave.  The waves' K picks meet in LDS, wave 0
// picks the K best of those 16 K.
constexpr uint32_t kGwFewBig = 1024;
template <bool TAX, bool BIG>
__global__ __launch_bounds__(BIG ? 1024 : 256) void gw_sorted_cands_kernel(BatchView b, DeviceTable tab, Workspace ws, uint32_t K, const uint32_t* __restrict__ taxkey,
                                                                           mc_candidate_dev* __restrict__ cands)
{
    constexpr uint32_t kWaves = BIG ? 16u : 4u;                // BIG: sixteen waves share one list (16 x K picks fit one wave's lanes, K <= 4)
    __shared__ uint32_t ringS[kWaves][128];                    // the list's last two chunks of 64 numbers (position & 127)
    __shared__ mc_candidate_dev pickS[kWaves][kLaneK];
    __shared__ uint32_t ptaxS[kWaves][kLaneK];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* ring = ringS[wave];
    const uint4* __restrict__ work = reinterpret_cast<const uint4*>(ws.midList) + (size_t)7 * b.n;
    const uint32_t* __restrict__ side = ws.sideList + (size_t)3 * b.n;
    auto finish = [&](uint32_t q, uint32_t strong) {
        if (lane == 0) {
            if (strong < K) { ws.hitScan[q] = ws.qstat[q].hits; ws.qflag[q] = kFlagCands; }
            else ws.qflag[q] = kFlagDone;
        }
    };
    if constexpr (!BIG) {
        const uint32_t nWaves = gridDim.x * 4, w0 = blockIdx.x * 4 + wave;
        const uint32_t nmine = ws.midCount[13];
        // how many lists are longer than kGwBigSorted (descending order: a binary search, the same in every wave)
        uint32_t lo = 0, hi = nmine;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (work[side[mid]].z > kGwBigSorted) lo = mid + 1u; else hi = mid; }
        const uint32_t nbig = min(lo, kGwFewBig);                  // the longest of them
        if (w0 == 0 && lane == 0) ws.midCount[18] = nbig;
        for (uint32_t i = nbig + w0; i < nmine; i += nWaves) {
            const uint32_t w = side[i];
            const uint4 rec = work[w];
            const uint32_t q = rec.x, n = rec.z, maxWin = rec.w;
            LaneCand top[kLaneK];
            uint32_t toptax[kLaneK];
#pragma unroll
            for (uint32_t j = 0; j < kLaneK; ++j) { top[j].tgt = 0xFFFFFFFFu; top[j].hits = 0; top[j].beg = 0; top[j].end = 0; toptax[j] = 0; }
            gw_sorted_scan<TAX>(ws.bigPool2 + rec.y, 0u, n, maxWin - 1u, tab, taxkey, K, ring, lane, top, toptax);
            mc_candidate_dev* out = cands + (size_t)q * K;
            const uint32_t strong = gw_sorted_rounds<TAX>(top, toptax, K, [&](uint32_t rnd, const mc_candidate_dev& ev, uint32_t) { if (lane == 0) out[rnd] = ev; });
            finish(q, strong);
        }
    } else {
        const uint32_t nbig = ws.midCount[18];
        for (uint32_t i = blockIdx.x; i < nbig; i += gridDim.x) {
            const uint4 rec = work[side[i]];
            const uint32_t q = rec.x, n = rec.z, maxWin = rec.w;
            const uint32_t chunks = (n + 63u) / 64u, cpw = (chunks + kWaves - 1u) / kWaves;
            const uint32_t first = min(n, wave * cpw * 64u), end = min(n, (wave + 1u) * cpw * 64u);
            LaneCand top[kLaneK];
            uint32_t toptax[kLaneK];
#pragma unroll
            for (uint32_t j = 0; j < kLaneK; ++j) { top[j].tgt = 0xFFFFFFFFu; top[j].hits = 0; top[j].beg = 0; top[j].end = 0; toptax[j] = 0; }
            if (first < end) gw_sorted_scan<TAX>(ws.bigPool2 + rec.y, first, end, maxWin - 1u, tab, taxkey, K, ring, lane, top, toptax);
            gw_sorted_rounds<TAX>(top, toptax, K, [&](uint32_t rnd, const mc_candidate_dev& ev, uint32_t tax) {
                if (lane == 0) { pickS[wave][rnd] = ev; ptaxS[wave][rnd] = tax; }
            });
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (uint32_t j = 0; j < kLaneK; ++j) { top[j].tgt = 0xFFFFFFFFu; top[j].hits = 0; top[j].beg = 0; top[j].end = 0; toptax[j] = 0; }
                if (lane < kWaves * K) {
                    const mc_candidate_dev e = pickS[lane / K][lane % K];
                    top[0].tgt = e.tgt; top[0].hits = e.hits; top[0].beg = e.beg; top[0].end = e.end;
                    toptax[0] = ptaxS[lane / K][lane % K];
                }
                mc_candidate_dev* out = cands + (size_t)q * K;
                const uint32_t strong = gw_sorted_rounds<TAX>(top, toptax, K, [&](uint32_t rnd, const mc_candidate_dev& ev, uint32_t) { if (lane == 0) out[rnd] = ev; });
                finish(q, strong);
            }
            __syncthreads();
        }
    }
}


static uint32_t gw_env(const char* name, uint32_t dflt)
{
    const char* e = std::getenv(name);
    return e ? (uint32_t)std::max(1, std::atoi(e)) : dflt;
}

void launch_gw_cands(uint32_t stage, const BatchView& b, const SketchParams& sp, const DeviceTable& tab, const Workspace& ws, uint32_t maxCand,
                     const uint32_t* taxkey, void* cands, hipStream_t st)
{
    if (b.n == 0) return;
    mc_candidate_dev* c = (mc_candidate_dev*)cands;
    const uint32_t fgrid = big_filter_grid(b.n, true, ws.filterBpc);               // blocks of 4 waves: the pool is cut into one slice per wave
    if (stage == 0) {
        // the filter with the counting of lists up to 512 numbers fused in (gw_filter_count_kernel); "gw_fuse" 0: the two kernels apart
        if (ws.gwFuse == 0) hipLaunchKernelGGL((gw_filter_kernel<4, 14, 5>), dim3(fgrid), dim3(256), 0, st, b, tab, ws);   // (compiled for five waves per SIMD: 96 registers)
        // (round 6: occupancy is what this kernel answers to -- the compiler sizes its registers by what the LDS allows, so the LDS was cut:
        // SIX waves per SIMD with 384 instead of 512 kept numbers in LDS (26 KB per block, 80 registers, three spilled): 13.3 -> 12.0 ms per
        // 5 x 10^6 reads; SEVEN with the round table in the filter bits' place until the loads are out and the counting's distinct slots in the
        // kept numbers' (22 KB, 72 registers, five spilled): 11.65 ms, the step 17.9 -> 16.4 ms.  "gw_fuse" 6 / 5: the six- / five-wave instances)
        else if (ws.gwFuse == 6 && taxkey) hipLaunchKernelGGL((gw_filter_count_kernel<4, 14, true, 6>), dim3(fgrid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, c);
        else if (ws.gwFuse == 6) hipLaunchKernelGGL((gw_filter_count_kernel<4, 14, false, 6>), dim3(fgrid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, c);
        else if (ws.gwFuse == 5 && taxkey) hipLaunchKernelGGL((gw_filter_count_kernel<4, 14, true, 4>), dim3(fgrid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, c);
        else if (ws.gwFuse == 5) hipLaunchKernelGGL((gw_filter_count_kernel<4, 14, false, 4>), dim3(fgrid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, c);
        else if (taxkey) hipLaunchKernelGGL((gw_filter_count_kernel<4, 14, true, 7>), dim3(fgrid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, c);
        else hipLaunchKernelGGL((gw_filter_count_kernel<4, 14, false, 7>), dim3(fgrid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, c);
    } else if (stage == 3) {
        // reads of up to 2 x kGwRounds rounds (read pairs): two register batches
        // ("gw_fuse" 5 also brings back the pair filter of rounds 3-5: both filter halves of 2^15 bits, 38 KB per block, four waves per SIMD)
        if (ws.gwFuse == 5) hipLaunchKernelGGL((gw_filter2_kernel<4, 15>), dim3(fgrid), dim3(256), 0, st, b, tab, ws);
        else hipLaunchKernelGGL((gw_filter2_kernel<4, 15, 13>), dim3(fgrid), dim3(256), 0, st, b, tab, ws);
    } else if (stage == 7) {
        // the records the register filters left -> the stream filter's list, longest reads first
        const uint32_t cgrid = std::min<uint32_t>((b.n + 255) / 256, 2048u);
        hipLaunchKernelGGL(gw_compact_kernel, dim3(cgrid), dim3(256), 0, st, ws, b.n, 0u);
        if (ws.orderScratch) { size_t tb = ws.orderTemp; (void)launch_gw_order(0, ws, b.n, b.n, ws.orderScratch, tb, st); }
    } else if (stage == 8) {
        // reads with more than kGwSmallH locations: the ones beyond kGwBigH first (one block of sixteen waves per read, 2^19 + 2^17 filter bits:
        // blocks x 16 = the same waves, the same pool slices; tuning switch "gw_big_h"; 0xFFFFFFFF: one instance for all reads, as round 3)
        const uint32_t nSlices = 4u * fgrid;
        if (ws.gwBigH != 0xFFFFFFFFu) hipLaunchKernelGGL((gw_filter_stream_kernel<16, 19, 17, true>), dim3((nSlices + 15u) / 16u), dim3(1024), 0, st, b, tab, ws, ws.gwBigH, 0xFFFFFFFFu, nSlices);
    } else if (stage == 11) {
        const uint32_t nSlices = 4u * fgrid;
        const uint32_t midH = std::min(ws.gwMidH, ws.gwBigH);
        if (midH) hipLaunchKernelGGL((gw_filter_stream_kernel<2, 16, 13>), dim3(2 * fgrid), dim3(128), 0, st, b, tab, ws, 0u, midH, nSlices);
    } else if (stage == 9) {
        // ... the others: 2^17 + 2^15 filter bits per block of two waves (20 KB), twice the blocks
        // (2^16 + 2^15 bits instead: more waves per CU, but more false positives to sort -- 612 against 676 Mreads/min on configs[4]'s reads;
        // four waves per block and pair of filters -- 24 waves per CU: 5.01 -> 4.87 ms per 250 000 long reads: left at two; a single-pass
        // instance in front of this kernel was measured slower: DESIGN section 10 of round 4)
        const uint32_t nSlices = 4u * fgrid;
        const uint32_t cgrid = std::min<uint32_t>((b.n + 255) / 256, 2048u);
        // (stage 11, "gw_mid_h" > 0: the reads of up to that many locations through an instance with 2^16 + 2^13 filter bits -- 9 KB instead of
        // 20 per block, six waves per SIMD where this instance runs at three.  Measured at 8 192: filters 3.35 -> 3.0 ms per 250 000 long reads,
        // and the sort, the scan and the counting of what the smaller filters keep too much +0.35: off by default)
        const uint32_t midH = std::min(ws.gwMidH, ws.gwBigH);      // (stage 11, before this one)
        hipLaunchKernelGGL((gw_filter_stream_kernel<2, 17, 15>), dim3(2 * fgrid), dim3(128), 0, st, b, tab, ws, midH, ws.gwBigH, nSlices);
        hipLaunchKernelGGL(gw_compact_kernel, dim3(cgrid), dim3(256), 0, st, ws, b.n, 1u);
    } else if (stage == 1 || stage == 10) {
        // filtered lists up to 256 (stage 1: 4 KB of LDS per wave), then 257 .. 512 (stage 10)
        // (eight blocks per CU are resident; a grid of exactly that many left the waves with 9 or 10 steps of 64 records each and the CU waiting
        // for the last one: 5.9 ms per 5 x 10^6 reads; 16 / 24 / 32 blocks per CU: 5.23 / 5.19 / 5.15)
        static const uint32_t bpcEnv = gw_env("MC_BIG_COUNT_BPC", 24u);
        const uint32_t bpc = ws.countBpc > 0 ? (uint32_t)ws.countBpc : bpcEnv;
        // (second instance: 40 KB of LDS per block = four blocks per CU at a time; its grid in whole rounds of four)
        static const uint32_t bpc1 = gw_env("MC_BIG_COUNT1_BPC", 16u);
        const uint32_t grid = std::min<uint32_t>(256 * bpc, (b.n + 3) / 4), grid1 = std::min<uint32_t>(256 * bpc1, (b.n + 3) / 4);
        if (stage == 1) {
            if (taxkey) hipLaunchKernelGGL((gw_count_kernel<9, 4, true>), dim3(grid), dim3(256), 0, st, b, sp.s, tab, ws, maxCand, taxkey, c, 0u);
            else        hipLaunchKernelGGL((gw_count_kernel<9, 4, false>), dim3(grid), dim3(256), 0, st, b, sp.s, tab, ws, maxCand, taxkey, c, 0u);
        } else {
            if (taxkey) hipLaunchKernelGGL((gw_count_kernel<10, 4, true>), dim3(grid1), dim3(256), 0, st, b, sp.s, tab, ws, maxCand, taxkey, c, 256u);
            else        hipLaunchKernelGGL((gw_count_kernel<10, 4, false>), dim3(grid1), dim3(256), 0, st, b, sp.s, tab, ws, maxCand, taxkey, c, 256u);
        }
    } else if (stage == 4) {                                   // candidates of the sorted lists (after launch_gw_segsort)
        const uint32_t grid = std::min<uint32_t>(256 * 8, (b.n + 3) / 4);
        if (taxkey) {
            hipLaunchKernelGGL((gw_sorted_cands_kernel<true, false>), dim3(grid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, c);
            hipLaunchKernelGGL((gw_sorted_cands_kernel<true, true>), dim3(std::min<uint32_t>(grid, 512u)), dim3(1024), 0, st, b, tab, ws, maxCand, taxkey, c);
        } else {
            hipLaunchKernelGGL((gw_sorted_cands_kernel<false, false>), dim3(grid), dim3(256), 0, st, b, tab, ws, maxCand, taxkey, c);
            hipLaunchKernelGGL((gw_sorted_cands_kernel<false, true>), dim3(std::min<uint32_t>(grid, 512u)), dim3(1024), 0, st, b, tab, ws, maxCand, taxkey, c);
        }
    } else if (stage == 2) {
        static const uint32_t bpc2 = gw_env("MC_BIG_COUNT2_BPC", 4u);  // 32 KB per block of two waves
        const uint32_t grid = std::min<uint32_t>(256 * bpc2, (b.n + 1) / 2);
        if (taxkey) hipLaunchKernelGGL((gw_count_kernel<11, 2, true>), dim3(grid), dim3(128), 0, st, b, sp.s, tab, ws, maxCand, taxkey, c, 512u);
        else        hipLaunchKernelGGL((gw_count_kernel<11, 2, false>), dim3(grid), dim3(128), 0, st, b, sp.s, tab, ws, maxCand, taxkey, c, 512u);
    }
}

}  // namespace mcamd
