// tools/gather_width.hip -- MEASUREMENT TOOL (round 4): the random-gather rate of the box by request width -- units of 16 / 32 / 64 / 128 bytes fetched by
// 1 / 2 / 4 / 8 lanes at random places of a large buffer (hipcc --offload-arch=gfx950 -O3 -o gather_width tools/gather_width.hip; ./gather_width <MiB>).
// Result on MI355X (64 GiB): 38 / 47 / 47 / 47 G units per second: requests, not bytes, are what random lookups cost.
// Round 5: + units of 256 bytes (16 lanes) and the LIST shape of the filter kernels -- runs of 49 numbers (196 bytes) at random 4-byte
// offsets, read in rounds of 64 bytes by 4 lanes x 16 bytes -- with the lines / sectors a run touches computed exactly, for the
// calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ on these shapes (profiles/r05_fetch_calibration.md).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }
// LANES lanes share one unit of LANES * 16 bytes; U units in flight per lane group
template <int LANES, int U>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ tab, uint64_t nunits, uint32_t iters, uint32_t* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t id = tid / LANES;
            const uint32_t h1 = mix32(id * 0x9E3779B1u + (it * U + u) * 0x85EBCA77u + 999u), h2 = mix32(h1 ^ 0x5bd1e995u);
            const uint64_t r = __umul64hi(((uint64_t)h1 << 32) | h2, nunits);
            v[u] = tab[r * LANES + (tid % LANES)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].w;
    }
    if (acc == 0x12345678u) out[tid] = acc;
}
// the filter kernels' shape: a GROUP of 16 lanes reads one run of RUN numbers (4-byte aligned start) as ceil(RUN / 16) rounds of 64 bytes, four
// lanes x 16 bytes per round; counts the 128-byte lines and 64-byte sectors every run touches (exact, from the addresses)
struct __attribute__((packed, aligned(4))) U4u { uint32_t x, y, z, w; };
template <int RUN>
__global__ __launch_bounds__(256) void klist(const uint32_t* __restrict__ tab, uint64_t nwords, uint32_t iters, uint32_t* __restrict__ out, unsigned long long* __restrict__ touched)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, grp = tid >> 4, l16 = tid & 15u;
    uint32_t acc = 0;
    unsigned long long lines = 0, sectors = 0;
    constexpr int ROUNDS = (RUN + 15) / 16;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t h1 = mix32(grp * 0x9E3779B1u + it * 0x85EBCA77u + 999u), h2 = mix32(h1 ^ 0x5bd1e995u);
        const uint64_t start = __umul64hi(((uint64_t)h1 << 32) | h2, nwords - RUN - 16);
        if (l16 == 0) {
            const uint64_t a = start * 4, e = a + RUN * 4 - 1;
            lines += (e >> 7) - (a >> 7) + 1; sectors += (e >> 6) - (a >> 6) + 1;
        }
#pragma unroll
        for (int r0 = 0; r0 < ROUNDS; r0 += 4) {                  // 16 lanes = 4 rounds per load instruction
            const int r = r0 + (int)(l16 >> 2);
            const uint32_t off = r * 16 + (l16 & 3u) * 4;
            if (off < RUN) { const U4u t = *reinterpret_cast<const U4u*>(tab + start + off); acc += t.x ^ t.w; }
        }
    }
    if (acc == 0x12345678u) out[tid] = acc;
    if (l16 == 0) { atomicAdd(touched, lines); atomicAdd(touched + 1, sectors); }
}
// Round 6: the DIRECT-ADDRESS index of SURVEY 7 as a benchmark -- 2^32 entries of 8 bytes (size | list index: 32 GiB), the feature IS the
// index: one 8-byte load per lookup by ONE lane, no key compare, no chain; U lookups in flight per lane.  What the hash table's 40
// requests per read (32 buckets + chains + the features' own lines) would become: 32.
template <int U>
__global__ __launch_bounds__(256) void kdirect(const uint2* __restrict__ tab, uint64_t nentries, uint32_t iters, uint32_t* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t h1 = mix32(tid * 0x9E3779B1u + (it * U + u) * 0x85EBCA77u + 999u), h2 = mix32(h1 ^ 0x5bd1e995u);
            v[u] = tab[__umul64hi(((uint64_t)h1 << 32) | h2, nentries)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y;
    }
    if (acc == 0x12345678u) out[tid] = acc;
}
template <int U>
double measure_direct(const uint4* tab, size_t bytes, uint32_t* out, int bpc)
{
    const uint32_t blocks = 256 * bpc, iters = 256 / U;
    const uint64_t nentries = bytes / 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((kdirect<U>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint2*>(tab), nentries, 2u, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((kdirect<U>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint2*>(tab), nentries, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 256 * iters * U / (ms * 1e-3);
}
template <int LANES, int U>
double measure(const uint4* tab, size_t bytes, uint32_t* out, int bpc)
{
    const uint32_t blocks = 256 * bpc, iters = 256 / U;
    const uint64_t nunits = bytes / (16 * LANES);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<LANES, U>), dim3(blocks), dim3(256), 0, 0, tab, nunits, 2u, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<LANES, U>), dim3(blocks), dim3(256), 0, 0, tab, nunits, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 256 * iters * U / LANES / (ms * 1e-3);
}
int main(int argc, char** argv)
{
    const uint64_t bytes = (argc > 1 ? (uint64_t)atoll(argv[1]) : 65536ull) << 20;
    uint4* tab = nullptr; uint32_t* out = nullptr;
    if (hipMalloc(&tab, bytes) != hipSuccess) return 1;
    hipMalloc(&out, 256u * 32 * 256 * 4);
    hipMemset(tab, 1, bytes); hipDeviceSynchronize();
    {
        unsigned long long* touched = nullptr; hipMalloc(&touched, 16); hipMemset(touched, 0, 16);
        const uint32_t blocks = 256 * 16, iters = 64;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL((klist<49>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint32_t*>(tab), bytes / 4, 2u, out, touched);
        hipDeviceSynchronize(); hipMemset(touched, 0, 16);
        hipEventRecord(a);
        hipLaunchKernelGGL((klist<49>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint32_t*>(tab), bytes / 4, iters, out, touched);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned long long h[2]; hipMemcpy(h, touched, 16, hipMemcpyDeviceToHost);
        const double runs = (double)blocks * 16 * iters;
        printf("list shape (runs of 49 numbers = 196 bytes at random 4-byte offsets, rounds of 64 bytes by 4 lanes): %.0f runs in %.3f ms = %.2f G runs/s; "
               "touched per run: %.3f lines of 128 B, %.3f sectors of 64 B; %.2f TB/s if every touched line moves 128 B, %.2f TB/s if touched sectors move\n",
               runs, ms, runs / ms / 1e6, h[0] / runs, h[1] / runs, h[0] * 128.0 / ms / 1e9, h[1] * 64.0 / ms / 1e9);
        printf("256B x16 lanes: U4 %.1f G units/s  (128B x8: %.1f; x 128 B / x 256 B = %.2f / %.2f TB/s)\n", measure<16, 4>(tab, bytes, out, 16) / 1e9,
               measure<8, 4>(tab, bytes, out, 16) / 1e9, measure<8, 4>(tab, bytes, out, 16) * 128 / 1e12, measure<16, 4>(tab, bytes, out, 16) * 256 / 1e12);
    }
    for (int bpc : {8, 16, 32})
        printf("direct-address index, 8-byte entries by one lane (bpc %d): U4 %.1f U8 %.1f U16 %.1f G lookups/s\n", bpc,
               measure_direct<4>(tab, bytes, out, bpc) / 1e9, measure_direct<8>(tab, bytes, out, bpc) / 1e9, measure_direct<16>(tab, bytes, out, bpc) / 1e9);
    for (int bpc : {8, 16}) {
        printf("bpc %d: 16B x1 lane: U4 %.1f U8 %.1f | 32B x2 lanes: U4 %.1f U8 %.1f | 64B x4 lanes: U4 %.1f U8 %.1f | 128B x8 lanes: U4 %.1f U8 %.1f  (G units/s)\n", bpc,
               measure<1, 4>(tab, bytes, out, bpc) / 1e9, measure<1, 8>(tab, bytes, out, bpc) / 1e9,
               measure<2, 4>(tab, bytes, out, bpc) / 1e9, measure<2, 8>(tab, bytes, out, bpc) / 1e9,
               measure<4, 4>(tab, bytes, out, bpc) / 1e9, measure<4, 8>(tab, bytes, out, bpc) / 1e9,
               measure<8, 4>(tab, bytes, out, bpc) / 1e9, measure<8, 8>(tab, bytes, out, bpc) / 1e9);
    }
    return 0;
}
