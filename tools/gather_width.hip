// tools/gather_width.hip -- MEASUREMENT TOOL (round 4): the random-gather rate of the box by request width -- units of 16 / 32 / 64 / 128 bytes fetched by
// 1 / 2 / 4 / 8 lanes at random places of a large buffer (hipcc --offload-arch=gfx950 -O3 -o gather_width tools/gather_width.hip; ./gather_width <MiB>).
// Result on MI355X (64 GiB): 38 / 47 / 47 / 47 G units per second: requests, not bytes, are what random lookups cost.
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
    for (int bpc : {8, 16}) {
        printf("bpc %d: 16B x1 lane: U4 %.1f U8 %.1f | 32B x2 lanes: U4 %.1f U8 %.1f | 64B x4 lanes: U4 %.1f U8 %.1f | 128B x8 lanes: U4 %.1f U8 %.1f  (G units/s)\n", bpc,
               measure<1, 4>(tab, bytes, out, bpc) / 1e9, measure<1, 8>(tab, bytes, out, bpc) / 1e9,
               measure<2, 4>(tab, bytes, out, bpc) / 1e9, measure<2, 8>(tab, bytes, out, bpc) / 1e9,
               measure<4, 4>(tab, bytes, out, bpc) / 1e9, measure<4, 8>(tab, bytes, out, bpc) / 1e9,
               measure<8, 4>(tab, bytes, out, bpc) / 1e9, measure<8, 8>(tab, bytes, out, bpc) / 1e9);
    }
    return 0;
}
