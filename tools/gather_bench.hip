// tools/gather_bench.hip -- random-gather roofline of the box (SURVEY.md §8d "random-access efficiency").
// Every G-lane group reads one G*16-byte segment at a pseudo-random, segment-aligned offset of a table;
// U independent loads are in flight per lane.  Prints segments/s and GB/s for several table sizes.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }

template <int G, int U>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ tab, uint32_t nseg, uint32_t iters, uint32_t* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / G, sub = tid % G;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t h = mix32(grp * 0x9E3779B1u + (it * U + u) * 0x85EBCA77u + 12345u);
            uint32_t seg = (uint32_t)(((uint64_t)h * nseg) >> 32);
            v[u] = tab[(size_t)seg * G + sub];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[tid] = acc;
}

template <int G, int U>
void run(const uint4* tab, size_t bytes, uint32_t* out, int blocksPerCU)
{
    const uint32_t nseg = (uint32_t)(bytes / (G * 16));
    const uint32_t blocks = 256 * blocksPerCU;
    const uint32_t iters = 2048 / U;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather<G, U>), dim3(blocks), dim3(256), 0, 0, tab, nseg, 8u, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((gather<G, U>), dim3(blocks), dim3(256), 0, 0, tab, nseg, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double segs = (double)blocks * 256 / G * iters * U;
    printf("table %7.0f MB  seg %3d B  U=%d  blocks/CU=%d : %7.2f Gseg/s  %8.1f GB/s  (%.2f ms)\n", bytes / 1048576.0, G * 16, U, blocksPerCU,
           segs / ms / 1e6, segs * G * 16 / ms / 1e6, ms);
}

int main()
{
    const size_t maxBytes = 32ull << 30;
    uint4* tab; uint32_t* out;
    hipMalloc(&tab, maxBytes); hipMalloc(&out, 256u * 8 * 256 * 4);
    hipMemset(tab, 1, maxBytes);
    for (size_t mb : {228ull, 1024ull, 8192ull, 32768ull}) {
        const size_t bytes = mb << 20;
        run<8, 1>(tab, bytes, out, 8); run<8, 2>(tab, bytes, out, 8); run<8, 4>(tab, bytes, out, 8); run<8, 8>(tab, bytes, out, 8);
        run<8, 4>(tab, bytes, out, 4);
        run<4, 1>(tab, bytes, out, 8); run<4, 4>(tab, bytes, out, 8); run<4, 8>(tab, bytes, out, 8);
        run<2, 4>(tab, bytes, out, 8); run<1, 4>(tab, bytes, out, 8);
        run<16, 4>(tab, bytes, out, 8);
    }
    return 0;
}
