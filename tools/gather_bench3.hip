// tools/gather_bench3.hip -- lane-private gathers: every LANE reads PART or ALL of its own random 128-byte line
// (NV x 16 B, back to back).  Does the L1 merge the 8 requests to one in-flight line?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }

template <int NV, int U>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ tab, uint32_t nseg, uint32_t iters, uint32_t* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[U][NV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t h = mix32(tid * 0x9E3779B1u + (it * U + u) * 0x85EBCA77u + 999u);
            uint32_t seg = (uint32_t)(((uint64_t)h * nseg) >> 32);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[u][j] = tab[(size_t)seg * 8 + j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < NV; ++j) acc += v[u][j].x ^ v[u][j].w;
    }
    if (acc == 0x12345678u) out[tid] = acc;
}
template <int NV, int U>
void run(const uint4* tab, size_t bytes, uint32_t* out, int bpc)
{
    const uint32_t nseg = (uint32_t)(bytes / 128), blocks = 256 * bpc, iters = 256 / U;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NV, U>), dim3(blocks), dim3(256), 0, 0, tab, nseg, 2u, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NV, U>), dim3(blocks), dim3(256), 0, 0, tab, nseg, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double lines = (double)blocks * 256 * iters * U;
    printf("lane-private: %d x16B per line, U=%d, blocks/CU=%d : %7.2f Glines/s  (%.2f ms)\n", NV, U, bpc, lines / ms / 1e6, ms);
}
int main(int argc, char** argv)
{
    // table size in MiB (default 366 = the configs[1] table; pass e.g. 16384 to leave the 256 MB infinity cache and the TLB reach)
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 366ull) << 20;
    printf("table: %zu MiB\n", bytes >> 20);
    uint4* tab; uint32_t* out;
    hipMalloc(&tab, bytes); hipMalloc(&out, 256u * 8 * 256 * 4);
    hipMemset(tab, 1, bytes);
    run<1, 4>(tab, bytes, out, 8); run<2, 4>(tab, bytes, out, 8); run<4, 2>(tab, bytes, out, 8); run<8, 1>(tab, bytes, out, 8);
    run<8, 2>(tab, bytes, out, 4); run<8, 1>(tab, bytes, out, 4); run<2, 4>(tab, bytes, out, 4); run<2, 8>(tab, bytes, out, 4);
    return 0;
}
