// tools/gather_bench2.hip -- which part of the probe loop costs what?  Variants of an 8-lane x 16-byte
// random gather: occupancy, index loaded from memory (dependent), ballot resolve, compact store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }

// MODE bit0: indices come from memory (coalesced 128 B per wave-iteration), bit1: ballot resolve, bit2: store 8 B per hit
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ tab, uint32_t nseg, const uint32_t* __restrict__ idx, uint32_t iters,
                                         uint64_t* __restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3;
    const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t feat;
        if (MODE & 1) feat = idx[((size_t)wid * iters + it) * 32 + (lane & 31)];
        else feat = mix32((wid * iters + it) * 32 + (lane & 31) + 77);
        uint4 v[4]; uint32_t f[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f[r] = __shfl(feat, r * 8 + grp);
            uint32_t seg = (uint32_t)(((uint64_t)mix32(f[r]) * nseg) >> 32);
            v[r] = tab[(size_t)seg * 8 + sub];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE & 2) {
                bool hit = v[r].x == f[r];
                uint64_t m = __ballot(hit);
                acc += __popcll(m);
                if ((MODE & 4) && hit) out[((size_t)wid * iters + it) * 32 + r * 8 + grp] = ((uint64_t)v[r].w << 32) | v[r].z;
            } else acc += v[r].x ^ v[r].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
void run(const uint4* tab, size_t bytes, const uint32_t* idx, uint64_t* out, int blocksPerCU, uint32_t totalQueries)
{
    const uint32_t nseg = (uint32_t)(bytes / 128);
    const uint32_t blocks = 256 * blocksPerCU, waves = blocks * 4;
    const uint32_t iters = totalQueries / waves;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, tab, nseg, idx, 4u, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, tab, nseg, idx, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double q = (double)waves * iters;
    printf("mode %d  blocks/CU %d : %7.3f ms per 1M queries (32 lines each), %6.2f Glines/s\n", MODE, blocksPerCU, ms * 1e6 / q, q * 32 / ms / 1e6);
}

int main()
{
    const size_t bytes = 366ull << 20;
    const uint32_t Q = 1u << 20;
    uint4* tab; uint32_t* idx; uint64_t* out;
    hipMalloc(&tab, bytes); hipMalloc(&idx, (size_t)Q * 32 * 4 + 4096); hipMalloc(&out, (size_t)Q * 32 * 8 + 4096);
    hipMemset(tab, 1, bytes); hipMemset(idx, 3, (size_t)Q * 32 * 4);
    for (int bpc : {2, 4, 8}) {
        run<0>(tab, bytes, idx, out, bpc, Q); run<1>(tab, bytes, idx, out, bpc, Q); run<3>(tab, bytes, idx, out, bpc, Q); run<7>(tab, bytes, idx, out, bpc, Q);
    }
    return 0;
}
