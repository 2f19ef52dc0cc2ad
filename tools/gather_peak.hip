// tools/gather_peak.hip -- the box's random-access peak, the second roofline of this path ("random-access efficiency
// vs measured random-gather peak of the box").  MEASUREMENT TOOL (bench.py loads it as libmcgather.so; also a stand-alone program):
// nothing but random reads of a large buffer in the access shapes the hot path uses,
//   0  lane-private 64-byte bucket   : every lane reads 4 x 16 B of its own random 64-byte half line   (probe_cands, tables <= 1 GiB)
//   1  quad-cooperative 64-byte bucket: four lanes read one random 64-byte bucket, 16 B each            (probe_cands<QUAD>)
//   2  wave-coalesced list            : a wave reads 64 consecutive u64 (512 B) at a random 8-byte aligned place (big_cands sweeps)
// Result: requests per second in units of 64-byte requests (what TCC_EA0_RDREQ counts, profiles/r01_fetch_calibration.md).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace {
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }

template <int SHAPE, int U>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ tab, uint64_t nunits, uint32_t iters, uint32_t* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[U][SHAPE == 0 ? 4 : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t id = SHAPE == 0 ? tid : SHAPE == 1 ? (tid >> 2) : (tid >> 6);
            const uint32_t h1 = mix32(id * 0x9E3779B1u + (it * U + u) * 0x85EBCA77u + 999u), h2 = mix32(h1 ^ 0x5bd1e995u);
            const uint64_t r = __umul64hi(((uint64_t)h1 << 32) | h2, nunits);            // uniform in [0, nunits), no division
            if (SHAPE == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[u][j] = tab[r * 4 + j];                      // unit = 64-byte bucket
            } else if (SHAPE == 1) v[u][0] = tab[r * 4 + (tid & 3u)];
            else v[u][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint64_t*>(tab) + r + (lane & ~1u));   // unit = u64; lanes pair up for 16 B
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < (SHAPE == 0 ? 4 : 1); ++j) acc += v[u][j].x ^ v[u][j].w;
    }
    if (acc == 0x12345678u) out[tid] = acc;
}

template <int SHAPE, int U>
double measure(const uint4* tab, size_t bytes, uint32_t* out, int blocksPerCu)
{
    const uint32_t blocks = 256 * blocksPerCu, iters = 128 / U;
    const uint64_t nunits = SHAPE == 2 ? bytes / 8 - 128 : bytes / 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather_kernel<SHAPE, U>), dim3(blocks), dim3(256), 0, 0, tab, nunits, 2u, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((gather_kernel<SHAPE, U>), dim3(blocks), dim3(256), 0, 0, tab, nunits, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    const double threads = (double)blocks * 256 * iters * U;
    // 64-byte requests: shape 0 one per lane, shape 1 one per quad, shape 2 eight (512 B; 9 when the start is not 64-byte aligned) per wave
    const double req = SHAPE == 0 ? threads : SHAPE == 1 ? threads / 4 : threads / 64 * 8.875;
    return req / (ms * 1e-3);
}
}  // namespace

// requests[0..2] = 64-byte requests per second of the three shapes (best of a few launch shapes each); buffer of `bytes` bytes
extern "C" int mcg_gather_peak(uint64_t bytes, double* requests)
{
    uint4* tab = nullptr; uint32_t* out = nullptr;
    if (hipMalloc(&tab, bytes) != hipSuccess) return -1;
    if (hipMalloc(&out, 256u * 16 * 256 * 4) != hipSuccess) { (void)hipFree(tab); return -1; }
    (void)hipMemset(tab, 1, bytes);
    (void)hipDeviceSynchronize();
    auto mx = [](double a, double b) { return a > b ? a : b; };
    requests[0] = mx(mx(measure<0, 2>(tab, bytes, out, 8), measure<0, 4>(tab, bytes, out, 8)), measure<0, 2>(tab, bytes, out, 12));
    requests[1] = mx(mx(measure<1, 4>(tab, bytes, out, 8), measure<1, 8>(tab, bytes, out, 8)), measure<1, 4>(tab, bytes, out, 16));
    requests[2] = mx(mx(measure<2, 4>(tab, bytes, out, 8), measure<2, 8>(tab, bytes, out, 8)), measure<2, 8>(tab, bytes, out, 16));
    (void)hipFree(tab); (void)hipFree(out);
    return 0;
}

// ---- co-residency experiment (round 5, tools/coresidency_probe.py): a lookup-shaped kernel small enough for what five waves of
// gw_filter_count_kernel per SIMD leave of a CU (32 registers, no LDS), launched on a second stream while the filter runs.
// LANES lanes share one random unit of LANES x 16 bytes (1: a bucket's keys; 4: a whole 64-byte bucket), U units in flight per lane.
template <int LANES, int U>
__attribute__((amdgpu_num_vgpr(32), amdgpu_flat_work_group_size(64, 64))) __global__ void side_gather_kernel(const uint4* __restrict__ tab, uint64_t nunits, uint32_t iters, uint32_t* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n32 = (uint32_t)nunits;                         // (units below 2^32: one multiply per place)
    const char* const base = reinterpret_cast<const char*>(tab) + (LANES == 1 ? 0u : (tid % LANES) * 16u);
    uint32_t acc = 0, h = (tid / LANES) * 0x9E3779B1u + 999u;
#pragma unroll 1
    for (uint32_t it = 0; it < iters; ++it) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            h = mix32(h + 0x85EBCA77u);
            v[u] = *reinterpret_cast<const uint4*>(base + (uint64_t)__umulhi(h, n32) * 64u);
        }
        __builtin_amdgcn_sched_barrier(0);                         // (all U loads in flight before the first is waited for)
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (acc == 0x12345678u) out[tid] = acc;
}
// units = blocks x 64 x iters x U / LANES; shape 1 / 4 = LANES
extern "C" int mcg_side_launch(const void* tab, uint64_t bytes, int lanes, uint32_t blocks, uint32_t iters, void* out, void* stream)
{
    const uint64_t nunits = bytes / 64;
    hipStream_t st = (hipStream_t)stream;
    if (lanes == 1) hipLaunchKernelGGL((side_gather_kernel<1, 4>), dim3(blocks), dim3(64), 0, st, (const uint4*)tab, nunits, iters, (uint32_t*)out);
    else hipLaunchKernelGGL((side_gather_kernel<4, 4>), dim3(blocks), dim3(64), 0, st, (const uint4*)tab, nunits, iters, (uint32_t*)out);
    return (int)hipGetLastError();
}

#ifdef GATHER_PEAK_MAIN
int main(int argc, char** argv)
{
    const uint64_t bytes = (argc > 1 ? (uint64_t)atoll(argv[1]) : 32768ull) << 20;
    double r[3];
    if (mcg_gather_peak(bytes, r)) { printf("allocation failed\n"); return 1; }
    printf("{\"buffer_MiB\": %llu, \"lane_private_64B_Greq_s\": %.2f, \"quad_64B_Greq_s\": %.2f, \"wave_512B_list_Greq_s\": %.2f}\n",
           (unsigned long long)(bytes >> 20), r[0] / 1e9, r[1] / 1e9, r[2] / 1e9);
    return 0;
}
#endif
