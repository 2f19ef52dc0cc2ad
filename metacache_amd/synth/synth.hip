// metacache_amd/synth/synth.hip -- GPU side of the synthetic workload (synth_spec.h): targets and reads are written straight into
// HBM, so a 150 Gbp collection never exists anywhere as a whole.  WORKLOAD GENERATION ONLY -- not part of libmetacache_amd.so.
#include <hip/hip_runtime.h>

#include "synth_spec.h"

namespace {

// bases [first, first + n) of ONE target; 4 bases per thread, one 4-byte store
__global__ __launch_bounds__(256) void target_kernel(syn_target t, uint32_t first, uint32_t n, uint8_t* __restrict__ dst)
{
    const uint64_t i4 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    uint32_t w = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t c = i4 + j < n ? (uint32_t)syn_ascii(syn_code(&t, first + (uint32_t)i4 + j)) : 0u;
        w |= c << (8 * j);
    }
    if (i4 + 4 <= n) *reinterpret_cast<uint32_t*>(dst + i4) = w;
    else for (uint32_t j = 0; i4 + j < n; ++j) dst[i4 + j] = (uint8_t)(w >> (8 * j));
}

// many whole targets at once: target i goes to dst + offs[i]; grid.y = target
__global__ __launch_bounds__(256) void targets_kernel(const syn_target* __restrict__ targets, const uint64_t* __restrict__ offs,
                                                      uint8_t* __restrict__ dst)
{
    const syn_target t = targets[blockIdx.y];
    uint8_t* out = dst + offs[blockIdx.y];
    for (uint64_t i4 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i4 < t.length; i4 += (uint64_t)gridDim.x * 1024) {
        uint32_t w = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t c = i4 + j < t.length ? (uint32_t)syn_ascii(syn_code(&t, (uint32_t)i4 + j)) : 0u;
            w |= c << (8 * j);
        }
        *reinterpret_cast<uint32_t*>(out + i4) = w;      // offsets are 4-byte aligned, up to 3 zero bytes follow the last base
    }
}

// one thread per 4 characters of a row
__global__ __launch_bounds__(256) void reads_kernel(syn_read_params P, const syn_target* __restrict__ targets, uint64_t first, uint64_t n,
                                                    uint8_t* __restrict__ dst, uint8_t* __restrict__ dst2)
{
    const uint32_t words = P.row_bytes / 4;
    const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t i = id / words;
    const uint32_t j0 = (uint32_t)(id % words) * 4;
    if (i >= n) return;
    const uint64_t r = first + i;
    const syn_read_origin o = syn_origin(&P, targets, r);
    const syn_target t = targets[o.target];
    for (uint32_t m = 0; m < (P.paired ? 2u : 1u); ++m) {
        uint32_t w = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j)
            if (j0 + j < P.read_len) w |= (uint32_t)syn_read_char(&P, &t, o, r, m, j0 + j) << (8 * j);
        *reinterpret_cast<uint32_t*>((m ? dst2 : dst) + i * P.row_bytes + j0) = w;
    }
}

}  // namespace

extern "C" {

// all pointers are DEVICE pointers except where noted; stream = hipStream_t or NULL; returns 0 or the hipError_t
int mcs_target(const syn_target* hostTarget, uint32_t first, uint32_t n, void* dst, void* stream)
{
    if (!n) return 0;
    hipLaunchKernelGGL(target_kernel, dim3((uint32_t)(((uint64_t)n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, *hostTarget, first, n,
                       (uint8_t*)dst);
    return (int)hipGetLastError();
}

// dTargets[count] and dOffs[count] on the device; dst must hold offs[count-1] + round_up(length, 4) bytes
int mcs_targets(const syn_target* dTargets, const uint64_t* dOffs, uint32_t count, void* dst, void* stream)
{
    if (!count) return 0;
    hipLaunchKernelGGL(targets_kernel, dim3(256, count), dim3(256), 0, (hipStream_t)stream, dTargets, dOffs, (uint8_t*)dst);
    return (int)hipGetLastError();
}

int mcs_reads(const syn_read_params* hostParams, const syn_target* dTargets, uint64_t first, uint64_t n, void* dst, void* dst2, void* stream)
{
    if (!n) return 0;
    if (hostParams->row_bytes % 4 || hostParams->row_bytes < hostParams->read_len) return -1;
    const uint64_t threads = n * (hostParams->row_bytes / 4);
    hipLaunchKernelGGL(reads_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *hostParams, dTargets, first, n,
                       (uint8_t*)dst, (uint8_t*)dst2);
    return (int)hipGetLastError();
}

}  // extern "C"
