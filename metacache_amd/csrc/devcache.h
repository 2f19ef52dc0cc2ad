// metacache_amd/csrc/devcache.h -- large device allocations that come back: hipMalloc hands out scrubbed memory at ~18 GB/s on this
// platform (610 calls = 9.7 of the 11.3 s of HIP API time of a 37 Gbp build; a part group's 38 GB of tables: two of its 2.4 s), so the
// callers that allocate the same large buffers over and over -- the builder's per-shard scratch, the tables of one part group after
// the other (partset.cpp) -- give them back to a small cache while they hold it open and get them again from there.  Internal.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>

namespace mcamd {

// as hipMalloc / hipFree.  big_free keeps a block of 64 MB or more for the next big_malloc of (nearly) its size while the cache is held
// open (big_cache_hold) and has room (MC_DEVCACHE_GB per device, default 64; the block is parked after a hipDeviceSynchronize, as hipFree would wait); otherwise -- and for every pointer big_malloc did not hand out -- it
// is hipFree.  A big_malloc the device cannot serve releases the cache and tries again.
hipError_t big_malloc(void** p, size_t bytes);
hipError_t big_free(void* p);
// hipMalloc for everybody else (workspaces, staging buffers): an allocation the device cannot serve releases what the cache keeps and
// tries again -- tens of GB parked for the next part group must not make a 100 MB workspace fail
hipError_t dev_malloc(void** p, size_t bytes);
// +1 / -1; when the count is back at 0 everything kept is released
void big_cache_hold(int delta);
void big_cache_trim();

}  // namespace mcamd
