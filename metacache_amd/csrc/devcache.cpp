// metacache_amd/csrc/devcache.cpp -- see devcache.h
#include "devcache.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace mcamd {

namespace {

constexpr size_t kMinCached = 64ull << 20;
struct Block { void* p; size_t bytes; int dev; };
std::mutex g_mu;
std::unordered_map<void*, Block> g_live;           // blocks big_malloc handed out (kMinCached and more)
std::vector<Block> g_kept;
std::unordered_map<int, size_t> g_keptBytes;       // per device: MC_DEVCACHE_GB is a device's budget
int g_hold = 0;
uint64_t g_trims = 0;                               // generation of g_kept: big_free parks a block only into the generation it reserved room in

size_t budget()
{
    static const size_t b = [] { const char* e = std::getenv("MC_DEVCACHE_GB"); return (size_t)(e ? std::max(0, std::atoi(e)) : 64) << 30; }();
    return b;
}

void trim_locked(std::vector<Block>& out) { out.swap(g_kept); g_keptBytes.clear(); ++g_trims; }

void release(const std::vector<Block>& blocks)
{
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const Block& b : blocks) { (void)hipSetDevice(b.dev); (void)hipFree(b.p); }
    if (!blocks.empty()) (void)hipSetDevice(cur);
}

}  // namespace

hipError_t big_malloc(void** p, size_t bytes)
{
    if (bytes < kMinCached) return hipMalloc(p, bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        size_t best = g_kept.size();
        for (size_t i = 0; i < g_kept.size(); ++i)              // the smallest kept block that holds it and is not a quarter larger
            if (g_kept[i].dev == dev && g_kept[i].bytes >= bytes && g_kept[i].bytes <= bytes + bytes / 4 &&
                (best == g_kept.size() || g_kept[i].bytes < g_kept[best].bytes)) best = i;
        if (best != g_kept.size()) {
            const Block b = g_kept[best];
            g_kept.erase(g_kept.begin() + (std::ptrdiff_t)best);
            g_keptBytes[b.dev] -= b.bytes;
            g_live[b.p] = b;
            *p = b.p;
            return hipSuccess;
        }
    }
    const hipError_t e = dev_malloc(p, bytes);
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_mu); g_live[*p] = Block{*p, bytes, dev}; }
    return e;
}

hipError_t dev_malloc(void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {                                       // what is kept may be what is missing
        std::vector<Block> out;
        { std::lock_guard<std::mutex> lk(g_mu); trim_locked(out); }
        if (!out.empty()) { (void)hipGetLastError(); release(out); e = hipMalloc(p, bytes); }
    }
    return e;
}

hipError_t big_free(void* p)
{
    if (!p) return hipSuccess;
    Block b{nullptr, 0, 0};
    bool keep = false;
    uint64_t gen = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        gen = g_trims;
        auto it = g_live.find(p);
        if (it != g_live.end()) {
            b = it->second;
            g_live.erase(it);
            keep = g_hold > 0 && g_keptBytes[b.dev] + b.bytes <= budget();
            if (keep) g_keptBytes[b.dev] += b.bytes;              // (the room is reserved; the block joins the list once the device is idle)
        }
    }
    if (!keep) return hipFree(p);
    // hipFree waits for the device before the memory goes to anybody else; a kept block gets the same: whatever kernel or copy is still
    // queued on ANY stream of its device (the caller's, a pipe's, a loader thread's) is done before another big_malloc can hand it out
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != b.dev) (void)hipSetDevice(b.dev);
    const hipError_t e = hipDeviceSynchronize();
    if (cur != b.dev) (void)hipSetDevice(cur);
    // the lock was dropped for the wait: a trim (a failing allocation, the last holder letting go) may have emptied the cache meanwhile --
    // then the reservation is gone with it and the block goes back to the device instead of being parked uncounted
    bool park = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        park = g_hold > 0 && g_trims == gen;
        if (park) g_kept.push_back(b);
        else if (g_trims == gen) g_keptBytes[b.dev] -= b.bytes;
    }
    if (!park) { const hipError_t f = hipFree(p); return e != hipSuccess ? e : f; }
    return e;
}

void big_cache_trim()
{
    std::vector<Block> out;
    { std::lock_guard<std::mutex> lk(g_mu); trim_locked(out); }
    release(out);
}

void big_cache_hold(int delta)
{
    bool trim = false;
    { std::lock_guard<std::mutex> lk(g_mu); g_hold = std::max(0, g_hold + delta); trim = g_hold == 0; }
    if (trim) big_cache_trim();
}

}  // namespace mcamd
