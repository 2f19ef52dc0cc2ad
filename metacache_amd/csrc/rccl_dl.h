// metacache_amd/csrc/rccl_dl.h -- the few RCCL entry points the multi-GPU drivers (partset.cpp, keyset.cpp) use, loaded at run time
// (dlopen librccl.so.1): the library has no link-time dependency on RCCL, and a process that already carries one (PyTorch) shares it.
// rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer, ncclChar = 0, ncclUint32 = 3.  Internal.
#pragma once

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <mutex>
#include <string>

namespace mcamd {

struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream) = nullptr;
    int (*Send)(const void* send, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
    int (*Recv)(void* recv, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    static constexpr int kChar = 0, kUint32 = 3;
    int state = 0;                     // 0 = not tried, 1 = usable, -1 = failed (err says why): the OUTCOME is cached, not the handle
    std::mutex mtx;                    // a keyset and a partset may be opened from two threads at once
    bool load()
    {
        std::lock_guard<std::mutex> lock(mtx);
        if (state) return state > 0;
        state = load_locked() ? 1 : -1;
        if (state < 0 && lib) { dlclose(lib); lib = nullptr; }
        return state > 0;
    }
    bool load_locked()
    {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = "RCCL not found (librccl.so.1)"; return false; }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) err = std::string("RCCL symbol missing: ") + n; return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather && Send && Recv;
    }
    std::string text(int r) const { return GetErrorString ? GetErrorString(r) : "error"; }
};
inline Rccl& rccl() { static Rccl r; return r; }

}  // namespace mcamd
