// Multi-GPU score gather of libiefvad: one RCCL exchange of per-snippet fp32 scores, rank order =
// the reference's sequential video order (/root/reference/test.py:123-129,153).  The reference itself has no
// collective (main.py:7 imports torch.distributed and never uses it); test-set videos shard across ranks with no
// state crossing them, so this is the only communication of the whole path.
//
// librccl is bound at run time with dlopen, preferring a copy the process has already loaded (PyTorch ships its own
// librccl.so.1; two RCCL runtimes in one process is what this avoids), so libiefvad.so itself has no link-time
// dependency on RCCL and a single-GPU consumer never loads it.  The types come from <rccl/rccl.h>.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>

struct RcclApi {
    void* so;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char* (*GetErrorString)(ncclResult_t);
    ncclResult_t (*GetVersion)(int*);
};

// Returns nullptr and fills `why` when librccl cannot be bound.
static const RcclApi* rccl_api(const char** why) {
    static RcclApi api;
    static int state = 0;   // 0 = not tried, 1 = bound, -1 = failed
    static const char* err = "";
    if (state == 0) {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* so = nullptr;
        for (int i = 0; i < 2 && !so; ++i) so = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);   // a copy already in the process
        for (int i = 0; i < 3 && !so; ++i) so = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!so) {
            err = "librccl.so.1 not found (dlopen)";
            state = -1;
        } else {
            api.so = so;
            bool ok = true;
#define RCCL_BIND(field, sym)                                        \
    do {                                                             \
        *(void**)(&api.field) = dlsym(so, sym);                      \
        if (!api.field) { ok = false; err = "librccl lacks " sym; }  \
    } while (0)
            RCCL_BIND(GetUniqueId, "ncclGetUniqueId");
            RCCL_BIND(CommInitRank, "ncclCommInitRank");
            RCCL_BIND(CommDestroy, "ncclCommDestroy");
            RCCL_BIND(CommCount, "ncclCommCount");
            RCCL_BIND(AllGather, "ncclAllGather");
            RCCL_BIND(Send, "ncclSend");
            RCCL_BIND(Recv, "ncclRecv");
            RCCL_BIND(GroupStart, "ncclGroupStart");
            RCCL_BIND(GroupEnd, "ncclGroupEnd");
            RCCL_BIND(GetErrorString, "ncclGetErrorString");
            RCCL_BIND(GetVersion, "ncclGetVersion");
#undef RCCL_BIND
            state = ok ? 1 : -1;
        }
    }
    if (why) *why = err;
    return state == 1 ? &api : nullptr;
}

struct iefvad_comm {
    ncclComm_t comm;
    int nranks;      // as RCCL reports it (ncclCommCount), not as the caller claimed
    int rank;
    int device;
};
