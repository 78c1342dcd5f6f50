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

#include <mutex>
#include <vector>

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

// Returns nullptr and fills `why` when librccl cannot be bound.  The binding happens once per process (std::call_once:
// two threads creating communicators at the same time must not both run the dlopen / dlsym sequence).
static const RcclApi* rccl_api(const char** why) {
    static RcclApi api;
    static bool bound = false;
    static const char* err = "";
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* so = nullptr;
        for (int i = 0; i < 2 && !so; ++i) so = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);   // a copy already in the process
        for (int i = 0; i < 3 && !so; ++i) so = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!so) {
            err = "librccl.so.1 not found (dlopen)";
            return;
        }
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
        bound = ok;
    });
    if (why) *why = err;
    return bound ? &api : nullptr;
}

// ---- the exchange, as data.  Rank r's slice lands at its running offset on every rank (rank order = the reference's
// sequential video order).  Equal counts: ONE all-gather.  Unequal counts: one grouped point-to-point exchange in which this
// rank sends its `my_count` elements to every peer that exists and receives peer p's `recv_count` elements at
// `recv_offset`; zero-length transfers are not issued (RCCL treats a 0-byte send / recv pair as a no-op on both sides only
// if BOTH sides skip it, which they do: every rank derives the plan from the same `counts`).  Pure host code, so that the
// bookkeeping is testable without a GPU or a second rank (iefvad_gather_plan, tests/test_cabi_cpu.py).
struct GatherStep { int peer; size_t send_count, recv_offset, recv_count; };
struct GatherPlan {
    bool equal;
    size_t my_offset, my_count, total;
    std::vector<GatherStep> steps;       // unequal counts only: one per peer, in rank order
};

// counts == nullptr: every rank contributes `count`.  Returns nullptr on success, else a message.
static const char* gather_plan(int nranks, int rank, const int64_t* counts, size_t count, GatherPlan* plan) {
    if (nranks < 1 || rank < 0 || rank >= nranks) return "rank outside the communicator";
    plan->steps.clear();
    plan->equal = true;
    if (!counts) {
        plan->my_count = count;
        plan->my_offset = (size_t)rank * count;
        plan->total = (size_t)nranks * count;
        return nullptr;
    }
    size_t off = 0;
    for (int r = 0; r < nranks; ++r) {
        if (counts[r] < 0) return "negative count";
        if (counts[r] != counts[0]) plan->equal = false;
        if (r == rank) plan->my_offset = off;
        off += (size_t)counts[r];
    }
    plan->total = off;
    plan->my_count = (size_t)counts[rank];
    if (plan->equal) return nullptr;
    off = 0;
    for (int r = 0; r < nranks; ++r) {
        if (r != rank) plan->steps.push_back(GatherStep{r, plan->my_count, off, (size_t)counts[r]});
        off += (size_t)counts[r];
    }
    return nullptr;
}

struct iefvad_comm {
    ncclComm_t comm;
    int nranks;      // as RCCL reports it (ncclCommCount), not as the caller claimed
    int rank;
    int device;
};
