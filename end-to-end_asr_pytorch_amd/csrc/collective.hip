// Gradient all-reduce for the data-parallel step, callable from the C launch loop (host code + one marker kernel; the reference is
// single-GPU and has no counterpart - SURVEY §8(e): each rank runs the whole step on its shard of the batch, gradients are summed).
//
// Two pieces:
//  * RCCL through its C API, bound at run time (dlopen / dlsym: the library torch already has in the process, so that there is one
//    RCCL and one set of its proxy threads per rank).  A communicator is made from a 128-byte unique id that rank 0 generates and the
//    caller carries to the other ranks (torch.distributed's store / a gloo broadcast: bootstrap only, never on the data path).
//  * asr_collective_mark: a one-thread kernel that does nothing, launched where a bucket of the flat gradient buffer is final.  In an
//    eagerly queued step it is followed by the all-reduce itself (asr_rccl_all_reduce_f32 on the same stream).  In a CAPTURED step it
//    is all there is: csrc/graph_exec.hip recognises the node by its function, keeps (buffer, count) from its argument block, and its
//    launch loop calls the all-reduce on that node's stream - the collective is part of the C loop, with the graph's own edges as its
//    ordering, and no Python runs between the step's first launch and its last.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and enums only: every function is looked up with dlsym
#include <stdio.h>
#include <string.h>

#include "asr_common.h"

__global__ void asr_collective_marker_kernel(float* buf, long long count, int tag) {
    (void)buf;
    (void)count;
    (void)tag;
}

const void* asr_collective_marker_func() { return reinterpret_cast<const void*>(&asr_collective_marker_kernel); }

namespace {

struct RcclApi {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
} api;

template <typename F>
bool bind(F& fn, const char* name) {
    fn = reinterpret_cast<F>(dlsym(api.dl, name));
    return fn != nullptr;
}

#define RCCL_CHECK(call)                                                                                          \
    do {                                                                                                          \
        ncclResult_t r__ = (call);                                                                                \
        if (r__ != ncclSuccess) {                                                                                 \
            asr_set_error("rccl: %s failed: %s", #call, api.GetErrorString ? api.GetErrorString(r__) : "?");      \
            return -20 - (int)r__;                                                                                \
        }                                                                                                         \
    } while (0)

}  // namespace

// path: the librccl to bind (NULL: the one already mapped into the process, else the loader's search path).  Idempotent.
extern "C" int asr_rccl_load(const char* path) {
    if (api.dl) return 0;
    const char* names[] = {path, "librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !api.dl; ++pass)
        for (const char* nm : names) {
            if (!nm) continue;
            api.dl = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (api.dl) break;
        }
    ASR_REQUIRE(api.dl, -1, "rccl_load: librccl not found (%s)", dlerror());
    const bool ok = bind(api.GetUniqueId, "ncclGetUniqueId") && bind(api.CommInitRank, "ncclCommInitRank") &&
                    bind(api.CommDestroy, "ncclCommDestroy") && bind(api.AllReduce, "ncclAllReduce") &&
                    bind(api.GetErrorString, "ncclGetErrorString");
    bind(api.CommAbort, "ncclCommAbort");
    bind(api.CommGetAsyncError, "ncclCommGetAsyncError");
    bind(api.GetVersion, "ncclGetVersion");
    if (!ok) {
        dlclose(api.dl);
        api = RcclApi();
        asr_set_error("rccl_load: the library lacks the collective entry points");
        return -2;
    }
    return 0;
}

extern "C" int asr_rccl_version(int* version) {
    ASR_REQUIRE(api.dl && api.GetVersion && version, -1, "rccl_version: library not loaded");
    RCCL_CHECK(api.GetVersion(version));
    return 0;
}

// out_id: 128 bytes (ASR_RCCL_ID_BYTES), generated on ONE rank and carried to the others by the caller.
extern "C" int asr_rccl_unique_id(void* out_id) {
    ASR_REQUIRE(api.dl && out_id, -1, "rccl_unique_id: library not loaded");
    ncclUniqueId id;
    RCCL_CHECK(api.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    memcpy(out_id, &id, sizeof(id));
    return 0;
}

// Collective over all ranks (blocks until every rank has called it).  The communicator is bound to the CURRENT device of the caller.
extern "C" int asr_rccl_comm_create(const void* id, int nranks, int rank, void** out_comm) {
    ASR_REQUIRE(api.dl && id && out_comm && nranks >= 1 && rank >= 0 && rank < nranks, -1, "rccl_comm_create: bad arguments");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    RCCL_CHECK(api.CommInitRank(&c, nranks, uid, rank));
    *out_comm = c;
    return 0;
}

extern "C" int asr_rccl_comm_destroy(void* comm) {
    if (!comm || !api.dl) return 0;
    RCCL_CHECK(api.CommDestroy(static_cast<ncclComm_t>(comm)));
    return 0;
}

// Tear the communicator down WITHOUT waiting for outstanding collectives (ncclCommAbort): peers blocked in a collective with this rank
// get an asynchronous error.  For a rank that cannot complete a step it has partly queued (asr_graphx_launch does this itself).
extern "C" int asr_rccl_comm_abort(void* comm) {
    if (!comm || !api.dl) return 0;
    if (!api.CommAbort) return asr_rccl_comm_destroy(comm);
    RCCL_CHECK(api.CommAbort(static_cast<ncclComm_t>(comm)));
    return 0;
}

// buf <- sum over ranks of buf (f32, in place), queued on `stream`; returns once it is queued.
extern "C" int asr_rccl_all_reduce_f32(void* comm, float* buf, long long count, void* stream) {
    ASR_REQUIRE(api.dl && comm && buf && count > 0, -1, "rccl_all_reduce: bad arguments");
    RCCL_CHECK(api.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
    return 0;
}

// 0: healthy; otherwise the communicator's asynchronous error (a peer died, a link error) with the message in asr_last_error.
extern "C" int asr_rccl_comm_check(void* comm) {
    ASR_REQUIRE(api.dl && comm, -1, "rccl_comm_check: bad arguments");
    if (!api.CommGetAsyncError) return 0;
    ncclResult_t async = ncclSuccess;
    RCCL_CHECK(api.CommGetAsyncError(static_cast<ncclComm_t>(comm), &async));
    if (async != ncclSuccess && async != ncclInProgress) {
        asr_set_error("rccl: asynchronous error on the communicator: %s", api.GetErrorString(async));
        return -20 - (int)async;
    }
    return 0;
}

// The ready point of a gradient bucket: buf[0..count) is final on `stream` once this node has run (see the file comment).
extern "C" int asr_collective_mark(float* buf, long long count, int tag, void* stream) {
    ASR_REQUIRE(buf && count > 0, -1, "collective_mark: bad arguments");
    hipLaunchKernelGGL(asr_collective_marker_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), buf, count, tag);
    ASR_LAUNCH_CHECK("asr_collective_mark");
    return 0;
}
