// RCCL communicator for the one exchange step inside the network executor: the per-BatchNorm sums of synchronized
// BatchNorm (lbc_net_set_sync_bn).  A ResNet-34 training step issues 80 of these all-reduces (40 BatchNorm sites, forward and backward; <= 1300 floats each), serialized with
// the kernels around them, so they go to RCCL straight from the executor's stream: no host framework in between
// (measured on one MI355X at 32 images/GPU: 2.6 ms of a 5.2 ms step when every reduction calls back into Python).
//
// RCCL is bound at run time (dlopen) instead of at link time: the host process normally has it loaded already
// (torch ships librccl.so and loads it for its "nccl" backend), and the library must stay loadable on machines without RCCL
// (the ABI tests, C hosts that never train data-parallel).  Only the five entry points below are used; their signatures are
// RCCL's public C API (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclCommDestroy, ncclGetErrorString).
#include "lbc_common.hpp"
#include "lbc_hip.h"

#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

namespace {
struct RcclId { char internal[LBC_COMM_ID_BYTES]; };       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;                                     // ncclComm_t
enum { kRcclSuccess = 0, kRcclSum = 0, kRcclFloat32 = 7 };  // ncclSuccess, ncclSum, ncclFloat32

struct Rccl {
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    char why[256] = "";
    Rccl()
    {
        // the copy the process already uses first (two RCCL instances in one process would each bring up their own
        // transports); then the usual sonames
        void* h = nullptr;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) { snprintf(why, sizeof(why), "librccl.so not found (%s)", dlerror()); return; }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(h, "ncclAllReduce"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        ok = GetUniqueId && CommInitRank && AllReduce && CommDestroy && GetErrorString;
        if (!ok) snprintf(why, sizeof(why), "librccl.so lacks an expected entry point");
    }
};
Rccl& rccl()
{
    static Rccl r;
    return r;
}
int rccl_fail(const char* what, int rc)
{
    lbc_set_error("%s: RCCL error %d (%s)", what, rc, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
    return LBC_ELAUNCH;
}
}  // namespace

struct lbc_comm {
    RcclComm comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int lbc_comm_unique_id(unsigned char* id)
{
    LBC_REQUIRE(id, "comm_unique_id: null id");
    LBC_REQUIRE(rccl().ok, "comm_unique_id: %s", rccl().why);
    RcclId u;
    const int rc = rccl().GetUniqueId(&u);
    if (rc != kRcclSuccess) return rccl_fail("comm_unique_id", rc);
    memcpy(id, u.internal, LBC_COMM_ID_BYTES);
    return LBC_OK;
}

int lbc_comm_create(const unsigned char* id, int rank, int world_size, lbc_comm** out)
{
    LBC_REQUIRE(id && out, "comm_create: null argument");
    LBC_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "comm_create: rank %d outside a world of %d", rank, world_size);
    LBC_REQUIRE(rccl().ok, "comm_create: %s", rccl().why);
    RcclId u;
    memcpy(u.internal, id, LBC_COMM_ID_BYTES);
    lbc_comm* c = new lbc_comm;
    c->rank = rank; c->world = world_size;
    const int rc = rccl().CommInitRank(&c->comm, world_size, u, rank);     // collective over the world; binds the current HIP device
    if (rc != kRcclSuccess) { delete c; return rccl_fail("comm_create", rc); }
    *out = c;
    return LBC_OK;
}

void lbc_comm_destroy(lbc_comm* c)
{
    if (!c) return;
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
}

int lbc_comm_world_size(const lbc_comm* c) { return c ? c->world : 0; }

// an lbc_allreduce_fn: ctx = the lbc_comm
int lbc_comm_allreduce_f32(void* ctx, float* buf, int count, lbc_stream_t stream)
{
    lbc_comm* c = static_cast<lbc_comm*>(ctx);
    LBC_REQUIRE(c && c->comm && buf && count > 0, "comm_allreduce: bad argument");
    const int rc = rccl().AllReduce(buf, buf, (size_t)count, kRcclFloat32, kRcclSum, c->comm, (hipStream_t)stream);
    if (rc != kRcclSuccess) return rccl_fail("comm_allreduce", rc);
    return LBC_OK;
}

}  // extern "C"
