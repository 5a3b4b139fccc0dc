// tdr_comm_*: the data-parallel exchange of the train step over RCCL / xGMI, behind the C ABI.
// Replaces what the reference gets from DistributedDataParallel (models/base_model.py:76-82: one summed, 1/world
// scaled all-reduce of all net_g gradients per step, plus the constructor's parameter broadcast) and from
// reduce_loss_dict (:361-372: the loss scalar reduced to rank 0).  One process per GPU, one communicator per process.
//
// RCCL is bound at run time (dlopen) instead of link time: a PyTorch-ROCm process already has its own librccl.so.1
// resident, and the communicator must live in THAT copy (one RCCL per process, one set of xGMI rings); a stand-alone
// C or C++ host gets /opt/rocm/lib/librccl.so.1.  Nothing here touches torch.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>

#include "../../include/tdr.h"
#include "tdr_common.h"

namespace {
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    if (r.ok || r.so) return r;
    const char* names[] = {getenv("TDR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // an already-resident copy first (RTLD_NOLOAD), then a fresh load
    for (int pass = 0; pass < 2 && !r.so; ++pass)
        for (const char* n : names)
            if (n && *n && (r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0)))) break;
    if (!r.so) return r;
    auto sym = [&](const char* n) { return dlsym(r.so, n); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Reduce && r.Broadcast && r.GetErrorString;
    return r;
}

#define TDR_RCCL_READY()                                                                          \
    Rccl& R = rccl();                                                                             \
    if (!R.ok) {                                                                                  \
        tdr_set_error("tdr_comm: librccl.so.1 could not be loaded (%s)", R.so ? "missing symbol" : dlerror()); \
        return TDR_ERR_UNSUPPORTED;                                                               \
    }
#define TDR_RCCL_CHECK(call, what)                                                      \
    do {                                                                                \
        ncclResult_t r__ = (call);                                                      \
        if (r__ != ncclSuccess) {                                                       \
            tdr_set_error("%s: %s", what, R.GetErrorString(r__));                       \
            return TDR_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)
}  // namespace

struct TdrComm {
    ncclComm_t comm;
    int rank, world;
};

extern "C" int tdr_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int tdr_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int tdr_comm_unique_id(void* id_out) {
    TDR_REQUIRE(id_out, "tdr_comm_unique_id: null pointer");
    TDR_RCCL_READY();
    ncclUniqueId id;
    TDR_RCCL_CHECK(R.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return TDR_OK;
}

extern "C" int tdr_comm_init(TdrComm** comm, int rank, int world, const void* unique_id) {
    TDR_REQUIRE(comm && unique_id && world >= 1 && rank >= 0 && rank < world, "tdr_comm_init: bad argument");
    TDR_RCCL_READY();
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c;
    TDR_RCCL_CHECK(R.CommInitRank(&c, world, id, rank), "ncclCommInitRank");     // binds to the caller's current HIP device
    *comm = new TdrComm{c, rank, world};
    return TDR_OK;
}

extern "C" int tdr_comm_allreduce(TdrComm* comm, float* buf, int64_t count, int average, void* stream) {
    TDR_REQUIRE(comm && buf && count > 0, "tdr_comm_allreduce: bad argument");
    TDR_RCCL_READY();
    TDR_RCCL_CHECK(R.AllReduce(buf, buf, (size_t)count, ncclFloat32, average ? ncclAvg : ncclSum, comm->comm, (hipStream_t)stream),
                   "ncclAllReduce");
    return TDR_OK;
}

extern "C" int tdr_comm_reduce(TdrComm* comm, float* buf, int64_t count, int root, void* stream) {
    TDR_REQUIRE(comm && buf && count > 0 && root >= 0 && root < comm->world, "tdr_comm_reduce: bad argument");
    TDR_RCCL_READY();
    TDR_RCCL_CHECK(R.Reduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, root, comm->comm, (hipStream_t)stream), "ncclReduce");
    return TDR_OK;
}

extern "C" int tdr_comm_broadcast(TdrComm* comm, float* buf, int64_t count, int root, void* stream) {
    TDR_REQUIRE(comm && buf && count > 0 && root >= 0 && root < comm->world, "tdr_comm_broadcast: bad argument");
    TDR_RCCL_READY();
    TDR_RCCL_CHECK(R.Broadcast(buf, buf, (size_t)count, ncclFloat32, root, comm->comm, (hipStream_t)stream), "ncclBroadcast");
    return TDR_OK;
}

extern "C" int tdr_comm_rank(const TdrComm* comm) { return comm ? comm->rank : -1; }
extern "C" int tdr_comm_world(const TdrComm* comm) { return comm ? comm->world : -1; }

extern "C" int tdr_comm_destroy(TdrComm* comm) {
    if (!comm) return TDR_OK;
    TDR_RCCL_READY();
    ncclResult_t r = R.CommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) {
        tdr_set_error("ncclCommDestroy: %s", R.GetErrorString(r));
        return TDR_ERR_HIP;
    }
    return TDR_OK;
}
