// gsx_comm.hip -- the gradient exchange for hosts that have no torch.distributed: a thin layer over RCCL (the reference's
// process group, scripts/train.py:65-73, and what its DDP wrapper and trainer do with it: the gradient average after
// backward, the parameter broadcast of training/trainer.py:83-87).  The Python mirror of this package does NOT come
// through here -- it uses torch.distributed, whose "nccl" backend is RCCL -- so librccl is not a load-time dependency of
// the library: it is opened on the first gsx_comm_* call (by soname, so a process that already has RCCL loaded shares it).
// Host code only.  One communicator per process and GPU; collectives are in place, on the caller's stream.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>
#include "gsx_common.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (R.lib) break;
        }
        if (!R.lib) return;
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(dlsym(R.lib, "ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(dlsym(R.lib, "ncclCommInitRank"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.lib, "ncclCommDestroy"));
        R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(dlsym(R.lib, "ncclAllReduce"));
        R.Broadcast = reinterpret_cast<decltype(R.Broadcast)>(dlsym(R.lib, "ncclBroadcast"));
        R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.AllReduce && R.Broadcast;
    });
    return R;
}

struct Comm { ncclComm_t comm; int rank, world; };

inline int nccl_status(ncclResult_t r) { return r == ncclSuccess ? GSX_OK : GSX_E_COMM; }

}  // namespace

GSX_API int gsx_comm_available(void) { return rccl().ok ? 1 : 0; }

GSX_API int gsx_comm_unique_id(void* id_out)
{
    if (!id_out) return GSX_E_INVALID;
    if (!rccl().ok) return GSX_E_COMM;
    static_assert(sizeof(ncclUniqueId) == GSX_COMM_ID_BYTES, "GSX_COMM_ID_BYTES must be RCCL's NCCL_UNIQUE_ID_BYTES");
    return nccl_status(rccl().GetUniqueId(static_cast<ncclUniqueId*>(id_out)));
}

GSX_API int gsx_comm_init(int32_t rank, int32_t world, const void* unique_id, void** comm_out)
{
    if (!unique_id || !comm_out || world < 1 || rank < 0 || rank >= world) return GSX_E_INVALID;
    if (!rccl().ok) return GSX_E_COMM;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    Comm* c = new Comm{nullptr, rank, world};
    const ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);      // binds the calling thread's current device
    if (r != ncclSuccess) { delete c; return GSX_E_COMM; }
    *comm_out = c;
    return GSX_OK;
}

GSX_API int gsx_allreduce_avg(void* comm, float* buf, int64_t n, void* stream)
{
    if (!comm || n < 0 || (n > 0 && !buf)) return GSX_E_INVALID;
    if (n == 0) return GSX_OK;
    Comm* c = static_cast<Comm*>(comm);
    return nccl_status(rccl().AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclAvg, c->comm, static_cast<hipStream_t>(stream)));
}

GSX_API int gsx_bcast(void* comm, float* buf, int64_t n, int32_t root, void* stream)
{
    if (!comm || n < 0 || (n > 0 && !buf)) return GSX_E_INVALID;
    Comm* c = static_cast<Comm*>(comm);
    if (root < 0 || root >= c->world) return GSX_E_INVALID;
    if (n == 0) return GSX_OK;
    return nccl_status(rccl().Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->comm, static_cast<hipStream_t>(stream)));
}

GSX_API int gsx_comm_destroy(void* comm)
{
    if (!comm) return GSX_OK;
    Comm* c = static_cast<Comm*>(comm);
    const ncclResult_t r = rccl().CommDestroy(c->comm);
    delete c;
    return nccl_status(r);
}
