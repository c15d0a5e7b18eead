// rsx_comm.hip -- RCCL called from the library: the one exchange of a sharded training step (SURVEY section 8e:
// "an RCCL all-reduce over xGMI on item gradients once per step") without the interpreter in the timed region.
//
// The reference has no multi-device code at all (main.py:24-27 pins one device); what is preserved is its batch
// mean (models/MF.py:105): with user rows sharded, every gradient carries 1 / (sum of the ranks' batches).
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that has already loaded a copy -- torch.distributed's
// "nccl" backend IS RCCL on ROCm and ships its own librccl.so.1 -- gets THAT copy (same soname), so there is one
// collective library per process; a process without torch gets the ROCm one.  Only the classic entry points are used
// (unique id, init rank, all-reduce, reduce-scatter, all-gather, destroy), whose signatures have not changed across
// the 2.x series.  One process per GPU, one communicator per process; the unique id travels through whatever
// bootstrap the host side has (torch.distributed's store in this package: plumbing, 128 bytes once).
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "rsx_common.h"

namespace {

// the part of <rccl/rccl.h> this file needs (kept local so the build does not depend on the header's version)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                       // ncclSuccess = 0
constexpr int kNcclSum = 0, kNcclFloat32 = 7;

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);          // the copy the process already has, if any
    for (int k = 0; h == nullptr && k < 3; ++k) h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) return;
    g_rccl.h = h;
#define RSX_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name))
    RSX_SYM(GetUniqueId, "ncclGetUniqueId");
    RSX_SYM(CommInitRank, "ncclCommInitRank");
    RSX_SYM(CommDestroy, "ncclCommDestroy");
    RSX_SYM(AllReduce, "ncclAllReduce");
    RSX_SYM(ReduceScatter, "ncclReduceScatter");
    RSX_SYM(AllGather, "ncclAllGather");
    RSX_SYM(GetErrorString, "ncclGetErrorString");
#undef RSX_SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce &&
                g_rccl.ReduceScatter && g_rccl.AllGather;
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) {
        rsx_set_error("RCCL (librccl.so.1) could not be loaded: %s", g_rccl.h ? "missing entry points" : dlerror());
        return nullptr;
    }
    return &g_rccl;
}

}  // namespace

struct rsx_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

#define RSX_NCCL(call, what)                                                                              \
    do {                                                                                                  \
        ncclResult_t r__ = (call);                                                                        \
        if (r__ != 0) {                                                                                   \
            rsx_set_error("%s: %s failed: %s", __func__, what, R->GetErrorString ? R->GetErrorString(r__) : "rccl error"); \
            return RSX_E_HIP;                                                                             \
        }                                                                                                 \
    } while (0)

RSX_API int rsx_comm_unique_id(void *id_out)
{
    RSX_CHECK_ARG(id_out != nullptr, "null output");
    const Rccl *R = rccl();
    if (R == nullptr) return RSX_E_HIP;
    ncclUniqueId id;
    RSX_NCCL(R->GetUniqueId(&id), "ncclGetUniqueId");
    static_assert(sizeof(id) == RSX_COMM_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof(id));
    return RSX_OK;
}

RSX_API int rsx_comm_create(const void *id, int rank, int world, rsx_comm **out)
{
    RSX_CHECK_ARG(id != nullptr && out != nullptr, "null pointer");
    RSX_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "rank must be in [0, world)");
    const Rccl *R = rccl();
    if (R == nullptr) return RSX_E_HIP;
    rsx_comm *c = new (std::nothrow) rsx_comm();
    if (c == nullptr) { rsx_set_error("rsx_comm_create: out of memory"); return RSX_E_INVALID; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    (void)hipGetDevice(&c->device);
    ncclResult_t r = R->CommInitRank(&c->comm, world, uid, rank);
    if (r != 0) {
        rsx_set_error("rsx_comm_create: ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
                      R->GetErrorString ? R->GetErrorString(r) : "rccl error");
        delete c;
        return RSX_E_HIP;
    }
    c->rank = rank; c->world = world;
    *out = c;
    return RSX_OK;
}

RSX_API void rsx_comm_destroy(rsx_comm *c)
{
    if (c == nullptr) return;
    const Rccl *R = rccl();
    if (R != nullptr && c->comm != nullptr) (void)R->CommDestroy(c->comm);
    delete c;
}

RSX_API int rsx_comm_info(const rsx_comm *c, int *rank, int *world)
{
    RSX_CHECK_ARG(c != nullptr, "null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return RSX_OK;
}

// in-place sum over the ranks of n floats (asynchronous on `stream`)
int rsx_comm_all_reduce(rsx_comm *c, float *buf, int64_t n, hipStream_t st)
{
    RSX_CHECK_ARG(c != nullptr && buf != nullptr && n >= 0, "bad argument");
    if (n == 0) return RSX_OK;
    const Rccl *R = rccl();
    if (R == nullptr) return RSX_E_HIP;
    RSX_NCCL(R->AllReduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, c->comm, st), "ncclAllReduce");
    return RSX_OK;
}

// buf holds world * n floats; on return buf[rank * n .. (rank + 1) * n) is the sum over the ranks of that slice
int rsx_comm_reduce_scatter(rsx_comm *c, float *buf, int64_t n, hipStream_t st)
{
    RSX_CHECK_ARG(c != nullptr && buf != nullptr && n >= 0, "bad argument");
    if (n == 0) return RSX_OK;
    const Rccl *R = rccl();
    if (R == nullptr) return RSX_E_HIP;
    RSX_NCCL(R->ReduceScatter(buf, buf + (size_t)c->rank * n, (size_t)n, kNcclFloat32, kNcclSum, c->comm, st), "ncclReduceScatter");
    return RSX_OK;
}

// buf holds world * n floats; every rank contributes buf[rank * n ..) and receives all slices, in place
int rsx_comm_all_gather(rsx_comm *c, float *buf, int64_t n, hipStream_t st)
{
    RSX_CHECK_ARG(c != nullptr && buf != nullptr && n >= 0, "bad argument");
    if (n == 0) return RSX_OK;
    const Rccl *R = rccl();
    if (R == nullptr) return RSX_E_HIP;
    RSX_NCCL(R->AllGather(buf + (size_t)c->rank * n, buf, (size_t)n, kNcclFloat32, c->comm, st), "ncclAllGather");
    return RSX_OK;
}

RSX_API int rsx_comm_all_reduce_f32(rsx_comm *c, float *buf_dev, int64_t n, rsx_stream_t stream)
{
    return rsx_comm_all_reduce(c, buf_dev, n, (hipStream_t)stream);
}
