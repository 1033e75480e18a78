// RCCL (xGMI) communicator for the data-parallel training epoch: the gradient buckets of a step are all-reduced from
// inside the executor's own launch loop, on a communication stream next to the backward pass, so that a data-parallel
// step costs the host no more than a single-device step (one call per epoch from Python).
//
// RCCL is NOT a link-time dependency of libfavae_hip.so: the process already holds one (PyTorch ships its own librccl,
// the ROCm tree has another), and two RCCL copies in one process must not be mixed.  The entry points are resolved at
// run time from the copy that is already loaded (dlopen with RTLD_NOLOAD), else from an explicit path, else by name.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <mutex>
#include <new>
#include "common.h"
#include "favae_hip.h"
#include "comm_internal.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl(const char* path) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return FAVAE_OK;
    void* h = nullptr;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);        // the copy this process already uses (PyTorch's)
    if (!h && path && path[0]) h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        favae_set_error("favae_comm: librccl not found (%s)", dlerror());
        return FAVAE_ERR_ARG;
    }
    Rccl r;
    r.lib = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
        favae_set_error("favae_comm: librccl lacks an entry point");
        return FAVAE_ERR_ARG;
    }
    g_rccl = r;
    return FAVAE_OK;
}

#define FAVAE_RCCL(call)                                                                                        \
    do {                                                                                                        \
        ncclResult_t r_ = (call);                                                                               \
        if (r_ != ncclSuccess) {                                                                                \
            favae_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_));            \
            return FAVAE_ERR_HIP;                                                                               \
        }                                                                                                       \
    } while (0)

}  // namespace

struct favae_comm {
    ncclComm_t comm;
    int rank, nranks;
};

extern "C" int favae_comm_unique_id(const char* rccl_path, void* id128) {
    FAVAE_CHECK_ARG(id128, "favae_comm_unique_id: null pointer");
    int rc = load_rccl(rccl_path);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    FAVAE_RCCL(g_rccl.GetUniqueId((ncclUniqueId*)id128));
    return FAVAE_OK;
}

extern "C" int favae_comm_init(favae_comm** out, const char* rccl_path, const void* id128, int rank, int nranks) {
    FAVAE_CHECK_ARG(out && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "favae_comm_init: bad argument");
    int rc = load_rccl(rccl_path);
    if (rc) return rc;
    favae_comm* c = new (std::nothrow) favae_comm();
    FAVAE_CHECK_ARG(c, "favae_comm_init: out of host memory");
    c->rank = rank; c->nranks = nranks; c->comm = nullptr;
    ncclUniqueId id;
    memcpy((void*)&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);      // collective: every rank calls it
    if (r != ncclSuccess) {
        favae_set_error("favae_comm_init: ncclCommInitRank -> %s", g_rccl.GetErrorString(r));
        delete c;
        return FAVAE_ERR_HIP;
    }
    *out = c;
    return FAVAE_OK;
}

extern "C" int favae_comm_destroy(favae_comm* c) {
    if (!c) return FAVAE_OK;
    (void)hipDeviceSynchronize();
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return FAVAE_OK;
}

extern "C" int favae_allreduce_f32(favae_comm* c, float* buf, long long n, int average, void* stream) {
    FAVAE_CHECK_ARG(c && buf && n > 0, "favae_allreduce_f32: bad argument");
    FAVAE_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream));
    return FAVAE_OK;
}

extern "C" int favae_allreduce_f64(favae_comm* c, double* buf, long long n, int average, void* stream) {
    FAVAE_CHECK_ARG(c && buf && n > 0, "favae_allreduce_f64: bad argument");
    FAVAE_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat64, average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream));
    return FAVAE_OK;
}

// ---- internal (vae_model.hip) ----------------------------------------------------------------------------------------
int favae_comm_size(favae_comm* c) { return c->nranks; }
