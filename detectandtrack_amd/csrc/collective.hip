// The one exchange step of the path: the gradient all-reduce of data-parallel training (reference: NCCLAllreduce / muji.Allreduce
// over the parameter gradients, lib/modeling/model_builder.py:938-942).  Thin wrappers over RCCL, which is loaded on first use
// (dlopen; an RCCL already mapped by the host -- PyTorch's -- is re-used, so the process never holds two copies).  The host owns
// the rendezvous: rank 0 makes the 128-byte id (dat_comm_unique_id) and ships it to the other ranks however it launches them
// (torch.distributed broadcast, a file, MPI); every rank then calls dat_comm_init_rank.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "dat_common.h"

namespace {

// the slice of rccl.h this file needs (the header is not included: no link-time dependency on RCCL)
typedef struct { char internal[128]; } rccl_unique_id;
typedef void* rccl_comm;
typedef int (*fn_get_unique_id)(rccl_unique_id*);
typedef int (*fn_comm_init_rank)(rccl_comm*, int, rccl_unique_id, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
typedef int (*fn_comm_destroy)(rccl_comm);
typedef const char* (*fn_error_string)(int);
constexpr int RCCL_FLOAT32 = 7, RCCL_SUM = 0;      // ncclFloat32, ncclSum (rccl.h:448-466)

struct Rccl {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
    std::string why;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);       // the host's copy, if mapped
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { r.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return; }
        r.get_unique_id = (fn_get_unique_id)dlsym(r.lib, "ncclGetUniqueId");
        r.comm_init_rank = (fn_comm_init_rank)dlsym(r.lib, "ncclCommInitRank");
        r.all_reduce = (fn_all_reduce)dlsym(r.lib, "ncclAllReduce");
        r.comm_destroy = (fn_comm_destroy)dlsym(r.lib, "ncclCommDestroy");
        r.error_string = (fn_error_string)dlsym(r.lib, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) { r.why = "RCCL symbols missing"; r.lib = nullptr; }
    });
    return &r;
}

#define DAT_RCCL(ctx, r, call, what)                                                                                       \
    do {                                                                                                                   \
        const int rc_ = (call);                                                                                            \
        if (rc_ != 0) DAT_FAIL(ctx, DAT_ERR_LAUNCH, "%s: %s", what, (r)->error_string ? (r)->error_string(rc_) : "RCCL error"); \
    } while (0)

}  // namespace

struct dat_comm {
    rccl_comm comm;
    int nranks, rank;
};

extern "C" {

int dat_comm_unique_id(dat_ctx* ctx, void* id128) {
    DAT_ENFORCE(ctx, id128, "comm_unique_id: null argument");
    Rccl* r = rccl();
    DAT_ENFORCE(ctx, r->lib, "comm_unique_id: %s", r->why.c_str());
    rccl_unique_id id;
    DAT_RCCL(ctx, r, r->get_unique_id(&id), "ncclGetUniqueId");
    memcpy(id128, id.internal, sizeof(id.internal));
    return DAT_OK;
}

int dat_comm_init_rank(dat_ctx* ctx, const void* id128, int nranks, int rank, dat_comm** out) {
    DAT_ENFORCE(ctx, id128 && out && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init_rank: bad argument (rank %d of %d)", rank, nranks);
    Rccl* r = rccl();
    DAT_ENFORCE(ctx, r->lib, "comm_init_rank: %s", r->why.c_str());
    rccl_unique_id id;
    memcpy(id.internal, id128, sizeof(id.internal));
    rccl_comm c = nullptr;
    DAT_RCCL(ctx, r, r->comm_init_rank(&c, nranks, id, rank), "ncclCommInitRank");     // on the calling thread's current device
    *out = new dat_comm{c, nranks, rank};
    return DAT_OK;
}

int dat_allreduce_bucket(dat_ctx* ctx, dat_stream s, dat_comm* comm, float* buf, size_t count) {
    DAT_ENFORCE(ctx, comm && (buf || count == 0), "allreduce_bucket: null argument");
    if (count == 0) return DAT_OK;
    Rccl* r = rccl();
    DAT_RCCL(ctx, r, r->all_reduce(buf, buf, count, RCCL_FLOAT32, RCCL_SUM, comm->comm, (hipStream_t)s), "ncclAllReduce");   // in place
    return DAT_OK;
}

int dat_comm_destroy(dat_comm* comm) {
    if (!comm) return DAT_OK;
    Rccl* r = rccl();
    const int rc = r->lib ? r->comm_destroy(comm->comm) : 0;
    delete comm;
    return rc == 0 ? DAT_OK : DAT_ERR_LAUNCH;
}

}  // extern "C"
