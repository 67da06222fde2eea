// comm.hip — built-in RCCL implementation of the all-reduce hook used by sharded global BA
// (SURVEY §8e).  librccl is dlopen'ed on first use so that single-GPU users carry no dependency.
// The only collective of the path: sum-all-reduce of fp64 buffers (camera system S|E, error scalars,
// gathered squared errors), in place, ordered on the caller's HIP stream.
#include <dlfcn.h>

#include "common.h"

namespace {
struct NcclUniqueId {
    char internal[128];
};
typedef void* NcclComm;
typedef int (*fn_get_unique_id)(NcclUniqueId*);
typedef int (*fn_comm_init_rank)(NcclComm*, int, NcclUniqueId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_comm_destroy)(NcclComm);
typedef const char* (*fn_get_error_string)(int);

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_get_error_string get_error_string = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.handle) return PTAM_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        ptam_set_error("cannot dlopen librccl: %s", dlerror());
        return PTAM_E_COMM;
    }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.get_error_string = (fn_get_error_string)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) {
        ptam_set_error("librccl lacks a required symbol");
        dlclose(h);
        return PTAM_E_COMM;
    }
    g_rccl.handle = h;
    return PTAM_OK;
}
const char* rccl_err(int rc) { return g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "?"; }
}   // namespace

struct ptam_rccl {
    NcclComm comm;
    int device;
    int rank, world;
};

extern "C" {

int ptam_rccl_unique_id(uint8_t id_out[128]) {
    ARG_TRY(id_out);
    int rc = rccl_load();
    if (rc) return rc;
    NcclUniqueId id;
    const int e = g_rccl.get_unique_id(&id);
    if (e != 0) {
        ptam_set_error("ncclGetUniqueId failed: %s", rccl_err(e));
        return PTAM_E_COMM;
    }
    std::memcpy(id_out, id.internal, 128);
    return PTAM_OK;
}

int ptam_rccl_create(ptam_ctx* ctx, const uint8_t id[128], int rank, int world, ptam_rccl** out) {
    ARG_TRY(ctx && id && out && world >= 1 && rank >= 0 && rank < world);
    int rc = rccl_load();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    NcclUniqueId uid;
    std::memcpy(uid.internal, id, 128);
    ptam_rccl* c = new ptam_rccl();
    c->device = ctx->device;
    c->rank = rank;
    c->world = world;
    const int e = g_rccl.comm_init_rank(&c->comm, world, uid, rank);
    if (e != 0) {
        ptam_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl_err(e));
        delete c;
        return PTAM_E_COMM;
    }
    *out = c;
    return PTAM_OK;
}

int ptam_rccl_destroy(ptam_rccl* c) {
    if (!c) return PTAM_OK;
    hipSetDevice(c->device);
    if (g_rccl.comm_destroy) g_rccl.comm_destroy(c->comm);
    delete c;
    return PTAM_OK;
}

// signature-compatible with ptam_allreduce_f64_fn: user = ptam_rccl*
int ptam_rccl_allreduce_f64(void* comm, double* dptr, size_t count, void* stream) {
    ptam_rccl* c = (ptam_rccl*)comm;
    if (!c || !dptr) return PTAM_E_ARG;
    if (count == 0) return PTAM_OK;
    const int e = g_rccl.all_reduce(dptr, dptr, count, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->comm, (hipStream_t)stream);
    if (e != 0) {
        ptam_set_error("ncclAllReduce(%zu doubles) failed: %s", count, rccl_err(e));
        return PTAM_E_COMM;
    }
    return PTAM_OK;
}

}   // extern "C"
