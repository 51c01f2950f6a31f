// X1 -- the one collective of the slab-parallel path: an all-gather of the per-slab result
// vectors over RCCL (xGMI) at the end of a job (SURVEY 8e / 8b `xc_comm_*`).  One process per
// GPU; the 128-byte unique id is created on rank 0 (xc_comm_unique_id) and distributed by the
// launcher (bench.py broadcasts it through the torch.distributed store).  librccl is loaded
// lazily with dlopen so that single-GPU users do not pay for it.
#include "xc_internal.h"
#include <dlfcn.h>
#include <string.h>

namespace xc {
namespace {

// minimal RCCL surface (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*fn_get_id)(UniqueId*);
typedef int (*fn_init_rank)(Comm*, int, UniqueId, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int /*ncclDataType_t*/, Comm, hipStream_t);
typedef int (*fn_destroy)(Comm);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* h = nullptr;
    fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr; fn_errstr errstr = nullptr;
};
Rccl g_rccl;

int load_rccl(xc_ctx* ctx)
{
    if (g_rccl.h) return XC_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(ctx, XC_EHIP, std::string("xc_comm: cannot load librccl: ") + dlerror());
    g_rccl.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.allgather || !g_rccl.destroy)
        return fail(ctx, XC_EHIP, "xc_comm: librccl lacks a required symbol");
    g_rccl.h = h;
    return XC_OK;
}

int rccl_fail(xc_ctx* ctx, int rc, const char* what)
{
    std::string m = std::string("RCCL error in ") + what + ": " + (g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    return fail(ctx, XC_EHIP, m);
}

}  // namespace
}  // namespace xc

using namespace xc;

extern "C" {

int xc_comm_unique_id(xc_ctx* ctx, void* out_id128)
{
    if (!ctx || !out_id128) return fail(ctx, XC_EBADARG, "xc_comm_unique_id: bad arguments");
    int rc = load_rccl(ctx); if (rc != XC_OK) return rc;
    UniqueId id;
    const int r = g_rccl.get_id(&id);
    if (r != 0) return rccl_fail(ctx, r, "ncclGetUniqueId");
    memcpy(out_id128, &id, sizeof(id));
    return XC_OK;
}

int xc_comm_init(xc_ctx* ctx, int nranks, int rank, const void* id128)
{
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, XC_EBADARG, "xc_comm_init: bad arguments");
    if (ctx->comm) return fail(ctx, XC_EBADARG, "xc_comm_init: communicator already initialised");
    int rc = load_rccl(ctx); if (rc != XC_OK) return rc;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hipfail(ctx, e, "hipSetDevice");
    UniqueId id; memcpy(&id, id128, sizeof(id));
    Comm c = nullptr;
    const int r = g_rccl.init_rank(&c, nranks, id, rank);
    if (r != 0) return rccl_fail(ctx, r, "ncclCommInitRank");
    ctx->comm = c; ctx->comm_nranks = nranks; ctx->comm_rank = rank;
    return XC_OK;
}

int xc_comm_allgather_dev(xc_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank)
{
    if (!ctx || !ctx->comm) return fail(ctx, XC_EBADARG, "xc_comm_allgather: no communicator (call xc_comm_init)");
    if (!send || !recv) return fail(ctx, XC_EBADARG, "xc_comm_allgather: NULL buffer");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hipfail(ctx, e, "hipSetDevice");
    const int r = g_rccl.allgather(send, recv, bytes_per_rank, 0 /* ncclInt8 / ncclChar */, ctx->comm, ctx->stream);
    if (r != 0) return rccl_fail(ctx, r, "ncclAllGather");
    return XC_OK;
}

int xc_comm_finalize(xc_ctx* ctx)
{
    if (!ctx) return fail(nullptr, XC_EBADARG, "null context");
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        const int r = g_rccl.destroy(ctx->comm);
        ctx->comm = nullptr;
        if (r != 0) return rccl_fail(ctx, r, "ncclCommDestroy");
    }
    return XC_OK;
}

}  // extern "C"
