// X1 -- the one collective of the slab-parallel path: the gather of the per-slab result vectors at the end of a job
// (SURVEY 8e / 8b `xc_comm_*`).  One process per GPU.  Two device carriers:
//   * RCCL over xGMI: the 128-byte unique id is created on rank 0 (xc_comm_unique_id) and distributed by the launcher's
//     rendezvous (xcontour_amd.distributed.SocketGroup); ncclAllGather (every rank gets everything) or a gather to ONE root
//     (grouped ncclSend / ncclRecv: 1 / world of the all-gather's traffic).  librccl is loaded lazily with dlopen so that
//     single-GPU users do not pay for it.
//   * HIP IPC (no RCCL needed): the root exports its receive buffer (hipIpcGetMemHandle), every rank opens it and PUSHES its
//     block with a device-to-device copy on its own comm stream (a peer write over xGMI on a real node, a same-device copy when
//     several ranks share one GPU in a rehearsal) -- ordered behind its own kernels by an event, so no cross-process
//     synchronisation is needed until the one rendezvous barrier at the end of the job.
// Both run on the context's COMM stream (a third stream beside compute and upload): a launch set's block leaves while the next
// launch set computes, and only the last set's block is exposed (xc_comm_wait_compute / xc_compute_wait_comm are the fences).
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
typedef int (*fn_sendrecv)(void*, size_t, int /*ncclDataType_t*/, int /*peer*/, Comm, hipStream_t);
typedef int (*fn_group)(void);
typedef int (*fn_comm_int)(const Comm, int*);
typedef int (*fn_version)(int*);

struct Rccl {
    void* h = nullptr;
    fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr; fn_errstr errstr = nullptr;
    fn_sendrecv send = nullptr, recv = nullptr; fn_group group_start = nullptr, group_end = nullptr; fn_destroy abort = nullptr;
    fn_comm_int count = nullptr, user_rank = nullptr, cu_device = nullptr; fn_version version = nullptr;
    char path[512] = {0};          // the file the symbols came from (dladdr)
};
Rccl g_rccl;

int load_rccl(xc_ctx* ctx)
{
    if (g_rccl.h) return XC_OK;
    // The librccl that belongs to the HIP runtime THIS library is bound to: a process may hold two ROCm stacks (PyTorch wheels bundle
    // their own libamdhip64 / libhsa-runtime64 / librccl; whichever libamdhip64 was loaded first serves this library), and a librccl
    // bound to the OTHER runtime sees no initialised device (round 6: ncclCommInitRank "no ROCm-capable device is detected" when torch
    // was imported after the context was created).  So: the directory of the libamdhip64 our own HIP calls resolve to comes first.
    void* h = nullptr;
    {
        Dl_info di;
        if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t sl = dir.rfind('/');
            if (sl != std::string::npos) {
                dir.resize(sl + 1);
                h = dlopen((dir + "librccl.so.1").c_str(), RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen((dir + "librccl.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
            }
        }
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(ctx, XC_EHIP, std::string("xc_comm: cannot load librccl: ") + dlerror());
    g_rccl.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    g_rccl.send = (fn_sendrecv)dlsym(h, "ncclSend"); g_rccl.recv = (fn_sendrecv)dlsym(h, "ncclRecv");
    g_rccl.group_start = (fn_group)dlsym(h, "ncclGroupStart"); g_rccl.group_end = (fn_group)dlsym(h, "ncclGroupEnd");
    g_rccl.abort = (fn_destroy)dlsym(h, "ncclCommAbort");
    g_rccl.count = (fn_comm_int)dlsym(h, "ncclCommCount"); g_rccl.user_rank = (fn_comm_int)dlsym(h, "ncclCommUserRank");
    g_rccl.cu_device = (fn_comm_int)dlsym(h, "ncclCommCuDevice"); g_rccl.version = (fn_version)dlsym(h, "ncclGetVersion");
    { Dl_info di; if (g_rccl.get_id && dladdr((void*)g_rccl.get_id, &di) && di.dli_fname) snprintf(g_rccl.path, sizeof(g_rccl.path), "%s", di.dli_fname); }
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

// the comm stream and its two fence events, created on first use
int ensure_comm_stream(xc_ctx* ctx)
{
    if (ctx->comm_stream) return XC_OK;
    XC_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    XC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_in, hipEventDisableTiming));
    XC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_out, hipEventDisableTiming));
    return XC_OK;
}

}  // namespace
}  // namespace xc

using namespace xc;

#define XC_COMM_CTX(ctx) do { if (!(ctx)) return fail(nullptr, XC_EBADARG, "null context"); \
                              hipError_t _e = hipSetDevice((ctx)->device); \
                              if (_e != hipSuccess) return hipfail((ctx), _e, "hipSetDevice"); } while (0)

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

// The communicator is CREATED without touching any context (ncclCommInitRank blocks until every rank has joined -- forever, if one
// never does -- so callers run it in a helper thread under a deadline: that thread must not write into a context the main thread
// keeps using) and ATTACHED to a context by the thread that owns the context, once the call has returned.
int xc_comm_create(int device, int nranks, int rank, const void* id128, void** out_comm, char* err, size_t errlen)
{
    auto say = [&](const std::string& m) { if (err && errlen) snprintf(err, errlen, "%s", m.c_str()); };
    if (err && errlen) err[0] = 0;
    if (!id128 || !out_comm || nranks < 1 || rank < 0 || rank >= nranks) { say("xc_comm_create: bad arguments"); return XC_EBADARG; }
    *out_comm = nullptr;
    if (!g_rccl.h) {
        xc_ctx tmp;                                            // (load_rccl reports through a context: a throw-away one)
        const int rc = load_rccl(&tmp);
        if (rc != XC_OK) { say(tmp.err); return rc; }
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { say(std::string("hipSetDevice: ") + hipGetErrorString(e)); return XC_EHIP; }
    UniqueId id; memcpy(&id, id128, sizeof(id));
    Comm c = nullptr;
    // RCCL checks the process-wide "last error" of the HIP runtime in places: an error that some EARLIER call of this process returned
    // (and its caller handled) must not fail the communicator -- it is cleared here, and named if the call fails all the same
    const hipError_t stale = hipGetLastError();
    const int r = g_rccl.init_rank(&c, nranks, id, rank);
    if (r != 0) {
        std::string m = std::string("RCCL error in ncclCommInitRank: ") + (g_rccl.errstr ? g_rccl.errstr(r) : "?");
        if (stale != hipSuccess) m += std::string(" (pending before the call: ") + hipGetErrorName(stale) + ")";
        const hipError_t now = hipGetLastError();
        if (now != hipSuccess) m += std::string(" (HIP last error after it: ") + hipGetErrorName(now) + ")";
        say(m); return XC_EHIP;
    }
    *out_comm = c;
    return XC_OK;
}

int xc_comm_attach(xc_ctx* ctx, void* comm, int nranks, int rank)
{
    if (!ctx || !comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, XC_EBADARG, "xc_comm_attach: bad arguments");
    if (ctx->comm) return fail(ctx, XC_EBADARG, "xc_comm_attach: the context already has a communicator");
    ctx->comm = comm; ctx->comm_nranks = nranks; ctx->comm_rank = rank;
    return XC_OK;
}

int xc_comm_release(void* comm)                       // a communicator that was never attached (its creator gave up waiting for it)
{
    if (!comm) return XC_OK;
    if (g_rccl.abort) return g_rccl.abort(comm) == 0 ? XC_OK : XC_EHIP;
    return g_rccl.destroy && g_rccl.destroy(comm) == 0 ? XC_OK : XC_EHIP;
}

int xc_comm_init(xc_ctx* ctx, int nranks, int rank, const void* id128)
{
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, XC_EBADARG, "xc_comm_init: bad arguments");
    if (ctx->comm) return fail(ctx, XC_EBADARG, "xc_comm_init: communicator already initialised");
    void* c = nullptr; char err[512];
    const int rc = xc_comm_create(ctx->device, nranks, rank, id128, &c, err, sizeof(err));
    if (rc != XC_OK) return fail(ctx, rc, err);
    return xc_comm_attach(ctx, c, nranks, rank);
}

// What RCCL ITSELF says about the communicator of this context: how many ranks it spans, which of them this process is, which device
// it drives, the library's version and file -- read back from the library (ncclCommCount / ncclCommUserRank / ncclCommCuDevice /
// ncclGetVersion), not from what the caller passed in.  comm_count = 0: the context has no RCCL communicator (another carrier).
int xc_comm_info(xc_ctx* ctx, xc_comm_info_t* out)
{
    if (!ctx || !out) return fail(ctx, XC_EBADARG, "xc_comm_info: bad arguments");
    memset(out, 0, sizeof(*out));
    out->comm_rank = -1; out->comm_device = -1;
    out->ctx_device = ctx->device;
    if (g_rccl.h) {
        if (g_rccl.version) (void)g_rccl.version(&out->rccl_version);
        snprintf(out->rccl_path, sizeof(out->rccl_path), "%s", g_rccl.path);
    }
    if (!ctx->comm) return XC_OK;
    int v = 0;
    if (g_rccl.count && g_rccl.count(ctx->comm, &v) == 0) out->comm_count = v;
    if (g_rccl.user_rank && g_rccl.user_rank(ctx->comm, &v) == 0) out->comm_rank = v;
    if (g_rccl.cu_device && g_rccl.cu_device(ctx->comm, &v) == 0) out->comm_device = v;
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

// ---- gather to ONE root over RCCL: grouped ncclSend (every other rank) / ncclRecv (the root, one per peer) on the comm stream.
// The root's own block is copied device to device.  recv + r * rank_stride is where rank r's `bytes` land (root only).
int xc_comm_gather_dev(xc_ctx* ctx, const void* send, size_t bytes, void* recv, size_t rank_stride, int root)
{
    XC_COMM_CTX(ctx);
    if (!ctx->comm) return fail(ctx, XC_EBADARG, "xc_comm_gather: no communicator (call xc_comm_init)");
    if (!g_rccl.send || !g_rccl.recv || !g_rccl.group_start || !g_rccl.group_end)
        return fail(ctx, XC_EHIP, "xc_comm_gather: librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
    if (!send || root < 0 || root >= ctx->comm_nranks || (ctx->comm_rank == root && !recv) || rank_stride < bytes)
        return fail(ctx, XC_EBADARG, "xc_comm_gather: bad arguments");
    { const int rc = ensure_comm_stream(ctx); if (rc != XC_OK) return rc; }
    if (bytes == 0) return XC_OK;
    if (ctx->comm_rank != root) {
        const int r = g_rccl.send(const_cast<void*>(send), bytes, 0 /* ncclInt8 */, root, ctx->comm, ctx->comm_stream);
        if (r != 0) return rccl_fail(ctx, r, "ncclSend");
        return XC_OK;
    }
    XC_HIP(ctx, hipMemcpyAsync((char*)recv + (size_t)root * rank_stride, send, bytes, hipMemcpyDeviceToDevice, ctx->comm_stream));
    int r = g_rccl.group_start();
    if (r != 0) return rccl_fail(ctx, r, "ncclGroupStart");
    for (int p = 0; p < ctx->comm_nranks; ++p) {
        if (p == root) continue;
        r = g_rccl.recv((char*)recv + (size_t)p * rank_stride, bytes, 0, p, ctx->comm, ctx->comm_stream);
        if (r != 0) { (void)g_rccl.group_end(); return rccl_fail(ctx, r, "ncclRecv"); }
    }
    r = g_rccl.group_end();
    if (r != 0) return rccl_fail(ctx, r, "ncclGroupEnd");
    return XC_OK;
}

// ---- HIP IPC: a device allocation of one process mapped into another (no RCCL involved)
int xc_ipc_export(xc_ctx* ctx, const void* dptr, void* out_handle64)
{
    XC_COMM_CTX(ctx);
    if (!dptr || !out_handle64) return fail(ctx, XC_EBADARG, "xc_ipc_export: bad arguments");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    XC_HIP(ctx, hipIpcGetMemHandle(&h, const_cast<void*>(dptr)));
    memcpy(out_handle64, &h, sizeof(h));
    return XC_OK;
}

int xc_ipc_open(xc_ctx* ctx, const void* handle64, void** out_dptr)
{
    XC_COMM_CTX(ctx);
    if (!handle64 || !out_dptr) return fail(ctx, XC_EBADARG, "xc_ipc_open: bad arguments");
    hipIpcMemHandle_t h; memcpy(&h, handle64, sizeof(h));
    *out_dptr = nullptr;
    XC_HIP(ctx, hipIpcOpenMemHandle(out_dptr, h, hipIpcMemLazyEnablePeerAccess));
    return XC_OK;
}

int xc_ipc_close(xc_ctx* ctx, void* dptr)
{
    XC_COMM_CTX(ctx);
    if (!dptr) return XC_OK;
    if (ctx->comm_stream) XC_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));      // a push may still be writing through the mapping
    XC_HIP(ctx, hipIpcCloseMemHandle(dptr));
    return XC_OK;
}

// ---- the comm stream: device-to-device pushes and the fences against the compute stream
int xc_comm_memcpy_d2d(xc_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    XC_COMM_CTX(ctx);
    if (bytes && (!dst || !src)) return fail(ctx, XC_EBADARG, "xc_comm_memcpy_d2d: NULL pointer");
    { const int rc = ensure_comm_stream(ctx); if (rc != XC_OK) return rc; }
    if (bytes) XC_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->comm_stream));
    return XC_OK;
}

int xc_comm_wait_compute(xc_ctx* ctx)          // later comm-stream work waits for the compute work enqueued so far
{
    XC_COMM_CTX(ctx);
    { const int rc = ensure_comm_stream(ctx); if (rc != XC_OK) return rc; }
    XC_HIP(ctx, hipEventRecord(ctx->ev_comm_in, ctx->stream));
    XC_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_comm_in, 0));
    return XC_OK;
}

int xc_compute_wait_comm(xc_ctx* ctx)          // later compute-stream work (and xc_sync) waits for the comm work enqueued so far
{
    XC_COMM_CTX(ctx);
    { const int rc = ensure_comm_stream(ctx); if (rc != XC_OK) return rc; }
    XC_HIP(ctx, hipEventRecord(ctx->ev_comm_out, ctx->comm_stream));
    XC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_comm_out, 0));
    return XC_OK;
}

// 1 once everything enqueued on the compute AND comm streams has finished, 0 while it has not: a caller that must not block
// forever (the first contact with a collective) polls this against its own deadline instead of calling xc_sync
int xc_streams_idle(xc_ctx* ctx, int* out_idle)
{
    XC_COMM_CTX(ctx);
    if (!out_idle) return fail(ctx, XC_EBADARG, "xc_streams_idle: out is NULL");
    *out_idle = 0;
    hipError_t e = hipStreamQuery(ctx->stream);
    if (e == hipSuccess && ctx->comm_stream) e = hipStreamQuery(ctx->comm_stream);
    if (e == hipSuccess) { *out_idle = 1; return XC_OK; }
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return XC_OK; }
    return hipfail(ctx, e, "hipStreamQuery");
}

int xc_device_can_access_peer(int device, int peer, int* out_can)
{
    if (!out_can) return fail(nullptr, XC_EBADARG, "xc_device_can_access_peer: out is NULL");
    *out_can = 0;
    if (device == peer) { *out_can = 1; return XC_OK; }
    int c = 0;
    if (hipDeviceCanAccessPeer(&c, device, peer) != hipSuccess) { (void)hipGetLastError(); c = 0; }
    *out_can = c;
    return XC_OK;
}

// give up on a communicator whose collective does not finish (ncclCommAbort frees it without waiting for its kernels)
int xc_comm_abort(xc_ctx* ctx)
{
    if (!ctx) return fail(nullptr, XC_EBADARG, "null context");
    if (ctx->comm) {
        Comm c = ctx->comm; ctx->comm = nullptr;
        if (g_rccl.abort) { const int r = g_rccl.abort(c); if (r != 0) return rccl_fail(ctx, r, "ncclCommAbort"); }
    }
    return XC_OK;
}

int xc_comm_finalize(xc_ctx* ctx)
{
    if (!ctx) return fail(nullptr, XC_EBADARG, "null context");
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
        const int r = g_rccl.destroy(ctx->comm);
        ctx->comm = nullptr;
        if (r != 0) return rccl_fail(ctx, r, "ncclCommDestroy");
    }
    return XC_OK;
}

}  // extern "C"
