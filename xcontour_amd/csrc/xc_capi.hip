// C ABI of libxcontour_hip.so (declared in include/xcontour_hip.h): context, device
// memory, HIP-event timing, and the orchestration of the kernels in xc_hist.hip,
// xc_misc.hip, xc_lwa.hip.  No C++ type or exception crosses this boundary.
#include "xc_internal.h"
#include <string.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <new>
#include <mutex>
#include <set>
#include <utility>

namespace xc {

static thread_local std::string g_err;   // errors without a context (xc_create)

int fail(xc_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) {
        ctx->err = msg;
        // a call that fails delivers nothing: results still parked in the pinned output buffer must not reach arrays the caller may
        // free once it has seen the error
        ctx->pending_out.clear(); ctx->pending_in.clear(); ctx->pin_in_off = 0; ctx->pin_out_off = 0;
        // (an entry of the small-input cache filled during the failed call may never have been uploaded: forget what this call staged)
        for (auto& e : ctx->small_in) if (e.epoch == ctx->small_epoch) { e.host.clear(); e.epoch = ~0ull; }
        ++ctx->small_epoch;
    } else g_err = msg;
    return code;
}

int hipfail(xc_ctx* ctx, hipError_t e, const char* what)
{
    std::string m = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    (void)hipGetLastError();
    return fail(ctx, e == hipErrorOutOfMemory ? XC_ENOMEM : XC_EHIP, m);
}

static int grow(xc_ctx* ctx, void** p, size_t* have, size_t need)
{
    if (need <= *have) return XC_OK;
    size_t want = *have ? *have : (size_t)1 << 20;
    while (want < need) want *= 2;
    if (*p) {
        XC_HIP(ctx, hipStreamSynchronize(ctx->stream));    // nothing in flight may still use the old block
        XC_HIP(ctx, hipFree(*p));
        *p = nullptr; *have = 0;
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) { want = need; e = hipMalloc(p, want); }
    if (e != hipSuccess) return hipfail(ctx, e, "hipMalloc(scratch)");
    *have = want;
    return XC_OK;
}

int ensure_scratch(xc_ctx* ctx, size_t bytes) { return grow(ctx, &ctx->scratch, &ctx->scratch_bytes, bytes); }
int ensure_arena(xc_ctx* ctx, size_t bytes)   { return grow(ctx, &ctx->arena, &ctx->arena_bytes, bytes); }
int ensure_big(xc_ctx* ctx, size_t bytes)     { return grow(ctx, &ctx->big, &ctx->big_bytes, bytes); }

int ensure_ones(xc_ctx* ctx, size_t n)
{
    if (n <= ctx->ones_n) return XC_OK;
    if (ctx->ones) { XC_HIP(ctx, hipStreamSynchronize(ctx->stream)); XC_HIP(ctx, hipFree(ctx->ones)); ctx->ones = nullptr; ctx->ones_n = 0; }
    size_t want = 4096; while (want < n) want *= 2;
    XC_HIP(ctx, hipMalloc((void**)&ctx->ones, want * sizeof(double)));
    std::vector<double> h(want, 1.0);
    XC_HIP(ctx, hipMemcpy(ctx->ones, h.data(), want * sizeof(double), hipMemcpyHostToDevice));
    ctx->ones_n = want;
    return XC_OK;
}

int ensure_big_lds(xc_ctx* ctx, const void* kernel, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(kernel, ctx->device);
    if (done.count(key)) return XC_OK;
    XC_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert(key);
    return XC_OK;
}

// chained min/max partials (xc_keff_desc.q_next) describe the bytes [mm_q, mm_q + nslab*ny*nx*esize): any write into
// that range through the library, or freeing it, drops them
static void mm_touch(xc_ctx* ctx, const void* p, size_t bytes)
{
    if (!ctx->mm_valid || !p) return;
    const char* a0 = (const char*)ctx->mm_q;
    const char* a1 = a0 + (size_t)ctx->mm_nslab * ctx->mm_ny * ctx->mm_nx * (ctx->mm_dtype == XC_F32 ? 4 : 8);
    const char* b0 = (const char*)p;
    if (b0 < a1 && b0 + (bytes ? bytes : 1) > a0) ctx->mm_valid = 0;
}

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
static inline size_t esize(int dtype) { return dtype == XC_F32 ? 4 : 8; }

// bump allocator over the staging arena
struct Stage {
    xc_ctx* ctx; char* base; size_t off = 0;
    explicit Stage(xc_ctx* c) : ctx(c), base((char*)c->arena) {}
    void* take(size_t bytes) { void* p = base + off; off += al(bytes); return p; }
};

// device mirror of the host bytes [h, h + n), if the caller registered an array that contains them (xc_keep_resident)
static const void* resident_lookup(const xc_ctx* ctx, const void* h, size_t n)
{
    const char* p = (const char*)h;
    for (const auto& e : ctx->resident)
        if (p >= e.host && p + n <= e.host + e.bytes) return (const char*)e.dev + (p - e.host);
    return nullptr;
}

constexpr size_t kPinBytes = (size_t)4 << 20;        // each bounce buffer
constexpr size_t kPinSmall = (size_t)1 << 20;        // transfers up to this size take the bounce buffers
constexpr size_t kCopyKernelMax = (size_t)64 << 10;  // ... and INPUTS up to this size are moved by k_copy_small, up to eight arrays per launch, instead of one DMA copy each (results: knobs.copy_out_kb)

static inline double now_s()
{
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int ensure_pins(xc_ctx* ctx)
{
    if (ctx->pin_in) return XC_OK;
    // coherent (fine-grained) on request, not by the runtime's default: kernels read pin_in and write pin_out themselves (k_copy_small, the
    // direct results), and what the host sees after the stream wait must not depend on HIP_HOST_COHERENT
    XC_HIP(ctx, hipHostMalloc((void**)&ctx->pin_in, kPinBytes, hipHostMallocCoherent));
    hipError_t e = hipHostMalloc((void**)&ctx->pin_out, kPinBytes, hipHostMallocCoherent);
    if (e != hipSuccess) { (void)hipHostFree(ctx->pin_in); ctx->pin_in = nullptr; return hipfail(ctx, e, "hipHostMalloc"); }
    return XC_OK;
}

// host bytes -> device on the compute stream.  Small blocks: memcpy into the pinned input buffer + an asynchronous copy (the slot is
// free again after the call's xc_sync); larger ones: the runtime's own staged copy straight from the caller's memory.
static int h2d_raw(xc_ctx* ctx, void* d, const void* h, size_t n)
{
    if (n <= kPinSmall && ensure_pins(ctx) == XC_OK && ctx->pin_in_off + n <= kPinBytes) {
        char* p = ctx->pin_in + ctx->pin_in_off;
        ctx->pin_in_off += (n + 63) & ~(size_t)63;
        memcpy(p, h, n);
        if (ctx->knobs.copy_kernel && n <= kCopyKernelMax) { if (n) ctx->pending_in.push_back({d, p, n}); return XC_OK; }   // leaves with flush_in
        XC_HIP(ctx, hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, ctx->stream));
        return XC_OK;
    }
    XC_HIP(ctx, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->stream));
    return XC_OK;
}

// the staged small inputs go to the device now: called by every host-form entry point between its staging and its first launch (and by
// xc_sync, so that nothing staged can outlive the call)
static int flush_in(xc_ctx* ctx)
{
    int rc = XC_OK;
    for (size_t i = 0; i < ctx->pending_in.size() && rc == XC_OK; i += 8) {
        SmallCopies c; int m = 0;
        for (; m < 8 && i + m < ctx->pending_in.size(); ++m) {
            const auto& e = ctx->pending_in[i + m];
            c.src[m] = e.pinned; c.dst[m] = e.dev; c.bytes[m] = (unsigned)e.bytes;
        }
        for (int k = m; k < 8; ++k) { c.src[k] = nullptr; c.dst[k] = nullptr; c.bytes[k] = 0; }
        rc = launch_copy_small(ctx, c, m);
    }
    ctx->pending_in.clear();
    return rc;
}
// ... and the small results the call has asked for leave the device in one launch (xc_sync, before it waits for the stream)
static int flush_out(xc_ctx* ctx)
{
    SmallCopies c; int m = 0, rc = XC_OK;
    for (auto& po : ctx->pending_out) {
        if (!po.dev) continue;
        c.src[m] = po.dev; c.dst[m] = const_cast<void*>(po.pinned); c.bytes[m] = (unsigned)po.bytes; po.dev = nullptr;
        if (++m == 8) { if (rc == XC_OK) rc = launch_copy_small(ctx, c, m); m = 0; }
    }
    if (m) {
        for (int k = m; k < 8; ++k) { c.src[k] = nullptr; c.dst[k] = nullptr; c.bytes[k] = 0; }
        if (rc == XC_OK) rc = launch_copy_small(ctx, c, m);
    }
    return rc;
}

// every host-form entry point stages its inputs through here: bytes that have a device mirror are copied from the mirror
// (device to device, ~50 us for a cfg2 slab) instead of crossing PCIe again (~1 ms)
static int h2d(xc_ctx* ctx, void* d, const void* h, size_t n)
{
    const double t0 = now_s();
    int rc = XC_OK;
    const void* m = ctx->resident.empty() ? nullptr : resident_lookup(ctx, h, n);
    if (m) { hipError_t e = hipMemcpyAsync(d, m, n, hipMemcpyDeviceToDevice, ctx->stream); if (e != hipSuccess) rc = hipfail(ctx, e, "hipMemcpyAsync(d2d)"); }
    else rc = h2d_raw(ctx, d, h, n);
    ctx->tr_h2d += now_s() - t0;
    return rc;
}
// the big read-only inputs of the Keff sequence skip even that copy: the kernel reads the mirror itself (`slot`: the arena bytes
// reserved for the upload, unused then)
static int stage_in(xc_ctx* ctx, void* slot, const void* h, size_t n, const void** dev)
{
    if (!ctx->resident.empty())
        if (const void* m = resident_lookup(ctx, h, n)) { *dev = m; return XC_OK; }
    *dev = slot;
    const double t0 = now_s();
    const int rc = h2d_raw(ctx, slot, h, n);
    ctx->tr_h2d += now_s() - t0;
    return rc;
}
// a SMALL read-only input (see xc_ctx::SmallIn): the device copy of an earlier call when the bytes are the same, else an upload into a cache
// entry (through the pinned buffer and the copy kernel, like every small input) that later calls can hit
static int stage_small(xc_ctx* ctx, void* slot, const void* h, size_t n, const void** dev)
{
    if (!ctx->resident.empty())
        if (const void* m = resident_lookup(ctx, h, n)) { *dev = m; return XC_OK; }
    if (ctx->knobs.copy_kernel && n > 0 && n <= kCopyKernelMax) {
        const double t0 = now_s();
        xc_ctx::SmallIn* lru = nullptr;
        for (auto& e : ctx->small_in) {
            if (e.dev && e.host.size() == n && memcmp(e.host.data(), h, n) == 0) {
                e.used = ++ctx->small_clock; e.epoch = ctx->small_epoch; ++ctx->small_hits;
                *dev = e.dev; ctx->tr_h2d += now_s() - t0;
                return XC_OK;
            }
            if (e.epoch != ctx->small_epoch && (!lru || e.used < lru->used)) lru = &e;
        }
        if (lru && ensure_pins(ctx) == XC_OK && ctx->pin_in_off + n <= kPinBytes) {
            if (!lru->dev) { hipError_t he = hipMalloc(&lru->dev, kCopyKernelMax); if (he != hipSuccess) { lru->dev = nullptr; (void)hipGetLastError(); } }
            if (lru->dev) {
                lru->host.assign((const char*)h, (const char*)h + n);
                lru->used = ++ctx->small_clock; lru->epoch = ctx->small_epoch; ++ctx->small_misses;
                char* p = ctx->pin_in + ctx->pin_in_off;
                ctx->pin_in_off += (n + 63) & ~(size_t)63;
                memcpy(p, h, n);
                ctx->pending_in.push_back({lru->dev, p, n});
                *dev = lru->dev; ctx->tr_h2d += now_s() - t0;
                return XC_OK;
            }
        }
    }
    *dev = slot;
    const double t0 = now_s();
    const int rc = h2d_raw(ctx, slot, h, n);
    ctx->tr_h2d += now_s() - t0;
    return rc;
}
// device -> the caller's host array.  Small results wait in the pinned output buffer and are handed over by xc_sync (EVERY host-form
// entry point ends in xc_sync): the copies of a call are asynchronous and its stream is waited for once.
static int d2h(xc_ctx* ctx, void* h, const void* d, size_t n)
{
    const double t0 = now_s();
    int rc = XC_OK;
    if (n <= kPinSmall && ensure_pins(ctx) == XC_OK && ctx->pin_out_off + n <= kPinBytes) {
        char* p = ctx->pin_out + ctx->pin_out_off;
        ctx->pin_out_off += (n + 63) & ~(size_t)63;
        if (ctx->knobs.copy_kernel && n <= (size_t)ctx->knobs.copy_out_kb << 10) {
            if (n) ctx->pending_out.push_back({h, p, n, d});         // fetched by flush_out: every caller goes on to xc_sync without another launch on `d`
        } else {
            hipError_t e = hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, ctx->stream);
            if (e != hipSuccess) rc = hipfail(ctx, e, "hipMemcpyAsync(d2h)");
            else ctx->pending_out.push_back({h, p, n, nullptr});
        }
    } else {
        hipError_t e = hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) rc = hipfail(ctx, e, "hipMemcpyAsync(d2h)");
    }
    ctx->tr_d2h += now_s() - t0;
    return rc;
}

// where the kernels of a host-form call write a SMALL result the caller wants in `h`: straight into the pinned output buffer -- the device
// writes host memory through the same pointer, xc_sync hands it over, and the stream carries no copy of any kind for it.  Only for outputs
// that are written once and never read back by a kernel.  nullptr: too large / buffer full / switched off -- arena bytes and d2h() then.
static void* out_direct(xc_ctx* ctx, void* h, size_t n)
{
    if (!h || !ctx->knobs.copy_kernel || n == 0 || n > ((size_t)ctx->knobs.copy_out_kb << 10) || ensure_pins(ctx) != XC_OK || ctx->pin_out_off + n > kPinBytes) return nullptr;
    char* p = ctx->pin_out + ctx->pin_out_off;
    ctx->pin_out_off += (n + 63) & ~(size_t)63;
    ctx->pending_out.push_back({h, p, n, nullptr});
    return p;
}

}  // namespace xc

using namespace xc;

#define XC_TRY(expr) do { int _rc = (expr); if (_rc != XC_OK) return _rc; } while (0)
#define XC_CTX(ctx) do { if (!(ctx)) return fail(nullptr, XC_EBADARG, "null context"); \
                         hipError_t _e = hipSetDevice((ctx)->device); \
                         if (_e != hipSuccess) return hipfail((ctx), _e, "hipSetDevice"); } while (0)

extern "C" {

const char* xc_version(void) { return "xcontour_hip 0.1.0 (gfx950)"; }

int xc_device_count(int* out_count)
{
    if (!out_count) return fail(nullptr, XC_EBADARG, "xc_device_count: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *out_count = n;
    return XC_OK;
}

const char* xc_last_error(xc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int xc_create(int device_id, xc_ctx** out)
{
    if (!out) return fail(nullptr, XC_EBADARG, "xc_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        return fail(nullptr, XC_ENODEV, "xc_create: no HIP device visible (this library has no CPU fallback)");
    }
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, XC_EBADARG, "xc_create: device_id out of range");
    xc_ctx* ctx = new (std::nothrow) xc_ctx();
    if (!ctx) return fail(nullptr, XC_ENOMEM, "xc_create: out of host memory");
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        int rc = hipfail(nullptr, e, "hipSetDevice/hipGetDeviceProperties"); delete ctx; return rc;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string m = std::string("xc_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        delete ctx; return fail(nullptr, XC_ENODEV, m);
    }
    // (some boxes of the pool report an EMPTY marketing name: the field a reader checks first must still say what ran)
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s, %d CUs, %.0f GB)", prop.name[0] ? prop.name : "AMD Instinct [name not reported by the driver]",
             prop.gcnArchName, prop.multiProcessorCount, (double)prop.totalGlobalMem / 1e9);
    {
        // the only place the library reads the environment: K3 geometry knobs for experiments (xc_internal.h, HistKnobs)
        auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        HistKnobs& k = ctx->knobs;
        k.xcd_map = env_int("XC_HIST_XCDMAP", 1); k.tile_map = env_int("XC_HIST_TILEMAP", 1); k.vec4 = env_int("XC_HIST_VEC4", -1); k.e32 = env_int("XC_HIST_E32", 1);
        k.threads = env_int("XC_HIST_THREADS", 0); k.ncopy = env_int("XC_HIST_NCOPY", 0); k.rows = env_int("XC_HIST_ROWS", 0);
        k.bps = env_int("XC_HIST_BPS", 0);
        k.cross_ncopy = env_int("XC_CROSS_NCOPY", 0); k.cross_blocks = env_int("XC_CROSS_BLOCKS", 0);
        k.copy_kernel = env_int("XC_COPY_KERNEL", 1); k.copy_out_kb = env_int("XC_COPY_OUT_KB", 256); if (k.copy_out_kb < 1 || k.copy_out_kb > 1024) k.copy_out_kb = 256; k.single = env_int("XC_KEFF_SINGLE", 1); k.single_timeout_us = env_int("XC_KEFF_SINGLE_TIMEOUT_US", 50000); k.single_map = env_int("XC_KEFF_SINGLE_MAP", 0);
        k.sort_range = env_int("XC_SORT_RANGE", 1); k.lwa_fast = env_int("XC_LWA_FAST", 1); k.k1_nt = env_int("XC_K1_NT", 0); k.lwa_strip = env_int("XC_LWA_STRIP", 1);
    }
    ctx->cus = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        int rc = hipfail(nullptr, e, "hipStreamCreate"); delete ctx; return rc;
    }
    if ((e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_copy, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_compute, hipEventDisableTiming)) != hipSuccess) {
        int rc = hipfail(nullptr, e, "hipStreamCreate(copy)"); delete ctx; return rc;
    }
    if ((e = hipEventCreate(&ctx->ev_hist0)) != hipSuccess || (e = hipEventCreate(&ctx->ev_hist1)) != hipSuccess) {
        int rc = hipfail(nullptr, e, "hipEventCreate"); delete ctx; return rc;
    }
    *out = ctx;
    return XC_OK;
}

int xc_destroy(xc_ctx* ctx)
{
    if (!ctx) return XC_OK;
    (void)hipSetDevice(ctx->device);
    (void)xc_comm_finalize(ctx);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) { (void)hipStreamSynchronize(ctx->comm_stream); (void)hipStreamDestroy(ctx->comm_stream); }
    if (ctx->ev_comm_in) (void)hipEventDestroy(ctx->ev_comm_in);
    if (ctx->ev_comm_out) (void)hipEventDestroy(ctx->ev_comm_out);
    if (ctx->pinned_flag) (void)hipHostFree(ctx->pinned_flag);
    for (auto& e : ctx->small_in) if (e.dev) (void)hipFree(e.dev);
    if (ctx->pin_in) (void)hipHostFree(ctx->pin_in);
    if (ctx->pin_out) (void)hipHostFree(ctx->pin_out);
    if (ctx->lwa_flag) (void)hipFree(ctx->lwa_flag);
    if (ctx->single_ws) (void)hipFree(ctx->single_ws);
    if (ctx->single_stamps) (void)hipFree(ctx->single_stamps);
    for (auto& e : ctx->resident) (void)hipFree(e.dev);
    if (ctx->ev_copy) (void)hipEventDestroy(ctx->ev_copy);
    if (ctx->ev_compute) (void)hipEventDestroy(ctx->ev_compute);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->big) (void)hipFree(ctx->big);
    if (ctx->ones) (void)hipFree(ctx->ones);
    for (int i = 0; i < 2; ++i) if (ctx->mmnext[i]) (void)hipFree(ctx->mmnext[i]);
    if (ctx->ev_hist0) (void)hipEventDestroy(ctx->ev_hist0);
    if (ctx->ev_hist1) (void)hipEventDestroy(ctx->ev_hist1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return XC_OK;
}

int xc_device_name(xc_ctx* ctx, char* buf, size_t buflen)
{
    if (!ctx || !buf || buflen == 0) return fail(ctx, XC_EBADARG, "xc_device_name: bad arguments");
    snprintf(buf, buflen, "%s", ctx->name);
    return XC_OK;
}

int xc_device_cus(xc_ctx* ctx, int* out_cus)
{
    if (!ctx || !out_cus) return fail(ctx, XC_EBADARG, "xc_device_cus: bad arguments");
    *out_cus = ctx->cus;
    return XC_OK;
}

int xc_sync(xc_ctx* ctx)
{
    XC_CTX(ctx);
    { const int rc = flush_in(ctx); if (rc != XC_OK) return rc; }
    { const int rc = flush_out(ctx); if (rc != XC_OK) return rc; }
    const double t0 = now_s();
    // (round 6: polling hipStreamQuery for the first 150 us instead of blocking at once changes nothing -- 369 against 374 us for the
    // reference's call sequence at its demo size: the runtime's own wait already spins)
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    const double t1 = now_s();
    ctx->tr_sync += t1 - t0;
    // results parked in the pinned output buffer go to the caller's arrays now; both bounce buffers are free again
    if (e == hipSuccess) for (const auto& po : ctx->pending_out) memcpy(po.host, po.pinned, po.bytes);
    ctx->pending_out.clear(); ctx->pending_in.clear();
    ctx->pin_in_off = 0; ctx->pin_out_off = 0;
    ++ctx->small_epoch;
    ctx->tr_d2h += now_s() - t1;
    if (e != hipSuccess) return hipfail(ctx, e, "hipStreamSynchronize");
    return XC_OK;
}

int xc_trace(xc_ctx* ctx, int reset, double* out3)
{
    if (!ctx) return fail(nullptr, XC_EBADARG, "null context");
    if (out3) { out3[0] = ctx->tr_h2d; out3[1] = ctx->tr_d2h; out3[2] = ctx->tr_sync; }
    if (reset) ctx->tr_h2d = ctx->tr_d2h = ctx->tr_sync = 0.0;
    return XC_OK;
}

void* xc_stream(xc_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int xc_malloc(xc_ctx* ctx, size_t bytes, void** out_dptr)
{
    XC_CTX(ctx);
    if (!out_dptr) return fail(ctx, XC_EBADARG, "xc_malloc: out is NULL");
    *out_dptr = nullptr;
    XC_HIP(ctx, hipMalloc(out_dptr, bytes ? bytes : 1));
    return XC_OK;
}

int xc_keep_resident(xc_ctx* ctx, const void* host_ptr, size_t bytes)
{
    XC_CTX(ctx);
    if (!host_ptr || bytes == 0) return fail(ctx, XC_EBADARG, "xc_keep_resident: bad arguments");
    XC_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    XC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->resident.size(); ++i)
        if (ctx->resident[i].host == (const char*)host_ptr) {       // registered before: refresh (the caller changed the array)
            auto& e = ctx->resident[i];
            hipError_t he = hipSuccess;
            if (bytes > e.bytes) {
                (void)hipFree(e.dev); e.dev = nullptr; e.bytes = 0;
                he = hipMalloc(&e.dev, bytes);
            }
            if (he == hipSuccess) he = hipMemcpy(e.dev, host_ptr, bytes, hipMemcpyHostToDevice);
            if (he != hipSuccess) {                                 // never leave a dead or half-refreshed mirror registered
                if (e.dev) (void)hipFree(e.dev);
                ctx->resident.erase(ctx->resident.begin() + (long)i);
                return hipfail(ctx, he, "xc_keep_resident: refresh");
            }
            e.bytes = bytes;                                        // (a shorter array now: the tail of the old mirror is no longer valid)
            // most recently refreshed first: an overlapping registration (an array and a sub-slab of it at another base
            // pointer) resolves to the mirror that was uploaded last
            if (i != 0) { auto me = e; ctx->resident.erase(ctx->resident.begin() + (long)i); ctx->resident.insert(ctx->resident.begin(), me); }
            return XC_OK;
        }
    void* dev = nullptr;
    XC_HIP(ctx, hipMalloc(&dev, bytes));
    hipError_t e = hipMemcpy(dev, host_ptr, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(dev); return hipfail(ctx, e, "xc_keep_resident: upload"); }
    ctx->resident.insert(ctx->resident.begin(), {(const char*)host_ptr, bytes, dev});       // newest first (see the refresh path)
    return XC_OK;
}

int xc_resident_lookup(xc_ctx* ctx, const void* host_ptr, size_t bytes, void** out_dev)
{
    if (!ctx || !out_dev) return fail(ctx, XC_EBADARG, "xc_resident_lookup: bad arguments");
    *out_dev = (host_ptr && bytes && !ctx->resident.empty()) ? const_cast<void*>(resident_lookup(ctx, host_ptr, bytes)) : nullptr;
    return XC_OK;
}

int xc_release_resident(xc_ctx* ctx, const void* host_ptr)
{
    XC_CTX(ctx);
    XC_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));        // an asynchronous upload may still be reading a mirror
    XC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->resident.size();) {
        if (!host_ptr || ctx->resident[i].host == (const char*)host_ptr) {
            (void)hipFree(ctx->resident[i].dev);
            ctx->resident.erase(ctx->resident.begin() + (long)i);
        } else ++i;
    }
    return XC_OK;
}

int xc_free(xc_ctx* ctx, void* dptr)
{
    XC_CTX(ctx);
    if (!dptr) return XC_OK;
    mm_touch(ctx, dptr, (size_t)1 << 62);         // an allocation that starts at or below the cached batch may contain it
    XC_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    XC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    XC_HIP(ctx, hipFree(dptr));
    return XC_OK;
}

int xc_memcpy_h2d(xc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes)
{
    XC_CTX(ctx);
    if (bytes && (!dst_dev || !src_host)) return fail(ctx, XC_EBADARG, "xc_memcpy_h2d: NULL pointer");
    mm_touch(ctx, dst_dev, bytes);
    const void* m = ctx->resident.empty() ? nullptr : resident_lookup(ctx, src_host, bytes);   // a registered array: from its device mirror
    XC_HIP(ctx, hipMemcpyAsync(dst_dev, m ? m : src_host, bytes, m ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    XC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return XC_OK;
}

int xc_memcpy_h2d_async(xc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes)
{
    XC_CTX(ctx);
    if (bytes && (!dst_dev || !src_host)) return fail(ctx, XC_EBADARG, "xc_memcpy_h2d_async: NULL pointer");
    mm_touch(ctx, dst_dev, bytes);
    const void* m = ctx->resident.empty() ? nullptr : resident_lookup(ctx, src_host, bytes);
    XC_HIP(ctx, hipMemcpyAsync(dst_dev, m ? m : src_host, bytes, m ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->copy_stream));
    return XC_OK;
}

int xc_stream_wait_copies(xc_ctx* ctx)
{
    XC_CTX(ctx);
    XC_HIP(ctx, hipEventRecord(ctx->ev_copy, ctx->copy_stream));
    XC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copy, 0));
    return XC_OK;
}

int xc_copies_wait_stream(xc_ctx* ctx)
{
    XC_CTX(ctx);
    XC_HIP(ctx, hipEventRecord(ctx->ev_compute, ctx->stream));
    XC_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_compute, 0));
    return XC_OK;
}

int xc_memcpy_d2h(xc_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes)
{
    XC_CTX(ctx);
    if (bytes && (!dst_host || !src_dev)) return fail(ctx, XC_EBADARG, "xc_memcpy_d2h: NULL pointer");
    XC_TRY(d2h(ctx, dst_host, src_dev, bytes));
    return xc_sync(ctx);
}

int xc_memset(xc_ctx* ctx, void* dptr, int value, size_t bytes)
{
    XC_CTX(ctx);
    if (bytes && !dptr) return fail(ctx, XC_EBADARG, "xc_memset: NULL pointer");
    mm_touch(ctx, dptr, bytes);
    XC_HIP(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return XC_OK;
}

int xc_event_create(xc_ctx* ctx, void** out_event)
{
    XC_CTX(ctx);
    if (!out_event) return fail(ctx, XC_EBADARG, "xc_event_create: out is NULL");
    hipEvent_t ev;
    XC_HIP(ctx, hipEventCreate(&ev));
    *out_event = (void*)ev;
    return XC_OK;
}

int xc_event_destroy(xc_ctx* ctx, void* event)
{
    XC_CTX(ctx);
    if (event) XC_HIP(ctx, hipEventDestroy((hipEvent_t)event));
    return XC_OK;
}

int xc_event_record(xc_ctx* ctx, void* event)
{
    XC_CTX(ctx);
    if (!event) return fail(ctx, XC_EBADARG, "xc_event_record: NULL event");
    XC_HIP(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
    return XC_OK;
}

int xc_event_record_copies(xc_ctx* ctx, void* event)      // on the COPY stream: completes when the uploads issued so far have landed
{
    XC_CTX(ctx);
    if (!event) return fail(ctx, XC_EBADARG, "xc_event_record_copies: NULL event");
    XC_HIP(ctx, hipEventRecord((hipEvent_t)event, ctx->copy_stream));
    return XC_OK;
}

int xc_event_query(xc_ctx* ctx, void* event, int* out_done)
{
    XC_CTX(ctx);
    if (!event || !out_done) return fail(ctx, XC_EBADARG, "xc_event_query: NULL argument");
    const hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipSuccess) { *out_done = 1; return XC_OK; }
    *out_done = 0;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return XC_OK; }
    return hipfail(ctx, e, "hipEventQuery");
}

int xc_event_elapsed_ms(xc_ctx* ctx, void* start, void* stop, float* out_ms)
{
    XC_CTX(ctx);
    if (!start || !stop || !out_ms) return fail(ctx, XC_EBADARG, "xc_event_elapsed_ms: NULL argument");
    XC_HIP(ctx, hipEventSynchronize((hipEvent_t)stop));
    XC_HIP(ctx, hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop));
    return XC_OK;
}

int xc_set_kernel_timing(xc_ctx* ctx, int enable)
{
    if (!ctx) return fail(nullptr, XC_EBADARG, "null context");
    ctx->timing = enable ? 1 : 0; ctx->ev_valid = 0;
    return XC_OK;
}

int xc_set_hist_events(xc_ctx* ctx, void* start_event, void* stop_event)
{
    if (!ctx) return fail(nullptr, XC_EBADARG, "null context");
    ctx->user_ev0 = (hipEvent_t)start_event; ctx->user_ev1 = (hipEvent_t)stop_event;
    return XC_OK;
}

static int hist_ev_begin(xc_ctx* ctx)
{
    if (ctx->user_ev0) XC_HIP(ctx, hipEventRecord(ctx->user_ev0, ctx->stream));
    else if (ctx->timing) XC_HIP(ctx, hipEventRecord(ctx->ev_hist0, ctx->stream));
    return XC_OK;
}

static int hist_ev_end(xc_ctx* ctx)
{
    if (ctx->user_ev0) {
        if (ctx->user_ev1) XC_HIP(ctx, hipEventRecord(ctx->user_ev1, ctx->stream));
        ctx->user_ev0 = ctx->user_ev1 = nullptr;
    } else if (ctx->timing) {
        XC_HIP(ctx, hipEventRecord(ctx->ev_hist1, ctx->stream)); ctx->ev_valid = 1;
    }
    return XC_OK;
}

int xc_last_hist_ms(xc_ctx* ctx, float* out_ms)
{
    XC_CTX(ctx);
    if (!out_ms) return fail(ctx, XC_EBADARG, "xc_last_hist_ms: out is NULL");
    if (!ctx->ev_valid) return fail(ctx, XC_EBADARG, "xc_last_hist_ms: no timed histogram launch (call xc_set_kernel_timing(ctx,1) first)");
    XC_HIP(ctx, hipEventSynchronize(ctx->ev_hist1));
    XC_HIP(ctx, hipEventElapsedTime(out_ms, ctx->ev_hist0, ctx->ev_hist1));
    return XC_OK;
}

// ------------------------------------------------------------------------------------ K1
int xc_minmax_dev(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ncell, double* out_minmax)
{
    XC_CTX(ctx);
    if (!q || !out_minmax || nslab < 1 || ncell < 1) return fail(ctx, XC_EBADARG, "xc_minmax: bad arguments");
    XC_TRY(ensure_scratch(ctx, al((size_t)nslab * kMinmaxBlocks * 2 * sizeof(double))));
    double* part = (double*)ctx->scratch;
    XC_TRY(launch_minmax_partial(ctx, q, q_dtype, nslab, ncell, part));
    return launch_minmax_final(ctx, part, nslab, minmax_blocks(ncell, nslab), out_minmax);
}

int xc_minmax(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ncell, double* out_minmax)
{
    XC_CTX(ctx);
    if (!q || !out_minmax || nslab < 1 || ncell < 1) return fail(ctx, XC_EBADARG, "xc_minmax: bad arguments");
    if (q_dtype != XC_F32 && q_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_minmax: q_dtype must be XC_F32 or XC_F64");
    const size_t qb = (size_t)nslab * ncell * esize(q_dtype), ob = (size_t)nslab * 2 * sizeof(double);
    XC_TRY(ensure_arena(ctx, al(qb) + al(ob)));
    Stage st(ctx);
    const void* dq; double* dout;
    XC_TRY(stage_in(ctx, st.take(qb), q, qb, &dq));
    dout = (double*)st.take(ob);
    XC_TRY(flush_in(ctx));
    XC_TRY(xc_minmax_dev(ctx, dq, q_dtype, nslab, ncell, dout));
    XC_TRY(d2h(ctx, out_minmax, dout, ob));
    return xc_sync(ctx);
}

// ------------------------------------------------------------------------------------ levels
int xc_levels_dev(xc_ctx* ctx, const double* minmax, int q_dtype, int64_t nslab, int N, int increase,
                  int ctr_dtype, int right_edge, double* ctr, double* edges, int32_t* status)
{
    XC_CTX(ctx);
    return launch_levels(ctx, minmax, q_dtype, nslab, N, increase, ctr_dtype, right_edge, ctr, edges, status);
}

int xc_levels(xc_ctx* ctx, const double* minmax, int q_dtype, int64_t nslab, int N, int increase,
              int ctr_dtype, int right_edge, double* ctr, double* edges, int32_t* status)
{
    XC_CTX(ctx);
    if (!minmax || !ctr || !edges || !status || N < 2 || nslab < 1) return fail(ctx, XC_EBADARG, "xc_levels: bad arguments (need N >= 2)");
    const size_t mb = (size_t)nslab * 2 * 8, cb = (size_t)nslab * N * 8, eb = (size_t)nslab * (N + 1) * 8, sb = (size_t)nslab * 4;
    XC_TRY(ensure_arena(ctx, al(mb) + al(cb) + al(eb) + al(sb)));
    Stage st(ctx);
    double* dm = (double*)st.take(mb); double* dc = (double*)st.take(cb);
    double* de = (double*)st.take(eb); int32_t* ds = (int32_t*)st.take(sb);
    XC_TRY(h2d(ctx, dm, minmax, mb));
    XC_TRY(flush_in(ctx));
    XC_TRY(launch_levels(ctx, dm, q_dtype, nslab, N, increase, ctr_dtype, right_edge, dc, de, ds));
    XC_TRY(d2h(ctx, ctr, dc, cb)); XC_TRY(d2h(ctx, edges, de, eb)); XC_TRY(d2h(ctx, status, ds, sb));
    return xc_sync(ctx);
}

// cal_contours(int levels) (core.py:205-249) in ONE call: min / max of every slab (K1), the levels (and edges) from them, ONE result
// hand-over and ONE stream synchronisation -- the facade used to make two host-form calls (xc_minmax, then xc_levels with the extrema
// sent back up) with a blocking transfer for each of their five small arrays.  Any output may be NULL except out_ctr.
int xc_contours(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ncell, int N, int increase, int ctr_dtype, int right_edge,
                double* out_minmax, double* out_ctr, double* out_edges, int32_t* out_status)
{
    XC_CTX(ctx);
    if (!q || !out_ctr || nslab < 1 || ncell < 1 || N < 2) return fail(ctx, XC_EBADARG, "xc_contours: bad arguments (need N >= 2)");
    if (q_dtype != XC_F32 && q_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_contours: q_dtype must be XC_F32 or XC_F64");
    const size_t qb = (size_t)nslab * ncell * esize(q_dtype);
    const size_t mb = (size_t)nslab * 2 * 8, cb = (size_t)nslab * N * 8, eb = (size_t)nslab * (N + 1) * 8, sb = (size_t)nslab * 4;
    XC_TRY(ensure_arena(ctx, al(qb) + al(mb) + al(cb) + al(eb) + al(sb)));
    Stage st(ctx);
    const void* dq;
    XC_TRY(stage_in(ctx, st.take(qb), q, qb, &dq));
    double* dm = (double*)st.take(mb);                       // (read by the level kernel: stays on the device)
    double* pc = (double*)out_direct(ctx, out_ctr, cb); double* pe = (double*)out_direct(ctx, out_edges, eb);
    int32_t* ps = (int32_t*)out_direct(ctx, out_status, sb);
    double* dc = pc ? pc : (double*)st.take(cb);
    double* de = pe ? pe : (double*)st.take(eb); int32_t* ds = ps ? ps : (int32_t*)st.take(sb);
    XC_TRY(flush_in(ctx));
    // K1's partials are reduced by the level kernel itself: two launches for the whole call
    XC_TRY(ensure_scratch(ctx, al((size_t)nslab * kMinmaxBlocks * 2 * sizeof(double))));
    XC_TRY(launch_minmax_partial(ctx, dq, q_dtype, nslab, ncell, (double*)ctx->scratch));
    XC_TRY(launch_levels(ctx, (const double*)ctx->scratch, q_dtype, nslab, N, increase, ctr_dtype, right_edge, dc, de, ds,
                         minmax_blocks(ncell, nslab), dm));
    if (out_minmax) XC_TRY(d2h(ctx, out_minmax, dm, mb));
    if (!pc) XC_TRY(d2h(ctx, out_ctr, dc, cb));
    if (out_edges && !pe) XC_TRY(d2h(ctx, out_edges, de, eb));
    if (out_status && !ps) XC_TRY(d2h(ctx, out_status, ds, sb));
    return xc_sync(ctx);
}

// The 2-cells-per-lane kernels issue 16-byte (f64) / 8-byte (f32) loads from EVERY streamed array:
// OR the low address bits of all of them so that hist_geometry can fall back to 1 cell per lane
// when any base is not suitably aligned (e.g. a slab offset inside a caller's buffer).
static const void* vec_align_bits(const void* q, int q_dtype, const void* q_next, const double* dA, int dA_rank,
                                  const void* const* integ, const int32_t* integ_dtype, int nint)
{
    uintptr_t bits = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(q_next);
    // f64 arrays need 16-byte alignment; express it in q's units: for f32 q (8-byte requirement) an
    // f64 array that is only 8-byte aligned must still force the fallback
    auto need16 = [&](const void* p) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        if (a % 16 != 0) bits |= (q_dtype == XC_F32) ? 4 : 8;   // make (bits % (2*esz)) != 0
    };
    if (dA && (dA_rank == XC_DA_PLANE || dA_rank == XC_DA_SLAB)) need16(dA);
    for (int i = 0; i < nint; ++i) {
        if (integ_dtype[i] == XC_F64) need16(integ[i]);
        else if (reinterpret_cast<uintptr_t>(integ[i]) % 8 != 0) bits |= (q_dtype == XC_F32) ? 4 : 8;
    }
    return reinterpret_cast<const void*>(bits);
}

// ------------------------------------------------------------------------------------ K3 + K5
static int check_hist_desc(xc_ctx* ctx, const xc_hist_desc* d)
{
    if (!d) return fail(ctx, XC_EBADARG, "xc_hist: desc is NULL");
    if (!d->q || !d->edges) return fail(ctx, XC_EBADARG, "xc_hist: q / edges is NULL");
    if (d->q_dtype != XC_F32 && d->q_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_hist: q_dtype must be XC_F32 or XC_F64");
    if (d->nslab < 1 || d->ny < 1 || d->nx < 1) return fail(ctx, XC_EBADARG, "xc_hist: nslab, ny, nx must be >= 1");
    if (d->nedge < 2) return fail(ctx, XC_EBADARG, "xc_hist: need at least 2 edges");
    if (d->dA_rank < XC_DA_NONE || d->dA_rank > XC_DA_SLAB) return fail(ctx, XC_EBADARG, "xc_hist: bad dA_rank");
    if (d->dA_rank != XC_DA_NONE && !d->dA) return fail(ctx, XC_EBADARG, "xc_hist: dA is NULL");
    if (d->nint < 0 || d->nint > XC_MAX_INTEGRANDS) return fail(ctx, XC_EBADARG, "xc_hist: nint must be 0..2");
    for (int i = 0; i < d->nint; ++i) {
        if (!d->integrand[i]) return fail(ctx, XC_EBADARG, "xc_hist: integrand is NULL");
        if (d->integrand_dtype[i] != XC_F32 && d->integrand_dtype[i] != XC_F64) return fail(ctx, XC_EBADARG, "xc_hist: bad integrand dtype");
    }
    if (d->grad && (!d->rdx || !d->rdy)) return fail(ctx, XC_EBADARG, "xc_hist: grad needs rdx and rdy");
    if (d->grad && d->negate) return fail(ctx, XC_EBADARG, "xc_hist: negate is not available together with grad");
    if (!d->pdf && !d->counts && !d->cdf) return fail(ctx, XC_EBADARG, "xc_hist: no output requested");
    return XC_OK;
}

// Deterministic sums in ONE histogram pass (xc_binning.h, xc_hist_det.hip): the bounds that fix every channel's accumulator window
// come first -- max |dA| from the caller or from a min / max pass over the dA array, the extrema of every supplied integrand
// (and of the tracer, for the in-kernel gradient with explicit edges) from K1 passes of their own -- then the pass, then the exact
// reduction of the blocks' limbs into f.red_h / f.red_c; launch_finalize runs its second stage only.
// `work`: det_bounds_bytes(nslab) of scratch for those extrema.
static size_t det_bounds_bytes(int64_t nslab)
{
    // K1 partials of one array at a time + (min, max) pairs: dA, the tracer, XC_MAX_INTEGRANDS integrands
    return al((size_t)nslab * kMinmaxBlocks * 2 * sizeof(double)) + (3 + XC_MAX_INTEGRANDS) * al((size_t)nslab * 2 * sizeof(double));
}

static int det_one_pass(xc_ctx* ctx, int q_dtype, int nint, int grad, const HistGeom& g, int64_t nslab, int64_t ny, int64_t nx,
                        HistArgs& a, int* c0, char* work, double dA_max_host, const void* const* integ, const int* integ_dtype, FinalArgs& f)
{
    double* part = (double*)work;
    char* pairs = work + al((size_t)nslab * kMinmaxBlocks * 2 * sizeof(double));
    const size_t pb = al((size_t)nslab * 2 * sizeof(double));
    auto extrema = [&](const void* p, int dtype, int64_t ns, int64_t ncell, double* out) -> int {
        // (the bounds of the accumulator windows: the largest FINITE magnitudes -- an infinite weight flags its own bin, it must not push
        // every finite one out of the window; the oracle takes its bounds the same way)
        XC_TRY(launch_minmax_partial(ctx, p, dtype, ns, ncell, part, nullptr, 0, true));
        return launch_minmax_final(ctx, part, ns, minmax_blocks(ncell, ns), out);
    };
    a.det_dA_max = -1.0; a.det_dA_max_dev = nullptr; a.det_dA_stride = 0;
    if (dA_max_host > 0.0 && dA_max_host < __builtin_inf()) a.det_dA_max = dA_max_host;
    else if (a.dA == ctx->ones) a.det_dA_max = 1.0;
    else {
        double* out = (double*)pairs;
        const int64_t ns = a.dA_rank == XC_DA_SLAB ? nslab : 1, ncell = a.dA_rank == XC_DA_ROW ? ny : ny * nx;
        XC_TRY(extrema(a.dA, XC_F64, ns, ncell, out));
        a.det_dA_max_dev = out; a.det_dA_stride = a.dA_rank == XC_DA_SLAB ? 1 : 0;
    }
    a.det_q_mm = nullptr;
    if (grad && !a.levels_mode) {
        double* out = (double*)(pairs + pb);
        XC_TRY(extrema(a.q, q_dtype, nslab, ny * nx, out));
        a.det_q_mm = out;
    }
    for (int i = 0; i < nint; ++i) {
        double* out = (double*)(pairs + (2 + i) * pb);
        XC_TRY(extrema(integ[i], integ_dtype[i], nslab, ny * nx, out));
        a.det_int_mm[i] = out;
    }
    a.det_c0_out = c0;
    XC_TRY(launch_hist_det3(ctx, q_dtype, nint, grad, g, nslab, a));
    XC_TRY(launch_det3_reduce(ctx, nslab, g.bps, f.nch, f.nbin, a.part_h, a.part_c, c0, f.red_h, f.red_c));
    f.skip_reduce = 1;
    return XC_OK;
}

int xc_hist_dev(xc_ctx* ctx, const xc_hist_desc* d)
{
    XC_CTX(ctx);
    XC_TRY(check_hist_desc(ctx, d));
    const int nbin = (int)(d->nedge - 1), nch = 1 + d->nint + (d->grad ? 1 : 0);
    HistGeom g;
    const int det = d->deterministic ? 1 : 0;
    XC_TRY(hist_geometry(ctx, d->q_dtype, d->nslab, d->ny, d->nx, nbin, nch,
                         vec_align_bits(d->q, d->q_dtype, nullptr, d->dA, d->dA_rank, d->integrand, d->integrand_dtype, d->nint), &g, 0, det));
    const size_t ph = al((size_t)d->nslab * g.bps * (det ? det_limbs_total(nch) : nch) * nbin * sizeof(double));
    const size_t pc = al((size_t)d->nslab * g.bps * nbin * sizeof(unsigned));
    const size_t rh = al((size_t)d->nslab * nch * nbin * sizeof(double));
    const size_t rc = al((size_t)d->nslab * nbin * sizeof(unsigned long long));
    XC_TRY(ensure_scratch(ctx, ph + pc + rh + rc + (det ? al((size_t)d->nslab * nch * sizeof(int)) + det_bounds_bytes(d->nslab) : 0)));
    ctx->mm_valid = 0;                          // (the scratch may have moved)
    HistArgs a; memset(&a, 0, sizeof(a));
    a.q = d->q; a.dA = d->dA; a.dA_rank = d->dA_rank;
    if (d->dA_rank == XC_DA_NONE) { XC_TRY(ensure_ones(ctx, (size_t)d->ny)); a.dA = ctx->ones; a.dA_rank = XC_DA_ROW; }
    for (int i = 0; i < d->nint; ++i) { a.integ[i] = d->integrand[i]; a.integ_f32[i] = d->integrand_dtype[i] == XC_F32; }
    a.edges = d->edges; a.levels_mode = 0; a.nbin = nbin; a.edges_per_slab = d->edges_per_slab;
    a.last_closed = d->last_closed; a.negate = d->negate; a.q_f32 = d->q_dtype == XC_F32;
    a.dA_pos_finite = (d->dA_rank == XC_DA_NONE) ? 1 : d->dA_pos_finite;
    a.prod_f32 = d->prod_f32;
    a.rdx = d->rdx; a.rdy = d->rdy; a.periodic_x = d->periodic_x;
    a.ny = d->ny; a.nx = d->nx; a.nstrip = g.nstrip; a.ncopy = g.ncopy;
    a.part_h = (double*)ctx->scratch; a.part_c = (d->counts || d->deterministic) ? (unsigned*)((char*)ctx->scratch + ph) : nullptr;   // (no counts wanted: no count adds)
    FinalArgs f; memset(&f, 0, sizeof(f));
    f.part_h = a.part_h; f.part_c = a.part_c; f.bps = g.bps; f.nch = nch; f.nbin = nbin;
    f.red_h = (double*)((char*)ctx->scratch + ph + pc); f.red_c = (unsigned long long*)((char*)ctx->scratch + ph + pc + rh);
    f.lt = d->lt; f.reverse = d->reverse; f.pdf = d->pdf; f.counts = d->counts; f.cdf = d->cdf;
    XC_TRY(hist_ev_begin(ctx));
    if (det) {
        char* xb = (char*)ctx->scratch + ph + pc + rh + rc;
        XC_TRY(det_one_pass(ctx, d->q_dtype, d->nint, d->grad, g, d->nslab, d->ny, d->nx, a, (int*)xb,
                            xb + al((size_t)d->nslab * nch * sizeof(int)), 0.0, d->integrand, d->integrand_dtype, f));
    } else {
        XC_TRY(launch_hist(ctx, d->q_dtype, d->nint, d->grad, g, d->nslab, a));
    }
    XC_TRY(hist_ev_end(ctx));
    return launch_finalize(ctx, d->nslab, f);
}

int xc_hist(xc_ctx* ctx, const xc_hist_desc* hd)
{
    XC_CTX(ctx);
    XC_TRY(check_hist_desc(ctx, hd));
    const int64_t S = hd->nslab, ny = hd->ny, nx = hd->nx, ne = hd->nedge;
    const int nbin = (int)(ne - 1), nch = 1 + hd->nint + (hd->grad ? 1 : 0);
    const int64_t nes = hd->edges_per_slab ? S : 1;
    // reference: 'non monotonic bins' (core.py:1233-1251); np.digitize needs monotone edges.
    // NaN edges are let through (an all-NaN slab yields NaN levels and an empty histogram).
    for (int64_t s = 0; s < nes; ++s)
        for (int64_t k = 1; k < ne; ++k) {
            const double e0 = hd->edges[s * ne + k - 1], e1 = hd->edges[s * ne + k];
            if (e1 <= e0) return fail(ctx, XC_EEDGES, "non monotonic bins");
        }
    const size_t cells = (size_t)S * ny * nx;
    const size_t qb = cells * esize(hd->q_dtype), eb = (size_t)nes * ne * 8;
    size_t dab = 0;
    if (hd->dA_rank == XC_DA_ROW) dab = (size_t)ny * 8;
    else if (hd->dA_rank == XC_DA_PLANE) dab = (size_t)ny * nx * 8;
    else if (hd->dA_rank == XC_DA_SLAB) dab = cells * 8;
    size_t ib[XC_MAX_INTEGRANDS] = {0, 0};
    for (int i = 0; i < hd->nint; ++i) ib[i] = cells * esize(hd->integrand_dtype[i]);
    const size_t rb = hd->grad ? (size_t)ny * 8 : 0;
    const size_t pb = (size_t)S * nch * nbin * 8, cb = (size_t)S * nbin * 8;
    XC_TRY(ensure_arena(ctx, al(qb) + al(eb) + al(dab) + al(ib[0]) + al(ib[1]) + 2 * al(rb) + 2 * al(pb) + al(cb)));
    Stage st(ctx);
    xc_hist_desc d = *hd;
    XC_TRY(stage_in(ctx, st.take(qb), hd->q, qb, &d.q));
    { const void* p; XC_TRY(stage_small(ctx, st.take(eb), hd->edges, eb, &p)); d.edges = (const double*)p; }
    if (dab) { const void* p; XC_TRY(stage_in(ctx, st.take(dab), hd->dA, dab, &p)); d.dA = (const double*)p; }
    for (int i = 0; i < hd->nint; ++i) XC_TRY(stage_in(ctx, st.take(ib[i]), hd->integrand[i], ib[i], &d.integrand[i]));
    if (hd->grad) {
        const void* p;
        XC_TRY(stage_small(ctx, st.take(rb), hd->rdx, rb, &p)); d.rdx = (const double*)p;
        XC_TRY(stage_small(ctx, st.take(rb), hd->rdy, rb, &p)); d.rdy = (const double*)p;
    }
    // (the finalize kernel writes the three results once and reads none of them back: small ones go straight to the pinned buffer)
    void* pp = out_direct(ctx, hd->pdf, pb); void* pcd = out_direct(ctx, hd->cdf, pb); void* pn = out_direct(ctx, hd->counts, cb);
    d.pdf = hd->pdf ? (pp ? (double*)pp : (double*)st.take(pb)) : nullptr;
    d.cdf = hd->cdf ? (pcd ? (double*)pcd : (double*)st.take(pb)) : nullptr;
    d.counts = hd->counts ? (pn ? (uint64_t*)pn : (uint64_t*)st.take(cb)) : nullptr;
    XC_TRY(flush_in(ctx));
    XC_TRY(xc_hist_dev(ctx, &d));
    if (hd->pdf && !pp) XC_TRY(d2h(ctx, hd->pdf, d.pdf, pb));
    if (hd->cdf && !pcd) XC_TRY(d2h(ctx, hd->cdf, d.cdf, pb));
    if (hd->counts && !pn) XC_TRY(d2h(ctx, hd->counts, d.counts, cb));
    return xc_sync(ctx);
}

// ------------------------------------------------------------------------------------ K2
int xc_rowsum_dev(xc_ctx* ctx, const void* mask, int mask_dtype, const double* dA, int dA_rank,
                  int64_t ny, int64_t nx, int multiply, double* out_rows)
{
    XC_CTX(ctx);
    return launch_rowsum(ctx, mask, mask_dtype, dA, dA_rank, ny, nx, multiply, out_rows);
}

int xc_rowsum(xc_ctx* ctx, const void* mask, int mask_dtype, const double* dA, int dA_rank,
              int64_t ny, int64_t nx, int multiply, double* out_rows)
{
    XC_CTX(ctx);
    if (!out_rows || ny < 1 || nx < 1) return fail(ctx, XC_EBADARG, "xc_rowsum: bad arguments");
    if (mask && mask_dtype != XC_F32 && mask_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_rowsum: bad mask dtype");
    const size_t mb = mask ? (size_t)ny * nx * esize(mask_dtype) : 0;
    const size_t dab = dA_rank == XC_DA_ROW ? (size_t)ny * 8 : dA_rank == XC_DA_PLANE ? (size_t)ny * nx * 8 : 0;
    if (dab && !dA) return fail(ctx, XC_EBADARG, "xc_rowsum: dA is NULL");
    XC_TRY(ensure_arena(ctx, al(mb) + al(dab) + al((size_t)ny * 8)));
    Stage st(ctx);
    void* dm = nullptr; double* dd = nullptr;
    // (resident inputs are read where they are: no device-to-device copy into the arena)
    if (mb) { const void* p; XC_TRY(stage_in(ctx, st.take(mb), mask, mb, &p)); dm = const_cast<void*>(p); }
    if (dab) { const void* p; XC_TRY(stage_in(ctx, st.take(dab), dA, dab, &p)); dd = (double*)const_cast<void*>(p); }
    double* po = (double*)out_direct(ctx, out_rows, (size_t)ny * 8);
    double* dout = po ? po : (double*)st.take((size_t)ny * 8);
    XC_TRY(flush_in(ctx));
    XC_TRY(launch_rowsum(ctx, dm, mask_dtype, dd, dA_rank, ny, nx, multiply, dout));
    if (!po) XC_TRY(d2h(ctx, out_rows, dout, (size_t)ny * 8));
    return xc_sync(ctx);
}

// ------------------------------------------------------------------------------------ K4
int xc_grad2_dev(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                 const double* rdx, const double* rdy, int periodic_x, double* out)
{
    XC_CTX(ctx);
    return launch_grad2(ctx, q, q_dtype, nslab, ny, nx, rdx, rdy, periodic_x, out);
}

int xc_grad2(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
             const double* rdx, const double* rdy, int periodic_x, double* out)
{
    XC_CTX(ctx);
    if (!q || !rdx || !rdy || !out || nslab < 1 || ny < 1 || nx < 1) return fail(ctx, XC_EBADARG, "xc_grad2: bad arguments");
    if (q_dtype != XC_F32 && q_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_grad2: bad dtype");
    const size_t cells = (size_t)nslab * ny * nx, qb = cells * esize(q_dtype), ob = cells * 8, rb = (size_t)ny * 8;
    XC_TRY(ensure_arena(ctx, al(qb) + al(ob) + 2 * al(rb)));
    Stage st(ctx);
    void* dq = st.take(qb); double* dx = (double*)st.take(rb); double* dy = (double*)st.take(rb); double* dout = (double*)st.take(ob);
    const void* pq;                                          // (a tracer with a device mirror is read where it is)
    XC_TRY(stage_in(ctx, dq, q, qb, &pq)); XC_TRY(h2d(ctx, dx, rdx, rb)); XC_TRY(h2d(ctx, dy, rdy, rb));
    XC_TRY(flush_in(ctx));
    XC_TRY(launch_grad2(ctx, pq, q_dtype, nslab, ny, nx, dx, dy, periodic_x, dout));
    XC_TRY(d2h(ctx, out, dout, ob));
    return xc_sync(ctx);
}

// ------------------------------------------------------------------------------------ K9
int xc_crossing_dev(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                    int pad_x, int pad_mode, const double* contours, int ncont, int contours_per_slab,
                    const void* area, int area_dtype, int area_per_slab, int stride, int full_width,
                    double* out_len, uint64_t* out_cnt)
{
    XC_CTX(ctx);
    return launch_crossing(ctx, q, q_dtype, nslab, ny, nx, pad_x, pad_mode, contours, ncont, contours_per_slab,
                           area, area_dtype, area_per_slab, stride, full_width, out_len, out_cnt);
}

int xc_crossing(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                int pad_x, int pad_mode, const double* contours, int ncont, int contours_per_slab,
                const void* area, int area_dtype, int area_per_slab, int stride, int full_width,
                double* out_len, uint64_t* out_cnt)
{
    XC_CTX(ctx);
    if (!q || !contours || !area || (!out_len && !out_cnt) || nslab < 1 || ny < 1 || nx < 1 || ncont < 1)
        return fail(ctx, XC_EBADARG, "xc_crossing: bad arguments");
    if ((q_dtype != XC_F32 && q_dtype != XC_F64) || (area_dtype != XC_F32 && area_dtype != XC_F64))
        return fail(ctx, XC_EBADARG, "xc_crossing: bad dtype");
    const int64_t nc = contours_per_slab ? nslab : 1;
    for (int64_t s = 0; s < nc; ++s)
        for (int k = 0; k < ncont; ++k) {
            const double c = contours[s * ncont + k];
            if (c != c || (k > 0 && c < contours[s * ncont + k - 1]))
                return fail(ctx, XC_EEDGES, "xc_crossing: contours must be ascending without NaN");
        }
    const size_t cells = (size_t)nslab * ny * nx, qb = cells * esize(q_dtype);
    const size_t ab = (area_per_slab ? cells : (size_t)ny * nx) * esize(area_dtype);
    const size_t cb = (size_t)nc * ncont * 8, ob = (size_t)nslab * ncont * 8;
    XC_TRY(ensure_arena(ctx, al(qb) + al(ab) + al(cb) + 2 * al(ob)));
    Stage st(ctx);
    void* dq = st.take(qb); void* da = st.take(ab); double* dc = (double*)st.take(cb);
    double* dl = out_len ? (double*)st.take(ob) : nullptr;
    uint64_t* dn = out_cnt ? (uint64_t*)st.take(ob) : nullptr;
    const void* pq; const void* pa;                          // (tracer / areas with a device mirror are read where they are)
    XC_TRY(stage_in(ctx, dq, q, qb, &pq)); XC_TRY(stage_in(ctx, da, area, ab, &pa)); XC_TRY(h2d(ctx, dc, contours, cb));
    XC_TRY(flush_in(ctx));
    XC_TRY(launch_crossing(ctx, pq, q_dtype, nslab, ny, nx, pad_x, pad_mode, dc, ncont, contours_per_slab,
                           pa, area_dtype, area_per_slab, stride, full_width, dl, dn));
    if (out_len) XC_TRY(d2h(ctx, out_len, dl, ob));
    if (out_cnt) XC_TRY(d2h(ctx, out_cnt, dn, ob));
    return xc_sync(ctx);
}

// ------------------------------------------------------------------------------------ K7
int xc_lwa_dev(xc_ctx* ctx, const void* q, int q_dtype, const double* Q, const double* coord,
               const double* dA, int dA_rank, double dA_max, const double* M, int M_rank,
               int64_t nslab, int64_t ny, int64_t nx, int increase, int part, int variant,
               const int32_t* mask_idx, int nmask, double* out_lwa, int8_t* out_masks)
{
    XC_CTX(ctx);
    return launch_lwa(ctx, q, q_dtype, Q, coord, dA, dA_rank, dA_max, M, M_rank, nslab, ny, nx,
                      increase, part, variant, mask_idx, nmask, out_lwa, out_masks);
}

int xc_lwa(xc_ctx* ctx, const void* q, int q_dtype, const double* Q, const double* coord,
           const double* dA, int dA_rank, double dA_max, const double* M, int M_rank,
           int64_t nslab, int64_t ny, int64_t nx, int increase, int part, int variant,
           const int32_t* mask_idx, int nmask, double* out_lwa, int8_t* out_masks)
{
    XC_CTX(ctx);
    if (!q || !Q || !coord || !dA || !out_lwa || nslab < 1 || ny < 2 || nx < 1) return fail(ctx, XC_EBADARG, "xc_lwa: bad arguments");
    if (q_dtype != XC_F32 && q_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_lwa: bad dtype");
    if (nmask < 0 || (nmask > 0 && (!mask_idx || !out_masks))) return fail(ctx, XC_EBADARG, "xc_lwa: mask arguments");
    for (int i = 0; i < nmask; ++i)
        if (mask_idx[i] < 0 || mask_idx[i] >= ny) return fail(ctx, XC_EBADARG, "indices in mask_idx out of boundary");
    const size_t cells = (size_t)nslab * ny * nx, plane = (size_t)ny * nx;
    const size_t qb = cells * esize(q_dtype), Qb = (size_t)nslab * ny * 8, cb = (size_t)ny * 8;
    const size_t dab = dA_rank == XC_DA_ROW ? cb : plane * 8;
    const size_t Mb = M_rank == XC_DA_NONE ? 0 : (M_rank == XC_DA_ROW ? cb : plane * 8);
    const size_t ob = cells * 8, mib = (size_t)nmask * 4, mob = (size_t)nmask * cells;
    XC_TRY(ensure_arena(ctx, al(qb) + al(Qb) + al(cb) + al(dab) + al(Mb) + al(ob) + al(mib) + al(mob)));
    Stage st(ctx);
    void* dq = st.take(qb); double* dQ = (double*)st.take(Qb); double* dc = (double*)st.take(cb);
    double* dd = (double*)st.take(dab); double* dM = Mb ? (double*)st.take(Mb) : nullptr;
    double* dout = (double*)st.take(ob);
    int32_t* dmi = nmask ? (int32_t*)st.take(mib) : nullptr; int8_t* dmo = nmask ? (int8_t*)st.take(mob) : nullptr;
    // the read-only planes are used where they are when they have a device mirror (the weights of a resident object: no device-to-device
    // copy per call); Q and the coordinate are small (pinned buffer + copy kernel)
    const void* pq; const void* pd; const void* pM = nullptr;
    XC_TRY(stage_in(ctx, dq, q, qb, &pq)); XC_TRY(h2d(ctx, dQ, Q, Qb)); XC_TRY(h2d(ctx, dc, coord, cb)); XC_TRY(stage_in(ctx, dd, dA, dab, &pd));
    if (Mb) XC_TRY(stage_in(ctx, dM, M, Mb, &pM));
    if (nmask) XC_TRY(h2d(ctx, dmi, mask_idx, mib));
    XC_TRY(flush_in(ctx));
    XC_TRY(launch_lwa(ctx, pq, q_dtype, dQ, dc, (const double*)pd, dA_rank, dA_max, (const double*)pM, M_rank, nslab, ny, nx, increase, part, variant,
                      dmi, nmask, dout, dmo));
    XC_TRY(d2h(ctx, out_lwa, dout, ob));
    if (nmask) XC_TRY(d2h(ctx, out_masks, dmo, mob));
    return xc_sync(ctx);
}

// ------------------------------------------------------------------------------------ K8
int xc_sort_profile_batch_dev(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype, int mask_per_slab,
                              const double* dA, int dA_rank, int64_t nslab, int64_t ny, int64_t nx, int negate,
                              const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                              double* out_Q, double* out_qsorted, double* out_acum, uint32_t* out_nvalid, double* out_bpe)
{
    XC_CTX(ctx);
    if (ny < 1 || nx < 1 || nslab < 1) return fail(ctx, XC_EBADARG, "xc_sort_profile: bad shape");
    XC_TRY(ensure_scratch(ctx, sort_workspace_bytes(ny * nx, nslab)));
    return launch_sort_profile(ctx, q, q_dtype, mask, mask_dtype, mask_per_slab, dA, dA_rank, nslab, ny, nx, negate,
                               targets, J, tbl, coord, ntbl, ctx->scratch, out_Q, out_qsorted, out_acum, out_nvalid, out_bpe);
}

int xc_sort_profile_batch(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype, int mask_per_slab,
                          const double* dA, int dA_rank, int64_t nslab, int64_t ny, int64_t nx, int negate,
                          const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                          double* out_Q, double* out_qsorted, double* out_acum, uint32_t* out_nvalid, double* out_bpe)
{
    XC_CTX(ctx);
    if (!q || ny < 1 || nx < 1 || nslab < 1 || J < 0) return fail(ctx, XC_EBADARG, "xc_sort_profile: bad arguments");
    if (q_dtype != XC_F32 && q_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_sort_profile: bad dtype");
    if (mask && mask_dtype != XC_F32 && mask_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_sort_profile: bad mask dtype");
    const size_t S = (size_t)nslab, n = (size_t)ny * nx;
    const size_t qb = S * n * esize(q_dtype), mb = mask ? (mask_per_slab ? S : 1) * n * esize(mask_dtype) : 0;
    const size_t dab = dA_rank == XC_DA_ROW ? (size_t)ny * 8 : dA_rank == XC_DA_PLANE ? n * 8 : dA_rank == XC_DA_SLAB ? S * n * 8 : 0;
    if (dab && !dA) return fail(ctx, XC_EBADARG, "xc_sort_profile: dA is NULL");
    const size_t tb = (size_t)J * 8, Qb = S * tb, tbb = (out_bpe ? (size_t)ntbl * 8 : 0);
    XC_TRY(ensure_arena(ctx, al(qb) + al(mb) + al(dab) + al(tb) + al(Qb) + 2 * al(tbb) + 2 * al(S * n * 8) + al(S * 4) + al(S * 8)));
    Stage st(ctx);
    void* dq = st.take(qb); XC_TRY(h2d(ctx, dq, q, qb));
    void* dm = nullptr; if (mb) { dm = st.take(mb); XC_TRY(h2d(ctx, dm, mask, mb)); }
    double* dd = nullptr; if (dab) { dd = (double*)st.take(dab); XC_TRY(h2d(ctx, dd, dA, dab)); }
    double* dt = nullptr; double* dQ = nullptr;
    if (J > 0 && out_Q) { dt = (double*)st.take(tb); dQ = (double*)st.take(Qb); XC_TRY(h2d(ctx, dt, targets, tb)); }
    double *dtbl = nullptr, *dcrd = nullptr;
    if (out_bpe) {
        if (!tbl || !coord || ntbl < 2) return fail(ctx, XC_EBADARG, "xc_sort_profile: BPE needs tbl/coord");
        dtbl = (double*)st.take(tbb); dcrd = (double*)st.take(tbb);
        XC_TRY(h2d(ctx, dtbl, tbl, tbb)); XC_TRY(h2d(ctx, dcrd, coord, tbb));
    }
    double* dqs = out_qsorted ? (double*)st.take(S * n * 8) : nullptr;
    double* dac = out_acum ? (double*)st.take(S * n * 8) : nullptr;
    uint32_t* dnv = (uint32_t*)st.take(S * 4);
    double* dbpe = out_bpe ? (double*)st.take(S * 8) : nullptr;
    XC_TRY(flush_in(ctx));
    XC_TRY(xc_sort_profile_batch_dev(ctx, dq, q_dtype, dm, mask_dtype, mask_per_slab, dd, dA_rank, nslab, ny, nx, negate,
                                     dt, dQ ? J : 0, dtbl, dcrd, ntbl, dQ, dqs, dac, dnv, dbpe));
    if (dQ) XC_TRY(d2h(ctx, out_Q, dQ, Qb));
    if (dqs) XC_TRY(d2h(ctx, out_qsorted, dqs, S * n * 8));
    if (dac) XC_TRY(d2h(ctx, out_acum, dac, S * n * 8));
    if (out_nvalid) XC_TRY(d2h(ctx, out_nvalid, dnv, S * 4));
    if (dbpe) XC_TRY(d2h(ctx, out_bpe, dbpe, S * 8));
    return xc_sync(ctx);
}

int xc_sort_profile_dev(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype,
                        const double* dA, int dA_rank, int64_t ny, int64_t nx, int negate,
                        const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                        double* out_Q, double* out_qsorted, double* out_acum, uint32_t* out_nvalid, double* out_bpe)
{
    if (dA_rank == XC_DA_SLAB) return fail(ctx, XC_EBADARG, "xc_sort_profile: dA_rank must be NONE, ROW or PLANE");
    return xc_sort_profile_batch_dev(ctx, q, q_dtype, mask, mask_dtype, 0, dA, dA_rank, 1, ny, nx, negate, targets, J, tbl, coord, ntbl,
                                     out_Q, out_qsorted, out_acum, out_nvalid, out_bpe);
}

int xc_sort_profile(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype,
                    const double* dA, int dA_rank, int64_t ny, int64_t nx, int negate,
                    const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                    double* out_Q, double* out_qsorted, double* out_acum, uint32_t* out_nvalid, double* out_bpe)
{
    if (dA_rank == XC_DA_SLAB) return fail(ctx, XC_EBADARG, "xc_sort_profile: dA_rank must be NONE, ROW or PLANE");
    return xc_sort_profile_batch(ctx, q, q_dtype, mask, mask_dtype, 0, dA, dA_rank, 1, ny, nx, negate, targets, J, tbl, coord, ntbl,
                                 out_Q, out_qsorted, out_acum, out_nvalid, out_bpe);
}

int xc_set_lwa_exact(xc_ctx* ctx, int exact)
{
    if (!ctx) return fail(nullptr, XC_EBADARG, "null context");
    if (exact < 0 || exact > 3) return fail(ctx, XC_EBADARG, "xc_set_lwa_exact: mode must be 0 (automatic), 1 (band walk), 2 (interval kernel, checked) or 3 (interval kernel, premises vouched for)");
    ctx->lwa_exact = exact;
    return XC_OK;
}

int xc_last_lwa_path(xc_ctx* ctx, int* out_path)
{
    if (!ctx || !out_path) return fail(ctx, XC_EBADARG, "xc_last_lwa_path: bad arguments");
    if (ctx->last_lwa_path < 0) {                                      // decided by the device-side check of the last call: read its flag
        unsigned f = 0;
        XC_HIP(ctx, hipSetDevice(ctx->device));
        XC_HIP(ctx, hipMemcpyAsync(&f, ctx->lwa_flag, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
        XC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->last_lwa_path = (f == ctx->lwa_epoch) ? 2 : 1;
    }
    *out_path = ctx->last_lwa_path;
    return XC_OK;
}

int xc_last_sort_path(xc_ctx* ctx, int* out_path)
{
    if (!ctx || !out_path) return fail(ctx, XC_EBADARG, "xc_last_sort_path: bad arguments");
    *out_path = ctx->last_sort_path;
    return XC_OK;
}

// ------------------------------------------------------------------------------------ K5 / K6 alone
int xc_keff_epilogue_dev(xc_ctx* ctx, const double* pdf, const double* ctr, int ctr_dtype, int64_t nslab, int N,
                         int increase, int lt, const double* tbl, const double* tbl_coord, int ntbl,
                         const double* preY, int npre, double nkeff_mask, double lmin_scale,
                         double* area, double* intgrdS, double* latEq, double* dqdA, double* dintSdA,
                         double* Leq2, double* Lmin, double* nkeff, double* interp)
{
    XC_CTX(ctx);
    if (!pdf || !ctr || !tbl || !tbl_coord) return fail(ctx, XC_EBADARG, "xc_keff_epilogue: pdf / ctr / tbl / tbl_coord must be given");
    if (nslab < 1 || N < 2 || ntbl < 2 || npre < 0) return fail(ctx, XC_EBADARG, "xc_keff_epilogue: need nslab >= 1, N >= 2, ntbl >= 2");
    if (ctr_dtype != XC_F32 && ctr_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_keff_epilogue: bad ctr_dtype");
    if (npre > 0 && interp && !preY) return fail(ctx, XC_EBADARG, "xc_keff_epilogue: preY is NULL");
    // the reduce stage sums ONE "partial" per slab (the pdf itself); its count channel reads zeros
    const int nch = 2;
    const size_t S = (size_t)nslab;
    const size_t pc = al(S * N * sizeof(unsigned)), rh = al(S * nch * N * sizeof(double)), rc = al(S * N * sizeof(unsigned long long));
    XC_TRY(ensure_scratch(ctx, pc + rh + rc));
    char* base = (char*)ctx->scratch;
    XC_HIP(ctx, hipMemsetAsync(base, 0, pc, ctx->stream));
    ctx->mm_valid = 0;                          // the scratch may have moved
    FinalArgs f; memset(&f, 0, sizeof(f));
    f.part_h = pdf; f.part_c = (const unsigned*)base; f.bps = 1; f.nch = nch; f.nbin = N;
    f.red_h = (double*)(base + pc); f.red_c = (unsigned long long*)(base + pc + rh);
    f.lt = lt; f.reverse = !increase;
    f.keff = 1; f.ctr_f32 = ctr_dtype == XC_F32; f.ctr = ctr;
    f.tbl = tbl; f.tbl_coord = tbl_coord; f.ntbl = ntbl;
    f.preY = preY; f.npre = interp ? npre : 0;
    f.nkeff_mask = nkeff_mask; f.lmin_scale = lmin_scale;
    f.o_area = area; f.o_intS = intgrdS; f.o_latEq = latEq; f.o_dqdA = dqdA; f.o_dSdA = dintSdA;
    f.o_Leq2 = Leq2; f.o_Lmin = Lmin; f.o_nkeff = nkeff; f.o_interp = interp;
    return launch_finalize(ctx, nslab, f);
}

int xc_keff_epilogue(xc_ctx* ctx, const double* pdf, const double* ctr, int ctr_dtype, int64_t nslab, int N,
                     int increase, int lt, const double* tbl, const double* tbl_coord, int ntbl,
                     const double* preY, int npre, double nkeff_mask, double lmin_scale,
                     double* area, double* intgrdS, double* latEq, double* dqdA, double* dintSdA,
                     double* Leq2, double* Lmin, double* nkeff, double* interp)
{
    XC_CTX(ctx);
    if (!pdf || !ctr || !tbl || !tbl_coord || nslab < 1 || N < 2 || ntbl < 2 || npre < 0)
        return fail(ctx, XC_EBADARG, "xc_keff_epilogue: bad arguments");
    const size_t S = (size_t)nslab, vb = S * N * 8, pb = 2 * vb, tb = (size_t)ntbl * 8, yb = (size_t)npre * 8, ib = S * 9 * npre * 8;
    XC_TRY(ensure_arena(ctx, al(pb) + al(vb) + 2 * al(tb) + al(yb) + 8 * al(vb) + al(ib)));
    Stage st(ctx);
    double* dp = (double*)st.take(pb); double* dc = (double*)st.take(vb);
    double* dt = (double*)st.take(tb); double* dy = (double*)st.take(tb);
    double* dpre = npre > 0 ? (double*)st.take(yb) : nullptr;
    double* o[8]; for (int i = 0; i < 8; ++i) o[i] = (double*)st.take(vb);
    double* di = (interp && npre > 0) ? (double*)st.take(ib) : nullptr;
    XC_TRY(h2d(ctx, dp, pdf, pb)); XC_TRY(h2d(ctx, dc, ctr, vb));
    XC_TRY(h2d(ctx, dt, tbl, tb)); XC_TRY(h2d(ctx, dy, tbl_coord, tb));
    if (dpre) { if (!preY) return fail(ctx, XC_EBADARG, "xc_keff_epilogue: preY is NULL"); XC_TRY(h2d(ctx, dpre, preY, yb)); }
    XC_TRY(flush_in(ctx));
    XC_TRY(xc_keff_epilogue_dev(ctx, dp, dc, ctr_dtype, nslab, N, increase, lt, dt, dy, ntbl, dpre, npre, nkeff_mask, lmin_scale,
                                o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], di));
    double* host[8] = {area, intgrdS, latEq, dqdA, dintSdA, Leq2, Lmin, nkeff};
    for (int i = 0; i < 8; ++i) if (host[i]) XC_TRY(d2h(ctx, host[i], o[i], vb));
    if (interp && di) XC_TRY(d2h(ctx, interp, di, ib));
    return xc_sync(ctx);
}

// ------------------------------------------------------------------------------------ fused Keff pipeline
constexpr int XC_EAGAIN = -1000;      // internal: "not this path" (never leaves the library)

static int keff_single(xc_ctx* ctx, const xc_keff_desc* d)
{
    const int N = d->N;
    const int vstride = d->out_stride ? d->out_stride : N;
    const double* dA = d->dA; int dA_rank = d->dA_rank;
    if (dA_rank == XC_DA_NONE) { XC_TRY(ensure_ones(ctx, (size_t)d->ny)); dA = ctx->ones; dA_rank = XC_DA_ROW; }
    SingleGeom g;
    if (!single_geometry(ctx, d->q_dtype, d->nslab, d->ny, d->nx, N, d->q, dA, dA_rank, &g)) return XC_EAGAIN;
    if (!ctx->single_ws) {
        XC_HIP(ctx, hipMalloc(&ctx->single_ws, 2 * sizeof(SingleSet)));
        XC_HIP(ctx, hipMemsetAsync(ctx->single_ws, 0, 2 * sizeof(SingleSet), ctx->stream));
        ctx->single_launches = 0; ctx->single_dirty_bins[0] = ctx->single_dirty_bins[1] = 0;
    }
    const unsigned cur = ctx->single_launches & 1u;
    SingleSet* sets = (SingleSet*)ctx->single_ws;
    SingleArgs a; memset(&a, 0, sizeof(a));
    a.q = d->q; a.dA = dA; a.dA_rank = dA_rank;
    a.dA_pos_finite = (d->dA_rank == XC_DA_NONE) ? 1 : d->dA_pos_finite;
    a.rdx = d->rdx; a.rdy = d->rdy; a.periodic_x = d->periodic_x;
    a.ny = d->ny; a.nx = d->nx; a.nslab = (int)d->nslab;
    a.nbin = N; a.ncopy = g.ncopy; a.increase = d->increase; a.q_f32 = d->q_dtype == XC_F32; a.ctr_f32 = d->ctr_dtype == XC_F32;
    a.right_edge = d->right_edge; a.last_closed = d->right_edge == XC_EDGE_NUMPY; a.want_counts = d->counts ? 1 : 0;
    a.inv_nm1 = 1.0 / (double)(N - 1); a.inv_n = 1.0 / (double)N;
    a.G = g.G; a.nstrip = g.nstrip; a.cps = g.cps; a.rpc = g.rpc; a.strip_fast = ctx->knobs.single_map;
    a.cur = sets + cur; a.other = sets + (1u - cur); a.other_dirty_bins = ctx->single_dirty_bins[1u - cur];
    a.ctr_out = d->ctr; a.ctr_stride = vstride; a.status = d->status;
    a.timeout_ticks = (unsigned long long)(ctx->knobs.single_timeout_us > 0 ? ctx->knobs.single_timeout_us : 1) * 100ull;   // 100 MHz wall clock
    a.stamps = ctx->single_stamps;
    FinalArgs f; memset(&f, 0, sizeof(f));
    f.bps = 1; f.nch = 2; f.nbin = N; f.skip_reduce = 1;
    f.red_h = a.cur->acc_h; f.red_c = a.cur->acc_c;
    f.lt = d->lt; f.reverse = !d->increase;
    f.counts = d->counts;
    f.keff = 1; f.ctr_f32 = a.ctr_f32; f.ctr = d->ctr; f.vstride = vstride;
    f.tbl = d->tbl; f.tbl_coord = d->tbl_coord; f.ntbl = (int)d->ny;
    f.preY = d->preY; f.npre = d->interp ? d->npre : 0;
    f.nkeff_mask = d->nkeff_mask; f.lmin_scale = d->lmin_scale;
    f.o_area = d->area; f.o_intS = d->intgrdS; f.o_latEq = d->latEq; f.o_dqdA = d->dqdA; f.o_dSdA = d->dintSdA;
    f.o_Leq2 = d->Leq2; f.o_Lmin = d->Lmin; f.o_nkeff = d->nkeff; f.o_interp = d->interp;
    f.abort_flag = &a.cur->abort; f.status_out = d->status;
    f.dbg = ctx->single_stamps ? ctx->single_stamps + 8 : nullptr;            // (slots 8.. of workgroup 0, slab 0)
    ctx->mm_valid = 0;                                        // (no chained min/max comes out of this path)
    XC_TRY(hist_ev_begin(ctx));
    XC_TRY(launch_keff_single(ctx, d->q_dtype, a, g));
    XC_TRY(hist_ev_end(ctx));
    XC_TRY(launch_finalize(ctx, d->nslab, f));
    ctx->single_dirty_bins[cur] = N;                          // what the NEXT launch clears in this set
    ctx->single_dirty_bins[1u - cur] = 0;
    ++ctx->single_launches;
    ctx->last_keff_path = 1;
    return XC_OK;
}

int xc_last_keff_path(xc_ctx* ctx, int* out_path)
{
    if (!ctx || !out_path) return fail(ctx, XC_EBADARG, "xc_last_keff_path: bad arguments");
    *out_path = ctx->last_keff_path;
    return XC_OK;
}

// diagnostics: wall-clock stamps (100 MHz) of thread 0 of every workgroup of the single-read kernel at its phase boundaries;
// enable != 0 allocates [kSingleMaxSlabs][cus][12] uint64 and returns the device pointer, 0 frees it
int xc_dbg_single_stamps(xc_ctx* ctx, int enable, void** out_dev, int* out_slots)
{
    XC_CTX(ctx);
    if (ctx->single_stamps) { XC_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->single_stamps); ctx->single_stamps = nullptr; }
    if (enable) {
        const size_t bytes = (size_t)kSingleMaxSlabs * ctx->cus * kSingleStampSlots * sizeof(unsigned long long);
        XC_HIP(ctx, hipMalloc((void**)&ctx->single_stamps, bytes));
        XC_HIP(ctx, hipMemset(ctx->single_stamps, 0, bytes));
    }
    if (out_dev) *out_dev = ctx->single_stamps;
    if (out_slots) *out_slots = kSingleStampSlots;
    return XC_OK;
}

int xc_keff_dev(xc_ctx* ctx, const xc_keff_desc* d)
{
    XC_CTX(ctx);
    if (!d) return fail(ctx, XC_EBADARG, "xc_keff: desc is NULL");
    if (!d->q || !d->ctr || !d->area || !d->tbl || !d->tbl_coord) return fail(ctx, XC_EBADARG, "xc_keff: q/ctr/area/tbl/tbl_coord must be given");
    if (d->q_dtype != XC_F32 && d->q_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_keff: bad q_dtype");
    if (d->ctr_dtype != XC_F32 && d->ctr_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_keff: bad ctr_dtype");
    if (d->nslab < 1 || d->ny < 2 || d->nx < 1 || d->N < 2) return fail(ctx, XC_EBADARG, "xc_keff: need nslab>=1, ny>=2, nx>=1, N>=2");
    if (d->dA_rank < XC_DA_NONE || d->dA_rank > XC_DA_SLAB || (d->dA_rank != XC_DA_NONE && !d->dA)) return fail(ctx, XC_EBADARG, "xc_keff: bad dA");
    if (d->grad ? (!d->rdx || !d->rdy) : !d->grdS) return fail(ctx, XC_EBADARG, "xc_keff: need rdx/rdy (grad=1) or grdS (grad=0)");
    if (d->npre < 0 || (d->npre > 0 && d->interp && !d->preY)) return fail(ctx, XC_EBADARG, "xc_keff: preY is NULL");
    if (d->out_stride != 0 && d->out_stride < d->N) return fail(ctx, XC_EBADARG, "xc_keff: out_stride must be 0 (dense) or >= N");
    const int vstride = d->out_stride ? d->out_stride : d->N;
    const int N = d->N, nch = 2;
    const int det = d->deterministic ? 1 : 0;
    const void* q_next = d->q_next;                          // (deterministic sums: carried by the fixed-point pass)
    ctx->last_keff_path = 0;
    // ONE slab (the reference's callers hand planes over one at a time): the single-read kernel (xc_keff1.hip) where the slab fits the
    // register tiles of the chip -- min/max, levels and histogram in ONE launch, the tracer read ONCE.  (Two slabs: only when forced --
    // the kernel runs them one after the other, 64.7 us against the chain's 59.3.)
    if (d->single_read != XC_SINGLE_NEVER && ctx->knobs.single != 0 && !det && d->grad &&
        d->nslab <= (d->single_read == XC_SINGLE_FORCE ? kSingleMaxSlabs : 1)) {
        const int rc = keff_single(ctx, d);
        if (rc != XC_EAGAIN) return rc;                      // (XC_EAGAIN: the shape does not suit it; nothing was enqueued)
    }
    HistGeom g;
    {
        const void* gi[1] = {d->grdS}; const int32_t gt[1] = {d->grdS_dtype};
        const bool da2d = d->dA_rank == XC_DA_PLANE || d->dA_rank == XC_DA_SLAB;
        const bool fast_layout = d->grad && da2d && d->periodic_x && d->dA_pos_finite && d->right_edge != XC_EDGE_NUMPY;
        // float32 tracer with a supplied float32 squared gradient (the reference's own workflow): four cells per lane too
        const bool supplied_f32 = !d->grad && da2d && d->q_dtype == XC_F32 && d->grdS_dtype == XC_F32 &&
                                  reinterpret_cast<uintptr_t>(d->grdS) % 16 == 0;
        XC_TRY(hist_geometry(ctx, d->q_dtype, d->nslab, d->ny, d->nx, N, nch,
                             vec_align_bits(d->q, d->q_dtype, d->q_next, d->dA, d->dA_rank, gi, gt, d->grad ? 0 : 1), &g,
                             fast_layout ? 1 : (supplied_f32 ? 2 : 0), det));
    }
    const size_t mb = al((size_t)d->nslab * kMinmaxBlocks * 2 * sizeof(double));
    // (deterministic sums: a block's partial is the limbs of its superaccumulators instead of one double per channel)
    const size_t ph = al((size_t)d->nslab * g.bps * (det ? det_limbs_total(nch) : nch) * N * sizeof(double));
    const size_t pc = al((size_t)d->nslab * g.bps * N * sizeof(unsigned));
    const size_t rh = al((size_t)d->nslab * nch * N * sizeof(double));
    const size_t rc = al((size_t)d->nslab * N * sizeof(unsigned long long));
    const size_t dx = det ? al((size_t)d->nslab * nch * sizeof(int)) + det_bounds_bytes(d->nslab) : 0;
    XC_TRY(ensure_scratch(ctx, mb + ph + pc + rh + rc + dx));
    double* mmpart = (double*)ctx->scratch;
    double* part_h = (double*)((char*)ctx->scratch + mb);
    unsigned* part_c = (unsigned*)((char*)ctx->scratch + mb + ph);
    int mmP = minmax_blocks(d->ny * d->nx, d->nslab);
    bool accum = false;

    // min/max partials: either produced by the previous call's histogram pass (q_next) or by K1 now
    if (ctx->mm_valid && ctx->mm_q == d->q && ctx->mm_nslab == d->nslab && ctx->mm_ny == d->ny &&
        ctx->mm_nx == d->nx && ctx->mm_dtype == d->q_dtype && ctx->mm_gen == d->q_gen) {
        mmpart = ctx->mmnext[ctx->mm_cur]; mmP = ctx->mm_P;
    } else {
        // FEW slabs (more than 64 blocks per slab: a single cfg2 slab has ~200): the blocks add into per-slab accumulators, cleared by
        // this K1 launch, and the finalize kernel reads them directly -- three dependent launches instead of four
        accum = !det && g.bps > 64;
        XC_TRY(launch_minmax_partial(ctx, d->q, d->q_dtype, d->nslab, d->ny * d->nx, mmpart,
                                     accum ? (double*)((char*)ctx->scratch + mb + ph + pc) : nullptr, accum ? (int64_t)((rh + rc) / 8) : 0));
    }
    ctx->mm_valid = 0;
    double* mm_next = nullptr;
    if (q_next) {
        const int nb = 1 - ctx->mm_cur;
        XC_TRY(grow(ctx, (void**)&ctx->mmnext[nb], &ctx->mmnext_bytes[nb], al((size_t)d->nslab * g.bps * 2 * sizeof(double))));
        mm_next = ctx->mmnext[nb];
    }

    HistArgs a; memset(&a, 0, sizeof(a));
    a.q = d->q; a.dA = d->dA; a.dA_rank = d->dA_rank;
    if (d->dA_rank == XC_DA_NONE) { XC_TRY(ensure_ones(ctx, (size_t)d->ny)); a.dA = ctx->ones; a.dA_rank = XC_DA_ROW; }
    if (!d->grad) { a.integ[0] = d->grdS; a.integ_f32[0] = d->grdS_dtype == XC_F32; }
    a.q_next = q_next; a.mm_next = mm_next;
    a.dA_pos_finite = (d->dA_rank == XC_DA_NONE) ? 1 : d->dA_pos_finite;
    a.mmpart = mmpart; a.P = mmP; a.levels_mode = 1; a.nbin = N;
    a.last_closed = d->right_edge == XC_EDGE_NUMPY;
    a.increase = d->increase; a.q_f32 = d->q_dtype == XC_F32; a.ctr_f32 = d->ctr_dtype == XC_F32;
    a.right_edge = d->right_edge; a.inv_nm1 = 1.0 / (double)(N - 1);
    a.prod_f32 = d->prod_f32;
    a.rdx = d->rdx; a.rdy = d->rdy; a.periodic_x = d->periodic_x;
    a.ny = d->ny; a.nx = d->nx; a.nstrip = g.nstrip; a.ncopy = g.ncopy;
    if (!d->counts && !det) part_c = nullptr;            // nobody asked for counts: the histogram pass skips their LDS adds (a third of its atomics)
    a.part_h = part_h; a.part_c = part_c; a.ctr_out = d->ctr; a.ctr_stride = vstride; a.status = d->status;
    FinalArgs f; memset(&f, 0, sizeof(f));
    f.part_h = part_h; f.part_c = part_c; f.bps = g.bps; f.nch = nch; f.nbin = N;
    f.red_h = (double*)((char*)ctx->scratch + mb + ph + pc); f.red_c = (unsigned long long*)((char*)ctx->scratch + mb + ph + pc + rh);
    if (accum) { a.acc_h = f.red_h; a.acc_c = d->counts ? f.red_c : nullptr; f.skip_reduce = 1; }
    XC_TRY(hist_ev_begin(ctx));
    if (det) {
        char* xb = (char*)ctx->scratch + mb + ph + pc + rh + rc;
        const void* gi[1] = {d->grdS}; const int gt[1] = {d->grdS_dtype};
        XC_TRY(det_one_pass(ctx, d->q_dtype, d->grad ? 0 : 1, d->grad, g, d->nslab, d->ny, d->nx, a, (int*)xb,
                            xb + al((size_t)d->nslab * nch * sizeof(int)), d->dA_max, gi, gt, f));
    } else {
        XC_TRY(launch_hist(ctx, d->q_dtype, d->grad ? 0 : 1, d->grad, g, d->nslab, a));
    }
    XC_TRY(hist_ev_end(ctx));
    if (q_next) {
        ctx->mm_cur = 1 - ctx->mm_cur; ctx->mm_valid = 1; ctx->mm_P = g.bps; ctx->mm_q = q_next;
        ctx->mm_nslab = d->nslab; ctx->mm_ny = d->ny; ctx->mm_nx = d->nx; ctx->mm_dtype = d->q_dtype;
        ctx->mm_gen = d->q_gen;
    }

    f.lt = d->lt; f.reverse = !d->increase;       // decreasing levels -> flip to level order (core.py:454-455)
    f.counts = d->counts;
    f.keff = 1; f.ctr_f32 = a.ctr_f32; f.ctr = d->ctr; f.vstride = vstride;
    f.tbl = d->tbl; f.tbl_coord = d->tbl_coord; f.ntbl = (int)d->ny;
    f.preY = d->preY; f.npre = d->interp ? d->npre : 0;
    f.nkeff_mask = d->nkeff_mask; f.lmin_scale = d->lmin_scale;
    f.o_area = d->area; f.o_intS = d->intgrdS; f.o_latEq = d->latEq; f.o_dqdA = d->dqdA; f.o_dSdA = d->dintSdA;
    f.o_Leq2 = d->Leq2; f.o_Lmin = d->Lmin; f.o_nkeff = d->nkeff; f.o_interp = d->interp;
    return launch_finalize(ctx, d->nslab, f);
}

// ------------------------------------------------------------------------------------ synthetic slabs
int xc_synth_dev(xc_ctx* ctx, void* out, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                 const double* lat_deg, const double* lon_deg, uint64_t seed, int variant)
{
    XC_CTX(ctx);
    if (out && nslab > 0 && ny > 0 && nx > 0) mm_touch(ctx, out, (size_t)nslab * ny * nx * esize(q_dtype));
    return launch_synth(ctx, out, q_dtype, nslab, ny, nx, lat_deg, lon_deg, seed, variant);
}

}  // extern "C"
