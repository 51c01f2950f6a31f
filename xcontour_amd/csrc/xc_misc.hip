// K1 min/max, stand-alone levels, K5/K6 finalize + Keff epilogue, K2 row sums,
// K4 stand-alone |grad q|^2, synthetic slab generator.  gfx950 only.
#include "xc_internal.h"

namespace xc {

namespace {

__device__ __forceinline__ double dnan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ double dinf() { return __longlong_as_double(0x7ff0000000000000LL); }

// =====================================================================================
// K1  NaN-skipping min / max    (tracer.min/max(dim=dimVs), core.py:224-225)
// One streaming pass, 16-byte loads, 4 in flight per lane; per-block partial pairs are
// written with plain stores and reduced in fixed order by the consumer (k_minmax_final
// or the K3 prologue): deterministic, no atomics, no memset.
// =====================================================================================
template <typename T> struct V16;
template <> struct V16<double> { using type = double2; static constexpr int n = 2; typedef double ext __attribute__((ext_vector_type(2))); };
template <> struct V16<float>  { using type = float4;  static constexpr int n = 4; typedef float ext __attribute__((ext_vector_type(4))); };
// one 16-byte load with the non-temporal hint: the tracer is read once here (a pure read stream reaches 6.5-7.1 TB/s with the
// hint against 5.9-6.3 without it, tools/probe/bw_probe.hip)
template <typename T>
__device__ __forceinline__ typename V16<T>::type ld16_nt(const typename V16<T>::type* p)
{
    const typename V16<T>::ext t = __builtin_nontemporal_load(reinterpret_cast<const typename V16<T>::ext*>(p));
    typename V16<T>::type r;
    if constexpr (V16<T>::n == 2) { r.x = t.x; r.y = t.y; } else { r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w; }
    return r;
}

__device__ __forceinline__ void mm(double& mn, double& mx, double v) { mn = fmin(mn, v); mx = fmax(mx, v); }
__device__ __forceinline__ void mmv(double& mn, double& mx, const double2& v) { mm(mn, mx, v.x); mm(mn, mx, v.y); }
__device__ __forceinline__ void mmv(double& mn, double& mx, const float4& v)
{
    // reduce in f32 first (exact), then widen
    const float lo = fminf(fminf(v.x, v.y), fminf(v.z, v.w));
    const float hi = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    // fminf/fmaxf return the non-NaN operand; all four NaN -> NaN, skipped by fmin/fmax below
    mn = fmin(mn, (double)lo); mx = fmax(mx, (double)hi);
}

// FIN: +-infinity is skipped like NaN (the deterministic sums derive the windows of their accumulators from these extrema: the bound
// must be the largest FINITE magnitude, as in the oracle -- an infinite one would push every finite weight out of the window)
// NT: the streaming hint (a pass over more bytes than the 256 MiB Infinity Cache holds: nothing read here is read again before it would be
// evicted anyway).  Without it -- a launch of a few slabs -- the tracer stays in the Infinity Cache for the histogram pass that follows.
template <bool FIN> __device__ __forceinline__ double2 fin_only(double2 v)
{
    if (FIN) { if (fabs(v.x) == dinf()) v.x = dnan(); if (fabs(v.y) == dinf()) v.y = dnan(); }
    return v;
}
template <bool FIN> __device__ __forceinline__ float4 fin_only(float4 v)
{
    if (FIN) {
        const float nanf_ = __int_as_float(0x7fc00000), inff = __int_as_float(0x7f800000);
        if (fabsf(v.x) == inff) v.x = nanf_; if (fabsf(v.y) == inff) v.y = nanf_;
        if (fabsf(v.z) == inff) v.z = nanf_; if (fabsf(v.w) == inff) v.w = nanf_;
    }
    return v;
}
template <bool FIN> __device__ __forceinline__ double fin_only(double v) { return (FIN && fabs(v) == dinf()) ? dnan() : v; }

template <typename T, bool NT, bool FIN = false>
__global__ __launch_bounds__(256)
void k_minmax_partial(const T* __restrict__ q, int64_t ncell, double* __restrict__ part, double* __restrict__ zero, int64_t nzero)
{
    using V = typename V16<T>::type;
    constexpr int VN = V16<T>::n;
    const int P = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    // (round 5) a few words some LATER kernel of the chain wants cleared (the accumulators a few-slab histogram pass adds into): done
    // here, by the first threads of the grid, instead of by a memset launch of its own between two dependent kernels
    if (zero)
        for (int64_t w = ((int64_t)blockIdx.y * P + b) * 256 + tid; w < nzero; w += (int64_t)gridDim.y * P * 256) zero[w] = 0.0;
    const T* qs = q + (size_t)blockIdx.y * ncell;
    double mn = dinf(), mx = -dinf();

    // leading scalars up to 16-byte alignment, vector body, trailing scalars
    int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(qs) & 15)) & 15) / sizeof(T));
    if (head > ncell) head = ncell;
    const int64_t nvec = (ncell - head) / VN;
    const V* qv = reinterpret_cast<const V*>(qs + head);
    const int64_t per = (nvec + P - 1) / P;
    const int64_t v0 = (int64_t)b * per;
    const int64_t v1 = (v0 + per < nvec) ? v0 + per : nvec;
    int64_t i = v0 + tid;
    for (; i + 7 * 256 < v1; i += 8 * 256) {
        V a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = NT ? ld16_nt<T>(qv + i + u * 256) : qv[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) mmv(mn, mx, fin_only<FIN>(a[u]));
    }
    for (; i < v1; i += 256) { const V a0 = NT ? ld16_nt<T>(qv + i) : qv[i]; mmv(mn, mx, fin_only<FIN>(a0)); }
    if (b == 0) {
        for (int64_t j = tid; j < head; j += 256) mm(mn, mx, fin_only<FIN>((double)qs[j]));
        for (int64_t j = head + nvec * VN + tid; j < ncell; j += 256) mm(mn, mx, fin_only<FIN>((double)qs[j]));
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
    __shared__ double s[8];
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) { s[2 * wave] = mn; s[2 * wave + 1] = mx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) { mn = fmin(mn, s[2 * w]); mx = fmax(mx, s[2 * w + 1]); }
        double* o = part + ((size_t)blockIdx.y * P + b) * 2;
        o[0] = mn; o[1] = mx;
    }
}

__global__ __launch_bounds__(256)
void k_minmax_final(const double* __restrict__ part, int P, double* __restrict__ out)
{
    const int tid = threadIdx.x;
    const double* mp = part + (size_t)blockIdx.x * P * 2;
    double mn = dinf(), mx = -dinf();
    for (int i = tid; i < P; i += 256) { mn = fmin(mn, mp[2 * i]); mx = fmax(mx, mp[2 * i + 1]); }
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
    __shared__ double s[8];
    if ((tid & 63) == 0) { s[2 * (tid >> 6)] = mn; s[2 * (tid >> 6) + 1] = mx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) { mn = fmin(mn, s[2 * w]); mx = fmax(mx, s[2 * w + 1]); }
        if (mn == dinf() && mx == -dinf()) { mn = dnan(); mx = dnan(); }
        out[2 * blockIdx.x] = mn; out[2 * blockIdx.x + 1] = mx;
    }
}

// =====================================================================================
// stand-alone levels + edges (same arithmetic as the K3 prologue; core.py:228-246, 1296-1305)
// =====================================================================================
// P > 0: `minmax` holds K1's per-block partials [slab][P][2] -- reduced here (the arithmetic of k_minmax_final), one launch less in
// xc_contours; the slab's pair goes to minmax_out when that is not NULL.  Every output word is written ONCE and none is read back: ctr /
// edges / status may be pinned host memory (round 6: the results of a small call are written straight into the buffer xc_sync hands over).
__global__ __launch_bounds__(256)
void k_levels(const double* __restrict__ minmax, int P, double* __restrict__ minmax_out, int N, int increase, int q_f32, int ctr_f32,
              int right_edge, double inv_nm1, double* __restrict__ ctr, double* __restrict__ edges,
              int32_t* __restrict__ status)
{
    const int slab = blockIdx.x, tid = threadIdx.x;
    __shared__ double s[8], s_mm[2];
    if (P > 0) {
        const double* mp = minmax + (size_t)slab * P * 2;
        double a = dinf(), b = -dinf();
        for (int i = tid; i < P; i += 256) { a = fmin(a, mp[2 * i]); b = fmax(b, mp[2 * i + 1]); }
        for (int o = 32; o > 0; o >>= 1) { a = fmin(a, __shfl_xor(a, o)); b = fmax(b, __shfl_xor(b, o)); }
        if ((tid & 63) == 0) { s[2 * (tid >> 6)] = a; s[2 * (tid >> 6) + 1] = b; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) { a = fmin(a, s[2 * w]); b = fmax(b, s[2 * w + 1]); }
            if (a == dinf() && b == -dinf()) { a = dnan(); b = dnan(); }
            s_mm[0] = a; s_mm[1] = b;
            if (minmax_out) { minmax_out[2 * slab] = a; minmax_out[2 * slab + 1] = b; }
        }
    } else if (tid == 0) { s_mm[0] = minmax[2 * slab]; s_mm[1] = minmax[2 * slab + 1]; }
    __syncthreads();
    const double mn = s_mm[0], mx = s_mm[1];
    double* e = edges + (size_t)slab * (N + 1);
    const double start = increase ? mn : mx, stop = increase ? mx : mn;
    const double d = q_f32 ? (double)__fsub_rn((float)stop, (float)start) : __dsub_rn(stop, start);
    const double steps = __dmul_rn(inv_nm1, d);
    auto level = [&](int k) {
        double c = __dadd_rn(__dmul_rn(steps, (double)k), start);
        if (ctr_f32) c = (double)(float)c;
        return c;
    };
    const bool bump = right_edge == XC_EDGE_XHISTOGRAM;
    int bad = 0;
    for (int k = tid; k < N; k += 256) {
        const double c = level(k);
        ctr[(size_t)slab * N + k] = c;
        const int idx = increase ? k + 1 : N - k;
        e[idx] = (idx == N && bump) ? (ctr_f32 ? (double)__fadd_rn((float)c, (float)1e-8) : __dadd_rn(c, 1e-8)) : c;
        if (k > 0) bad |= (c == level(k - 1));
    }
    bad = __syncthreads_or(bad);
    if (tid == 0) {
        const double lo = level(increase ? 0 : N - 1), hi = level(increase ? N - 1 : 0);
        if (ctr_f32) {
            const float step = __fdiv_rn(__fsub_rn((float)hi, (float)lo), (float)(N - 1));
            e[0] = (double)__fsub_rn((float)lo, step);
        } else {
            const double step = __ddiv_rn(__dsub_rn(hi, lo), (double)(N - 1));
            e[0] = __dsub_rn(lo, step);
        }
        status[slab] = bad ? 1 : 0;
    }
}

#include "xc_finalize.h"

// Stage 1: sum the per-block partial histograms of K3 in a fixed order.  8 values per
// 256-thread block, 32 lanes per value (each lane owns partials l, l+32, ...), then a
// fixed xor-shuffle tree: deterministic, fully parallel (no serial latency chain).
__global__ __launch_bounds__(256)
void k_reduce_partials(const double* __restrict__ part_h, const unsigned* __restrict__ part_c,
                       int bps, int nvh, int nbin, double* __restrict__ red_h,
                       unsigned long long* __restrict__ red_c)
{
    const int slab = blockIdx.y, tid = threadIdx.x, l = tid & 31;
    const int v = blockIdx.x * 8 + (tid >> 5);
    if (v < nvh) {
        const double* p = part_h + (size_t)slab * bps * nvh + v;
        double sum = 0.0;
        for (int b = l; b < bps; b += 32) sum += p[(size_t)b * nvh];
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (l == 0) red_h[(size_t)slab * nvh + v] = sum;
    } else if (part_c && v < nvh + nbin) {
        const int k = v - nvh;
        const unsigned* p = part_c + (size_t)slab * bps * nbin + k;
        unsigned long long sum = 0;
        for (int b = l; b < bps; b += 32) sum += p[(size_t)b * nbin];
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (l == 0) red_c[(size_t)slab * nbin + k] = sum;
    }
}

// Stage 2: one block per slab (the body lives in xc_finalize.h).
__global__ __launch_bounds__(1024)
void k_finalize(const FinalArgs a)
{
    extern __shared__ __align__(16) double sm_lds[];
    finalize_body(a, (int)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, sm_lds);
}

// =====================================================================================
// K2  per-row sums of dA * [mask == 1]        (cal_area_eqCoord_table_hist, core.py:176-193)
// one wave per row, 4 rows per block
// =====================================================================================
template <typename TM>
__global__ __launch_bounds__(256)
void k_rowsum(const TM* __restrict__ mask, const double* __restrict__ dA, int dA_rank,
              int64_t ny, int64_t nx, int multiply, double* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= ny) return;
    double sum = 0.0;
    const double rowv = (dA_rank == XC_DA_ROW) ? dA[row] : 1.0;
    for (int64_t x = lane; x < nx; x += 64) {
        const double w = (dA_rank == XC_DA_PLANE) ? dA[row * nx + x] : rowv;
        if (multiply) {                                   // (mask*dA).sum(skipna), core.py:130-133
            const double t = mask ? (double)mask[row * nx + x] * w : w;
            if (t == t) sum += t;
        } else {
            const bool in = mask ? (mask[row * nx + x] == (TM)1) : true;     // .where(mask==1), core.py:178
            if (in) sum += w;
        }
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) out[row] = sum;
}

// =====================================================================================
// K4  stand-alone |grad q|^2 (oracle grad2_sphere; no reference call site)
// =====================================================================================
// a thread walks GRAD_ROWS rows down its column with the south / centre / north values rolling through registers:
// one new (coalesced) load per cell for the meridional difference, the zonal neighbours are cache hits
constexpr int GRAD_ROWS = 16;
template <typename T>
__global__ __launch_bounds__(256)
void k_grad2(const T* __restrict__ q, int64_t ny, int64_t nx, const double* __restrict__ rdx,
             const double* __restrict__ rdy, int periodic_x, double* __restrict__ out)
{
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= nx) return;
    const int64_t y0 = (int64_t)blockIdx.y * GRAD_ROWS, y1 = (y0 + GRAD_ROWS < ny) ? y0 + GRAD_ROWS : ny;
    const T* qs = q + (size_t)blockIdx.z * ny * nx;
    double* os = out + (size_t)blockIdx.z * ny * nx;
    const int64_t xw = (x == 0) ? (periodic_x ? nx - 1 : 0) : x - 1;
    const int64_t xe = (x == nx - 1) ? (periodic_x ? 0 : nx - 1) : x + 1;
    const double f = (!periodic_x && (x == 0 || x == nx - 1)) ? 2.0 : 1.0;
    double qS = (double)qs[(y0 > 0 ? y0 - 1 : 0) * nx + x], qC = (double)qs[y0 * nx + x];
#pragma unroll 4
    for (int64_t y = y0; y < y1; ++y) {
        const double qN = (double)qs[(y < ny - 1 ? y + 1 : ny - 1) * nx + x];
        const double gx = __dmul_rn(__dmul_rn(__dsub_rn((double)qs[y * nx + xe], (double)qs[y * nx + xw]), rdx[y]), f);
        const double gy = __dmul_rn(__dsub_rn(qN, (y > 0) ? qS : qC), rdy[y]);
        os[y * nx + x] = __dadd_rn(__dmul_rn(gx, gx), __dmul_rn(gy, gy));
        qS = qC; qC = qN;
    }
}

// =====================================================================================
// synthetic PV-like slabs, counter-based RNG (splitmix64 finaliser -> Box-Muller)
// =====================================================================================
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t r) { return ((double)(r >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

template <typename T>
__global__ __launch_bounds__(256)
void k_synth(T* __restrict__ out, int64_t ny, int64_t nx, const double* __restrict__ lat,
             const double* __restrict__ lon, uint64_t seed, int variant)
{
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= nx) return;
    const uint64_t sseed = seed + blockIdx.z;
    const double d2r = 0.017453292519943295;
    const double phi = lat[y] * d2r, lam = lon[x] * d2r;
    const uint64_t cell = (uint64_t)(y * nx + x);
    const uint64_t r1 = mix64(mix64(sseed) ^ (2 * cell)), r2 = mix64(mix64(sseed) ^ (2 * cell + 1));
    const double eps = sqrt(-2.0 * log(u01(r1))) * cos(6.283185307179586 * u01(r2));
    double v;
    if (variant == 1) v = eps;
    else if (variant == 2) v = sin(phi);
    else {
        double wavy = 0.0;
        for (int k = 1; k <= 6; ++k) {
            const uint64_t rk = mix64(mix64(sseed ^ 0xA5A5A5A5ull) + k);
            const double ak = 0.5 + 0.5 * u01(rk), th = 6.283185307179586 * u01(mix64(rk));
            wavy += ak * cos(k * lam + th);
        }
        const double c = cos(phi);
        v = sin(phi) + 0.25 * wavy * c * c + 0.02 * eps;
    }
    out[(size_t)blockIdx.z * ny * nx + y * nx + x] = (T)v;
}

}  // namespace

// ------------------------------------------------------------------------------------ launchers
int launch_minmax_partial(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ncell, double* part, double* zero, int64_t nzero,
                          bool finite_only)
{
    if (!q || !part || nslab < 1 || ncell < 1) return fail(ctx, XC_EBADARG, "xc_minmax: bad arguments");
    if (nslab > 65535) return fail(ctx, XC_EBADARG, "xc_minmax: at most 65535 slabs per launch");
    dim3 grid((unsigned)minmax_blocks(ncell, nslab), (unsigned)nslab);      // partials: [nslab][minmax_blocks(ncell, nslab)][2]
    const bool nt = (double)nslab * (double)ncell * (q_dtype == XC_F64 ? 8.0 : 4.0) > 128.0 * 1048576.0 || ctx->knobs.k1_nt > 0;
    if (q_dtype == XC_F64) {
        if (finite_only) hipLaunchKernelGGL((k_minmax_partial<double, false, true>), grid, dim3(256), 0, ctx->stream, (const double*)q, ncell, part, zero, nzero);
        else if (nt) hipLaunchKernelGGL((k_minmax_partial<double, true>), grid, dim3(256), 0, ctx->stream, (const double*)q, ncell, part, zero, nzero);
        else hipLaunchKernelGGL((k_minmax_partial<double, false>), grid, dim3(256), 0, ctx->stream, (const double*)q, ncell, part, zero, nzero);
    } else if (q_dtype == XC_F32) {
        if (finite_only) hipLaunchKernelGGL((k_minmax_partial<float, false, true>), grid, dim3(256), 0, ctx->stream, (const float*)q, ncell, part, zero, nzero);
        else if (nt) hipLaunchKernelGGL((k_minmax_partial<float, true>), grid, dim3(256), 0, ctx->stream, (const float*)q, ncell, part, zero, nzero);
        else hipLaunchKernelGGL((k_minmax_partial<float, false>), grid, dim3(256), 0, ctx->stream, (const float*)q, ncell, part, zero, nzero);
    }
    else return fail(ctx, XC_EBADARG, "xc_minmax: q_dtype must be XC_F32 or XC_F64");
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

int launch_minmax_final(xc_ctx* ctx, const double* part, int64_t nslab, int P, double* out)
{
    hipLaunchKernelGGL(k_minmax_final, dim3((unsigned)nslab), dim3(256), 0, ctx->stream, part, P, out);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

int launch_levels(xc_ctx* ctx, const double* minmax, int q_dtype, int64_t nslab, int N, int increase,
                  int ctr_dtype, int right_edge, double* ctr, double* edges, int32_t* status, int P, double* minmax_out)
{
    if (!minmax || !ctr || !edges || !status || N < 2 || nslab < 1)
        return fail(ctx, XC_EBADARG, "xc_levels: bad arguments (need N >= 2)");
    const double inv = 1.0 / (double)(N - 1);
    hipLaunchKernelGGL(k_levels, dim3((unsigned)nslab), dim3(256), 0, ctx->stream, minmax, P, minmax_out, N, increase,
                       q_dtype == XC_F32, ctr_dtype == XC_F32, right_edge, inv, ctr, edges, status);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

int launch_finalize(xc_ctx* ctx, int64_t nslab, const FinalArgs& a_in)
{
    FinalArgs a = a_in;
    if (a.vstride <= 0) a.vstride = a.nbin;
    hipStream_t st = ctx->stream;
    size_t lds = ((size_t)2 * a.nch * a.nbin + (a.keff ? 7 * (size_t)a.nbin : 0)) * sizeof(double);
    a.big = nullptr; a.big_stride = 0;
    if (lds > kLdsBudget) {               // work arrays in global memory instead
        a.big_stride = lds / sizeof(double);
        const int rc = ensure_big(ctx, (size_t)nslab * lds);
        if (rc != XC_OK) return rc;
        a.big = (double*)ctx->big;
        lds = 0;
    }
    { const int rc = ensure_big_lds(ctx, reinterpret_cast<const void*>(k_finalize), (int)kLdsBudget + 4096); if (rc != XC_OK) return rc; }
    a.tbl_in_lds = 0;
    if (!a.big && a.keff && lds + (size_t)2 * a.ntbl * sizeof(double) <= 64 * 1024) {
        a.tbl_in_lds = 1;
        lds += (size_t)2 * a.ntbl * sizeof(double);
    }
    const int nvh = a.nch * a.nbin;
    dim3 g1((unsigned)((nvh + a.nbin + 7) / 8), (unsigned)nslab);
    // few partials per slab (the Keff pipeline: ~20): stage 1 runs inside stage 2's workgroup, one launch less
    a.fuse_reduce = (!a.skip_reduce && a.bps <= 64 && !a.big) ? 1 : 0;
    if (!a.skip_reduce && !a.fuse_reduce) {
        hipLaunchKernelGGL(k_reduce_partials, g1, dim3(256), 0, st, a.part_h, a.part_c, a.bps, nvh, a.nbin,
                           a.red_h, a.red_c);
        XC_HIP(ctx, hipGetLastError());
    }
    // (the Keff epilogue runs its two chains per contour in the two halves of the workgroup: 512 threads for up to 256 contours)
    int nthr = (a.keff && a.o_interp && a.npre > 256) ? 1024 : (a.keff ? 512 : 256);
    if (a.fuse_reduce) {                 // one thread per partial sum to reduce: all loads of the reduction in flight at once
        const int want = ((nvh + (a.counts ? a.nbin : 0) + 63) / 64) * 64;
        nthr = want > 1024 ? 1024 : (want > nthr ? want : nthr);
    }
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)nslab), dim3(nthr), lds, st, a);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

int launch_rowsum(xc_ctx* ctx, const void* mask, int mask_dtype, const double* dA, int dA_rank,
                  int64_t ny, int64_t nx, int multiply, double* out_rows)
{
    if (!out_rows || ny < 1 || nx < 1) return fail(ctx, XC_EBADARG, "xc_rowsum: bad arguments");
    if (dA_rank != XC_DA_NONE && dA_rank != XC_DA_ROW && dA_rank != XC_DA_PLANE)
        return fail(ctx, XC_EBADARG, "xc_rowsum: dA_rank must be NONE, ROW or PLANE");
    if (dA_rank != XC_DA_NONE && !dA) return fail(ctx, XC_EBADARG, "xc_rowsum: dA is NULL");
    dim3 grid((unsigned)((ny + 3) / 4));
    if (!mask || mask_dtype == XC_F64)
        hipLaunchKernelGGL(k_rowsum<double>, grid, dim3(256), 0, ctx->stream, (const double*)mask, dA, dA_rank, ny, nx, multiply, out_rows);
    else if (mask_dtype == XC_F32)
        hipLaunchKernelGGL(k_rowsum<float>, grid, dim3(256), 0, ctx->stream, (const float*)mask, dA, dA_rank, ny, nx, multiply, out_rows);
    else return fail(ctx, XC_EBADARG, "xc_rowsum: mask_dtype must be XC_F32 or XC_F64");
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

int launch_grad2(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                 const double* rdx, const double* rdy, int periodic_x, double* out)
{
    if (!q || !rdx || !rdy || !out || nslab < 1 || ny < 1 || nx < 1)
        return fail(ctx, XC_EBADARG, "xc_grad2: bad arguments");
    if ((ny + GRAD_ROWS - 1) / GRAD_ROWS > 65535 || nslab > 65535) return fail(ctx, XC_EBADARG, "xc_grad2: ny / nslab too large");
    dim3 grid((unsigned)((nx + 255) / 256), (unsigned)((ny + GRAD_ROWS - 1) / GRAD_ROWS), (unsigned)nslab);
    if (q_dtype == XC_F64)
        hipLaunchKernelGGL(k_grad2<double>, grid, dim3(256), 0, ctx->stream, (const double*)q, ny, nx, rdx, rdy, periodic_x, out);
    else if (q_dtype == XC_F32)
        hipLaunchKernelGGL(k_grad2<float>, grid, dim3(256), 0, ctx->stream, (const float*)q, ny, nx, rdx, rdy, periodic_x, out);
    else return fail(ctx, XC_EBADARG, "xc_grad2: q_dtype must be XC_F32 or XC_F64");
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

int launch_synth(xc_ctx* ctx, void* out, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                 const double* lat_deg, const double* lon_deg, uint64_t seed, int variant)
{
    if (!out || !lat_deg || !lon_deg || nslab < 1 || ny < 1 || nx < 1 || ny > 65535 || nslab > 65535)
        return fail(ctx, XC_EBADARG, "xc_synth: bad arguments");
    dim3 grid((unsigned)((nx + 255) / 256), (unsigned)ny, (unsigned)nslab);
    if (q_dtype == XC_F64)
        hipLaunchKernelGGL(k_synth<double>, grid, dim3(256), 0, ctx->stream, (double*)out, ny, nx, lat_deg, lon_deg, seed, variant);
    else if (q_dtype == XC_F32)
        hipLaunchKernelGGL(k_synth<float>, grid, dim3(256), 0, ctx->stream, (float*)out, ny, nx, lat_deg, lon_deg, seed, variant);
    else return fail(ctx, XC_EBADARG, "xc_synth: q_dtype must be XC_F32 or XC_F64");
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

// ---- the small transfers of a host-form call, one KERNEL per direction (round 6).  Every hipMemcpyAsync between a pinned bounce buffer
// and device memory is an operation of its own on the stream (a DMA packet or a blit launch, with the dependency packets around it): at the
// reference's demo size (15 x 241 x 480) the three or four result vectors of a call cost more stream time than its kernels.  pin_in /
// pin_out are hipHostMalloc'd: the device reads and writes them through the same pointers; a kernel's stores to host memory are visible to
// the host once the stream has been waited for, and the host's memcpy into pin_in happened before the launch.
__global__ __launch_bounds__(256)
void k_copy_small(const SmallCopies c)
{
    const int e = (int)blockIdx.y;
    const unsigned n = c.bytes[e];
    const char* s = (const char*)c.src[e];
    char* d = (char*)c.dst[e];
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const uintptr_t bits = (uintptr_t)s | (uintptr_t)d | (uintptr_t)n;
    if ((bits & 15) == 0) {
        for (unsigned i = t; i < (n >> 4); i += nt) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    } else if ((bits & 3) == 0) {
        for (unsigned i = t; i < (n >> 2); i += nt) reinterpret_cast<unsigned*>(d)[i] = reinterpret_cast<const unsigned*>(s)[i];
    } else {
        for (unsigned i = t; i < n; i += nt) d[i] = s[i];
    }
}

int launch_copy_small(xc_ctx* ctx, const SmallCopies& c, int count)
{
    if (count < 1 || count > 8) return fail(ctx, XC_EBADARG, "launch_copy_small: 1..8 copies per launch");
    // one sweep: every 16-byte piece has its own thread, so a read of pinned host memory is ONE round trip over PCIe (4 blocks: 6.5 us
    // for 24 KB, two dependent trips)
    unsigned mx = 0;
    for (int i = 0; i < count; ++i) mx = c.bytes[i] > mx ? c.bytes[i] : mx;
    const unsigned bx = mx <= 4096 ? 1u : (mx + 4095) / 4096;
    hipLaunchKernelGGL(k_copy_small, dim3(bx > 64 ? 64 : bx, (unsigned)count), dim3(256), 0, ctx->stream, c);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

}  // namespace xc
