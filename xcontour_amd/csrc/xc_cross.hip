// K9 -- box-counting contour crossing (gfx950).
//
// Replaces Contour2D.cal_contour_crossing (reference core.py:640-693) and its per-(slab, contour)
// numba kernel _contour_crossing (core.py:1490-1566).  The reference runs one full pass over the
// padded slab PER CONTOUR; here one pass serves all N contours of a slab:
//   box (j, i) covers fine cells [j*s, j*s+s) x [i*s, i*s+s), i.e. corner rows j*s..j*s+s and
//   corner columns i*s..i*s+s of the X-padded slab; with mn / mx the NaN-skipping min / max of
//   those corners, contour c crosses the box  <=>  (some corner <= c) and (some corner > c)
//   <=>  mn <= c < mx  (core.py:1531-1557).  For ascending contours that is the index range
//   [lower_bound(mn), lower_bound(mx)), each member of which receives
//   sqrt(areaPad[j, i]) * stride (core.py:1560: area indexed with the COARSE indices) -- NaN
//   skipped like np.nansum (1564) -- and one count.
// The reference's column loop runs over range(Jn-1) (core.py:1521); `nbi` carries that choice
// (the launcher computes it), the kernel is agnostic.
//
// Mapping: a block owns a static, strided set of (32 box rows x 256 box columns) tiles of one
// slab; lanes run along X; a thread walks down its box column carrying the min / max of the row
// shared by consecutive boxes, so every corner row is loaded once per tile.  Sums live in LDS
// (ds_add_f64 / ds_add_u32), one partial vector per block, reduced in fixed order afterwards.
// HBM-bound: tracer (4|8 B) + area (4|8 B) per fine cell at stride 1, independent of N.
#include "xc_internal.h"

namespace xc {
namespace {

constexpr int CROSS_RB = 32;     // box rows per tile
constexpr int CROSS_TPB = 256;   // threads = box columns per tile

// source column of padded column c >= nx (np.pad semantics on the last axis); -1 = NaN fill
__device__ __forceinline__ int64_t pad_source(int64_t c, int64_t nx, int mode)
{
    switch (mode) {
    case XC_PAD_EDGE: return nx - 1;
    case XC_PAD_WRAP: return c % nx;
    case XC_PAD_REFLECT: { if (nx == 1) return 0; const int64_t p = 2 * (nx - 1), m = c % p; return m < nx ? m : p - m; }
    case XC_PAD_SYMMETRIC: { const int64_t p = 2 * nx, m = c % p; return m < nx ? m : p - 1 - m; }
    default: return -1;
    }
}

template <typename T>
__device__ __forceinline__ double load_padded(const T* __restrict__ row, int64_t c, int64_t nx, int mode)
{
    if (c >= nx) { c = pad_source(c, nx, mode); if (c < 0) return __longlong_as_double(0x7ff8000000000000LL); }
    return (double)row[c];
}

// NaN-skipping min / max of corner columns c0..c0+s of one row, folded into (mn, mx)
template <typename T>
__device__ __forceinline__ void row_segment(const T* __restrict__ row, int64_t c0, int s, int64_t nx, int mode,
                                            double& mn, double& mx)
{
    for (int d = 0; d <= s; ++d) {
        const double v = load_padded(row, c0 + d, nx, mode);
        mn = fmin(mn, v); mx = fmax(mx, v);          // fmin / fmax return the non-NaN operand
    }
}

__device__ __forceinline__ int lower_bound(const double* __restrict__ c, int n, double v)
{
    int lo = 0, hi = n;                               // first k with c[k] >= v
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (c[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

template <typename TQ>
__global__ __launch_bounds__(CROSS_TPB)
void k_crossing(const TQ* __restrict__ q, int64_t ny, int64_t nx, int pad_mode,
                const double* __restrict__ contours, int N, int contours_per_slab,
                const void* __restrict__ area, int area_f32, int area_per_slab,
                int s, int64_t nbj, int64_t nbi, int64_t ntj, int64_t nti, int bps,
                double* __restrict__ part_len, unsigned* __restrict__ part_cnt)
{
    extern __shared__ double sm[];
    double* s_c = sm;                      // [N] ascending contours of this slab
    double* s_len = sm + N;                // [N]
    unsigned* s_cnt = (unsigned*)(sm + 2 * (size_t)N);   // [N]
    const int tid = threadIdx.x;
    const int64_t slab = blockIdx.y;
    const double* cs = contours + (contours_per_slab ? (size_t)slab * N : 0);
    for (int k = tid; k < N; k += CROSS_TPB) { s_c[k] = cs[k]; s_len[k] = 0.0; s_cnt[k] = 0u; }
    __syncthreads();

    const TQ* qs = q + (size_t)slab * ny * nx;
    const size_t aoff = area_per_slab ? (size_t)slab * ny * nx : 0;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    const double fs = (double)s;

    for (int64_t tile = blockIdx.x; tile < ntj * nti; tile += bps) {
        const int64_t tj = tile / nti, ti = tile - tj * nti;
        const int64_t i = ti * CROSS_TPB + tid;
        if (i >= nbi) continue;                       // no block-wide barrier inside the loop
        const int64_t j0 = tj * CROSS_RB, j1 = (j0 + CROSS_RB < nbj) ? j0 + CROSS_RB : nbj;
        const int64_t c0 = i * s;
        double cmn = inf, cmx = -inf;                 // the corner row shared with the previous box
        row_segment(qs + (size_t)(j0 * s) * nx, c0, s, nx, pad_mode, cmn, cmx);
        for (int64_t j = j0; j < j1; ++j) {
            double mn = cmn, mx = cmx;
            for (int r = 1; r < s; ++r) row_segment(qs + (size_t)(j * s + r) * nx, c0, s, nx, pad_mode, mn, mx);
            cmn = inf; cmx = -inf;
            row_segment(qs + (size_t)(j * s + s) * nx, c0, s, nx, pad_mode, cmn, cmx);
            mn = fmin(mn, cmn); mx = fmax(mx, cmx);
            // area of the box: the padded area array at the COARSE indices (core.py:1560)
            double a;
            {
                int64_t ac = i;
                bool nanfill = false;
                if (ac >= nx) { ac = pad_source(ac, nx, pad_mode); nanfill = ac < 0; if (nanfill) ac = 0; }
                const size_t idx = aoff + (size_t)j * nx + ac;
                // f32 area: np.sqrt rounds in f32; f64 sqrt then one rounding to f32 is the correctly rounded f32 root (53 >= 2*24+2)
                a = area_f32 ? (double)(float)__dsqrt_rn((double)((const float*)area)[idx]) : __dsqrt_rn(((const double*)area)[idx]);
                if (nanfill) a = __longlong_as_double(0x7ff8000000000000LL);
            }
            if (mn < mx) {
                const int klo = lower_bound(s_c, N, mn);
                if (klo < N && s_c[klo] < mx) {
                    const int khi = lower_bound(s_c, N, mx);
                    const double w = __dmul_rn(a, fs);
                    for (int k = klo; k < khi; ++k) {
                        if (w == w) atomicAdd(&s_len[k], w);
                        atomicAdd(&s_cnt[k], 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    const size_t pb = ((size_t)slab * bps + blockIdx.x) * N;
    for (int k = tid; k < N; k += CROSS_TPB) { part_len[pb + k] = s_len[k]; part_cnt[pb + k] = s_cnt[k]; }
}

__global__ __launch_bounds__(256)
void k_crossing_reduce(const double* __restrict__ part_len, const unsigned* __restrict__ part_cnt, int bps, int N,
                       double* __restrict__ out_len, unsigned long long* __restrict__ out_cnt)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const size_t slab = blockIdx.y;
    double len = 0.0; unsigned long long cnt = 0;
    for (int b = 0; b < bps; ++b) {
        len += part_len[(slab * bps + b) * N + k];
        cnt += part_cnt[(slab * bps + b) * N + k];
    }
    if (out_len) out_len[slab * N + k] = len;
    if (out_cnt) out_cnt[slab * N + k] = cnt;
}

}  // namespace

// Coarse shape (core.py:1510-1511): np.round == round-half-even of the true quotient.
static int64_t coarse(int64_t n, int s) { return (int64_t)nearbyint((double)n / (double)s); }

int launch_crossing(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                    int pad_x, int pad_mode, const double* contours, int N, int contours_per_slab,
                    const void* area, int area_dtype, int area_per_slab, int stride, int full_width,
                    double* out_len, uint64_t* out_cnt)
{
    if (!q || !contours || !area || (!out_len && !out_cnt) || nslab < 1 || ny < 1 || nx < 1 || N < 1)
        return fail(ctx, XC_EBADARG, "xc_crossing: bad arguments");
    if (q_dtype != XC_F32 && q_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_crossing: q_dtype must be XC_F32 or XC_F64");
    if (area_dtype != XC_F32 && area_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_crossing: area_dtype must be XC_F32 or XC_F64");
    if (stride < 1 || pad_x < 0) return fail(ctx, XC_EBADARG, "xc_crossing: stride must be >= 1 and pad_x >= 0");
    if (pad_mode < XC_PAD_EDGE || pad_mode > XC_PAD_SYMMETRIC) return fail(ctx, XC_EBADARG, "xc_crossing: unknown pad_mode");
    if (nslab > 65535) return fail(ctx, XC_EBADARG, "xc_crossing: nslab too large");
    const size_t lds = (size_t)N * 20 + 8;
    if (lds > kLdsBudget) return fail(ctx, XC_EBADARG, "xc_crossing: too many contours for one pass");
    const int64_t Jn = coarse(ny, stride), In = coarse(nx + pad_x, stride);
    const int64_t nbj = Jn - 1, nbi = full_width ? In - 1 : (Jn < In ? Jn : In) - 1;
    if (nbj < 1 || nbi < 1) {
        if (out_len) XC_HIP(ctx, hipMemsetAsync(out_len, 0, (size_t)nslab * N * 8, ctx->stream));
        if (out_cnt) XC_HIP(ctx, hipMemsetAsync(out_cnt, 0, (size_t)nslab * N * 8, ctx->stream));
        return XC_OK;
    }
    // the last corner row / column every box touches must exist in the padded slab
    if (nbj * stride > ny - 1 || nbi * stride > nx + pad_x - 1) return fail(ctx, XC_EBADARG, "xc_crossing: boxes leave the padded slab");
    const int64_t ntj = (nbj + CROSS_RB - 1) / CROSS_RB, nti = (nbi + CROSS_TPB - 1) / CROSS_TPB;
    int64_t bps = 2048 / nslab; if (bps < 8) bps = 8; if (bps > ntj * nti) bps = ntj * nti;
    const size_t pl = (size_t)nslab * bps * N * 8, pc = (size_t)nslab * bps * N * 4;
    {
        const int rc = ensure_scratch(ctx, ((pl + 255) & ~(size_t)255) + pc);
        if (rc != XC_OK) return rc;
    }
    double* part_len = (double*)ctx->scratch;
    unsigned* part_cnt = (unsigned*)((char*)ctx->scratch + ((pl + 255) & ~(size_t)255));
    const dim3 grid((unsigned)bps, (unsigned)nslab);
#define XC_CROSS(T) do { \
        if (lds > 64 * 1024) XC_HIP(ctx, hipFuncSetAttribute((const void*)k_crossing<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_crossing<T>), grid, dim3(CROSS_TPB), lds, ctx->stream, (const T*)q, ny, nx, pad_mode, contours, N, \
                           contours_per_slab, area, area_dtype == XC_F32, area_per_slab, stride, nbj, nbi, ntj, nti, (int)bps, \
                           part_len, part_cnt); } while (0)
    if (q_dtype == XC_F64) XC_CROSS(double); else XC_CROSS(float);
#undef XC_CROSS
    XC_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_crossing_reduce, dim3((unsigned)((N + 255) / 256), (unsigned)nslab), dim3(256), 0, ctx->stream,
                       part_len, part_cnt, (int)bps, N, out_len, (unsigned long long*)out_cnt);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

}  // namespace xc
