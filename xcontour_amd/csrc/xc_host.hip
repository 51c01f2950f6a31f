// Host-only helpers of the facade (no device, no context): the O(N) contour-space algebra of the reference that is too small for a
// launch and too slow as a dozen numpy calls -- at the reference's demo size (15 levels x 201 contours) `cal_gradient_wrt_area` spent
// 29 us per call on dispatching ~3 000 numbers through numpy.  The arithmetic is numpy's, operation for operation and dtype for dtype
// (this file is compiled with -ffp-contract=off like the rest); tests/test_host_logic.py compares bit for bit.
#include "xc_internal.h"
#include <cmath>

namespace {

// is the 1-D coordinate x[0..n) equally spaced in ITS OWN dtype (np.diff(x) == np.diff(x)[0] everywhere)?  *dx = that first difference
template <typename T>
bool uniform_step(const T* x, int64_t n, double* dx)
{
    const T d0 = x[1] - x[0];
    for (int64_t i = 1; i + 1 < n; ++i) if (!((T)(x[i + 1] - x[i]) == d0)) return false;
    *dx = (double)d0;
    return !(d0 != d0);
}
static bool uniform_step_any(const void* x, int dtype, int64_t n, double* dx)
{
    return dtype == XC_F32 ? uniform_step((const float*)x, n, dx) : uniform_step((const double*)x, n, dx);
}

// np.gradient(f, x, axis=-1, edge_order=1), uniform branch, one row: the result in f's dtype T (numpy computes the quotients in T when x's
// dtype is not wider than T: the caller guarantees it)
template <typename T>
inline void grad_row(const T* f, int64_t n, T dx, T two_dx, T* out)
{
    for (int64_t i = 1; i + 1 < n; ++i) out[i] = (T)(f[i + 1] - f[i - 1]) / two_dx;
    out[0] = (T)(f[1] - f[0]) / dx;
    out[n - 1] = (T)(f[n - 1] - f[n - 2]) / dx;
}

template <typename TV, typename TA, typename TO>
void grad_ratio(const TV* var, int64_t vs, const TA* area, int64_t as, int64_t nlead, int64_t n, int area_rows, double dxv, double dxa, TO* out, TV* gv, TA* ga)
{
    const TV dv = (TV)dxv, dv2 = (TV)(2.0 * dxv);        // `2. * dx`: exact in either precision
    const TA da = (TA)dxa, da2 = (TA)(2.0 * dxa);
    for (int64_t r = 0; r < nlead; ++r) {
        grad_row(var + r * vs, n, dv, dv2, gv);
        if (r == 0 || area_rows > 1) grad_row(area + (area_rows > 1 ? r * as : 0), n, da, da2, ga);
        for (int64_t i = 0; i < n; ++i) out[r * n + i] = (TO)gv[i] / (TO)ga[i];
    }
}

}  // namespace

extern "C" int xc_host_gradient_wrt_area(const void* var, int var_dtype, const void* var_coord, int var_coord_dtype,
                                         const void* area, int area_dtype, const void* area_coord, int area_coord_dtype,
                                         int64_t nlead, int64_t n, int64_t area_rows, int64_t var_row_stride, int64_t area_row_stride, void* out)
{
    if (!var || !var_coord || !area || !area_coord || !out || nlead < 1 || n < 2) return XC_EBADARG;
    if ((var_dtype != XC_F32 && var_dtype != XC_F64) || (area_dtype != XC_F32 && area_dtype != XC_F64)) return XC_EBADARG;
    if ((var_coord_dtype != XC_F32 && var_coord_dtype != XC_F64) || (area_coord_dtype != XC_F32 && area_coord_dtype != XC_F64)) return XC_EBADARG;
    if (area_rows != 1 && area_rows != nlead) return XC_EBADARG;
    // a float64 coordinate under a float32 array: numpy's result depends on its promotion rules (NEP 50 or not) -- left to numpy
    if ((var_dtype == XC_F32 && var_coord_dtype == XC_F64) || (area_dtype == XC_F32 && area_coord_dtype == XC_F64)) return 1;
    double dxv, dxa;
    if (!uniform_step_any(var_coord, var_coord_dtype, n, &dxv) || !uniform_step_any(area_coord, area_coord_dtype, n, &dxa)) return 1;
    double stack[2 * 1024];
    double* tmp = stack;
    if (n > 1024) { tmp = (double*)malloc(sizeof(double) * 2 * (size_t)n); if (!tmp) return XC_ENOMEM; }
    void* gv = tmp; void* ga = tmp + n;
    if (var_dtype == XC_F32 && area_dtype == XC_F32)
        grad_ratio((const float*)var, var_row_stride, (const float*)area, area_row_stride, nlead, n, (int)area_rows, dxv, dxa, (float*)out, (float*)gv, (float*)ga);
    else if (var_dtype == XC_F32)
        grad_ratio((const float*)var, var_row_stride, (const double*)area, area_row_stride, nlead, n, (int)area_rows, dxv, dxa, (double*)out, (float*)gv, (double*)ga);
    else if (area_dtype == XC_F32)
        grad_ratio((const double*)var, var_row_stride, (const float*)area, area_row_stride, nlead, n, (int)area_rows, dxv, dxa, (double*)out, (double*)gv, (float*)ga);
    else
        grad_ratio((const double*)var, var_row_stride, (const double*)area, area_row_stride, nlead, n, (int)area_rows, dxv, dxa, (double*)out, (double*)gv, (double*)ga);
    if (tmp != stack) free(tmp);
    return XC_OK;
}

// ---- histogram edges from contour levels: _histogram's dummy first edge (core.py:1296-1305) and xhistogram's `+ 1e-8` on the last one, in the
// levels' OWN dtype, for every slab of a stack; the checks of the reference on the way ('non monotonic bins', core.py:1233-1251; one
// direction for every slab).  What the facade did with sixteen numpy calls per binning call.
namespace {
template <typename T>
int edges_from_levels(const T* b, int64_t nslab, int64_t N, int right_edge, double* edges, int* increasing)
{
    for (int64_t s = 0; s < nslab; ++s)
        for (int64_t k = 1; k < N; ++k)
            if (b[s * N + k] == b[s * N + k - 1]) return xc::fail(nullptr, XC_EEDGES, "non monotonic bins");
    if (N < 2) return xc::fail(nullptr, XC_EBADARG, "need at least two contour levels");
    const bool binc = b[0] < b[N - 1];
    for (int64_t s = 1; s < nslab; ++s) {
        if ((b[s * N] < b[s * N + N - 1]) == binc) continue;
        bool has_nan = false;                               // an all-NaN slab (no valid cell) has no direction: let through
        for (int64_t k = 0; k < N && !has_nan; ++k) has_nan = b[s * N + k] != b[s * N + k];
        if (!has_nan) return xc::fail(nullptr, XC_EBADARG, "not every time or level is increasing/decreasing");
    }
    const T n1 = (T)(N - 1);
    for (int64_t s = 0; s < nslab; ++s) {
        const T* r = b + s * N;
        double* e = edges + s * (N + 1);
        const T first = r[0], last = r[N - 1];
        T e0, eN;
        if (binc) { for (int64_t k = 0; k < N; ++k) e[k + 1] = (double)r[k]; e0 = first - (T)((T)(last - first) / n1); eN = last; }
        else { for (int64_t k = 0; k < N; ++k) e[k + 1] = (double)r[N - 1 - k]; e0 = last - (T)((T)(first - last) / n1); eN = first; }
        e[0] = (double)e0;
        if (right_edge == XC_EDGE_XHISTOGRAM) e[N] = (double)(T)(eN + (T)1e-8);
    }
    *increasing = binc ? 1 : 0;
    return XC_OK;
}
}  // namespace

extern "C" int xc_host_edges_from_levels(const void* levels, int levels_dtype, int64_t nslab, int64_t N, int right_edge,
                                         double* out_edges, int* out_increasing)
{
    if (!levels || !out_edges || !out_increasing || nslab < 1 || N < 1) return xc::fail(nullptr, XC_EBADARG, "xc_host_edges_from_levels: bad arguments");
    if (right_edge != XC_EDGE_NUMPY && right_edge != XC_EDGE_XHISTOGRAM) return xc::fail(nullptr, XC_EBADARG, "right_edge should be \"numpy\" or \"xhistogram\"");
    if (levels_dtype == XC_F32) return edges_from_levels((const float*)levels, nslab, N, right_edge, out_edges, out_increasing);
    if (levels_dtype == XC_F64) return edges_from_levels((const double*)levels, nslab, N, right_edge, out_edges, out_increasing);
    return xc::fail(nullptr, XC_EBADARG, "xc_host_edges_from_levels: levels_dtype must be XC_F32 or XC_F64");
}
