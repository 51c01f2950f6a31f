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
#include <stdlib.h>

namespace xc {
namespace {

constexpr int CROSS_RB = 32;     // box rows per tile
constexpr int CROSS_TPB = 256;   // threads = box columns per tile
#ifndef XC_CROSS_GROUP
#define XC_CROSS_GROUP 2
#endif
constexpr int CROSS_W1 = 252;    // box columns per tile at stride 1 (4 waves x 63 boxes, see k_crossing)

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

// lane i <- lane i+1 (DPP wave_shl:1); lane 63 keeps its own value
__device__ __forceinline__ double from_right_lane(double v)
{
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_update_dpp((int)(u & 0xffffffffu), (int)(u & 0xffffffffu), 0x130, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp((int)(u >> 32), (int)(u >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// number of contours < v, i.e. the klo with cx[klo] < v <= cx[klo+1]; cx = [-inf, c_0 .. c_{N-1}, +inf]
__device__ __forceinline__ int count_below(const double* __restrict__ cx, int N, double v)
{
    int lo = 0, hi = N;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cx[mid + 1] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// Equally spaced levels: number of contours < v from arithmetic.  The block measured how far the levels sit from their ideal
// positions (zlo = twice that, in units of the spacing, plus the rounding of t): when the fractional position of v lies
// outside that zone of either neighbouring level the arithmetic answer IS the count and no LDS read is needed; otherwise
// (v within a hair of a level, NaN, infinities) one adjacent-pair read verifies, bisection is the fallback.
__device__ __forceinline__ int count_below_uniform(const double* __restrict__ cx, int N, double v, double c_first,
                                                   double inv_step, double zlo)
{
    const double t = (v - c_first) * inv_step;
    int k = (int)fmin(fmax(t + 1.0, 0.0), (double)N);          // floor(t) + 1 clamped to [0, N]; NaN -> 0
    const double fr = __builtin_amdgcn_fract(t);
    if (!((fr > zlo) & (fr < 1.0 - zlo))) {
        const double c_lo = cx[k], c_hi = cx[k + 1];
        if (!((c_lo < v) & (v <= c_hi))) k = count_below(cx, N, v);
    }
    return k;
}

template <typename TA> __device__ __forceinline__ double sqrt_like_numpy(TA a);
// f32 area: np.sqrt rounds in f32; an f64 root rounded once more to f32 IS the correctly rounded f32 root (53 >= 2*24+2)
template <> __device__ __forceinline__ double sqrt_like_numpy<float>(float a) { return (double)(float)__dsqrt_rn((double)a); }
template <> __device__ __forceinline__ double sqrt_like_numpy<double>(double a) { return __dsqrt_rn(a); }

// One finished box: contours k with cx[k+1] in [mn, mx) are crossed.  `g` is the previous box's klo
// (walking down a column the tracer changes slowly: one adjacent-pair LDS read confirms the guess).
template <typename TA, bool CNT>
__device__ __forceinline__ void box_done(const double* __restrict__ cx, int N, double mn, double mx, TA araw, bool nanfill,
                                         double fs, double c_first, double inv_step, double zlo, int& g,
                                         double* __restrict__ my_len, unsigned* __restrict__ my_cnt, double* __restrict__ s_dir, int cshift)
{
    if (!(mn < mx)) return;                            // one value, or no valid corner at all
    if (inv_step > 0.0) {
        // Equally spaced levels (what cal_contours produces): BOTH ends of the crossed range follow from arithmetic -- klo = number
        // of levels < mn, khi = number of levels < mx -- each verified by one adjacent-pair LDS read only when the value
        // lies within the measured irregularity of the levels (count_below_uniform), bisection as the fallback.
        // The box then adds +w at klo and -w at khi: a DIFFERENCE array, two adds whatever the number of crossed levels (the
        // per-lane level loop of a noisy field -- 3.6 levels per box on the PV-like slabs, up to ~10 in a wave, 39 on white
        // noise -- was half of the kernel's instructions); the block turns the differences into sums with one pass over the
        // levels before it writes its partials (k_crossing epilogue).  Counts are exact (integer differences modulo 2^32);
        // lengths agree with the oracle to summation order, level by level.
        const int klo = count_below_uniform(cx, N, mn, c_first, inv_step, zlo);
        const int khi = count_below_uniform(cx, N, mx, c_first, inv_step, zlo);
        if (khi <= klo) return;                            // no level in [mn, mx): the common case on a smooth field ends here
        double w = __dmul_rn(sqrt_like_numpy<TA>(araw), fs);    // core.py:1560 (product in f64: numba types f32 * int64 as f64)
        if (nanfill) w = __longlong_as_double(0x7ff8000000000000LL);
#ifndef XC_CROSS_NOATOM
        if (w == w) {                                      // np.nansum skips NaN (negative or NaN area)
            if (w < __longlong_as_double(0x7ff0000000000000LL)) { atomicAdd(&my_len[(klo) << cshift], w); atomicAdd(&my_len[(khi) << cshift], -w); }
            else for (int t = klo; t < khi; ++t) atomicAdd(&s_dir[t], w);        // an infinite area: +inf / -inf differences would turn into NaN
        }
        if (CNT) { atomicAdd(&my_cnt[(klo) << cshift], 1u); atomicAdd(&my_cnt[(khi) << cshift], 0xffffffffu); }
#endif
        return;
    }
    // any ascending levels: start from the previous box's answer (walking down a column the tracer changes slowly), one
    // adjacent-pair LDS read verifies, bisection is the fallback; then scan the crossed levels
    const double c_lo = cx[g], c_hi = cx[g + 1];
    if (!((c_lo < mn) & (mn <= c_hi))) g = count_below(cx, N, mn);
    int k = g;
    double ck = cx[k + 1];
    if (!(ck < mx)) return;
    double w = __dmul_rn(sqrt_like_numpy<TA>(araw), fs);
    if (nanfill) w = __longlong_as_double(0x7ff8000000000000LL);
    const bool add = (w == w);
    do {                                               // the +inf sentinel ends the scan
#ifndef XC_CROSS_NOATOM
        if (add) atomicAdd(&my_len[(k) << cshift], w);
        if (CNT) atomicAdd(&my_cnt[(k) << cshift], 1u);
#else
        if (add && w == 1.2345) my_len[(k) << cshift] = w;         // diagnostic build: the loop without its atomics
#endif
        ++k; ck = cx[k + 1];
    } while (ck < mx);
}

// G finished boxes of a lane at once, equally spaced levels only (inv_step > 0): the same decisions as box_done, arranged in
// phases -- the LDS reads of a phase are in flight together, the G square roots are independent chains (measured on noisy
// cfg2 slabs: G = 2 is 1-2 us per slab faster than one box at a time, 4 the same, 8 slower: registers).
template <typename TA, bool CNT, int G>
__device__ __forceinline__ void boxes_group(const double* __restrict__ cx, int N, const double (&mn)[G], const double (&mx)[G],
                                            const TA (&araw)[G], const bool (&valid)[G], bool nanfill, double fs, double c_first,
                                            double inv_step, double zlo, double* __restrict__ my_len, unsigned* __restrict__ my_cnt,
                                            double* __restrict__ s_dir, int cshift)
{
    int klo[G], khi[G];
    bool cross[G];
    bool any = false;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        klo[i] = count_below_uniform(cx, N, mn[i], c_first, inv_step, zlo);
        khi[i] = count_below_uniform(cx, N, mx[i], c_first, inv_step, zlo);
        cross[i] = valid[i] && (mn[i] < mx[i]) && (khi[i] > klo[i]);                    // one value, no valid corner, no level in [mn, mx): nothing
        any = any || cross[i];
    }
    if (!any) return;                                                                   // the common case on a smooth field
    double w[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        w[i] = __dmul_rn(sqrt_like_numpy<TA>(araw[i]), fs);                             // core.py:1560
        if (nanfill) w[i] = __longlong_as_double(0x7ff8000000000000LL);
    }
#ifndef XC_CROSS_NOATOM
#pragma unroll
    for (int i = 0; i < G; ++i) {
        if (cross[i]) {
            if (w[i] == w[i]) {                            // np.nansum skips NaN (negative or NaN area)
                if (w[i] < __longlong_as_double(0x7ff0000000000000LL)) { atomicAdd(&my_len[klo[i] << cshift], w[i]); atomicAdd(&my_len[khi[i] << cshift], -w[i]); }
                else for (int t = klo[i]; t < khi[i]; ++t) atomicAdd(&s_dir[t], w[i]);
            }
            if (CNT) { atomicAdd(&my_cnt[klo[i] << cshift], 1u); atomicAdd(&my_cnt[khi[i] << cshift], 0xffffffffu); }
        }
    }
#endif
}

// CNT: also count the crossed boxes (exact integer output).  S: compile-time stride (loads of a batch
// of boxes are issued together), 0 = any stride at run time.  ncopy lane-privatised copies of the
// per-contour sums: neighbouring lanes (neighbouring columns) cross the SAME contour, and same-address
// LDS atomics serialise.
template <typename TQ, typename TA, bool CNT, int S>
__global__ __launch_bounds__(CROSS_TPB)
void k_crossing(const TQ* __restrict__ q, int64_t ny, int64_t nx, int pad_mode,
                const double* __restrict__ contours, int N, int contours_per_slab,
                const TA* __restrict__ area, int area_per_slab,
                int s_rt, int64_t nbj, int64_t nbi, int64_t ntj, int64_t nti, int bps, int ncopy, int np, int rbox,
                double* __restrict__ part_len, unsigned* __restrict__ part_cnt)
{
    extern __shared__ double sm[];
    double* s_cx = sm;                                   // [N + 2]  -inf, ascending contours of this slab, +inf
    double* s_len = sm + (N + 2);                        // [ncopy][np]
    unsigned* s_cnt = (unsigned*)(s_len + (size_t)ncopy * np);   // [ncopy][np]
    double* s_dir = (double*)(s_cnt + (size_t)ncopy * np + (((size_t)ncopy * np) & 1));   // [N] direct sums of boxes with an infinite weight
    const int tid = threadIdx.x;
    const int64_t slab = blockIdx.y;
    const double* cs = contours + (contours_per_slab ? (size_t)slab * N : 0);
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    for (int k = tid; k < N; k += CROSS_TPB) s_cx[k + 1] = cs[k];
    if (tid == 0) { s_cx[0] = -inf; s_cx[N + 1] = inf; }
    for (int k = tid; k < ncopy * np; k += CROSS_TPB) { s_len[k] = 0.0; if (CNT) s_cnt[k] = 0u; }
    for (int k = tid; k < N; k += CROSS_TPB) s_dir[k] = 0.0;
    __syncthreads();
    // cell k of copy c sits at [k * ncopy + c] (level-major, round 4): the lanes of a wave hold `ncopy` different copies and
    // levels a few positions apart, i.e. addresses spread over ncopy * (a few) consecutive words -- distinct banks.  Rounds 1-3
    // kept the copies apart by an odd pitch (copies 0 / 3 / 6, 1 / 4 / 7 and 2 / 5 came out two banks apart) and the round-3
    // review blamed the 40 % conflict cycles of a noisy field on that.  Measured: 22.76 -> 22.72 us per noisy cfg2 slab, i.e.
    // NOTHING -- the conflict cycles are the eight lanes that share a copy adding to the SAME cell (neighbouring boxes cross the
    // same levels), which no layout removes; 4 / 16 / 32 copies: 25.8 / 25.5 / 25.1 us (fewer copies collide more, more copies
    // cost occupancy).  The layout stays (it is never worse and needs no pitch rule).
    const int cshift = __builtin_ctz((unsigned)ncopy);
    double* my_len = s_len + (tid & (ncopy - 1));
    unsigned* my_cnt = s_cnt + (tid & (ncopy - 1));

    const int s = S > 0 ? S : s_rt;
    const TQ* qs = q + (size_t)slab * ny * nx;
    const TA* as = area + (area_per_slab ? (size_t)slab * ny * nx : 0);
    const double fs = (double)s;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    const double c_first = s_cx[1];
    double inv_step = (N > 1) ? (double)(N - 1) / (s_cx[N] - c_first) : 0.0;
    if (!(inv_step > 0.0 && inv_step < inf)) inv_step = 0.0;
    double zlo = 0.5;
    {   // equally spaced?  (block-uniform answer) -- and how exactly: the largest distance of a level from its ideal position
        int ok = inv_step > 0.0;
        double dev = 0.0;
        for (int k = tid; k < N && ok; k += CROSS_TPB) {
            const double d = fabs((s_cx[k + 1] - c_first) * inv_step - (double)k);
            ok = d < 0.01; dev = fmax(dev, d);
        }
        if (!__syncthreads_and(ok)) inv_step = 0.0;
        for (int o = 32; o > 0; o >>= 1) dev = fmax(dev, __shfl_xor(dev, o));
        __shared__ double s_dev[CROSS_TPB / 64];
        if ((tid & 63) == 0) s_dev[tid >> 6] = dev;
        __syncthreads();
        dev = s_dev[0];
        for (int w = 1; w < CROSS_TPB / 64; ++w) dev = fmax(dev, s_dev[w]);
        zlo = 2.0 * dev + 1e-9;                                  // + the rounding of t itself (|t| <= ~N: 1e-13 at most)
    }
    int g = 0;

    for (int64_t tile = blockIdx.x; tile < ntj * nti; tile += bps) {
        const int64_t tj = tile / nti, ti = tile - tj * nti;
        if constexpr (S == -1) {
            // strides 2..63: lanes run along the FINE columns (coalesced, every cell loaded once per tile apart from
            // the span overlap); a lane first folds the s+1 corner rows of a box row vertically, then a sliding-window
            // min / max over s+1 neighbouring lanes (log2 shuffle steps) gives the box extrema at the lanes l % s == 0.
            // A wave spans nbw = 63 / s boxes (their last corner column is lane nbw*s <= 63).
            const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int nbw = 63 / s;
            const int64_t i0 = (ti * 4 + wave) * nbw;                  // first box of this wave's span
            if (i0 >= nbi) continue;                                   // wave-uniform; no block barrier in the loop
            const int64_t j0 = tj * rbox, j1 = (j0 + rbox < nbj) ? j0 + rbox : nbj;
            int64_t c = i0 * s + lane;                                 // padded fine column of this lane
            if (c > nbi * s) c = nbi * s;                              // (lanes beyond the last corner column feed no valid box)
            bool pn = false;
            if (c >= nx) { c = pad_source(c, nx, pad_mode); pn = c < 0; if (pn) c = 0; }
            const int bl = lane / s;
            const int64_t i = i0 + bl;
            const bool box = (lane - bl * s == 0) && bl < nbw && i < nbi;
            int64_t ac = i < nbi ? i : nbi - 1;
            bool nanfill = false;
            if (ac >= nx) { ac = pad_source(ac, nx, pad_mode); nanfill = ac < 0; if (nanfill) ac = 0; }
            const int w = s + 1;
            double cmn, cmx;
            { const double x = pn ? qnan : (double)qs[(size_t)(j0 * s) * nx + c]; cmn = fmin(inf, x); cmx = fmax(-inf, x); }
            // the fine rows below the first corner row as one stream of 8-row batches, double-buffered: the loads of the next batch
            // are in flight while this one is folded (one batch at a time left the wave idle for a memory round trip per box row:
            // cfg2 slabs at stride 8 / 16 / 32: 23.2 / 27 / 31.3 -> 13.4 / 14.8 / 15.7 us)
            const int64_t frow0 = j0 * s, nfr = (j1 - j0) * s;
            const TQ* col = qs + (size_t)frow0 * nx + c;
            auto load8 = [&](TQ (&v)[8], int64_t r) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int64_t rr = (r + k <= nfr) ? r + k : nfr; v[k] = col[(size_t)rr * nx]; }
            };
            double vmn = cmn, vmx = cmx;
            int inbox = 0;
            int64_t j = j0;
            auto fold8 = [&](const TQ (&v)[8], int64_t r) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (r + k > nfr) break;                                 // wave-uniform
                    const double x = pn ? qnan : (double)v[k];
                    vmn = fmin(vmn, x); vmx = fmax(vmx, x);
                    if (++inbox == s) {                                     // wave-uniform: the last corner row of this box row
                        cmn = fmin(inf, x); cmx = fmax(-inf, x);            // ... is the first of the next
                        int P = 1;                                          // sliding window of w lanes: doubling, then one overlap step
                        for (; 2 * P <= w; P *= 2) { vmn = fmin(vmn, __shfl_down(vmn, P)); vmx = fmax(vmx, __shfl_down(vmx, P)); }
                        if (w != P) { vmn = fmin(vmn, __shfl_down(vmn, w - P)); vmx = fmax(vmx, __shfl_down(vmx, w - P)); }
                        if (box) box_done<TA, CNT>(s_cx, N, vmn, vmx, as[(size_t)j * nx + ac], nanfill, fs, c_first, inv_step, zlo, g, my_len, my_cnt, s_dir, cshift);
                        ++j; inbox = 0; vmn = cmn; vmx = cmx;
                    }
                }
            };
            TQ A[8], B[8];
            load8(A, 1);
            for (int64_t r = 1; r <= nfr; r += 16) {
                if (r + 8 <= nfr) load8(B, r + 8);
                fold8(A, r);
                if (r + 8 > nfr) break;
                if (r + 16 <= nfr) load8(A, r + 16);
                fold8(B, r + 8);
            }
            continue;
        }
        if constexpr (S == 1) {
            // stride 1: a wave covers 63 boxes with 64 corner columns -- every lane loads ONE value per row and gets
            // its right neighbour from the next lane (DPP), lane 63 only supplies the last corner column
            constexpr int B = 8;
            const int lane = tid & 63, wave = tid >> 6;
            const int64_t i = ti * CROSS_W1 + wave * 63 + lane;
            const int64_t j0 = tj * CROSS_RB, j1 = (j0 + CROSS_RB < nbj) ? j0 + CROSS_RB : nbj;
            const bool box = lane < 63 && i < nbi;                    // lanes without a box still load and shift
            int64_t c = i < nbi ? i : nbi;                            // corner column (padded index), <= nbi exists
            bool pn = false;
            if (c >= nx) { c = pad_source(c, nx, pad_mode); pn = c < 0; if (pn) c = 0; }
            int64_t ac = i < nbi ? i : nbi - 1;
            bool nanfill = false;
            if (ac >= nx) { ac = pad_source(ac, nx, pad_mode); nanfill = ac < 0; if (nanfill) ac = 0; }
            double cmn, cmx;
            {
                const double x = pn ? qnan : (double)qs[(size_t)j0 * nx + c], xr = from_right_lane(x);
                cmn = fmin(fmin(inf, x), xr); cmx = fmax(fmax(-inf, x), xr);
            }
            for (int64_t jb = j0; jb < j1; jb += B) {
                TQ v[B]; TA av[B];
#pragma unroll
                for (int b = 0; b < B; ++b) {                         // all loads of the batch in flight together
                    const int64_t jj = (jb + b < j1) ? jb + b : j1 - 1;
                    av[b] = as[(size_t)jj * nx + ac];
                    v[b] = qs[(size_t)(jj + 1) * nx + c];
                }
                if (inv_step > 0.0) {                                  // block-uniform: equally spaced levels, boxes in groups
                    constexpr int G = XC_CROSS_GROUP;
#pragma unroll
                    for (int b0 = 0; b0 < B; b0 += G) {
                        double mnb[G], mxb[G]; TA ab[G]; bool vb[G];
#pragma unroll
                        for (int i = 0; i < G; ++i) {
                            const int b = b0 + i;
                            const double x = pn ? qnan : (double)v[b], xr = from_right_lane(x);
                            const double rmn = fmin(fmin(inf, x), xr), rmx = fmax(fmax(-inf, x), xr);
                            mnb[i] = fmin(cmn, rmn); mxb[i] = fmax(cmx, rmx);
                            cmn = rmn; cmx = rmx;
                            ab[i] = av[b]; vb[i] = box && (jb + b < j1);
                        }
                        boxes_group<TA, CNT, G>(s_cx, N, mnb, mxb, ab, vb, nanfill, fs, c_first, inv_step, zlo, my_len, my_cnt, s_dir, cshift);
                    }
                    continue;
                }
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    if (jb + b >= j1) break;                          // wave-uniform
                    const double x = pn ? qnan : (double)v[b], xr = from_right_lane(x);
                    const double rmn = fmin(fmin(inf, x), xr), rmx = fmax(fmax(-inf, x), xr);
                    const double mn = fmin(cmn, rmn), mx = fmax(cmx, rmx);
                    cmn = rmn; cmx = rmx;
                    if (box) box_done<TA, CNT>(s_cx, N, mn, mx, av[b], nanfill, fs, c_first, inv_step, zlo, g, my_len, my_cnt, s_dir, cshift);
                }
            }
            continue;
        }
        const int64_t i = ti * CROSS_TPB + tid;
        if (i >= nbi) continue;                       // no block-wide barrier inside the loop
        const int64_t j0 = tj * CROSS_RB, j1 = (j0 + CROSS_RB < nbj) ? j0 + CROSS_RB : nbj;
        const int64_t c0 = i * s;
        // column of the box area: the padded area array at the COARSE indices (core.py:1560)
        int64_t ac = i;
        bool nanfill = false;
        if (ac >= nx) { ac = pad_source(ac, nx, pad_mode); nanfill = ac < 0; if (nanfill) ac = 0; }
        double cmn = inf, cmx = -inf;                 // the corner row shared with the previous box
        row_segment(qs + (size_t)(j0 * s) * nx, c0, s, nx, pad_mode, cmn, cmx);
        if constexpr (S == 0) {
            for (int64_t j = j0; j < j1; ++j) {
                double mn = cmn, mx = cmx;
                for (int r = 1; r < s; ++r) row_segment(qs + (size_t)(j * s + r) * nx, c0, s, nx, pad_mode, mn, mx);
                cmn = inf; cmx = -inf;
                row_segment(qs + (size_t)(j * s + s) * nx, c0, s, nx, pad_mode, cmn, cmx);
                mn = fmin(mn, cmn); mx = fmax(mx, cmx);
                box_done<TA, CNT>(s_cx, N, mn, mx, as[(size_t)j * nx + ac], nanfill, fs, c_first, inv_step, zlo, g, my_len, my_cnt, s_dir, cshift);
            }
        } else if constexpr (S > 0) {
            constexpr int B = S == 1 ? 8 : S == 2 ? 4 : S == 3 ? 2 : 1;     // boxes per load batch
            int64_t src[S + 1]; bool pnan[S + 1];      // source columns of the S+1 corner columns (X padding resolved once)
#pragma unroll
            for (int d = 0; d <= S; ++d) {
                int64_t c = c0 + d; pnan[d] = false;
                if (c >= nx) { c = pad_source(c, nx, pad_mode); pnan[d] = c < 0; if (pnan[d]) c = 0; }
                src[d] = c;
            }
            for (int64_t jb = j0; jb < j1; jb += B) {
                TQ v[B][S][S + 1]; TA av[B];
#pragma unroll
                for (int b = 0; b < B; ++b) {          // all loads of the batch in flight together
                    const int64_t jj = (jb + b < j1) ? jb + b : j1 - 1;
                    av[b] = as[(size_t)jj * nx + ac];
#pragma unroll
                    for (int r = 0; r < S; ++r) {
                        const TQ* row = qs + (size_t)(jj * S + r + 1) * nx;
#pragma unroll
                        for (int d = 0; d <= S; ++d) v[b][r][d] = row[src[d]];
                    }
                }
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    if (jb + b >= j1) break;
                    double mn = cmn, mx = cmx;
#pragma unroll
                    for (int r = 0; r < S - 1; ++r)
#pragma unroll
                        for (int d = 0; d <= S; ++d) { const double x = pnan[d] ? qnan : (double)v[b][r][d]; mn = fmin(mn, x); mx = fmax(mx, x); }
                    cmn = inf; cmx = -inf;
#pragma unroll
                    for (int d = 0; d <= S; ++d) { const double x = pnan[d] ? qnan : (double)v[b][S - 1][d]; cmn = fmin(cmn, x); cmx = fmax(cmx, x); }
                    mn = fmin(mn, cmn); mx = fmax(mx, cmx);
                    box_done<TA, CNT>(s_cx, N, mn, mx, av[b], nanfill, fs, c_first, inv_step, zlo, g, my_len, my_cnt, s_dir, cshift);
                }
            }
        }
    }
    __syncthreads();
    // copies -> one compact array s_len[0 .. N] (s_cnt likewise), a chunk of CROSS_TPB levels at a time: the compact cells of a
    // chunk lie below every interleaved cell that is still unread (k <= k * ncopy), and a barrier separates a chunk's reads
    // from its writes
    for (int k0 = 0; k0 <= N; k0 += CROSS_TPB) {
        const int k = k0 + tid;
        double l = 0.0; unsigned n = 0;
        if (k <= N)
            for (int c = 0; c < ncopy; ++c) {
                const int cc = (c + tid) & (ncopy - 1);                // rotated start: the lanes of a wave read distinct banks
                l += s_len[((size_t)k << cshift) + cc]; if (CNT) n += s_cnt[((size_t)k << cshift) + cc];
            }
        __syncthreads();
        if (k <= N) { s_len[k] = l; if (CNT) s_cnt[k] = n; }
        __syncthreads();
    }
    if (inv_step > 0.0 && tid < 64) {
        // equally spaced levels: the cells hold DIFFERENCES D[0..N] (box_done) whose total is zero, so the sum at level k is
        // both the prefix D[0] + .. + D[k] and minus the suffix D[k+1] + .. + D[N].  The lower half of the levels takes the
        // prefix, the upper half the suffix: a level's rounding error then scales with the mass on ITS side of the range --
        // small where the sums themselves are small, and a level no box crosses at either end comes out as exactly 0.
        // One wave, fixed order: lane l owns a run of levels, the lane totals are scanned with shuffles.  (More than ~1000
        // contours: the runs no longer fit the registers, prefix for all levels.)
        const int H0 = (N + 1) / 2;
        const bool two = (N - H0 + 63) / 64 <= 8;
        const int H = two ? H0 : N;                             // levels [0, H): prefix; [H, N): suffix over the cells k+1 .. N
        {
            const int per = (H + 63) / 64;
            const int k0 = tid * per, k1 = (k0 + per < H) ? k0 + per : H;
            double run = 0.0; unsigned crun = 0u;
            for (int k = k0; k < k1; ++k) { run += s_len[k]; if (CNT) crun += s_cnt[k]; }
            for (int o = 1; o < 64; o <<= 1) {
                const double t = __shfl_up(run, o); const unsigned ct = __shfl_up(crun, o);
                if (tid >= o) { run += t; crun += ct; }
            }
            double base = __shfl_up(run, 1); unsigned cbase = __shfl_up(crun, 1);
            if (tid == 0) { base = 0.0; cbase = 0u; }
            for (int k = k0; k < k1; ++k) {                     // a lane reads and writes its own run only
                base += s_len[k]; s_len[k] = base;
                if (CNT) { cbase += s_cnt[k]; s_cnt[k] = cbase; }
            }
        }
        if (two) {
            // cells N, N-1, .. H+1 from the top; level k = cell - 1 receives -(D[cell] + .. + D[N]).  Lane l owns the cells
            // (c1, c0]; every lane first takes its run into registers (the result of a lane's last cell lands on the next
            // lane's first cell), then the results are written -- one wave, so the two phases are ordered.
            const int per = (N - H + 63) / 64;                  // <= 8
            const int c0 = N - tid * per;
            int c1 = c0 - per; if (c1 < H) c1 = H;
            double v[8]; unsigned cv[8];
            double run = 0.0; unsigned crun = 0u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c0 - i; const bool in = i < per && c > c1;
                v[i] = in ? s_len[c] : 0.0; cv[i] = (CNT && in) ? s_cnt[c] : 0u;
                run += v[i]; crun += cv[i];
            }
            for (int o = 1; o < 64; o <<= 1) {
                const double t = __shfl_up(run, o); const unsigned ct = __shfl_up(crun, o);
                if (tid >= o) { run += t; crun += ct; }
            }
            double base = __shfl_up(run, 1); unsigned cbase = __shfl_up(crun, 1);
            if (tid == 0) { base = 0.0; cbase = 0u; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = c0 - i;
                if (i < per && c > c1) {
                    base += v[i]; s_len[c - 1] = -base;
                    if (CNT) { cbase += cv[i]; s_cnt[c - 1] = 0u - cbase; }
                }
            }
        }
    }
    __syncthreads();
    const size_t pb = ((size_t)slab * bps + blockIdx.x) * N;
    for (int k = tid; k < N; k += CROSS_TPB) {
        // the difference form leaves, at a level none of this block's boxes crosses, the residue of cancelling sums (~1e-16 of
        // the mass on that side, either sign); the box count is exact, so with it such a level is written as the exact 0 the
        // reference's per-contour loop gives.  (Lengths without counts, out_cnt == NULL: agreement to ~1e-15 of the largest level.)
        const double l = s_len[k] + s_dir[k];
        part_len[pb + k] = (CNT && s_cnt[k] == 0u) ? 0.0 : l;
        if (CNT) part_cnt[pb + k] = s_cnt[k];
    }
}

// one block per (contour, slab): thread t adds the partials of blocks t, t+256, ... (fixed order), then a fixed
// tree over the 256 threads -- a single slab has up to 2048 block partials, far too many for one thread
// (201 threads walking 2048 dependent loads each took 250 us)
__global__ __launch_bounds__(256)
void k_crossing_reduce(const double* __restrict__ part_len, const unsigned* __restrict__ part_cnt, int bps, int N,
                       double* __restrict__ out_len, unsigned long long* __restrict__ out_cnt)
{
    const int k = blockIdx.x, tid = threadIdx.x;
    const size_t slab = blockIdx.y;
    double len = 0.0; unsigned long long cnt = 0;
    for (int b = tid; b < bps; b += 256) {
        len += part_len[(slab * bps + b) * N + k];
        if (out_cnt) cnt += part_cnt[(slab * bps + b) * N + k];
    }
    for (int o = 32; o > 0; o >>= 1) { len += __shfl_xor(len, o); cnt += __shfl_xor(cnt, o); }
    __shared__ double s_l[4];
    __shared__ unsigned long long s_c[4];
    if ((tid & 63) == 0) { s_l[tid >> 6] = len; s_c[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        if (out_len) out_len[slab * N + k] = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
        if (out_cnt) out_cnt[slab * N + k] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    }
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
    const int np = (N + 1) | 1;                           // N + 1 cells (the difference form writes at index N), odd row pitch: the copies of one contour fall in different banks
    int ncopy = 8;
    { const int v = ctx->knobs.cross_ncopy; if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) ncopy = v; }   // experiment knob (xc_create)
    while (ncopy > 1 && (size_t)(2 * N + 3) * 8 + (size_t)ncopy * np * 12 > 48 * 1024) ncopy >>= 1;   // several blocks per CU
    const size_t lds = (size_t)(2 * N + 3) * 8 + (size_t)ncopy * np * 12;
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
    // stride 1: DPP path; 2..5: one box per thread (loads of several boxes batched for 2 and 4); 6..63: lanes along
    // the fine columns (measured on cfg2 slabs: 25-35 us flat, where one box per thread needs 36 / 118 / 538 us at
    // strides 8 / 16 / 32 but only 16 / 18 / 14 us at 2 / 3 / 4); larger: one box per thread again
    const bool cols = stride >= 6 && stride <= 63;
    const int nbw = cols ? 63 / stride : 0;
    const int rbox = cols ? (64 / stride > 1 ? 64 / stride : 1) : CROSS_RB;
    const int64_t tw = stride == 1 ? CROSS_W1 : CROSS_TPB;
    const int64_t ntj = (nbj + rbox - 1) / rbox;
    const int64_t nti = cols ? ((nbi + nbw - 1) / nbw + 3) / 4 : (nbi + tw - 1) / tw;
    const int env_blocks = ctx->knobs.cross_blocks;                                                                // experiment knob (xc_create)
    int64_t bps = (env_blocks > 0 ? env_blocks : 2048) / nslab; if (bps < 8) bps = 8; if (bps > ntj * nti) bps = ntj * nti;
    const size_t pl = (size_t)nslab * bps * N * 8, pc = (size_t)nslab * bps * N * 4;
    {
        const int rc = ensure_scratch(ctx, ((pl + 255) & ~(size_t)255) + pc);
        if (rc != XC_OK) return rc;
    }
    double* part_len = (double*)ctx->scratch;
    unsigned* part_cnt = (unsigned*)((char*)ctx->scratch + ((pl + 255) & ~(size_t)255));
    const dim3 grid((unsigned)bps, (unsigned)nslab);
#define XC_CROSS4(TQ_, TA_, C_, S_) do { \
        if (lds > 64 * 1024) XC_HIP(ctx, hipFuncSetAttribute((const void*)k_crossing<TQ_, TA_, C_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_crossing<TQ_, TA_, C_, S_>), grid, dim3(CROSS_TPB), lds, ctx->stream, (const TQ_*)q, ny, nx, pad_mode, contours, N, \
                           contours_per_slab, (const TA_*)area, area_per_slab, stride, nbj, nbi, ntj, nti, (int)bps, \
                           ncopy, np, rbox, part_len, part_cnt); } while (0)
#define XC_CROSS3(TQ_, TA_, C_) do { if (stride == 1) XC_CROSS4(TQ_, TA_, C_, 1); else if (stride == 2) XC_CROSS4(TQ_, TA_, C_, 2); \
        else if (stride == 4) XC_CROSS4(TQ_, TA_, C_, 4); else if (cols) XC_CROSS4(TQ_, TA_, C_, -1); \
        else XC_CROSS4(TQ_, TA_, C_, 0); } while (0)
#define XC_CROSS2(TQ_, TA_) do { if (out_cnt) XC_CROSS3(TQ_, TA_, true); else XC_CROSS3(TQ_, TA_, false); } while (0)
    if (q_dtype == XC_F64) { if (area_dtype == XC_F64) XC_CROSS2(double, double); else XC_CROSS2(double, float); }
    else { if (area_dtype == XC_F64) XC_CROSS2(float, double); else XC_CROSS2(float, float); }
#undef XC_CROSS2
#undef XC_CROSS3
#undef XC_CROSS4
    XC_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_crossing_reduce, dim3((unsigned)N, (unsigned)nslab), dim3(256), 0, ctx->stream,
                       part_len, part_cnt, (int)bps, N, out_len, (unsigned long long*)out_cnt);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

}  // namespace xc
