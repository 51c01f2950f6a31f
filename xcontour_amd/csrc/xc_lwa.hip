// K7 -- local finite-amplitude wave activity / local APE (gfx950).
//
// Replaces the J-iteration python loop of Contour2D.cal_local_wave_activity
// (reference core.py:752-791): for every target row j and column x
//     lwa[j,x] = - sum_{y'} (q[y',x] - Q[j]) * mask3(j,y',x) * (dA[y',x]/dAmax) * M[y',x]
// with mask3 in {-1,0,1} from the sign of qe and the side of row j (core.py:757-766)
// and the part selection of core.py:773-784 (masked-out cells are NaN there and are
// skipped by the sum, i.e. contribute nothing).
//
// Mapping: lanes run along X (coalesced row reads), each thread keeps JT target rows in
// registers and streams the column once per JT targets; the y' loop is sequential, the
// same order numpy's axis-0 nansum uses, so results are reproducible bit for bit.
#include "xc_internal.h"

namespace xc {
namespace {


// a value every lane of the wave holds: hand it to the scalar unit
__device__ __forceinline__ double lane_uniform(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ int mask3(double qe, bool m, int increase)
{
    // core.py:759-766
    const bool neg = increase ? (qe > 0.0) : (qe < 0.0);   // -> -1 on the far side
    const bool pos = increase ? (qe < 0.0) : (qe > 0.0);   // -> +1 on the near side
    return (pos && m) ? 1 : (m ? 0 : (neg ? -1 : 0));
}

// Once per call, one block per (row, slab):
//  * wei = dA.squeeze() / max(dA) (core.py:723-724) does not depend on the target row: one division per cell
//    instead of one per (target row, cell);
//  * NaN-skipping min / max of every tracer row.  mask3(j, y', x) != 0 needs qe < 0 on the near side or
//    qe > 0 on the far side of row j, so a row whose extrema exclude that (almost all rows away from the
//    band where the tracer is displaced across Q[j]) contributes nothing to target j and is skipped with a
//    wave-uniform test -- without loading it.
constexpr int LWA_RB = 8;     // rows per load batch of k_lwa; rowinfo is padded by as many rows

// Once per call, one block per (row, slab):
//  * wei = dA.squeeze() / max(dA) (core.py:723-724) does not depend on the target row: one division per cell
//    instead of one per (target row, cell);
//  * rowinfo[slab][ny + LWA_RB][2] = {coord, Q} (padded, contiguous: wide scalar loads in k_lwa);
//  * stripmm[slab][strip][ny][2]: NaN-skipping min / max of every 64-column strip of every tracer row.
//    mask3(j, y', x) != 0 needs qe < 0 on the near side or qe > 0 on the far side of row j, so a strip row whose
//    extrema exclude that (almost all rows away from the band where the tracer is displaced across Q[j])
//    contributes nothing to the wave that owns the strip and is never loaded by it.
template <typename T>
__global__ __launch_bounds__(256)
void k_lwa_prep(const T* __restrict__ q, const double* __restrict__ Q, const double* __restrict__ coord,
                const double* __restrict__ dA, int dA_rank, double dA_max,
                int64_t ny, int64_t nx, int64_t nstrip, double* __restrict__ wei, double* __restrict__ rowinfo,
                double* __restrict__ stripmm)
{
    const int64_t y = blockIdx.x, slab = blockIdx.y;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double* ri = rowinfo + ((size_t)slab * (ny + LWA_RB) + y) * 2;
    if (y >= ny) {                                        // padding rows are never inside a band
        if (threadIdx.x == 0 && blockIdx.z == 0) { ri[0] = coord[ny - 1]; ri[1] = __longlong_as_double(0x7ff8000000000000LL); }
        return;
    }
    if (threadIdx.x == 0 && blockIdx.z == 0) { ri[0] = coord[y]; ri[1] = Q[(size_t)slab * ny + y]; }
    const T* row = q + ((size_t)slab * ny + y) * nx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blockIdx.z: chunk of 64 strips (a very wide, short plane would otherwise walk all its strips in one block)
    const int64_t st1 = ((int64_t)blockIdx.z + 1) * 64 < nstrip ? ((int64_t)blockIdx.z + 1) * 64 : nstrip;
    for (int64_t st = (int64_t)blockIdx.z * 64 + wave; st < st1; st += 4) {
        const int64_t x = st * 64 + lane;
        double mn = inf, mx = -inf;
        if (x < nx) {
            const double v = (double)row[x];
            mn = fmin(mn, v); mx = fmax(mx, v);
            if (slab == 0 && dA_rank != XC_DA_ROW) wei[y * nx + x] = __ddiv_rn(dA[y * nx + x], dA_max);
        }
        for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
        if (lane == 0) {
            double* sm = stripmm + (((size_t)slab * nstrip + st) * ny + y) * 2;
            sm[0] = mn; sm[1] = mx;
        }
    }
    if (slab == 0 && dA_rank == XC_DA_ROW && threadIdx.x == 0 && blockIdx.z == 0) wei[y] = __ddiv_rn(dA[y], dA_max);
}

// V2: cal_local_wave_activity2 (core.py:802-905): qe = q[row j] - Q[all rows], opposite sign convention.
// JT target rows per thread: 1 for small problems (more waves in flight), 4 when the slab is large
// (each thread re-streams its column once per JT targets).
template <typename T, bool V2, int JT>
__global__ __launch_bounds__(256)
void k_lwa(const T* __restrict__ q, const double* __restrict__ Q, const double* __restrict__ coord,
           const double* __restrict__ wei_, int dA_rank,
           const double* __restrict__ M, int M_rank, const double* __restrict__ rowinfo,
           const double* __restrict__ stripmm,
           int64_t ny, int64_t nx, int increase, int part, double* __restrict__ out)
{
    const int coord_incre = !(coord[ny - 1] < coord[0]);                 // core.py:736-738
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t x = (int64_t)blockIdx.x * 64 + lane;
    const int64_t j0 = ((int64_t)blockIdx.y * 4 + wave) * JT;
    if (j0 >= ny) return;
    const size_t so = (size_t)blockIdx.z * ny * nx;
    const T* qs = q + so;
    const double* Qs = Q + (size_t)blockIdx.z * ny;
    const bool active = x < nx;

    const double* rinfo = rowinfo + (size_t)blockIdx.z * (ny + LWA_RB) * 2;
    const double* smm = stripmm + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * ny * 2;      // this wave's strip
    double Qj[JT], cj[JT], acc[JT];
    double tlo[JT], thi[JT];            // wave-uniform: V1 the target level Q[j] (both), V2 min / max of the strip of tracer row j
#pragma unroll
    for (int t = 0; t < JT; ++t) {
        const int64_t j = (j0 + t < ny) ? j0 + t : ny - 1;
        tlo[t] = V2 ? smm[2 * j] : Qs[j]; thi[t] = V2 ? smm[2 * j + 1] : Qs[j];
        Qj[t] = V2 ? (active ? (double)qs[j * nx + x] : 0.0) : Qs[j];      // V2: the tracer on target row j
        cj[t] = coord[j]; acc[t] = 0.0;
    }
    const int inc_eff = V2 ? !increase : increase;                          // core.py:865-872 vs 759-766
    // keep the sign of the selected part: 'upper' keeps mask>0 if increase else mask<0 (core.py:775-784)
    const int keep = (part == 0) ? 0 : (((part == 1) == (increase != 0)) ? 1 : -1);

    // rows are consumed strictly in y' order (numpy's axis-0 nansum order), but the loads of RB rows
    // are issued together so that their latency overlaps
    constexpr int RB = LWA_RB;
    const int64_t xl = active ? x : 0;
    // Band of rows that can contribute to this wave's targets, found once with the lanes spread over y':
    // mask3(j, y', x) != 0 needs qe < 0 on the near side or qe > 0 on the far side of row j; the row extrema in
    // rowinfo bound qe, so rows outside [y0, y1) -- almost all rows away from where the tracer is displaced
    // across Q[j] -- are never loaded.  (Rows inside the band that cannot contribute still add nothing.)
    // The extrema are those of THIS wave's 64-column strip, so a meandering front costs each wave only its own part.
    int64_t y0 = ny, y1 = 0;
    for (int64_t yy = 0; yy < ny; yy += 64) {
        const int64_t y = (yy + lane < ny) ? yy + lane : ny - 1;
        const double rmin = smm[2 * y], rmax = smm[2 * y + 1], cyr = rinfo[2 * y], Qy = rinfo[2 * y + 1];
        bool nd = false;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
            // V1: qe = q[y',x] - Q[j] in [rmin - Q_j, rmax - Q_j];  V2: qe = q[j,x] - Q[y'] in [tlo - Q_y, thi - Q_y]
            const bool anypos = V2 ? (thi[t] > Qy) : (rmax > thi[t]);
            const bool anyneg = V2 ? (tlo[t] < Qy) : (rmin < tlo[t]);
            const bool m = coord_incre ? (cyr >= cj[t]) : (cyr <= cj[t]);
            nd |= m ? (inc_eff ? anyneg : anypos) : (inc_eff ? anypos : anyneg);
        }
        const unsigned long long hit = __ballot(nd && yy + lane < ny);
        if (hit) {
            const int64_t first = yy + (__ffsll((long long)hit) - 1), last = yy + 63 - __clzll((long long)hit);
            y0 = first < y0 ? first : y0; y1 = last + 1 > y1 ? last + 1 : y1;
        }
    }
    for (int64_t yb = y0 & ~(int64_t)(RB - 1); yb < y1; yb += RB) {
        double qv_[RB], wv_[RB], mv_[RB], cy_[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int64_t y = (yb + r < ny) ? yb + r : ny - 1;
            qv_[r] = V2 ? Qs[y] : (double)qs[y * nx + xl];
            cy_[r] = rinfo[2 * y];
            wv_[r] = (dA_rank == XC_DA_ROW) ? wei_[y] : wei_[y * nx + xl];
            mv_[r] = (M_rank == XC_DA_ROW) ? M[y] : M[y * nx + xl];
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (yb + r >= ny) break;
            const double qv = qv_[r], mv = mv_[r];
            const double cy = lane_uniform(cy_[r]);                                     // the same value in every lane: keep the side test scalar
            const double wei = wv_[r];                                                  // dA / max(dA), core.py:724
#pragma unroll
            for (int t = 0; t < JT; ++t) {
                // mask3 (core.py:759-766 / 865-872) with the side m of row y' WAVE-UNIFORM:  mask3 != 0  <=>  u > 0 with
                // u = -qe on the side where mask3 = +-1 needs qe < 0 and u = +qe on the other; qe * mask3 = -+u exactly
                // (a - b and b - a are exact negations, as are x * (-1) and -x), so the term of core.py:789 is
                // -+((u * wei) * M) bit for bit and only |term| is accumulated; the sign goes on at the end.
                const bool m = coord_incre ? (cy >= cj[t]) : (cy <= cj[t]);             // core.py:757 (scalar)
                if (keep != 0 && (keep > 0) != m) continue;                             // 'upper' / 'lower' keep one side (core.py:775-784)
                const double a = V2 ? Qj[t] : qv, b = V2 ? qv : Qj[t];                  // qe = a - b, core.py:860 / 754
                const double u = (m == (inc_eff != 0)) ? __dsub_rn(b, a) : __dsub_rn(a, b);
                // (kept as a branch: for most (j, y') pairs no lane contributes and the wave skips the products;
                //  a branch-free select version measured 89 vs 63 us on cfg3)
                if (u > 0.0) {
                    const double term = __dmul_rn(__dmul_rn(u, wei), mv);
                    if (term == term) acc[t] = __dadd_rn(acc[t], term);                 // nansum
                }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int t = 0; t < JT; ++t)
            if (j0 + t < ny) out[so + (size_t)(j0 + t) * nx + x] = inc_eff ? (acc[t] == 0.0 ? -0.0 : acc[t]) : -acc[t];   // -(sum of terms), core.py:789 (an empty sum is -0.0 there)
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void k_lwa_masks(const T* __restrict__ q, const double* __restrict__ Q, const double* __restrict__ coord,
                 int64_t ny, int64_t nx, int increase, int v2,
                 const int32_t* __restrict__ mask_idx, int nmask, int8_t* __restrict__ out)
{
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    const int slab = blockIdx.z / nmask, im = blockIdx.z % nmask;
    if (x >= nx) return;
    const int coord_incre = !(coord[ny - 1] < coord[0]);
    const int64_t j = mask_idx[im];
    const double qe = v2 ? __dsub_rn((double)q[(size_t)slab * ny * nx + j * nx + x], Q[(size_t)slab * ny + y])
                         : __dsub_rn((double)q[(size_t)slab * ny * nx + y * nx + x], Q[(size_t)slab * ny + j]);
    const bool m = coord_incre ? (coord[y] >= coord[j]) : (coord[y] <= coord[j]);
    out[(((size_t)slab * nmask + im) * ny + y) * nx + x] = (int8_t)mask3(qe, m, v2 ? !increase : increase);
}

}  // namespace

int launch_lwa(xc_ctx* ctx, const void* q, int q_dtype, const double* Q, const double* coord,
               const double* dA, int dA_rank, double dA_max, const double* M, int M_rank,
               int64_t nslab, int64_t ny, int64_t nx, int increase, int part, int variant,
               const int32_t* mask_idx, int nmask, double* out_lwa, int8_t* out_masks)
{
    if (!q || !Q || !coord || !dA || !out_lwa || nslab < 1 || ny < 2 || nx < 1)
        return fail(ctx, XC_EBADARG, "xc_lwa: bad arguments");
    if (dA_rank != XC_DA_ROW && dA_rank != XC_DA_PLANE) return fail(ctx, XC_EBADARG, "xc_lwa: dA_rank must be ROW or PLANE");
    if (M_rank != XC_DA_NONE && M_rank != XC_DA_ROW && M_rank != XC_DA_PLANE) return fail(ctx, XC_EBADARG, "xc_lwa: bad M_rank");
    if (M_rank != XC_DA_NONE && !M) return fail(ctx, XC_EBADARG, "xc_lwa: M is NULL");
    if (part < 0 || part > 2) return fail(ctx, XC_EBADARG, "xc_lwa: part must be 0 (all), 1 (upper) or 2 (lower)");
    if (nmask < 0 || (nmask > 0 && (!mask_idx || !out_masks))) return fail(ctx, XC_EBADARG, "xc_lwa: mask arguments");
    if (ny > 65535 || nslab * (nmask > 0 ? nmask : 1) > 65535) return fail(ctx, XC_EBADARG, "xc_lwa: ny / nslab too large");
    // small problems: one target row per thread so that the whole chip is busy
    // scratch: wei (same rank as dA) and the per-row tracer extrema; M defaults to dA itself (core.py:789 as written)
    const int64_t nw = dA_rank == XC_DA_ROW ? ny : ny * nx;
    const int64_t nstrip = (nx + 63) / 64;
    { const int rc = ensure_scratch(ctx, (size_t)nw * 8 + (size_t)nslab * (ny + LWA_RB) * 16 + (size_t)nslab * nstrip * ny * 16); if (rc != XC_OK) return rc; }
    double* wei = (double*)ctx->scratch;
    double* rowinfo = wei + nw;
    double* stripmm = rowinfo + (size_t)nslab * (ny + LWA_RB) * 2;
    if ((nstrip + 63) / 64 > 65535) return fail(ctx, XC_EBADARG, "xc_lwa: nx too large");
    const dim3 gp((unsigned)(ny + LWA_RB), (unsigned)nslab, (unsigned)((nstrip + 63) / 64));
    if (q_dtype == XC_F64)
        hipLaunchKernelGGL(k_lwa_prep<double>, gp, dim3(256), 0, ctx->stream, (const double*)q, Q, coord, dA, dA_rank, dA_max, ny, nx, nstrip, wei, rowinfo, stripmm);
    else if (q_dtype == XC_F32)
        hipLaunchKernelGGL(k_lwa_prep<float>, gp, dim3(256), 0, ctx->stream, (const float*)q, Q, coord, dA, dA_rank, dA_max, ny, nx, nstrip, wei, rowinfo, stripmm);
    else return fail(ctx, XC_EBADARG, "xc_lwa: q_dtype must be XC_F32 or XC_F64");
    if (M_rank == XC_DA_NONE) { M = dA; M_rank = dA_rank; }
    const bool small = (double)ny * (double)ny * (double)nx * (double)nslab < 2.0e8;
    const int jt = small ? 1 : 4;
    dim3 grid((unsigned)((nx + 63) / 64), (unsigned)((ny + 4 * jt - 1) / (4 * jt)), (unsigned)nslab);
#define XC_LWA2(T, V, J) hipLaunchKernelGGL((k_lwa<T, V, J>), grid, dim3(256), 0, ctx->stream, (const T*)q, Q, coord, wei, dA_rank, \
                           M, M_rank, rowinfo, stripmm, ny, nx, increase, part, out_lwa)
#define XC_LWA(T, V) do { if (small) XC_LWA2(T, V, 1); else XC_LWA2(T, V, 4); } while (0)
    if (variant != 0 && variant != 1) return fail(ctx, XC_EBADARG, "xc_lwa: variant must be 0 or 1");
    if (q_dtype == XC_F64) { if (variant) XC_LWA(double, true); else XC_LWA(double, false); }
    else if (q_dtype == XC_F32) { if (variant) XC_LWA(float, true); else XC_LWA(float, false); }
    else return fail(ctx, XC_EBADARG, "xc_lwa: q_dtype must be XC_F32 or XC_F64");
#undef XC_LWA
#undef XC_LWA2
    XC_HIP(ctx, hipGetLastError());
    if (nmask > 0) {
        dim3 g2((unsigned)((nx + 255) / 256), (unsigned)ny, (unsigned)(nslab * nmask));
        if (q_dtype == XC_F64)
            hipLaunchKernelGGL(k_lwa_masks<double>, g2, dim3(256), 0, ctx->stream, (const double*)q, Q, coord, ny, nx,
                               increase, variant, mask_idx, nmask, out_masks);
        else
            hipLaunchKernelGGL(k_lwa_masks<float>, g2, dim3(256), 0, ctx->stream, (const float*)q, Q, coord, ny, nx,
                               increase, variant, mask_idx, nmask, out_masks);
        XC_HIP(ctx, hipGetLastError());
    }
    return XC_OK;
}

}  // namespace xc
