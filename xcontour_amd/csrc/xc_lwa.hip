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
                double* __restrict__ stripmm, const unsigned* __restrict__ gate, unsigned epoch)
{
    if (gate && *gate != epoch) return;      // the interval kernel (K7F) took this call: its premises held (k_lwa_check)
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
           int64_t ny, int64_t nx, int increase, int part, double* __restrict__ out, const unsigned* __restrict__ gate, unsigned epoch)
{
    if (gate && *gate != epoch) return;      // (see k_lwa_prep)
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

// ---- small planes (the reference's own 256 x 512 field, X-Z sections): ONE launch, no prologue kernel.  A 1024-thread
// workgroup owns a 64-column strip and `tper` target rows, 16 at a time (one per wave, lanes along X):
//   (a) the whole strip of the tracer goes into LDS (row pitch 65: a thread can walk a row without bank conflicts), with the
//       (coord, Q) pairs and the per-row weights;  (b) 256 threads take the NaN-skipping extrema of the strip's rows;
//   (c) every wave finds the band of rows that can contribute to its target from those (as k_lwa does) and the workgroup takes
//       the union;  (d) wei = dA / max(dA) and a 2-D metric are staged for the union band only, `wchunk` rows at a time -- the
//       f64 divisions are done for the rows that matter, once per workgroup;  (e) the waves walk their bands out of LDS.
// Same arithmetic and order as k_lwa (bit-identical).  k_lwa_prep + k_lwa remain for planes whose strip does not fit the LDS.
constexpr int LWA_SW = 8;                       // waves per workgroup = target rows in flight per workgroup (cfg3: 256 workgroups, one per CU)
__host__ __device__ inline size_t lwa_strip_lds(int64_t ny, size_t tsize, bool wplane, bool mplane, int wchunk)
{
    size_t b = (size_t)ny * 6 * 8;                                            // coord, Q, min, max, row wei, row M
    if (wplane) b += (size_t)wchunk * 64 * 8;
    if (mplane) b += (size_t)wchunk * 64 * 8;
    b += (size_t)ny * 65 * tsize + 64;
    return (b + 15) & ~(size_t)15;
}

template <typename T, bool V2>
__global__ __launch_bounds__(64 * LWA_SW)
void k_lwa_strip(const T* __restrict__ q, const double* __restrict__ Q, const double* __restrict__ coord,
                 const double* __restrict__ dA, int dA_rank, double dA_max, const double* __restrict__ M, int M_rank,
                 int64_t ny_, int64_t nx_, int increase, int part, int tper, int wchunk, double* __restrict__ out,
                 const unsigned* __restrict__ gate, unsigned epoch)
{
    if (gate && *gate != epoch) return;      // (see k_lwa_prep)
    extern __shared__ __align__(16) double sm[];
    const int ny = (int)ny_, nx = (int)nx_;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool wplane = dA_rank == XC_DA_PLANE, mplane = M_rank == XC_DA_PLANE;
    double* s_c = sm;                 double* s_Q = s_c + ny;      double* s_mn = s_Q + ny;   double* s_mx = s_mn + ny;
    double* s_wr = s_mx + ny;         double* s_Mr = s_wr + ny;
    double* s_wei = s_Mr + ny;        double* s_Mp = s_wei + (wplane ? (size_t)wchunk * 64 : 0);
    int* s_band = reinterpret_cast<int*>(s_Mp + (mplane ? (size_t)wchunk * 64 : 0));           // [2] + padding to 64 bytes
    T* s_q = reinterpret_cast<T*>(reinterpret_cast<char*>(s_band) + 64);
    const size_t so = (size_t)blockIdx.z * ny * nx;
    const T* qs = q + so;
    const double* Qs = Q + (size_t)blockIdx.z * ny;
    const int x0 = blockIdx.x * 64, x = x0 + lane;
    const bool active = x < nx;
    const int xl = active ? x : nx - 1;
    const double inf = __longlong_as_double(0x7ff0000000000000LL), nan = __longlong_as_double(0x7ff8000000000000LL);

    // (a) the strip: wave w takes rows w, w + 16, ...; sixteen loads in flight per lane
    for (int yb = wave; yb < ny; yb += LWA_SW * 16) {
        T r[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int y = yb + LWA_SW * k; r[k] = qs[(size_t)(y < ny ? y : ny - 1) * nx + xl]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int y = yb + LWA_SW * k; if (y < ny) s_q[y * 65 + lane] = active ? r[k] : (T)nan; }   // beyond the plane: NaN -> no contribution
    }
    for (int y = tid; y < ny; y += 64 * LWA_SW) {
        s_c[y] = coord[y]; s_Q[y] = Qs[y];
        s_wr[y] = wplane ? 0.0 : __ddiv_rn(dA[y], dA_max);                    // core.py:723-724 (row weights: once per row)
        s_Mr[y] = mplane ? 0.0 : M[y];
    }
    __syncthreads();
    // (b) NaN-skipping extrema of the strip's rows
    for (int y = tid; y < ny; y += 64 * LWA_SW) {
        double mn[4] = {inf, inf, inf, inf}, mx[4] = {-inf, -inf, -inf, -inf};   // four independent chains: the LDS reads pipeline
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double v = (double)s_q[y * 65 + c + k]; mn[k] = fmin(mn[k], v); mx[k] = fmax(mx[k], v); }
        }
        s_mn[y] = fmin(fmin(mn[0], mn[1]), fmin(mn[2], mn[3])); s_mx[y] = fmax(fmax(mx[0], mx[1]), fmax(mx[2], mx[3]));
    }
    __syncthreads();

    const int coord_incre = !(s_c[ny - 1] < s_c[0]);                           // core.py:736-738
    const int inc_eff = V2 ? !increase : increase;                             // core.py:865-872 vs 759-766
    const int keep = (part == 0) ? 0 : (((part == 1) == (increase != 0)) ? 1 : -1);   // core.py:775-784
    const int jlo = blockIdx.y * tper, jhi = (jlo + tper < ny) ? jlo + tper : ny;
    for (int jb = jlo; jb < jhi; jb += LWA_SW) {
        const int j = jb + wave;                                               // this wave's target row (idle beyond jhi, but it joins the barriers)
        const bool live = j < jhi;
        const int jc = live ? j : ny - 1;
        // (c) the band of rows that can contribute to target j (lanes spread over y')
        int y0 = ny, y1 = 0;
        const double tlo = V2 ? s_mn[jc] : s_Q[jc], thi = V2 ? s_mx[jc] : s_Q[jc], cj = s_c[jc];
        for (int yy = 0; yy < ny && live; yy += 64) {
            const int y = (yy + lane < ny) ? yy + lane : ny - 1;
            const double rmin = s_mn[y], rmax = s_mx[y], cyr = s_c[y], Qy = s_Q[y];
            const bool anypos = V2 ? (thi > Qy) : (rmax > thi);
            const bool anyneg = V2 ? (tlo < Qy) : (rmin < tlo);
            const bool m = coord_incre ? (cyr >= cj) : (cyr <= cj);
            bool nd = m ? (inc_eff ? anyneg : anypos) : (inc_eff ? anypos : anyneg);
            if (keep != 0 && (keep > 0) != m) nd = false;                      // 'upper' / 'lower': one side only
            const unsigned long long hit = __ballot(nd && yy + lane < ny);
            if (hit) {
                const int first = yy + (__ffsll((long long)hit) - 1), last = yy + 63 - __clzll((long long)hit);
                y0 = first < y0 ? first : y0; y1 = last + 1 > y1 ? last + 1 : y1;
            }
        }
        const double Qj = V2 ? (double)s_q[jc * 65 + lane] : s_Q[jc];
        double acc = 0.0;
        // (d) plane weights: the union band of the 16 targets, staged `wchunk` rows at a time; row weights: nothing to stage,
        //     one "chunk" = the wave's own band, no barriers
        const bool staged = wplane || mplane;                                  // workgroup-uniform
        int Y0 = y0, Y1 = y1;
        if (staged) {
            if (tid == 0) { s_band[0] = ny; s_band[1] = 0; }
            __syncthreads();
            if (lane == 0 && live && y0 < y1) { atomicMin(&s_band[0], y0); atomicMax(&s_band[1], y1); }
            __syncthreads();
            Y0 = s_band[0]; Y1 = s_band[1];
        }
        const int step = staged ? wchunk : (Y1 > Y0 ? Y1 - Y0 : 1);
        for (int yc = Y0; yc < Y1; yc += step) {
            const int nr = (Y1 - yc < step) ? Y1 - yc : step;
            if (staged) {
                for (int i0 = tid; i0 < nr * 64; i0 += 64 * LWA_SW * 4) {
                    double dr[4], mr[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = i0 + 64 * LWA_SW * k;
                        const int r = (i < nr * 64 ? i : nr * 64 - 1) >> 6, c = i & 63;
                        const size_t g = (size_t)(yc + r) * nx + (x0 + c < nx ? x0 + c : nx - 1);
                        dr[k] = wplane ? dA[g] : 0.0; mr[k] = mplane ? M[g] : 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = i0 + 64 * LWA_SW * k;
                        if (i < nr * 64) {
                            if (wplane) s_wei[i] = __ddiv_rn(dr[k], dA_max);    // core.py:723-724, for the rows that matter
                            if (mplane) s_Mp[i] = mr[k];
                        }
                    }
                }
                __syncthreads();
            }
            // (e) this wave's rows of the chunk, in y' order -- only the rows whose extrema allow a contribution (the test of (c),
            //     a ballot per 64 rows; the span between the first and the last such row is mostly rows that cannot contribute)
            const int ya = y0 > yc ? y0 : yc, yb_ = y1 < yc + nr ? y1 : yc + nr;
            for (int yy = ya & ~63; yy < yb_ && live; yy += 64) {
                const int yl = yy + lane;
                const int y = yl < ny ? yl : ny - 1;
                const double rmin = s_mn[y], rmax = s_mx[y], cyr = s_c[y], Qy = s_Q[y];
                const bool anypos = V2 ? (thi > Qy) : (rmax > thi);
                const bool anyneg = V2 ? (tlo < Qy) : (rmin < tlo);
                const bool ml = coord_incre ? (cyr >= cj) : (cyr <= cj);
                bool nd = ml ? (inc_eff ? anyneg : anypos) : (inc_eff ? anypos : anyneg);
                if (keep != 0 && (keep > 0) != ml) nd = false;
                unsigned long long hit = __ballot(nd && yl >= ya && yl < yb_);
                while (hit) {
                    // four contributing rows per turn: their LDS reads are issued together, the arithmetic follows in row order
                    int yr[4]; bool on[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        on[k] = hit != 0ull;
                        yr[k] = on[k] ? yy + (__ffsll((long long)hit) - 1) : yr[k > 0 ? k - 1 : 0];
                        if (k == 0 && !on[0]) yr[0] = yy;
                        hit &= hit - (on[k] ? 1ull : 0ull);
                    }
                    double cy[4], qv[4], wv[4], mv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        cy[k] = s_c[yr[k]];
                        qv[k] = V2 ? s_Q[yr[k]] : (double)s_q[yr[k] * 65 + lane];
                        wv[k] = wplane ? s_wei[(yr[k] - yc) * 64 + lane] : s_wr[yr[k]];
                        mv[k] = mplane ? s_Mp[(yr[k] - yc) * 64 + lane] : s_Mr[yr[k]];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (on[k]) {
                            const bool m = coord_incre ? (cy[k] >= cj) : (cy[k] <= cj);      // core.py:757 (wave-uniform)
                            const double a = V2 ? Qj : qv[k], b = V2 ? qv[k] : Qj;           // qe = a - b (core.py:860 / 754); see k_lwa for the sign algebra
                            const double u = (m == (inc_eff != 0)) ? __dsub_rn(b, a) : __dsub_rn(a, b);
                            if (u > 0.0) {
                                const double term = __dmul_rn(__dmul_rn(u, wv[k]), mv[k]);
                                if (term == term) acc = __dadd_rn(acc, term);                 // nansum
                            }
                        }
                    }
                }
            }
            if (staged) __syncthreads();
        }
        if (live && active) out[so + (size_t)j * nx + x] = inc_eff ? (acc == 0.0 ? -0.0 : acc) : -acc;   // core.py:789
    }
}


// =====================================================================================
// K7F  large planes: O(ny log ny) per column instead of O(J * band)            (round 4)
// =====================================================================================
// With q' = s q, Q' = s Q (s = +1 if increase else -1; Q' non-decreasing in j -- it is the sorted reference state) the
// sum of core.py:789 is, per column x and target row j,
//     lwa[j, x] = s * (  sum_{y >= j, q'_y < Q'_j} (Q'_j - q'_y) W_y   [near side, mask3 = +1]
//                      + sum_{y <  j, q'_y > Q'_j} (q'_y - Q'_j) W_y ) [far side,  mask3 = -1],   W = (dA / max dA) * M.
// Because Q' is monotone, the targets a cell (y, x) contributes to form ONE interval of j: with b = #{j: Q'_j < q'_y} and
// a = #{j: Q'_j <= q'_y} (two bounds of one binary search) the cell is a near-side term of j in [a, y] (if a <= y) or a
// far-side term of j in [y + 1, b - 1] (if b >= y + 2), never both.  Either way it adds +W at index p (= a or b) and -W at
// index y + 1 of a difference array D0, and the same with (q'_y - c) W in D1 (c: a reference level that keeps the two big
// terms of the final difference small).  Prefix sums S0, S1 over j then give lwa = s ((Q'_j - c) S0_j - S1_j).
// One binary search and four LDS adds per cell; the band walk costs O(band) per (cell, target group).  The sums are formed
// in another order and through a difference of two products: agreement with the bit-exact kernels is ~1e-13 relative to
// the column's largest value (tests: 1e-9), not bit for bit -- so this path serves planes of more than kLwaFastMinRows rows
// (where the band walk takes milliseconds) and only after k_lwa_check has PROVED its premises (no NaN in Q, Q' monotone,
// the coordinate strictly monotone); xc_set_lwa_exact(ctx, 1) keeps the band walk everywhere.
constexpr int kLwaFastMinRows = 512;

__global__ __launch_bounds__(256)
void k_lwa_check(const double* __restrict__ Q, const double* __restrict__ coord, int ny, int increase, unsigned* __restrict__ flag, unsigned epoch)
{
    const double* Qs = Q + (size_t)blockIdx.x * ny;
    const double s = increase ? 1.0 : -1.0;
    const bool cinc = !(coord[ny - 1] < coord[0]);
    int bad = 0;
    for (int j = threadIdx.x; j < ny; j += 256) {
        const double v = Qs[j];
        bad |= !(fabs(v) < __longlong_as_double(0x7ff0000000000000LL));   // finite: NaN fails, and so does an infinite level ((Q'_j - c) * S0 = inf * 0 = NaN in the interval kernel where the reference sums to 0)
        if (j + 1 < ny) {
            bad |= !(s * Qs[j + 1] >= s * v);                          // (a NaN neighbour fails too)
            bad |= cinc ? !(coord[j + 1] > coord[j]) : !(coord[j + 1] < coord[j]);
        }
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicMax(flag, epoch);      // the word holds the epoch of the last call whose check failed
}

// (round 5) The kernel is PERSISTENT over column groups -- `gridDim.x` workgroups (one per CU: the difference arrays fill the LDS)
// walk the groups b, b + gridDim.x, ... -- so that
//   * Q' and the bucket table below are staged once per workgroup, not once per group;
//   * the cells of the NEXT group are requested before this group's prefix sums, transform and stores (a group used to pay its
//     ~5 us of load latency with nothing else in flight: one workgroup per CU);
//   * bracket search: a table G of LWA_NB + 1 row indices over equal-width value buckets of [Q'_0, Q'_last] -- G[k] = number of levels
//     whose bucket is below k, the bucket being the SAME monotone function of the value for levels and cells -- confines the lower
//     bound of a cell of bucket k to [G[k], G[k + 1]] exactly (no float consistency needed between an edge and a cell), so the 11
//     dependent LDS reads of a full binary search over 1801 levels become ~1 (measured by ablation: the search was 32 of the 108 us);
//   * the prefix sums use all sixteen waves (two waves per array: 128 pieces) instead of eight.
constexpr int LWA_NB = 4096;
template <typename T>
__global__ __launch_bounds__(1024)
void k_lwa_fast(const T* __restrict__ q, const double* __restrict__ Q, const double* __restrict__ dA, int dA_rank, double dA_max,
                const double* __restrict__ M, int M_rank, int ny, int64_t nx, int increase, int side, int CG, int64_t nvb,
                double* __restrict__ out, unsigned* __restrict__ gate, unsigned epoch)
{
    if (gate && *gate == epoch) return;          // k_lwa_check found a premise broken: the band walk enqueued behind this kernel runs instead (gate NULL: the caller vouches)
    extern __shared__ __align__(16) double sm[];
    const int tid = threadIdx.x, nthr = blockDim.x, slab = blockIdx.y;
    const int L = ny + 1;
    double* Qs = sm;                         // [ny]  Q' = s Q
    double* D0 = sm + L;                     // [CG][ny + 1]
    double* D1 = D0 + (size_t)CG * L;        // [CG][ny + 1]
    int* G = (int*)(D1 + (size_t)CG * L);    // [LWA_NB + 1]
    __shared__ double s_half[16];
    const double s = increase ? 1.0 : -1.0;
    const double* Qg = Q + (size_t)slab * ny;
    for (int j = tid; j < ny; j += nthr) Qs[j] = s * Qg[j];
    __syncthreads();
    const double cref = Qs[ny / 2];
    const double q0 = Qs[0], qw_ = Qs[ny - 1] - q0;
    const double bscale = (qw_ > 0.0) ? (double)LWA_NB / qw_ : 0.0;
    auto bucket = [&](double v) { return (int)fmin(fmax((v - q0) * bscale, 0.0), (double)(LWA_NB - 1)); };      // monotone in v; NaN -> 0
    for (int j = tid; j <= ny; j += nthr) {                            // Q' is sorted: level j opens the buckets (k_{j-1}, k_j]
        const int kp = j == 0 ? -1 : bucket(Qs[j - 1]), kj = j < ny ? bucket(Qs[j]) : LWA_NB;
        for (int k = kp + 1; k <= kj; ++k) G[k] = j;
    }
    const T* qs = q + (size_t)slab * ny * nx;
    double* os = out + (size_t)slab * ny * nx;
    // Column groups that share 128-byte lines (16 float64 columns = 16 / CG groups) go to ONE XCD: workgroups are dealt round-robin
    // over the eight XCDs (b and b + 8 share one), each XCD has its own L2, and a group touches only CG * 8 bytes of every line of
    // its rows -- with the plain order the four groups of a line ran on four XCDs and every line of the tracer, the weights and the
    // output crossed the fabric four times (measured: 0.176 -> 0.148 ms per cfg2-sized slab; two / one columns per workgroup: 0.217 / 0.362).
    // virtual block vb = 8 k + xcd  ->  group ((k / GQ) * 8 + xcd) * GQ + k % GQ, GQ = 16 / CG groups per line; vb runs over nvb
    // (a multiple of 8 GQ) in steps of gridDim.x (a multiple of 8: vb keeps its XCD) and the surplus groups are skipped.
    const int GQ = 16 / CG;
    auto group_x0 = [&](int64_t vb) { const int64_t kq = vb >> 3, xcd = vb & 7; return (((kq / GQ) * 8 + xcd) * GQ + (kq % GQ)) * CG; };
    // cells: row-major over (y, column of the group); CPT cells per thread and round: all their loads are issued first (a workgroup
    // of 1024 threads x 8 covers the 7204 cells of four cfg2 columns in ONE round of loads), then the CPT searches advance together
    constexpr int CPT = 8;
    T qraw[CPT];
    double da[CPT], mm_[CPT];
    const double inv_max = 1.0 / dA_max;
    const bool da_row = dA_rank == XC_DA_ROW, m_row = M_rank == XC_DA_NONE ? da_row : (M_rank == XC_DA_ROW);
    const double* Mp = M_rank == XC_DA_NONE ? dA : M;
    // cell i of a group: row i >> cshift, column i & (CG - 1) (CG is 4, 2 or 1: no integer division -- with a runtime `ncol` the four
    // index computations per cell were ~1300 instructions per thread and group, ~9 of a group's ~28 us); columns >= ncol of a ragged
    // last group are simply not there
    const int cshift = CG == 4 ? 2 : (CG == 2 ? 1 : 0), cmask = CG - 1;
    const int ncell = ny << cshift;
    // Addresses: a uniform base (the group's first column: scalar registers) + a 32-bit byte offset per cell; a plane-rank weight
    // shares the tracer's element offset, a row-rank one uses the row.  (Per-cell 64-bit addresses of three arrays, kept alive over the
    // group loop as loop invariants, cost 71 spilled VGPRs in the first persistent version.)  The launcher admits planes of < 2^29 cells.
    auto request = [&](int64_t x0, int ncol, int i0) {                 // the loads of one round of one group (clamped: never out of bounds)
        asm volatile("" : "+v"(i0));                                   // (not a loop invariant: the few index operations per cell are recomputed, not kept in 24 registers)
        const char* qb = (const char*)(qs + x0);
        const char* db = (const char*)(da_row ? dA : dA + x0);
        const char* mb = (const char*)(m_row ? Mp : Mp + x0);
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int i = i0 + u * nthr;
            const unsigned y = (unsigned)((i < ncell ? i : 0) >> cshift), c = (unsigned)((i & cmask) < ncol ? (i & cmask) : 0);
            const unsigned e = y * (unsigned)nx + c;                       // element offset from the group's first column
            qraw[u] = *(const T*)(qb + (size_t)(e * (unsigned)sizeof(T)));
            da[u] = *(const double*)(db + (size_t)((da_row ? y : e) * 8u));    // branch-free: the rank picks the INDEX (three loads per cell, no control flow)
            mm_[u] = *(const double*)(mb + (size_t)((m_row ? y : e) * 8u));    // (no M: the weight itself, from the same line -- core.py:789 with M = dA)
        }
    };
    int64_t vb = blockIdx.x;
    while (vb < nvb && group_x0(vb) >= nx) vb += gridDim.x;
    if (vb < nvb) { const int64_t x0 = group_x0(vb); request(x0, (int)((nx - x0 < CG) ? nx - x0 : CG), tid); }
    while (vb < nvb) {
        const int64_t x0 = group_x0(vb);
        const int ncol = (int)((nx - x0 < CG) ? nx - x0 : CG);
        int64_t vnext = vb + gridDim.x;
        while (vnext < nvb && group_x0(vnext) >= nx) vnext += gridDim.x;
        for (int i = tid; i < 2 * CG * L; i += nthr) D0[i] = 0.0;
        __syncthreads();                                               // (also: Qs, G of the prologue; the stores of the group before)
        for (int i0 = tid; i0 < ncell; i0 += CPT * nthr) {
            if (i0 != tid) request(x0, ncol, i0);                      // (planes of more than 2048 rows: further rounds, not prefetched)
            int lo[CPT], hi[CPT];
            double qv[CPT], wv[CPT];
            bool ok[CPT];
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const int i = i0 + u * nthr;
                ok[u] = i < ncell && (i & cmask) < ncol;
                qv[u] = s * (double)qraw[u];
                wv[u] = (da[u] * inv_max) * mm_[u];                       // (u * wei) * M of core.py:789, weights first (wei = dA / max: here times the reciprocal -- eight float64 divisions per thread and group were ~7 % of the kernel; this path is not the bit-exact one)
                // an INFINITE tracer cell is a premise this kernel cannot check ahead of time: +-inf into the difference arrays turns every
                // row behind the cell into inf - inf = NaN, where the reference's per-row sums stay finite.  Stamp the flag: the gated band
                // walk enqueued behind this kernel then runs and overwrites the plane (mode 3 has no gate: the caller vouched for finite cells)
                if (gate && ok[u] && fabs(qv[u]) == __longlong_as_double(0x7ff0000000000000LL)) atomicMax(gate, epoch);
                ok[u] = ok[u] && (qv[u] == qv[u]) && (wv[u] == wv[u]);    // NaN tracer / weight: the term is NaN and nansum skips it
                const int k = bucket(qv[u]);
                lo[u] = ok[u] ? G[k] : 0; hi[u] = ok[u] ? G[k + 1] : 0;   // lower bound: first j with Q'_j >= q'
            }
            for (int step = 0; step < 32; ++step) {                       // ceil(log2(ny + 1)) steps at most; ~1 behind the bucket table
                bool any = false;
#pragma unroll
                for (int u = 0; u < CPT; ++u)
                    if (lo[u] < hi[u]) { const int mid = (lo[u] + hi[u]) >> 1; if (Qs[mid] < qv[u]) lo[u] = mid + 1; else hi[u] = mid; any = true; }
                if (!any) break;
            }
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                if (!ok[u]) continue;
                const int i = i0 + u * nthr, y = i >> cshift, c = i & cmask;
                const int b = lo[u];
                int a = b;
                while (a < ny && Qs[a] == qv[u]) ++a;                      // upper bound: ties with a level are rare
                int p = -1;
                if (a <= y) { if (side != 2) p = a; }                      // near-side term of targets [a, y]
                else if (b >= y + 2) { if (side != 1) p = b; }             // far-side term of targets [y + 1, b - 1]
                if (p >= 0) {
                    double* d0 = D0 + (size_t)c * L;
                    double* d1 = D1 + (size_t)c * L;
                    const double w = wv[u], qw = (qv[u] - cref) * w;
                    atomicAdd(d0 + p, w);  atomicAdd(d0 + y + 1, -w);
                    atomicAdd(d1 + p, qw); atomicAdd(d1 + y + 1, -qw);
                }
            }
        }
        // the next group's first round of loads is requested HERE: its latency runs under this group's prefix sums and stores (while the
        // searches run the registers are needed: requested before them, 60 VGPRs spilled and the kernel was slower than without)
        __builtin_amdgcn_sched_barrier(0);                             // (the scheduler must not lift these loads above the searches)
        if (vnext < nvb) { const int64_t xn = group_x0(vnext); request(xn, (int)((nx - xn < CG) ? nx - xn : CG), tid); }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // prefix sums over j, in place: an array is cut into 64 * split contiguous pieces, every lane sums its piece (independent LDS
        // reads, one dependent add each), ONE wave scan of the piece totals (the second wave of an array adds the first one's
        // total), then the piece is written back with its offset.  Then lwa[j] = s ((Q'_j - c) S0_j - S1_j) overwrites D0.
        {
            const int wave = tid >> 6, lane = tid & 63, nw = nthr >> 6;
            const int split = (4 * ncol <= nw) ? 2 : 1;
            const int per = (ny + 64 * split - 1) / (64 * split);
            for (int t0 = 0; t0 < 2 * ncol * split; t0 += nw) {
                const int task = t0 + wave;
                const bool on = task < 2 * ncol * split;
                const int arr = on ? task / split : 0, h = task % split;
                double* d = (arr & 1 ? D1 : D0) + (size_t)(arr >> 1) * L;
                int j0 = (h * 64 + lane) * per, j1 = j0 + per;
                if (j1 > ny) j1 = ny;
                if (!on) j1 = j0;
                double tot = 0.0;
                for (int j = j0; j < j1; ++j) tot += d[j];
                double v = tot;                                            // inclusive scan of the piece totals over the lanes
                for (int o = 1; o < 64; o <<= 1) { const double tt = __shfl_up(v, o); if (lane >= o) v += tt; }
                double base = 0.0;
                if (split == 2) {                                          // (uniform over the workgroup: ncol is)
                    if (on && h == 0 && lane == 63) s_half[arr] = v;
                    __syncthreads();
                    if (on && h == 1) base = s_half[arr];
                    __syncthreads();
                }
                double run = base + (v - tot);                             // sum of the pieces before this lane's
                for (int j = j0; j < j1; ++j) { run += d[j]; d[j] = run; }
            }
        }
        __syncthreads();
        // lwa and its store in one sweep (a thread reads only the two sums of its own cell: nothing to wait for in between)
#pragma unroll 2
        for (int i = tid; i < ncell; i += nthr) {
            const int y = i >> cshift, c = i & cmask;
            if (c < ncol) os[(size_t)y * nx + x0 + c] = s * ((Qs[y] - cref) * D0[(size_t)c * L + y] - D1[(size_t)c * L + y]);
        }
        __syncthreads();                                               // D0 is cleared at the top of the next group
        vb = vnext;
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
    if (variant != 0 && variant != 1) return fail(ctx, XC_EBADARG, "xc_lwa: variant must be 0 or 1");
    if (q_dtype != XC_F32 && q_dtype != XC_F64) return fail(ctx, XC_EBADARG, "xc_lwa: q_dtype must be XC_F32 or XC_F64");
    const unsigned* gate = nullptr;          // set: the exact kernels below run only if the device-side check FAILED
    unsigned epoch = 0;
    ctx->last_lwa_path = 0;
    // ctx->lwa_exact (xc_set_lwa_exact): 0 automatic (planes of more than kLwaFastMinRows rows take the interval kernel behind its
    // device-side check), 1 the band walk everywhere, 2 the interval kernel for every plane (checked), 3 the same with the premises
    // vouched for by the caller (it looked at Q and the coordinate on the host): ONE launch, no check, no gated band walk behind it
    const int mode = ctx->lwa_exact;
    const bool want_fast = mode >= 2 || (mode == 0 && ctx->knobs.lwa_fast && (ny > kLwaFastMinRows || ctx->knobs.lwa_fast > 1));
    if (variant == 0 && want_fast && ny * nx < ((int64_t)1 << 29)) {           // (32-bit byte offsets inside a plane: k_lwa_fast)
        int CG = 0;
        for (int c : {4, 2, 1})
            if (!CG && (size_t)(1 + 2 * c) * (ny + 1) * 8 + (size_t)(LWA_NB + 1) * 4 <= kLdsBudget) CG = c;
        if (CG) {
            unsigned* flag = nullptr;
            if (mode != 3) {
                if (!ctx->lwa_flag) { XC_HIP(ctx, hipMalloc((void**)&ctx->lwa_flag, 256)); XC_HIP(ctx, hipMemset(ctx->lwa_flag, 0, 256)); ctx->lwa_epoch = 0; }
                flag = ctx->lwa_flag;
                epoch = ++ctx->lwa_epoch;        // a failed check stamps the word with its call's epoch: no memset per call
                hipLaunchKernelGGL(k_lwa_check, dim3((unsigned)nslab), dim3(256), 0, ctx->stream, Q, coord, (int)ny, increase, flag, epoch);
                XC_HIP(ctx, hipGetLastError());
            }
            const size_t lds = (size_t)(1 + 2 * CG) * (ny + 1) * 8 + (size_t)(LWA_NB + 1) * 4;
            // part (core.py:773-784): 'upper' keeps mask3 > 0 (the near side) if increase else mask3 < 0 (the far side)
            const int side = part == 0 ? 0 : (((part == 1) == (increase != 0)) ? 1 : 2);
            const int64_t ngrp = (nx + CG - 1) / CG, gq = 8 * (16 / CG);                 // (XCD-aware group order: k_lwa_fast)
            const int64_t nvb = ((ngrp + gq - 1) / gq) * gq;                                // virtual blocks: the groups, padded to whole lines per XCD
            // persistent: one workgroup per CU and slab at most (the LDS holds one), a multiple of 8 so that a workgroup keeps its XCD; a stack
            // of slabs fills the chip with its first slabs and the rest queue behind them
            int64_t pw = ((ctx->cus > 0 ? ctx->cus : 256) / 8) * 8;
            if (pw < 8) pw = 8;
            const dim3 grid((unsigned)(nvb < pw ? nvb : pw), (unsigned)nslab);
#define XC_LWAF(T) do { \
                const int rc = ensure_big_lds(ctx, reinterpret_cast<const void*>(k_lwa_fast<T>), (int)kLdsBudget + 4096); if (rc != XC_OK) return rc; \
                hipLaunchKernelGGL((k_lwa_fast<T>), grid, dim3(1024), lds, ctx->stream, (const T*)q, Q, dA, dA_rank, dA_max, \
                                   M, M_rank, (int)ny, nx, increase, side, CG, nvb, out_lwa, flag, epoch); } while (0)
            if (q_dtype == XC_F64) XC_LWAF(double); else XC_LWAF(float);
#undef XC_LWAF
            XC_HIP(ctx, hipGetLastError());
            if (mode == 3) { ctx->last_lwa_path = 1; goto masks; }           // vouched for: nothing else to enqueue
            gate = flag;
            ctx->last_lwa_path = -1;         // decided on the device: xc_last_lwa_path reads the flag
        }
    }
    {
        // ---- one launch with the 64-column strip of the tracer in LDS when it fits
        const double* Mt = M_rank == XC_DA_NONE ? dA : M;
        const int Mr = M_rank == XC_DA_NONE ? dA_rank : M_rank;
        const size_t tsz = q_dtype == XC_F32 ? 4 : 8;
        const bool wpl = dA_rank == XC_DA_PLANE, mpl = Mr == XC_DA_PLANE;
        int wchunk = 0;
        for (int c : {64, 32, 16})
            if (!wchunk && lwa_strip_lds(ny, tsz, wpl, mpl, c) <= kLdsBudget) wchunk = c;
        if (wchunk && ctx->knobs.lwa_strip && ny <= 0x7fff && nx <= 0x7fffffff / ny) {
            const int64_t nstrip = (nx + 63) / 64;
            const int64_t nb16 = (ny + LWA_SW - 1) / LWA_SW;                  // workgroups per strip with LWA_SW targets each
            // ~2 workgroups per CU; more target rows per workgroup when there is more work than that (the strip is staged once per workgroup)
            // measured on MI355X: this kernel wins while its grid does not fill the chip twice (cfg3 alone: 13 us against 5 + 19 for
            // prologue + streaming kernel); stacks that do are VALU-bound either way and the streaming kernel's four targets per
            // thread win (64 slabs: 226 against 272 us)
            const bool few = nstrip * nslab * nb16 <= 2 * (ctx->cus > 0 ? ctx->cus : 256) || ctx->knobs.lwa_strip > 1;
            const int64_t tper = LWA_SW, ts = nb16;
            if (few && nstrip <= 0x7fffffff && ts <= 65535 && nslab <= 65535) {
                const size_t lds = lwa_strip_lds(ny, tsz, wpl, mpl, wchunk);
                const dim3 grid((unsigned)nstrip, (unsigned)ts, (unsigned)nslab);
#define XC_LWAS(T, V) do { \
                    const int rc = ensure_big_lds(ctx, reinterpret_cast<const void*>(k_lwa_strip<T, V>), (int)kLdsBudget + 4096); if (rc != XC_OK) return rc; \
                    hipLaunchKernelGGL((k_lwa_strip<T, V>), grid, dim3(64 * LWA_SW), lds, ctx->stream, (const T*)q, Q, coord, dA, dA_rank, dA_max, \
                                       Mt, Mr, ny, nx, increase, part, (int)tper, wchunk, out_lwa, gate, epoch); } while (0)
                if (q_dtype == XC_F64) { if (variant) XC_LWAS(double, true); else XC_LWAS(double, false); }
                else { if (variant) XC_LWAS(float, true); else XC_LWAS(float, false); }
#undef XC_LWAS
                XC_HIP(ctx, hipGetLastError());
                goto masks;
            }
        }
    }
    {
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
        hipLaunchKernelGGL(k_lwa_prep<double>, gp, dim3(256), 0, ctx->stream, (const double*)q, Q, coord, dA, dA_rank, dA_max, ny, nx, nstrip, wei, rowinfo, stripmm, gate, epoch);
    else if (q_dtype == XC_F32)
        hipLaunchKernelGGL(k_lwa_prep<float>, gp, dim3(256), 0, ctx->stream, (const float*)q, Q, coord, dA, dA_rank, dA_max, ny, nx, nstrip, wei, rowinfo, stripmm, gate, epoch);
    else return fail(ctx, XC_EBADARG, "xc_lwa: q_dtype must be XC_F32 or XC_F64");
    if (M_rank == XC_DA_NONE) { M = dA; M_rank = dA_rank; }
    const bool small = (double)ny * (double)ny * (double)nx * (double)nslab < 2.0e8;
    const int jt = small ? 1 : 4;
    dim3 grid((unsigned)((nx + 63) / 64), (unsigned)((ny + 4 * jt - 1) / (4 * jt)), (unsigned)nslab);
#define XC_LWA2(T, V, J) hipLaunchKernelGGL((k_lwa<T, V, J>), grid, dim3(256), 0, ctx->stream, (const T*)q, Q, coord, wei, dA_rank, \
                           M, M_rank, rowinfo, stripmm, ny, nx, increase, part, out_lwa, gate, epoch)
#define XC_LWA(T, V) do { if (small) XC_LWA2(T, V, 1); else XC_LWA2(T, V, 4); } while (0)
    if (q_dtype == XC_F64) { if (variant) XC_LWA(double, true); else XC_LWA(double, false); }
    else { if (variant) XC_LWA(float, true); else XC_LWA(float, false); }
#undef XC_LWA
#undef XC_LWA2
    XC_HIP(ctx, hipGetLastError());
    }
masks:
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
