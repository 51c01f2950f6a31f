// K3 kernel body + its launcher template, shared by xc_hist.hip (the float64-atomics variants) and xc_hist_det.hip (the
// order-free fixed-point variant, DET = 3).  Included INSIDE `namespace xc { namespace {` of each translation unit.
#pragma once

#ifndef XC_U
#define XC_U 2
#endif
// Cache policy of the two tracer streams (bit 0: the binned batch, bit 1: the NEXT batch whose min / max rides along).
// Measured on MI355X (bench.py, 64 slabs per launch): the non-temporal hint on the NEXT stream -- read once, used for
// two compares, never needed again by this launch -- keeps the shared dA plane in the XCD L2s: 1.367 -> 1.285 ms per
// launch; the same hint on the binned stream costs 9 % (its halo / neighbour re-reads want the line), on both 7 %.
#ifndef XC_E32_NBUF
#define XC_E32_NBUF 2
#endif
#ifndef XC_HIST_QNT
#define XC_HIST_QNT 2
#endif
template <int VEC> struct RowsPerBatch { static constexpr int value = VEC >= 4 ? 1 : XC_U; };   // rows per prefetch batch (double-buffered): the same bytes in flight

// Grid mapping (1-D grid).  Workgroups go round-robin to the 8 XCDs (workgroup id % 8), each XCD has its own
// L2, and every slab of a launch reads the SAME dA rows in its block `bx`.  XCD-aware order (default): XCD x
// takes the row groups bx = x, x+8, ... and runs each for ALL slabs back to back -- with one 1024-thread block
// per CU the 32 CUs of an XCD hold one row group of 32 slabs at a time, so that part of the dA plane is
// fetched into that L2 once per launch instead of once per XCD per slab group.  Plain order (xcd_map == 0):
// slab fastest, blocks of one row group spread over all XCDs.
#define XC_NBLK (a.bps)

#include "xc_binning.h"

// TB: the type the tracer values are KEPT in while a row waits in its buffer: double (converted by the load), or float for
// the E32 variant (raw float32: half the registers per buffered row, converted when the row is consumed)
template <int VEC, int NINT, typename TB = double>
struct RowBuf {
    TB     q[VEC];          // GRAD: row (centre+1); else: the centre row itself
    TB     h;               // GRAD: lane 0 = left halo of that row, lane 63 = right halo
    double dA[VEC];
    double in[NINT > 0 ? NINT : 1][VEC];
    TB     qn[VEC];         // NEXT: the same cells of the next batch (min/max by-product)
};

// DA2D: dA is a [ny][nx] plane (vector loads); otherwise one value per row (scalar).
// NEXT: also stream the same cells of a.q_next and emit its per-block min/max partials.
// FAST (Keff layout only): periodic X and dA verified finite and >= 0 are COMPILE-time facts -- the wall selects and
// the fillna selects vanish from the row body.
// DET (deterministic sums, xc_hist_det.hip): 0 = float64 LDS atomics (sums depend on the order in which waves reach the
// LDS: last bits vary from run to run); 3 = ONE pass into a fixed-point superaccumulator per (bin, channel): every weight is
// rounded once to 49 bits and added as two integer chunks with ds_add_u64 to limbs on a fixed grid (xc_binning.h) -- integer
// addition is associative, so the per-bin sums do not depend on the order of arrival, the block geometry or the number of
// slabs per launch.  (Rounds 3-4: two passes, DET = 1 / 2 -- a per-bin scale needed the bin's maximum first.)
// E32 (float32 tracer AND float32 contour levels, the Keff FAST layout; round 4): everything that is exact in float32 stays in
// float32 -- the bin search (float32 values against float32 edges: the comparison np.digitize makes, so the counts are the same
// bits; the nearest-edge guess in float32 is good to ~1e-4 of a bin, the decision is the exact compare), the min / max of the
// NEXT stream, the rows waiting in their buffers.  The squared gradient and the weights stay float64 (oracle.grad2_sphere
// fixes that arithmetic).  Measured (bench.py --dtype f32, 64 slabs per launch): chained 0.809 -> 0.760 ms, unchained K3 0.699 ->
// 0.682.  A THIRD row of loads in flight with the freed registers (XC_E32_NBUF = 3) does not pay: unchained 0.684 (the kernel
// is not short of bytes in flight), chained 1.185 (45 spilled registers).
template <typename TQ, int VEC, int NINT, bool GRAD, bool DA2D, bool NEXT, bool FAST = false, int DET = 0, bool E32 = false>
__global__ __launch_bounds__(kHistThreads)
void k_hist(const HistArgs a)
{
    constexpr int NCH = 1 + NINT + (GRAD ? 1 : 0);
    constexpr int W = 64 * VEC;
    constexpr int U = RowsPerBatch<VEC>::value;
    constexpr int NBUF = E32 ? XC_E32_NBUF : 2;                            // ring of row batches: NBUF - 1 of them in flight while one is processed
    using TB = typename std::conditional<E32, float, double>::type;
    static_assert(!E32 || (std::is_same<TQ, float>::value && GRAD && NINT == 0 && FAST && DET == 0), "E32: float32 Keff layout only");
    extern __shared__ __align__(16) double smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction
    const int nwave = blockDim.x >> 6;
    int slab, bx;
    {
        const int L = (int)blockIdx.x;
        if (a.xcd_map) { const int j = L >> 3; slab = j % a.nslab_grid; bx = (L & 7) + 8 * (j / a.nslab_grid); }
        else { slab = L % a.nslab_grid; bx = L / a.nslab_grid; }
        if (bx >= a.bps) return;                                  // padding blocks of the XCD-aware order
    }
    const int nbx = XC_NBLK;
    const int N = a.nbin;
    const int ncopy = a.ncopy;
    const int epad = (N + 2) & ~1;
    double*   s_red   = smem;                                   // 64 doubles
    double*   s_edges = smem + 64;                              // N+1
    // one CELL per (bin, copy): the NCH sums and the count side by side (count = low word of the last slot), so a cell's
    // three adds share ONE address computation; bin N is a trash bin -- NaN, out-of-range and inactive cells add there
    // unconditionally instead of branching around the adds (round 2: 86 -> 7x VALU instructions per 128-cell wave-row)
    // DET == 3: a cell is, per channel, the limbs of its superaccumulator and a trash word (a chunk that falls below the window), then
    // the count word (low half: count, high half: one bit per channel that saw a non-finite weight)
    constexpr int CW = DET == 3 ? kDetWords * NCH + 1 : NCH + 1;
    double*   s_cell  = s_edges + epad;                         // [(N + 1) * ncopy][CW]
    const int hsz = (N + 1) * ncopy;

    // ------------------------------------------------------------------ this wave's share of (strip,row) pairs
    const int ny = (int)a.ny, nx = (int)a.nx;                    // host guarantees < 2^31
    const int64_t total = (int64_t)a.nstrip * ny;
    const int64_t nw = (int64_t)nbx * nwave;
    const int64_t wg = (int64_t)bx * nwave + wave;
    int64_t g0 = total * wg / nw;
    int64_t g1 = total * (wg + 1) / nw;
    if (a.nchunk > 0) {
        // Strip-fastest order: the waves of a workgroup sweep the SAME rows of ADJACENT strips side by side, so the halo
        // column a wave needs (one 128-byte line per row and side for 8 bytes) is a line its neighbour wave streams
        // at the same moment -- an L2 hit instead of a fabric fetch (measured: ~1 GB of the 7.7 GB per 64-slab launch).
        const int strip = (int)(wg % a.nstrip), chunk = (int)(wg / a.nstrip);
        if (chunk < a.nchunk) {
            g0 = (int64_t)strip * ny + (int64_t)ny * chunk / a.nchunk;
            g1 = (int64_t)strip * ny + (int64_t)ny * (chunk + 1) / a.nchunk;
        } else { g0 = 0; g1 = 0; }
    }

    const size_t slab_off = (size_t)slab * (size_t)ny * (size_t)nx;
    const size_t rowq = (size_t)nx * sizeof(TQ), rowd = (size_t)nx * sizeof(double);
    const TQ* __restrict__ qs = reinterpret_cast<const TQ*>(a.q) + slab_off;
    const TQ* __restrict__ qnx = NEXT ? reinterpret_cast<const TQ*>(a.q_next) + slab_off : nullptr;
    TB nmn = (TB)dinf(), nmx = (TB)-dinf();
    const double* __restrict__ dAp = (DA2D && a.dA_rank == XC_DA_SLAB) ? a.dA + slab_off : a.dA;
    const double* __restrict__ rdxp = a.rdx;
    const double* __restrict__ rdyp = a.rdy;
    const int copy = lane & (ncopy - 1);
    const int cshift = __builtin_ctz((unsigned)ncopy);            // ncopy is a power of two
    const unsigned cstride = (unsigned)(CW * 8) << cshift, cbase = (unsigned)copy * (unsigned)(CW * 8);
    const int periodic_x = FAST ? 1 : a.periodic_x;

    // segment state (wave-uniform scalars + per-lane 32-bit byte offsets inside a row)
    int y0 = 0, y1 = 0;
    unsigned xo_q = 0, xo_h = 0, xo_d = 0, xo_f = 0;
    bool active = false, full_strip = true;
    unsigned long long inactive_mask = 0ull;                      // lanes beyond the row end (ragged last strip)
    int rlane = 63;
    double fx[VEC];

    auto begin_segment = [&]() {
        const int     s  = (int)(g0 / ny);
        y0 = (int)(g0 - (int64_t)s * ny);
        const int64_t rem = g1 - g0;
        y1 = (y0 + rem < ny) ? (int)(y0 + rem) : ny;
        g0 += (y1 - y0);
        const int x0 = s * W;
        const int x  = x0 + lane * VEC;
        active = x < nx;
        const int xend = (x0 + W < nx) ? x0 + W : nx;
        rlane = (xend - x0) / VEC - 1;                                 // lane holding the strip's last valid cell
        full_strip = rlane == 63;
        inactive_mask = full_strip ? 0ull : (~0ull << (rlane + 1));
        // inactive lanes load from a clamped (valid) address; in a ragged last strip of a periodic
        // domain the first inactive lane loads columns 0.. so that its cell 0 IS the right halo
        const int xld = active ? x : ((periodic_x && lane == rlane + 1) ? 0 : nx - VEC);
        const int xl = (x0 == 0) ? (periodic_x ? nx - 1 : 0) : x0 - 1;
        const int xr = (xend == nx) ? (periodic_x ? 0 : nx - 1) : xend;
        const int xh = (lane == 0) ? xl : ((lane == 63) ? xr : xld);   // halo column (lanes 0 / 63 matter)
        xo_q = (unsigned)xld * (unsigned)sizeof(TQ); xo_h = (unsigned)xh * (unsigned)sizeof(TQ);
        xo_d = (unsigned)xld * 8u; xo_f = (unsigned)xld * 4u;
        // one-sided x differences at the walls of a non-periodic domain use spacing dx, not 2dx
#pragma unroll
        for (int c = 0; c < VEC; ++c)
            fx[c] = (!periodic_x && (x + c == 0 || x + c == nx - 1)) ? 2.0 : 1.0;
    };

    // per-row gradient metrics, lane-distributed: lane i holds row ymet0 + i (64 rows per refill),
    // read back with v_readlane -- 4 VGPRs per wave instead of 4 per buffered row
    double rdxv = 0.0, rdyv = 0.0;
    int ymet0 = 0;
    auto load_metrics = [&](int yfirst) {
        ymet0 = yfirst;
        const int y = (yfirst + lane < ny) ? yfirst + lane : ny - 1;
        rdxv = rdxp[y]; rdyv = rdyp[y];
    };

    // branch-free loads of one row: q row `yq` (+ halo), weights of row `yw`
    const char* qbase = reinterpret_cast<const char*>(qs);
    const char* nbase = reinterpret_cast<const char*>(qnx);
    const char* dbase = reinterpret_cast<const char*>(dAp);
    auto load_row = [&](RowBuf<VEC, NINT, TB>& r, int yq, int yw) {
        yq = yq < ny - 1 ? yq : ny - 1;
        yw = yw < ny - 1 ? yw : ny - 1;
        const char* qrow = qbase + (size_t)yq * rowq;                          // wave-uniform
#if (XC_HIST_QNT & 1)
        RowLoadNT<TQ, VEC>::ld(reinterpret_cast<const TQ*>(qrow + xo_q), r.q);
#else
        RowLoad<TQ, VEC>::ld(reinterpret_cast<const TQ*>(qrow + xo_q), r.q);
#endif
        if (GRAD) r.h = (TB)*reinterpret_cast<const TQ*>(qrow + xo_h);
#if (XC_HIST_QNT & 2)
        if (NEXT) RowLoadNT<TQ, VEC>::ld(reinterpret_cast<const TQ*>(nbase + (size_t)yw * rowq + xo_q), r.qn);
#else
        if (NEXT) RowLoad<TQ, VEC>::ld(reinterpret_cast<const TQ*>(nbase + (size_t)yw * rowq + xo_q), r.qn);
#endif
        if (DA2D) {
            RowLoad<double, VEC>::ld(reinterpret_cast<const double*>(dbase + (size_t)yw * rowd + xo_d), r.dA);
        } else {
            const double v = dAp[yw];
#pragma unroll
            for (int c = 0; c < VEC; ++c) r.dA[c] = v;
        }
#pragma unroll
        for (int i = 0; i < NINT; ++i) {
            if (a.integ_f32[i])
                RowLoad<float, VEC>::ld(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.integ[i]) +
                                        (slab_off + (size_t)yw * nx) * 4 + xo_f), r.in[i]);
            else
                RowLoad<double, VEC>::ld(reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.integ[i]) +
                                         (slab_off + (size_t)yw * nx) * 8 + xo_d), r.in[i]);
        }
    };
    auto load_batch = [&](RowBuf<VEC, NINT, TB> (&L)[U], int yb) {
#pragma unroll
        for (int i = 0; i < U; ++i) load_row(L[i], GRAD ? yb + i + 1 : yb + i, yb + i);
    };

    // issue the first loads of this wave BEFORE the edge prologue so that their HBM latency
    // overlaps the min/max reduction and the barriers below
    RowBuf<VEC, NINT, TB> R[NBUF][U];
    double qm[VEC], qcur[VEC];
    TB hcur = (TB)0, qc32[E32 ? VEC : 1];                        // E32: the centre row once more, raw, for the bin search
    // first rows of a segment: the two rows above the sweep (gradient window) and NBUF - 1 batches in flight
    auto prime = [&]() {
        begin_segment();
        if (GRAD) {
            load_metrics(y0);
            RowBuf<VEC, NINT, TB> t;
            load_row(t, y0 > 0 ? y0 - 1 : 0, y0);
#pragma unroll
            for (int c = 0; c < VEC; ++c) qm[c] = (double)t.q[c];
            load_row(t, y0, y0);
#pragma unroll
            for (int c = 0; c < VEC; ++c) { qcur[c] = (double)t.q[c]; if (E32) qc32[c] = t.q[c]; }
            hcur = t.h;
        }
#pragma unroll
        for (int b = 0; b < NBUF - 1; ++b)
            if (y0 + b * U < y1) load_batch(R[b], y0 + b * U);
    };
    bool have = g0 < g1;
    if (have) prime();

    for (int i = tid; i < CW * hsz; i += blockDim.x) s_cell[i] = 0.0;

    // ------------------------------------------------------------------ edges -> LDS
    int* s_c0 = reinterpret_cast<int*>(s_red + 56);             // DET == 3: the window constant of every channel (s_red[0 .. 47] hold the reductions)
    double det_rng = 0.0, det_rdm = 0.0;                          // DET == 3: tracer range and largest gradient metric of this slab
    if (DET == 3 && GRAD) {
        // the largest factor a tracer difference can meet in the stencil: max over the rows of |rdx| (x 2 at the walls of a
        // non-periodic domain) and |rdy|; fixed order of evaluation does not matter for a maximum
        double m = 0.0;
        for (int y = tid; y < ny; y += blockDim.x) {
            const double rx = fabs(rdxp[y]) * (periodic_x ? 1.0 : 2.0);
            m = fmax(m, fmax(rx, fabs(rdyp[y])));
        }
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        if (lane == 0) s_red[32 + wave] = m;
    }
    if (a.levels_mode) {
        // reduce the per-block partial min/max of K1 (fixed order: deterministic)
        const double* mp = a.mmpart + (size_t)slab * a.P * 2;
        double mn = dinf(), mx = -dinf();
        for (int i = tid; i < a.P; i += blockDim.x) { mn = fmin(mn, mp[2 * i]); mx = fmax(mx, mp[2 * i + 1]); }
        for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
        if (lane == 0) { s_red[2 * wave] = mn; s_red[2 * wave + 1] = mx; }
        __syncthreads();
        mn = s_red[0]; mx = s_red[1];
        for (int w = 1; w < nwave; ++w) { mn = fmin(mn, s_red[2 * w]); mx = fmax(mx, s_red[2 * w + 1]); }
        if (mn == dinf() && mx == -dinf()) { mn = dnan(); mx = dnan(); }     // all-NaN slab
        if (DET == 3 && GRAD) det_rng = a.q_f32 ? fabs((double)__fsub_rn((float)mx, (float)mn)) : fabs(__dsub_rn(mx, mn));
        for (int k = tid; k < N; k += blockDim.x) {
            const double c = level_value(mn, mx, k, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1);
            s_edges[a.increase ? k + 1 : N - k] = c;
            if (bx == 0 && a.ctr_out) a.ctr_out[(size_t)slab * a.ctr_stride + k] = c;
        }
        __syncthreads();
        if (tid == 0) {
            const double lo = s_edges[1], hi = s_edges[N];
            s_edges[0] = dummy_edge(lo, hi, N, a.ctr_f32);
            if (a.right_edge == XC_EDGE_XHISTOGRAM) s_edges[N] = bump_last_edge(hi, a.ctr_f32);
        }
        if (bx == 0 && a.status) {
            // reference raises 'non monotonic bins' when two adjacent levels coincide (core.py:1233)
            int bad = 0;
            for (int k = tid + 1; k < N; k += blockDim.x) bad |= (s_edges[k] == s_edges[k + 1]);
            bad = __syncthreads_or(bad);
            if (tid == 0) a.status[slab] = bad ? 1 : 0;
        }
        __syncthreads();
        if (bx == 0 && a.edges_out)
            for (int k = tid; k <= N; k += blockDim.x) a.edges_out[(size_t)slab * (N + 1) + k] = s_edges[k];
    } else {
        const double* e = a.edges + (a.edges_per_slab ? (size_t)slab * (N + 1) : 0);
        for (int k = tid; k <= N; k += blockDim.x) s_edges[k] = e[k];
        __syncthreads();
        if (DET == 3 && GRAD) {                                   // explicit edges: the tracer's extrema come from a K1 pass of their own
            const double mn = a.det_q_mm[2 * (size_t)slab], mx = a.det_q_mm[2 * (size_t)slab + 1];
            det_rng = a.q_f32 ? fabs((double)__fsub_rn((float)mx, (float)mn)) : fabs(__dsub_rn(mx, mn));
            if (!(det_rng == det_rng)) det_rng = 0.0;
        }
    }
    if (DET == 3) {
        // the window of every channel's accumulator, from bounds known before the pass (see xc_binning.h); every block of the slab
        // derives the same constants from the same inputs
        if (GRAD) { det_rdm = s_red[32]; for (int w = 1; w < nwave; ++w) det_rdm = fmax(det_rdm, s_red[32 + w]); }
        if (tid == 0) {
            double dmax = a.det_dA_max;
            if (!(dmax >= 0.0)) {
                const double* mm = a.det_dA_max_dev + 2 * (size_t)slab * a.det_dA_stride;
                dmax = fmax(fabs(mm[0]), fabs(mm[1]));
                if (!(dmax == dmax)) dmax = 0.0;                // no finite weight at all (the extrema skip NaN and +-inf)
            }
            s_c0[0] = det_c0_from_bound(dmax);
#pragma unroll
            for (int i = 0; i < NINT; ++i) {
                const double* mm = a.det_int_mm[i] + 2 * (size_t)slab;
                double imax = fmax(fabs(mm[0]), fabs(mm[1]));
                if (!(imax == imax)) imax = 0.0;
                double bnd = __dmul_rn(imax, dmax);
                if (a.prod_f32) bnd = __dmul_rn(bnd, 1.0000002384185791);        // (the float32 product may round up past the float64 one)
                s_c0[1 + i] = det_c0_from_bound(bnd);
            }
            if (GRAD) {
                const double B = __dmul_rn(det_rng, det_rdm);
                s_c0[NCH - 1] = det_c0_from_bound(__dmul_rn(__dmul_rn(__dmul_rn(B, B), 2.0), dmax));
            }
            if (bx == 0 && a.det_c0_out)
                for (int ch = 0; ch < NCH; ++ch) a.det_c0_out[(size_t)slab * NCH + ch] = s_c0[ch];
        }
        __syncthreads();
    }
    int c0r[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) c0r[ch] = DET == 3 ? __builtin_amdgcn_readfirstlane(s_c0[ch]) : 0;
    const double e0 = s_edges[0], eN = s_edges[N];
    const double inv = (double)N / (eN - e0);
    // E32: a float32 copy of the edges (they ARE float32 values: ctr_f32) behind the cells, for 4-byte reads
    float* s_e32 = reinterpret_cast<float*>(s_cell + (size_t)CW * hsz);
    const float e0f = (float)e0, invf = (float)inv;
    if (E32) { for (int k = tid; k <= N; k += blockDim.x) s_e32[k] = (float)s_edges[k]; }
    const int last_closed = FAST ? 0 : a.last_closed;           // FAST: the half-open (xhistogram) rule is a compile-time fact
    // Are the edges equally spaced to a quarter of a bin?  Then the NEAREST edge j = floor((v - e0) / h + 1/2) brackets v
    // between e[j-1] and e[j+1], and ONE exact comparison against e[j] gives np.digitize's answer (one 8-byte LDS read
    // and one compare per cell instead of two reads, two compares and a call on a miss).  Checked per slab, wave-uniform.
    bool uni;
    {
        const double hstep = (eN - e0) / (double)N;
        int bad = 0;
        for (int k = tid; k <= N; k += blockDim.x) bad |= !(fabs(s_edges[k] - (e0 + (double)k * hstep)) <= 0.25 * hstep);
        uni = !__syncthreads_or(bad);
    }
    const int negate = a.negate;
    const bool want_cnt = DET != 0 || a.part_c != nullptr;         // counts are an OUTPUT only: the float64-atomics pass skips their adds when none is wanted
    const bool wpos = FAST ? true : (a.dA_pos_finite != 0);
    double   acc[NCH];
    unsigned cnt = 0;
    int      cur = -1;                   // wave-uniform: bin of the register accumulators
    int      fp_skip = 0, fp_miss = 0;   // wave-uniform: rows left without the one-bin test / consecutive rows that failed it
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = 0.0;

    auto flush = [&]() {
        if (cnt) {
            double* cp = reinterpret_cast<double*>(reinterpret_cast<char*>(s_cell) + (__umul24((unsigned)cur, cstride) + cbase));
#pragma unroll
            for (int c = 0; c < NCH; ++c) lds_add(cp + c, acc[c]);
            if (want_cnt) lds_add(reinterpret_cast<unsigned*>(cp + NCH), cnt);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = 0.0;
        cnt = 0;
    };

    // one centre row: bins, weights, accumulate
    auto do_row = [&](const double (&qc)[VEC], const double (&qS)[VEC], const double (&qN)[VEC],
                      double hc, const double (&dAv)[VEC], const double (&inv_)[NINT > 0 ? NINT : 1][VEC],
                      double rdx, double rdy, const TB (&qraw)[E32 ? VEC : 1]) {
        unsigned k[VEC];                 // bin, or N (the trash bin) for a dropped cell
        double w[NCH][VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const double vb = (!GRAD && negate) ? -qc[c] : qc[c];
            int kb;
            if (E32) {
                // float32 value against float32 edges (only launched when the edges are equally spaced: xc_keff_dev checks
                // nothing -- `uni` is checked below and the non-uniform case takes the float64 search)
                const float vf = (float)qraw[E32 ? c : 0];
                if (uni) {
                    int j = (int)__builtin_fmaf(vf - e0f, invf, 0.5f);                  // NaN -> 0
                    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(j) : "v"(j), "s"(N));
                    kb = (vf >= s_e32[j]) ? j : j - 1;
                } else {
                    kb = find_bin(vb, s_edges, N, e0, eN, inv, last_closed);
                }
            } else
            if (uni) {
                int j = (int)__builtin_fma(vb - e0, inv, 0.5);                      // NaN -> 0
                asm("v_med3_i32 %0, %1, 0, %2" : "=v"(j) : "v"(j), "s"(N));         // clamp to [0, N]
                kb = (vb >= s_edges[j]) ? j : j - 1;                                // NaN -> -1; at or beyond the last edge -> N
                if (last_closed && vb == eN) kb = N - 1;
            } else {
                kb = find_bin(vb, s_edges, N, e0, eN, inv, last_closed);            // -1 when dropped
            }
            unsigned ku = (unsigned)kb < (unsigned)N ? (unsigned)kb : (unsigned)N;  // v_min_u32: -1 and N both land in the trash bin
            if (!full_strip) ku = active ? ku : (unsigned)N;                        // wave-uniform: selects only in a ragged strip
            k[c] = ku;
            const double dv = dAv[c];
            w[0][c] = wpos ? dv : ((dv != dv) ? 0.0 : dv);                    // fillna(0), core.py:449 (wpos: host checked dA finite)
#pragma unroll
            for (int i = 0; i < NINT; ++i) {
                double p = a.prod_f32 ? (double)__fmul_rn((float)inv_[i][c], (float)dv)
                                      : __dmul_rn(inv_[i][c], dv);            // integrand * dA, core.py:444
                w[1 + i][c] = (p != p) ? 0.0 : p;
            }
        }
        if (GRAD) {
            // x-neighbours: lane-1's last cell / lane+1's first cell; lane 0 keeps the left halo (hc of
            // lane 0), lane 63 the right halo (hc of lane 63); a ragged strip's right halo sits in the
            // first inactive lane's cell 0 (see begin_segment)
            const double fromL = lane_shift_keep<DPP_WAVE_SHR1>(qc[VEC - 1], hc);
            const double fromR = lane_shift_keep<DPP_WAVE_SHL1>(qc[0], hc);
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                double qW = (c == 0) ? fromL : qc[c > 0 ? c - 1 : 0];
                double qE = (c == VEC - 1) ? fromR : qc[c < VEC - 1 ? c + 1 : 0];
                if (!periodic_x) {                                        // wave-uniform branch: walls are one-sided
                    if (fx[c] == 2.0) { if (lane == 0 && c == 0) qW = qc[c]; else qE = qc[c]; }
                }
                double gx = __dmul_rn(__dsub_rn(qE, qW), rdx);
                if (!periodic_x) gx = __dmul_rn(gx, fx[c]);
                const double gy = __dmul_rn(__dsub_rn(qN[c], qS[c]), rdy);
                const double g2 = __dadd_rn(__dmul_rn(gx, gx), __dmul_rn(gy, gy));
                const double p = __dmul_rn(g2, dAv[c]);
                // NaN -> 0; with finite non-negative dA the product is >= 0 or NaN, so one v_max_f64 does it
                w[NCH - 1][c] = wpos ? fmax(p, 0.0) : ((p != p) ? 0.0 : p);
            }
        }
        // wave-uniform fast path: every valid cell of the row in one bin -> per-lane registers, no LDS traffic.  A field with
        // grid-scale noise never takes it, and the test itself (readfirstlane, compares, ballot) is a tenth of the row: after 8
        // consecutive failures the wave stops testing for 48 rows, then looks again (smooth fields never stop)
        // (round 5) ... and only in the chained (NEXT) kernels at all.  Ablation builds (profiles/r05_notes.md): no variant waits for its LDS adds
        // (removing every one of them changes nothing), so the register path saves nothing that matters, and carrying the test costs the
        // unchained float64 kernel 6 % on smooth and noisy fields alike (0.784 -> 0.734 / 0.797 -> 0.750 ms per 64 slabs); rows of land cells
        // piling onto the trash bin cost nothing measurable either (30 % land, unchained: 24.9 with the test, 24.1 us per slab without).  The
        // chained kernel keeps it: there it measured 1.4 % FASTER with the test (1.147 against 1.163 ms).
        constexpr bool ROWTEST = NEXT;
        bool one_bin = false, all_dropped = false;
        int rb = 0;
        if (ROWTEST && DET == 0 && fp_skip == 0) {       // (the order-free variants always go through the LDS)
            rb = __builtin_amdgcn_readfirstlane((int)k[0]);
            bool match = true;
#pragma unroll
            for (int c = 0; c < VEC; ++c) match = match && ((int)k[c] == rb);
            const bool same = (__ballot(match) | inactive_mask) == ~0ull;
            one_bin = rb < N && same;
            all_dropped = rb >= N && same;                // a row of NaN / out-of-range cells (land in ocean fields): nothing to add at all
            if (one_bin || all_dropped) fp_miss = 0;
            else if (++fp_miss >= 8) { fp_skip = 48; fp_miss = 7; }
        } else {
            --fp_skip;
        }
        if (all_dropped) {
            // (without this the 64 * VEC cells of such a row each did their LDS adds on the `ncopy` addresses of the trash bin)
        } else if (one_bin) {
            if (rb != cur) { flush(); cur = rb; }
            if (active) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) acc[ch] += w[ch][c];
                }
                cnt += VEC;
            }
        } else {
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                // byte offset of the cell = k * (8 CW ncopy) + copy * 8 CW: ONE full-rate v_mad_u32_u24 (k <= N and the stride are far below
                // 2^24).  (Round 6: it was shift-add, v_mul_lo_u32 by CW, add -- and a 32-bit v_mul_lo is a QUARTER-rate instruction on
                // CDNA: four issue slots per cell of a kernel that waits for its VALU work.)
                double* cp = reinterpret_cast<double*>(reinterpret_cast<char*>(s_cell) + (__umul24(k[c], cstride) + cbase));
                if (DET == 0) {
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) lds_add(cp + ch, w[ch][c]);
                    if (want_cnt) lds_add(reinterpret_cast<unsigned*>(cp + NCH), 1u);       // (wave-uniform: a third of the LDS atomics when nobody asked for counts)
                } else {
                    unsigned long long* cw = reinterpret_cast<unsigned long long*>(cp);
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        const double wv = w[ch][c];                             // (NaN went to 0 above)
                        unsigned long long hi, lo; int E;
                        const int jc = det_split(wv, c0r[ch], hi, lo, E);
                        if (__any(E == 0x7ff)) {                                // an infinite weight (wave-uniform, rare): the bin reports NaN
                            if (E == 0x7ff) atomicOr(reinterpret_cast<unsigned*>(cw + (CW - 1)) + 1, 1u << ch);
                        }
                        if (!wpos || (ch >= 1 && ch <= NINT)) {                 // weights that may be negative (a supplied integrand; dA not vouched for): two's complement chunks
                            const unsigned long long sm = (unsigned long long)(__double_as_longlong(wv) >> 63);
                            hi = (hi ^ sm) - sm; lo = (lo ^ sm) - sm;
                        }
                        unsigned long long* p = cw + (kDetWords * ch - 1) + jc;  // limb jc - 1, and the word behind it
                        lds_add(p, hi); lds_add(p + 1, lo);
                    }
                    lds_add(reinterpret_cast<unsigned*>(cw + (CW - 1)), 1u);
                }
            }
        }
    };

    // batch of U centre rows starting at yb; L holds (GRAD) q rows yb+1.. and weights rows yb..
    auto process_batch = [&](RowBuf<VEC, NINT, TB> (&L)[U], int yb) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (yb + i < y1) {
                if (NEXT && active) {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) {              // NaN never wins a compare: NaN-skipping
                        const TB v = L[i].qn[c];
                        if (E32) { nmn = fminf(nmn, v); nmx = fmaxf(nmx, v); }   // exact in float32
                        else { nmn = fmin(nmn, v); nmx = fmax(nmx, v); }         // fmin / fmax skip NaN
                    }
                }
                if (GRAD) {
                    if (yb + i - ymet0 >= 64) load_metrics(yb + i);               // wave-uniform, once per 64 rows
                    double qn64[VEC];
#pragma unroll
                    for (int c = 0; c < VEC; ++c) qn64[c] = (double)L[i].q[c];
                    do_row(qcur, qm, qn64, (double)hcur, L[i].dA, L[i].in,
                           lane_get(rdxv, yb + i - ymet0), lane_get(rdyv, yb + i - ymet0), qc32);
#pragma unroll
                    for (int c = 0; c < VEC; ++c) { qm[c] = qcur[c]; qcur[c] = qn64[c]; if (E32) qc32[c] = L[i].q[c]; }
                    hcur = L[i].h;
                } else {
                    if constexpr (!E32) do_row(L[i].q, L[i].q, L[i].q, 0.0, L[i].dA, L[i].in, 0.0, 0.0, qc32);
                }
            }
        }
    };

    while (have) {
        for (int yb = y0; yb < y1; yb += NBUF * U) {
#pragma unroll
            for (int b = 0; b < NBUF; ++b) {
                const int yy = yb + b * U;
                if (yy >= y1) break;                              // wave-uniform
                const int yl = yy + (NBUF - 1) * U;
                if (yl < y1) load_batch(R[(b + NBUF - 1) % NBUF], yl);
                process_batch(R[b], yy);
            }
        }
        have = g0 < g1;
        if (have) prime();                            // next segment (the range crossed a strip boundary)
    }
    flush();
    if (NEXT) {
        double dmn = (double)nmn, dmx = (double)nmx;
        for (int o = 32; o > 0; o >>= 1) { dmn = fmin(dmn, __shfl_xor(dmn, o)); dmx = fmax(dmx, __shfl_xor(dmx, o)); }
        if (lane == 0) { s_red[2 * wave] = dmn; s_red[2 * wave + 1] = dmx; }
    }
    __syncthreads();
    if (NEXT && tid == 0) {
        double dmn = s_red[0], dmx = s_red[1];
        for (int w = 1; w < nwave; ++w) { dmn = fmin(dmn, s_red[2 * w]); dmx = fmax(dmx, s_red[2 * w + 1]); }
        double* o = a.mm_next + ((size_t)slab * nbx + bx) * 2;
        o[0] = dmn; o[1] = dmx;
    }

    // ------------------------------------------------------------------ per-block partials (plain stores)
    const size_t pb = (size_t)slab * nbx + bx;
    if (DET == 3) {
        // one thread per (channel, bin): the limbs of the LDS copies added in copy order with carries, so that what leaves the block is
        // canonical -- every limb below the first in [0, 2^48), the first one signed -- and a few thousand blocks can be added in 64 bits
        constexpr int NL = det_total_limbs(NCH);
        unsigned long long* pl = reinterpret_cast<unsigned long long*>(a.part_h) + pb * NL * N;
        for (int i = tid; i < NCH * N; i += blockDim.x) {
            const int ch = i / N, b = i - ch * N;
            long long acc[kDetLimbsX] = {0, 0, 0, 0};
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(s_cell) + (size_t)b * ncopy * CW + kDetWords * ch;
            for (int c = 0; c < ncopy; ++c) {
                long long carry = 0;
#pragma unroll
                for (int l = kDetLimbsX - 1; l >= 0; --l) {
                    long long v = acc[l] + (long long)src[(size_t)c * CW + l] + carry;
                    carry = 0;
                    if (l > 0) { carry = v >> kDetLimbBits; v -= carry << kDetLimbBits; }
                    acc[l] = v;
                }
            }
#pragma unroll
            for (int l = 0; l < kDetLimbsX; ++l) pl[(size_t)(kDetLimbsX * ch + l) * N + b] = (unsigned long long)acc[l];
        }
        unsigned* pc3 = a.part_c + pb * N;
        for (int b = tid; b < N; b += blockDim.x) {
            unsigned sum = 0u, fl = 0u;
            for (int c = 0; c < ncopy; ++c) {
                const unsigned* p = reinterpret_cast<const unsigned*>(s_cell + (size_t)(b * ncopy + c) * CW + (CW - 1));
                sum += p[0]; fl |= p[1];
            }
            pc3[b] = sum | (fl << 28);                            // a block holds fewer than 2^28 cells: the flags ride in the top bits
        }
        return;
    }
    double* ph = a.part_h + pb * NCH * N;
    // few slabs in the launch (a.acc_h): hundreds of blocks share a slab, and summing their partials is a launch of its own in a chain
    // of four dependent ones -- the block adds what it has to the slab's accumulators instead (global float64 atomics, ~1 us for the
    // ~200 k adds of a cfg2 slab; a block meets a fraction of the bins, zeros are not sent)
    double* ah = a.acc_h ? a.acc_h + (size_t)slab * NCH * N : nullptr;
    // sum the lane-privatised copies; every thread starts at a rotated copy index so that the
    // 64 lanes of a wave hit distinct LDS banks (a fixed, thread-determined order)
    for (int i = tid; i < NCH * N; i += blockDim.x) {
        const int ch = i / N, b = i - ch * N;
        const double* src = s_cell + (size_t)b * ncopy * CW + ch;
        double sum = 0.0;
        for (int c = 0; c < ncopy; ++c) sum += src[(size_t)((c + tid) & (ncopy - 1)) * CW];
        if (ah) { if (sum != 0.0) atomicAdd(ah + i, sum); }
        else ph[i] = sum;
    }
    unsigned* pc = a.part_c + pb * N;
    if (a.part_c)
    for (int b = tid; b < N; b += blockDim.x) {
        unsigned sum = 0u;
        for (int c = 0; c < ncopy; ++c)
            sum += *reinterpret_cast<const unsigned*>(s_cell + (size_t)(b * ncopy + ((c + tid) & (ncopy - 1))) * CW + NCH);
        if (ah) { if (sum && a.acc_c) atomicAdd(a.acc_c + (size_t)slab * N + b, (unsigned long long)sum); }
        else pc[b] = sum;
    }
}


template <typename TQ, int VEC, int NINT, bool GRAD, bool DA2D, bool NEXT, bool FAST = false, int DET = 0, bool E32 = false>
int launch_three(xc_ctx* ctx, const HistGeom& g, int64_t nslab, const HistArgs& a)
{
    auto kern = k_hist<TQ, VEC, NINT, GRAD, DA2D, NEXT, FAST, DET, E32>;
    { const int rc = ensure_big_lds(ctx, reinterpret_cast<const void*>(kern), (int)kLdsBudget + 4096); if (rc != XC_OK) return rc; }
    HistArgs b = a;
    // the XCD-aware order needs whole groups of 8 row groups, otherwise it would leave XCDs idle (bps = 1 with many
    // small slabs would put every block on XCD 0): plain slab-fastest order then
    b.bps = g.bps; b.nslab_grid = (int)nslab; b.xcd_map = (ctx->knobs.xcd_map && g.bps % 8 == 0) ? 1 : 0;
    {
        const int64_t nw = (int64_t)g.bps * (g.threads / 64);
        const int64_t nchunk = nw / g.nstrip;
        // needs at least one chunk, at least 8 rows per chunk (two halo rows are loaded per chunk) and no more than a tenth of
        // the waves left without a chunk (waves beyond nchunk x nstrip idle; the even strip-major split uses them all)
        b.nchunk = (ctx->knobs.tile_map && nchunk >= 1 && a.ny / nchunk >= 8 && (nw - nchunk * g.nstrip) * 10 <= nw) ? (int)nchunk : 0;
    }
    const int64_t nblk = b.xcd_map ? (int64_t)8 * ((g.bps + 7) / 8) * nslab : (int64_t)g.bps * nslab;
    if (nblk > 0x7fffffff) return fail(ctx, XC_EBADARG, "xc_hist: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(g.threads), g.lds, ctx->stream, b);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

