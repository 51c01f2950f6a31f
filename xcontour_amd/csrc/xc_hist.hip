// K3 -- one-pass, multi-channel weighted histogram of tracer slabs (gfx950).
//
// Replaces xhistogram.xarray.histogram(var, bins=[edges], dim=dims, weights=w)
// (reference core.py:1284, 1307) for ALL weight channels of a Keff step at once,
// with |grad q|^2 optionally computed in-kernel from a rolling 3-row register
// window (no stand-alone gradient field is ever written).
//
// Layout / mapping (see DESIGN.md "K3"):
//   * a slab is [ny][nx], X fastest.  It is cut into column strips of W = 64*VEC
//     cells (VEC cells per lane, one 8/16-byte load per lane per row: fully
//     coalesced 512 B / 1 KiB per wave-instruction).
//   * a wave sweeps a chunk of consecutive rows top-to-bottom inside a strip, so
//     y-neighbours live in registers and x-neighbours come from the adjacent lane
//     (DPP) plus one halo load per row.  Wave w of a slab takes strip w % nstrip of
//     row chunk w / nstrip: the waves of a workgroup sweep the same rows of adjacent
//     strips side by side and a halo line is a line the neighbour wave streams at
//     the same moment (an L2 hit).  Fallback when a slab has fewer waves than
//     strips: the (strip, row) pairs linearised strip-major and divided evenly.
//   * bin search: nearest-edge guess + ONE exact comparison against the f64 edge
//     in LDS when the edges are equally spaced (checked per slab), else a bracket
//     test + fix-up (exactly np.digitize semantics, any ascending edges).
//   * accumulation: rows whose 64*VEC cells all fall in one bin (the common case
//     on smooth geophysical fields) accumulate in per-lane registers with no
//     cross-lane traffic; other rows use LDS atomics on `ncopy` lane-privatised
//     copies of one CELL per bin (sums and count side by side; bin N is a trash bin
//     for dropped cells).  Per-block partials are written with plain stores and
//     reduced in fixed order by k_reduce_partials: no global atomics.
#include "xc_internal.h"
#include <stdlib.h>

namespace xc {

namespace {

#ifndef XC_U
#define XC_U 2
#endif
// Cache policy of the two tracer streams (bit 0: the binned batch, bit 1: the NEXT batch whose min / max rides along).
// Measured on MI355X (bench.py, 64 slabs per launch): the non-temporal hint on the NEXT stream -- read once, used for
// two compares, never needed again by this launch -- keeps the shared dA plane in the XCD L2s: 1.367 -> 1.285 ms per
// launch; the same hint on the binned stream costs 9 % (its halo / neighbour re-reads want the line), on both 7 %.
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

// Diagnostic build only (-DXC_STAMPS): wave 0 / lane 0 of every block stores s_memrealtime
// (100 MHz) at phase boundaries into a buffer nothing else reads.  Never in the shipped .so.
#ifdef XC_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define XC_STAMP(i) do { if (g_stamps && tid == 0) g_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define XC_STAMP(i) do {} while (0)
#endif

#include "xc_binning.h"

template <int VEC, int NINT>
struct RowBuf {
    double q[VEC];          // GRAD: row (centre+1); else: the centre row itself
    double h;               // GRAD: lane 0 = left halo of that row, lane 63 = right halo
    double dA[VEC];
    double in[NINT > 0 ? NINT : 1][VEC];
    double qn[VEC];         // NEXT: the same cells of the next batch (min/max by-product)
};

// DA2D: dA is a [ny][nx] plane (vector loads); otherwise one value per row (scalar).
// NEXT: also stream the same cells of a.q_next and emit its per-block min/max partials.
// FAST (Keff layout only): periodic X and dA verified finite and >= 0 are COMPILE-time facts -- the wall selects and
// the fillna selects vanish from the row body.
template <typename TQ, int VEC, int NINT, bool GRAD, bool DA2D, bool NEXT, bool FAST = false>
__global__ __launch_bounds__(kHistThreads)
void k_hist(const HistArgs a)
{
    constexpr int NCH = 1 + NINT + (GRAD ? 1 : 0);
    constexpr int W = 64 * VEC;
    constexpr int U = RowsPerBatch<VEC>::value;
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
    constexpr int CW = NCH + 1;
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
    double nmn = dinf(), nmx = -dinf();
    const double* __restrict__ dAp = (DA2D && a.dA_rank == XC_DA_SLAB) ? a.dA + slab_off : a.dA;
    const double* __restrict__ rdxp = a.rdx;
    const double* __restrict__ rdyp = a.rdy;
    const int copy = lane & (ncopy - 1);
    const int cshift = __builtin_ctz((unsigned)ncopy);            // ncopy is a power of two
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
    auto load_row = [&](RowBuf<VEC, NINT>& r, int yq, int yw) {
        yq = yq < ny - 1 ? yq : ny - 1;
        yw = yw < ny - 1 ? yw : ny - 1;
        const char* qrow = qbase + (size_t)yq * rowq;                          // wave-uniform
#if (XC_HIST_QNT & 1)
        RowLoadNT<TQ, VEC>::ld(reinterpret_cast<const TQ*>(qrow + xo_q), r.q);
#else
        RowLoad<TQ, VEC>::ld(reinterpret_cast<const TQ*>(qrow + xo_q), r.q);
#endif
        if (GRAD) r.h = (double)*reinterpret_cast<const TQ*>(qrow + xo_h);
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
    auto load_batch = [&](RowBuf<VEC, NINT> (&L)[U], int yb) {
#pragma unroll
        for (int i = 0; i < U; ++i) load_row(L[i], GRAD ? yb + i + 1 : yb + i, yb + i);
    };

    XC_STAMP(0);
    // issue the first loads of this wave BEFORE the edge prologue so that their HBM latency
    // overlaps the min/max reduction and the barriers below
    RowBuf<VEC, NINT> A[U], B[U];
    double qm[VEC], qcur[VEC], hcur = 0.0;
    bool have = g0 < g1;
    if (have) {
        begin_segment();
        if (GRAD) {
            load_metrics(y0);
            RowBuf<VEC, NINT> t;
            load_row(t, y0 > 0 ? y0 - 1 : 0, y0);
#pragma unroll
            for (int c = 0; c < VEC; ++c) qm[c] = t.q[c];
            load_row(t, y0, y0);
#pragma unroll
            for (int c = 0; c < VEC; ++c) qcur[c] = t.q[c];
            hcur = t.h;
        }
        load_batch(A, y0);
    }

    for (int i = tid; i < CW * hsz; i += blockDim.x) s_cell[i] = 0.0;
    XC_STAMP(1);

    // ------------------------------------------------------------------ edges -> LDS
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
        for (int k = tid; k < N; k += blockDim.x) {
            const double c = level_value(mn, mx, k, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1);
            s_edges[a.increase ? k + 1 : N - k] = c;
            if (bx == 0 && a.ctr_out) a.ctr_out[(size_t)slab * N + k] = c;
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
    }
    const double e0 = s_edges[0], eN = s_edges[N];
    const double inv = (double)N / (eN - e0);
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
    const bool wpos = FAST ? true : (a.dA_pos_finite != 0);
    XC_STAMP(2);

    double   acc[NCH];
    unsigned cnt = 0;
    int      cur = -1;                   // wave-uniform: bin of the register accumulators
    int      fp_skip = 0, fp_miss = 0;   // wave-uniform: rows left without the one-bin test / consecutive rows that failed it
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = 0.0;

    auto flush = [&]() {
        if (cnt) {
            double* cp = s_cell + (unsigned)(cur * ncopy + copy) * (unsigned)CW;
#pragma unroll
            for (int c = 0; c < NCH; ++c) lds_add(cp + c, acc[c]);
            lds_add(reinterpret_cast<unsigned*>(cp + NCH), cnt);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = 0.0;
        cnt = 0;
    };

    // one centre row: bins, weights, accumulate
    auto do_row = [&](const double (&qc)[VEC], const double (&qS)[VEC], const double (&qN)[VEC],
                      double hc, const double (&dAv)[VEC], const double (&inv_)[NINT > 0 ? NINT : 1][VEC],
                      double rdx, double rdy) {
        unsigned k[VEC];                 // bin, or N (the trash bin) for a dropped cell
        double w[NCH][VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const double vb = (!GRAD && negate) ? -qc[c] : qc[c];
            int kb;
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
        bool one_bin = false;
        int rb = 0;
        if (fp_skip == 0) {
            rb = __builtin_amdgcn_readfirstlane((int)k[0]);
            bool match = true;
#pragma unroll
            for (int c = 0; c < VEC; ++c) match = match && ((int)k[c] == rb);
            one_bin = rb < N && (__ballot(match) | inactive_mask) == ~0ull;
            if (one_bin) fp_miss = 0;
            else if (++fp_miss >= 8) { fp_skip = 48; fp_miss = 7; }
        } else {
            --fp_skip;
        }
        if (one_bin) {
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
                double* cp = s_cell + ((k[c] << cshift) + (unsigned)XC_ROT(copy, k[c], ncopy)) * (unsigned)CW;      // 32-bit LDS offset
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) lds_add(cp + ch, w[ch][c]);
                lds_add(reinterpret_cast<unsigned*>(cp + NCH), 1u);
            }
        }
    };

    // batch of U centre rows starting at yb; L holds (GRAD) q rows yb+1.. and weights rows yb..
    auto process_batch = [&](RowBuf<VEC, NINT> (&L)[U], int yb) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (yb + i < y1) {
                if (NEXT && active) {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) {              // NaN never wins a compare: NaN-skipping
                        const double v = L[i].qn[c];
                        nmn = fmin(nmn, v); nmx = fmax(nmx, v);                  // fmin / fmax skip NaN
                    }
                }
                if (GRAD) {
                    if (yb + i - ymet0 >= 64) load_metrics(yb + i);               // wave-uniform, once per 64 rows
                    do_row(qcur, qm, L[i].q, hcur, L[i].dA, L[i].in,
                           lane_get(rdxv, yb + i - ymet0), lane_get(rdyv, yb + i - ymet0));
#pragma unroll
                    for (int c = 0; c < VEC; ++c) { qm[c] = qcur[c]; qcur[c] = L[i].q[c]; }
                    hcur = L[i].h;
                } else {
                    do_row(L[i].q, L[i].q, L[i].q, 0.0, L[i].dA, L[i].in, 0.0, 0.0);
                }
            }
        }
    };

    while (have) {
        for (int yb = y0; yb < y1; yb += 2 * U) {
            if (yb + U < y1) load_batch(B, yb + U);
            process_batch(A, yb);
            if (yb + U >= y1) break;
            if (yb + 2 * U < y1) load_batch(A, yb + 2 * U);
            process_batch(B, yb + U);
        }
        have = g0 < g1;
        if (have) {                                   // next segment (the range crossed a strip boundary)
            begin_segment();
            if (GRAD) {
                load_metrics(y0);
                RowBuf<VEC, NINT> t;
                load_row(t, y0 > 0 ? y0 - 1 : 0, y0);
#pragma unroll
                for (int c = 0; c < VEC; ++c) qm[c] = t.q[c];
                load_row(t, y0, y0);
#pragma unroll
                for (int c = 0; c < VEC; ++c) qcur[c] = t.q[c];
                hcur = t.h;
            }
            load_batch(A, y0);
        }
    }
    XC_STAMP(3);
    flush();
    if (NEXT) {
        for (int o = 32; o > 0; o >>= 1) { nmn = fmin(nmn, __shfl_xor(nmn, o)); nmx = fmax(nmx, __shfl_xor(nmx, o)); }
        if (lane == 0) { s_red[2 * wave] = nmn; s_red[2 * wave + 1] = nmx; }
    }
    __syncthreads();
    if (NEXT && tid == 0) {
        for (int w = 1; w < nwave; ++w) { nmn = fmin(nmn, s_red[2 * w]); nmx = fmax(nmx, s_red[2 * w + 1]); }
        double* o = a.mm_next + ((size_t)slab * nbx + bx) * 2;
        o[0] = nmn; o[1] = nmx;
    }
    XC_STAMP(4);

    // ------------------------------------------------------------------ per-block partials (plain stores)
    const size_t pb = (size_t)slab * nbx + bx;
    double* ph = a.part_h + pb * NCH * N;
    // sum the lane-privatised copies; every thread starts at a rotated copy index so that the
    // 64 lanes of a wave hit distinct LDS banks (a fixed, thread-determined order)
    for (int i = tid; i < NCH * N; i += blockDim.x) {
        const int ch = i / N, b = i - ch * N;
        const double* src = s_cell + (size_t)b * ncopy * CW + ch;
        double sum = 0.0;
        for (int c = 0; c < ncopy; ++c) sum += src[(size_t)((c + tid) & (ncopy - 1)) * CW];
        ph[i] = sum;
    }
    unsigned* pc = a.part_c + pb * N;
    for (int b = tid; b < N; b += blockDim.x) {
        unsigned sum = 0u;
        for (int c = 0; c < ncopy; ++c)
            sum += *reinterpret_cast<const unsigned*>(s_cell + (size_t)(b * ncopy + ((c + tid) & (ncopy - 1))) * CW + NCH);
        pc[b] = sum;
    }
    XC_STAMP(5);
}

#ifdef XC_STAMPS
}  // namespace
extern "C" int xc_dbg_set_hist_stamps(unsigned long long* p)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p));
}
namespace {
#endif

template <typename TQ, int VEC, int NINT, bool GRAD, bool DA2D, bool NEXT, bool FAST = false>
int launch_three(xc_ctx* ctx, const HistGeom& g, int64_t nslab, const HistArgs& a)
{
    auto kern = k_hist<TQ, VEC, NINT, GRAD, DA2D, NEXT, FAST>;
    { const int rc = ensure_big_lds(ctx, reinterpret_cast<const void*>(kern), (int)kLdsBudget + 4096); if (rc != XC_OK) return rc; }
    HistArgs b = a;
    static const int xcd_env = [] { const char* e = getenv("XC_HIST_XCDMAP"); return e ? atoi(e) : 1; }();
    // the XCD-aware order needs whole groups of 8 row groups, otherwise it would leave XCDs idle (bps = 1 with many
    // small slabs would put every block on XCD 0): plain slab-fastest order then
    b.bps = g.bps; b.nslab_grid = (int)nslab; b.xcd_map = (xcd_env && g.bps % 8 == 0) ? 1 : 0;
    {
        static const int tile_env = [] { const char* e = getenv("XC_HIST_TILEMAP"); return e ? atoi(e) : 1; }();
        const int64_t nw = (int64_t)g.bps * (g.threads / 64);
        const int64_t nchunk = nw / g.nstrip;
        // needs at least one chunk, at least 8 rows per chunk (two halo rows are loaded per chunk) and no more than a tenth of
        // the waves left without a chunk (waves beyond nchunk x nstrip idle; the even strip-major split uses them all)
        b.nchunk = (tile_env && nchunk >= 1 && a.ny / nchunk >= 8 && (nw - nchunk * g.nstrip) * 10 <= nw) ? (int)nchunk : 0;
    }
    const int64_t nblk = b.xcd_map ? (int64_t)8 * ((g.bps + 7) / 8) * nslab : (int64_t)g.bps * nslab;
    if (nblk > 0x7fffffff) return fail(ctx, XC_EBADARG, "xc_hist: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(g.threads), g.lds, ctx->stream, b);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

template <typename TQ, int VEC, int NINT, bool GRAD, bool DA2D>
int launch_two(xc_ctx* ctx, const HistGeom& g, int64_t nslab, const HistArgs& a)
{
    // the NEXT variant exists for the two channel layouts of the Keff pipeline only
    constexpr bool kHasNext = (NINT == 0 && GRAD) || (NINT == 1 && !GRAD);
    if constexpr (NINT == 0 && GRAD && VEC == 2) {                  // the Keff layout: compile-time periodic / fillna-free variant
        if (a.periodic_x && a.dA_pos_finite && !a.negate && !a.last_closed) {
            if (a.q_next) return launch_three<TQ, VEC, NINT, GRAD, DA2D, true, true>(ctx, g, nslab, a);
            return launch_three<TQ, VEC, NINT, GRAD, DA2D, false, true>(ctx, g, nslab, a);
        }
    }
    if (a.q_next) {
        if constexpr (kHasNext) return launch_three<TQ, VEC, NINT, GRAD, DA2D, true>(ctx, g, nslab, a);
        else return fail(ctx, XC_EBADARG, "xc_hist: q_next is only supported by the Keff pipeline layouts");
    }
    return launch_three<TQ, VEC, NINT, GRAD, DA2D, false>(ctx, g, nslab, a);
}

template <typename TQ, int VEC, int NINT, bool GRAD>
int launch_one(xc_ctx* ctx, const HistGeom& g, int64_t nslab, const HistArgs& a)
{
    const bool da2d = a.dA_rank == XC_DA_PLANE || a.dA_rank == XC_DA_SLAB;
    return da2d ? launch_two<TQ, VEC, NINT, GRAD, true>(ctx, g, nslab, a)
                : launch_two<TQ, VEC, NINT, GRAD, false>(ctx, g, nslab, a);
}

template <typename TQ, int VEC>
int launch_nint(xc_ctx* ctx, int nint, int grad, const HistGeom& g, int64_t nslab, const HistArgs& a)
{
    if (grad) {
        switch (nint) {
            case 0: return launch_one<TQ, VEC, 0, true>(ctx, g, nslab, a);
            case 1: return launch_one<TQ, VEC, 1, true>(ctx, g, nslab, a);
            case 2: return launch_one<TQ, VEC, 2, true>(ctx, g, nslab, a);
        }
    } else {
        switch (nint) {
            case 0: return launch_one<TQ, VEC, 0, false>(ctx, g, nslab, a);
            case 1: return launch_one<TQ, VEC, 1, false>(ctx, g, nslab, a);
            case 2: return launch_one<TQ, VEC, 2, false>(ctx, g, nslab, a);
        }
    }
    return fail(ctx, XC_EBADARG, "xc_hist: nint must be 0..2");
}

}  // namespace

// Geometry: strips, blocks per slab, LDS copies.  Host-side checks live here so that
// a kernel is never launched on shapes it does not assume (ny, nx >= 1; nx even for
// VEC = 2; 16-byte aligned rows for the vector loads).
int hist_geometry(xc_ctx* ctx, int q_dtype, int64_t nslab, int64_t ny, int64_t nx, int nbin, int nch,
                  const void* q, HistGeom* g, int keff_fast_layout)
{
    if (nslab < 1 || ny < 1 || nx < 1) return fail(ctx, XC_EBADARG, "xc_hist: nslab, ny, nx must be >= 1");
    if (nbin < 1) return fail(ctx, XC_EBADARG, "xc_hist: need at least 2 edges");
    if (nslab > 65535) return fail(ctx, XC_EBADARG, "xc_hist: at most 65535 slabs per launch");
    const size_t esz = (q_dtype == XC_F32) ? 4 : 8;
    // VEC = 2 needs every row start (and dA / integrand rows, all f64 or f32) 2-element aligned
    // (`q` here is the OR of every streamed pointer's low bits, see vec_align_bits)
    const bool even = (nx % 2) == 0 && (reinterpret_cast<uintptr_t>(q) % (2 * esz)) == 0;
    g->vec = even ? 2 : 1;
    {
        // Four cells per lane (256-column strips, one 16-byte load per lane and row for float32): instantiated for the Keff FAST
        // layout only (xc_keff_dev says when its call qualifies).  Measured on cfg2-sized stacks: float32 tracers 14.7 -> 13.2 us
        // per slab chained, 16.7 -> 15.5 unchained (half the per-row fixed work and halo loads per cell); float64 tracers LOSE
        // (chained 1.14 -> 1.44 ms per 64 slabs, cfg4 +17 %), so the default is float32 only.  XC_HIST_VEC4: 0 never, 1 always.
        static const int env_v4 = [] { const char* e = getenv("XC_HIST_VEC4"); return e ? atoi(e) : -1; }();
        const bool want = env_v4 < 0 ? (q_dtype == XC_F32) : (env_v4 != 0);
        if (want && keff_fast_layout && even && nx % 4 == 0 && nx >= 1024 && (reinterpret_cast<uintptr_t>(q) % 16) == 0) g->vec = 4;
        if (keff_fast_layout == 2 && q_dtype != XC_F32) g->vec = even ? 2 : 1;      // the supplied-grdS variant exists for float32 only
    }
    if (ny > 0x7fffffff || nx > 0x7fffffff || (int64_t)((nx + 127) / 128) * ny > 0x7fffffffLL)
        return fail(ctx, XC_EBADARG, "xc_hist: slab too large (ny, nx and strips*ny must fit 31 bits)");
    const int W = 64 * g->vec;
    g->nstrip = (int)((nx + W - 1) / W);
    g->nch = nch;
    // experiment knobs (environment, read once): XC_HIST_THREADS, XC_HIST_NCOPY, XC_HIST_ROWS
    static int env_threads = -1, env_ncopy = -1, env_rows = -1;
    if (env_threads < 0) {
        const char* e = getenv("XC_HIST_THREADS"); env_threads = e ? atoi(e) : 0;
        e = getenv("XC_HIST_NCOPY"); env_ncopy = e ? atoi(e) : 0;
        e = getenv("XC_HIST_ROWS"); env_rows = e ? atoi(e) : 0;
    }
    g->threads = (env_threads >= 64 && env_threads <= kHistThreads && env_threads % 64 == 0) ? env_threads : kHistThreads;
    // largest power-of-two copy count that fits the LDS budget
    int ncopy = (env_ncopy >= 1 && env_ncopy <= kMaxCopies && (env_ncopy & (env_ncopy - 1)) == 0) ? env_ncopy : kMaxCopies;
    const size_t fixed = (64 + ((nbin + 2) & ~1)) * sizeof(double);
    const size_t cell = (size_t)(nch + 1) * 8;             // nch sums + the count, side by side; nbin + 1 bins (the last is the trash bin)
    while (ncopy > 1 && fixed + (size_t)(nbin + 1) * ncopy * cell > kLdsBudget) ncopy >>= 1;
    if (fixed + (size_t)(nbin + 1) * ncopy * cell > kLdsBudget)
        return fail(ctx, XC_EBADARG, "xc_hist: too many bins x channels for the LDS histogram");
    g->ncopy = ncopy;
    g->lds = fixed + (size_t)(nbin + 1) * ncopy * cell;
    g->lds = (g->lds + 15) & ~(size_t)15;
    // blocks per slab
    const int64_t total = (int64_t)g->nstrip * ny;
    const int waves = g->threads / 64;
    const int cus = ctx->cus > 0 ? ctx->cus : 256;
    // (strip,row) pairs per wave when slabs are plentiful: long sweeps amortise the block prologue (min/max partials,
    // levels, LDS clear) and epilogue (copy reduction); scanned on MI355X: 64 -> 192 rows is +7 % on the chained cfg2
    // schedule and +9 % on cfg4-sized slabs, 256 and more lose to the tail of the last round
    const int rows = env_rows > 0 ? env_rows : 192;
    int64_t bps = (total + waves * rows - 1) / (waves * rows);
    if (bps * nslab < cus) bps = (cus + nslab - 1) / nslab;
    // one block per CU is resident (LDS): make the grid a whole number of CU-wide rounds so that
    // the last round is not partly empty (408 blocks on 256 CUs ran at 80 % efficiency)
    {
        const int64_t tot = bps * nslab, rounds = (tot + cus - 1) / cus;
        const int64_t b2 = (rounds * cus) / nslab;
        if (b2 >= bps) bps = b2;
    }
    {
        static const int env_bps = [] { const char* e = getenv("XC_HIST_BPS"); return e ? atoi(e) : 0; }();
        if (env_bps > 0) bps = env_bps;                    // experiment knob
    }
    const int64_t maxb = (total + waves - 1) / waves;      // at least one pair per wave
    if (bps > maxb) bps = maxb;
    if (bps < 1) bps = 1;
    g->bps = (int)bps;
    g->part_h_doubles = (size_t)nch * nbin;
    return XC_OK;
}

int launch_hist(xc_ctx* ctx, int q_dtype, int nint, int grad, const HistGeom& g, int64_t nslab, const HistArgs& a)
{
    if (g.vec == 4) {                                    // chosen by hist_geometry for the two Keff layouts only (xc_keff_dev)
        const bool da2d = a.dA_rank == XC_DA_PLANE || a.dA_rank == XC_DA_SLAB;
        if (!grad && nint == 1 && da2d && q_dtype == XC_F32 && a.integ_f32[0] && !a.negate)      // float32 tracer + supplied float32 grdS
            return a.q_next ? launch_three<float, 4, 1, false, true, true, false>(ctx, g, nslab, a)
                            : launch_three<float, 4, 1, false, true, false, false>(ctx, g, nslab, a);
        if (nint != 0 || !grad || !da2d || !a.periodic_x || !a.dA_pos_finite || a.negate || a.last_closed)
            return fail(ctx, XC_EBADARG, "xc_hist: the four-cell variant exists for the Keff layouts only");
        if (q_dtype == XC_F64)
            return a.q_next ? launch_three<double, 4, 0, true, true, true, true>(ctx, g, nslab, a)
                            : launch_three<double, 4, 0, true, true, false, true>(ctx, g, nslab, a);
        return a.q_next ? launch_three<float, 4, 0, true, true, true, true>(ctx, g, nslab, a)
                        : launch_three<float, 4, 0, true, true, false, true>(ctx, g, nslab, a);
    }
    if (q_dtype == XC_F64) {
        return g.vec == 2 ? launch_nint<double, 2>(ctx, nint, grad, g, nslab, a)
                          : launch_nint<double, 1>(ctx, nint, grad, g, nslab, a);
    } else if (q_dtype == XC_F32) {
        return g.vec == 2 ? launch_nint<float, 2>(ctx, nint, grad, g, nslab, a)
                          : launch_nint<float, 1>(ctx, nint, grad, g, nslab, a);
    }
    return fail(ctx, XC_EBADARG, "xc_hist: q_dtype must be XC_F32 or XC_F64");
}

}  // namespace xc
