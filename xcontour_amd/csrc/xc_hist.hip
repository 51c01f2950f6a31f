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

#include "xc_hist_kernel.h"

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
                  const void* q, HistGeom* g, int keff_fast_layout, int det)
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
        const int env_v4 = ctx->knobs.vec4;
        const bool want = env_v4 < 0 ? (q_dtype == XC_F32) : (env_v4 != 0);
        if (want && keff_fast_layout && even && nx % 4 == 0 && nx >= 1024 && (reinterpret_cast<uintptr_t>(q) % 16) == 0) g->vec = 4;
        if (keff_fast_layout == 2 && q_dtype != XC_F32) g->vec = even ? 2 : 1;      // the supplied-grdS variant exists for float32 only
        if (det) g->vec = even ? 2 : 1;                                              // the order-free variants: one or two cells per lane
    }
    if (ny > 0x7fffffff || nx > 0x7fffffff || (int64_t)((nx + 127) / 128) * ny > 0x7fffffffLL)
        return fail(ctx, XC_EBADARG, "xc_hist: slab too large (ny, nx and strips*ny must fit 31 bits)");
    const int W = 64 * g->vec;
    g->nstrip = (int)((nx + W - 1) / W);
    g->nch = nch;
    // experiment knobs (environment, read once in xc_create): XC_HIST_THREADS, XC_HIST_NCOPY, XC_HIST_ROWS, XC_HIST_BPS
    const int env_threads = ctx->knobs.threads, env_ncopy = ctx->knobs.ncopy, env_rows = ctx->knobs.rows;
    g->threads = (env_threads >= 64 && env_threads <= kHistThreads && env_threads % 64 == 0) ? env_threads : kHistThreads;
    // largest power-of-two copy count that fits the LDS budget
    int ncopy = (env_ncopy >= 1 && env_ncopy <= kMaxCopies && (env_ncopy & (env_ncopy - 1)) == 0) ? env_ncopy : kMaxCopies;
    // (the fixed-point pass keeps its (nbin + 1) x nch binary exponents behind the cells)
    // (and the float32 Keff layout a float32 copy of its nbin + 1 edges, same place)
    const size_t behind = det ? 0 : ((keff_fast_layout == 1 && q_dtype == XC_F32 && nbin <= kE32MaxBins) ? ((size_t)nbin + 2) / 2 : 0);
    const size_t fixed = (64 + ((nbin + 2) & ~1) + behind) * sizeof(double);
    // nch sums + the count, side by side; nbin + 1 bins (the last is the trash bin).  Deterministic sums: the limbs of every channel's
    // superaccumulator + a trash word + the count / flag word (xc_binning.h)
    const size_t cell = det ? (size_t)(kDetWords * nch + 1) * 8 : (size_t)(nch + 1) * 8;
    while (ncopy > 1 && fixed + (size_t)(nbin + 1) * ncopy * cell > kLdsBudget) ncopy >>= 1;
    if (fixed + (size_t)(nbin + 1) * ncopy * cell > kLdsBudget)
        return fail(ctx, XC_EBADARG, "xc_hist: too many bins x channels for the LDS histogram");
    g->ncopy = ncopy;
    g->lds = fixed + (size_t)(nbin + 1) * ncopy * cell;
    g->lds = (g->lds + 15) & ~(size_t)15;
    // blocks per slab
    const int64_t total = (int64_t)g->nstrip * ny;
    const int cus = ctx->cus > 0 ? ctx->cus : 256;
    // a small stack (the reference's demo files: 15 x 241 x 480) leaves a 16-wave workgroup three rows per wave: the prologue (LDS clear,
    // edges, barriers) is the kernel then -- half the waves, twice the rows (measured there: 14.0 -> 10.4 us with an integrand, 11.0 -> 9.0
    // without; 256 threads: 13.2 / 8.9)
    if (!(env_threads >= 64) && g->threads == kHistThreads) {
        int64_t b0 = (total + (kHistThreads / 64) * 192 - 1) / ((kHistThreads / 64) * 192);
        if (b0 * nslab < cus) b0 = (cus + nslab - 1) / nslab;
        if (total < b0 * (kHistThreads / 64) * 6) g->threads = kHistThreads / 2;
    }
    const int waves = g->threads / 64;
    // (strip,row) pairs per wave when slabs are plentiful: long sweeps amortise the block prologue (min/max partials,
    // levels, LDS clear) and epilogue (copy reduction); scanned on MI355X: 64 -> 192 rows is +7 % on the chained cfg2
    // schedule and +9 % on cfg4-sized slabs, 256 and more lose to the tail of the last round
    int rows = env_rows > 0 ? env_rows : 192;
    // deterministic sums: a limb of one LDS copy holds at most 32767 chunks of 2^48 before a signed 64-bit word could wrap; a copy
    // serves 64 / ncopy lanes of every wave, vec cells per lane and row (margin: the row chunks of the strip-fastest order are
    // cut a little unevenly)
    const int det_cap = det ? 28000 / (waves * (64 / ncopy) * g->vec) : 0;
    if (det) { if (det_cap < 1) return fail(ctx, XC_EBADARG, "xc_hist: deterministic sums: too many bins for the LDS"); if (rows > det_cap) rows = det_cap; }
    int64_t bps = (total + waves * rows - 1) / (waves * rows);
    if (bps * nslab < cus) bps = (cus + nslab - 1) / nslab;
    // one block per CU is resident (LDS): make the grid a whole number of CU-wide rounds so that
    // the last round is not partly empty (408 blocks on 256 CUs ran at 80 % efficiency)
    {
        const int64_t tot = bps * nslab, rounds = (tot + cus - 1) / cus;
        const int64_t b2 = (rounds * cus) / nslab;
        if (b2 >= bps) bps = b2;
    }
    if (ctx->knobs.bps > 0) bps = ctx->knobs.bps;           // experiment knob
    const int64_t maxb = (total + waves - 1) / waves;      // at least one pair per wave
    if (bps > maxb) bps = maxb;
    if (bps < 1) bps = 1;
    if (det && (total + bps * waves - 1) / (bps * waves) > det_cap + det_cap / 16)
        return fail(ctx, XC_EBADARG, "xc_hist: deterministic sums: this block geometry could overflow an accumulator limb (XC_HIST_BPS / XC_HIST_THREADS too small)");
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
        // float32 tracer AND float32 contour levels (the reference's default dtype): the E32 variant, unless XC_HIST_E32=0
        if (a.levels_mode && a.ctr_f32 && a.nbin <= kE32MaxBins && ctx->knobs.e32)
            return a.q_next ? launch_three<float, 4, 0, true, true, true, true, 0, true>(ctx, g, nslab, a)
                            : launch_three<float, 4, 0, true, true, false, true, 0, true>(ctx, g, nslab, a);
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
