// K3, deterministic sums: the order-free variants of the histogram pass (k_hist<..., DET = 1 / 2>, xc_hist_kernel.h)
// and the two small kernels between / after them.
//
// The reference's sums come out of np.bincount inside xhistogram (core.py:1284, 1307): the same input gives the same
// bits.  The default K3 adds float64 weights with LDS atomics, so the LAST bits of a sum depend on the order in which
// waves reach the LDS.  `xc_hist_desc.deterministic` / `xc_keff_desc.deterministic` select this path instead:
//
//   pass 1  k_hist<DET=1>   per (bin, channel): max |w| by ds_max_u64 on the bit patterns + exact counts
//   scales  k_det_scales    k = 62 - ceil(log2 count) - (ilogb(max) + 1):  count * max * 2^k < 2^62
//   pass 2  k_hist<DET=2>   n = rint(w * 2^k) as a 64-bit integer (xc_binning.h: fixed_point; one rounding, a function of
//                           the cell alone), added with ds_add_u64
//   reduce  k_det_reduce    the integer partials of the blocks summed exactly, converted once: sum = double(n_total) * 2^-k
//
// Integer addition is associative and commutative: the result does not depend on the order of arrival, on the block
// geometry or on how many slabs share a launch -- two runs, or a 1-rank and an 8-rank job, give the same bits.
// Precision: every weight keeps 62 - ceil(log2 count) bits below the LARGEST weight of its bin (a bin of 2^15 cells: 47
// bits, rounding error of the sum ~2^-55 of it -- float64 summation in any order is no better); a cell more than 2^47
// times smaller than its bin's maximum is rounded away, as it is by float64 addition to a sum that holds that maximum.
// A bin that saw an infinite weight yields NaN.
#include "xc_internal.h"

namespace xc {

namespace {

#include "xc_hist_kernel.h"

constexpr int kDetNonFinite = -0x40000000;      // exponent marker: ldexp(w, it) == 0, the reduction writes NaN

template <typename TQ, int VEC, int NINT, bool GRAD>
int det_two(xc_ctx* ctx, const HistGeom& g, int64_t nslab, const HistArgs& a, int det)
{
    const bool da2d = a.dA_rank == XC_DA_PLANE || a.dA_rank == XC_DA_SLAB;
    if (det == 1)
        return da2d ? launch_three<TQ, VEC, NINT, GRAD, true, false, false, 1>(ctx, g, nslab, a)
                    : launch_three<TQ, VEC, NINT, GRAD, false, false, false, 1>(ctx, g, nslab, a);
    // the fixed-point pass of the two Keff layouts (in-kernel gradient; one supplied integrand) can carry the NEXT batch's
    // min / max like the default kernel does (xc_keff_desc.q_next): the stand-alone K1 pass of the next call disappears
    if constexpr ((GRAD && NINT == 0) || (!GRAD && NINT == 1)) {
        if (a.q_next)
            return da2d ? launch_three<TQ, VEC, NINT, GRAD, true, true, false, 2>(ctx, g, nslab, a)
                        : launch_three<TQ, VEC, NINT, GRAD, false, true, false, 2>(ctx, g, nslab, a);
    }
    if (a.q_next) return fail(ctx, XC_EBADARG, "xc_hist: q_next rides in the Keff layouts only");
    return da2d ? launch_three<TQ, VEC, NINT, GRAD, true, false, false, 2>(ctx, g, nslab, a)
                : launch_three<TQ, VEC, NINT, GRAD, false, false, false, 2>(ctx, g, nslab, a);
}

template <typename TQ, int VEC>
int det_one(xc_ctx* ctx, int nint, int grad, const HistGeom& g, int64_t nslab, const HistArgs& a, int det)
{
    if (grad) {
        switch (nint) {
            case 0: return det_two<TQ, VEC, 0, true>(ctx, g, nslab, a, det);
            case 1: return det_two<TQ, VEC, 1, true>(ctx, g, nslab, a, det);
            case 2: return det_two<TQ, VEC, 2, true>(ctx, g, nslab, a, det);
        }
    } else {
        switch (nint) {
            case 0: return det_two<TQ, VEC, 0, false>(ctx, g, nslab, a, det);
            case 1: return det_two<TQ, VEC, 1, false>(ctx, g, nslab, a, det);
            case 2: return det_two<TQ, VEC, 2, false>(ctx, g, nslab, a, det);
        }
    }
    return fail(ctx, XC_EBADARG, "xc_hist: nint must be 0..2");
}

// 8 values per 256-thread block, 32 lanes per value (as k_reduce_partials).  v < nch*nbin: (channel, bin).
__global__ __launch_bounds__(256)
void k_det_scales(const unsigned long long* __restrict__ part_m, const unsigned* __restrict__ part_c,
                  int bps, int nch, int nbin, int* __restrict__ scale, unsigned long long* __restrict__ red_c)
{
    const int slab = blockIdx.y, tid = threadIdx.x, l = tid & 31;
    const int nvh = nch * nbin;
    const int v = blockIdx.x * 8 + (tid >> 5);
    if (v >= nvh) return;
    const int ch = v / nbin, k = v - ch * nbin;
    const unsigned long long* pm = part_m + (size_t)slab * bps * nvh + v;
    const unsigned* pc = part_c + (size_t)slab * bps * nbin + k;
    unsigned long long mx = 0ull, cnt = 0ull;
    for (int b = l; b < bps; b += 32) {
        const unsigned long long m = pm[(size_t)b * nvh];
        mx = m > mx ? m : mx;
        cnt += pc[(size_t)b * nbin];
    }
    for (int o = 16; o > 0; o >>= 1) {
        const unsigned long long m = __shfl_xor(mx, o);
        mx = m > mx ? m : mx;
        cnt += __shfl_xor(cnt, o);
    }
    if (l != 0) return;
    if (ch == 0) red_c[(size_t)slab * nbin + k] = cnt;
    const double M = __longlong_as_double((long long)mx);            // max |w| >= 0
    int kk;
    if (cnt == 0ull || M == 0.0) {
        kk = 0;                                                      // nothing to scale
    } else if (!(M < __longlong_as_double(0x7ff0000000000000LL))) {
        kk = kDetNonFinite;                                          // an infinite weight: the bin reports NaN
    } else {
        const int e = ilogb(M) + 1;                                  // M < 2^e
        const int L = cnt > 1ull ? 64 - __clzll((long long)(cnt - 1ull)) : 0;   // cnt <= 2^L
        kk = 62 - L - e;                                             // cnt * M * 2^kk < 2^62
    }
    scale[(size_t)slab * nvh + v] = kk;
}

__global__ __launch_bounds__(256)
void k_det_reduce(const unsigned long long* __restrict__ part_s, int bps, int nch, int nbin,
                  const int* __restrict__ scale, double* __restrict__ red_h)
{
    const int slab = blockIdx.y, tid = threadIdx.x, l = tid & 31;
    const int nvh = nch * nbin;
    const int v = blockIdx.x * 8 + (tid >> 5);
    if (v >= nvh) return;
    const unsigned long long* ps = part_s + (size_t)slab * bps * nvh + v;
    unsigned long long tot = 0ull;
    for (int b = l; b < bps; b += 32) tot += ps[(size_t)b * nvh];
    for (int o = 16; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (l != 0) return;
    const int kk = scale[(size_t)slab * nvh + v];
    red_h[(size_t)slab * nvh + v] = kk == kDetNonFinite ? __longlong_as_double(0x7ff8000000000000LL)
                                                        : ldexp((double)(long long)tot, -kk);     // |tot| < 2^63: one rounding to 53 bits
}

}  // namespace

int launch_hist_det(xc_ctx* ctx, int q_dtype, int nint, int grad, const HistGeom& g, int64_t nslab, const HistArgs& a, int det)
{
    if (det != 1 && det != 2) return fail(ctx, XC_EBADARG, "xc_hist: det must be 1 or 2");
    if (det == 2 && !a.det_scale) return fail(ctx, XC_EBADARG, "xc_hist: the fixed-point pass needs its exponents");
    if (a.q_next && (det != 2 || !a.mm_next)) return fail(ctx, XC_EBADARG, "xc_hist: q_next rides in the fixed-point pass only");
    if (g.vec == 4) return fail(ctx, XC_EBADARG, "xc_hist: no four-cell variant with deterministic sums");
    if (q_dtype == XC_F64)
        return g.vec == 2 ? det_one<double, 2>(ctx, nint, grad, g, nslab, a, det) : det_one<double, 1>(ctx, nint, grad, g, nslab, a, det);
    if (q_dtype == XC_F32)
        return g.vec == 2 ? det_one<float, 2>(ctx, nint, grad, g, nslab, a, det) : det_one<float, 1>(ctx, nint, grad, g, nslab, a, det);
    return fail(ctx, XC_EBADARG, "xc_hist: q_dtype must be XC_F32 or XC_F64");
}

int launch_det_scales(xc_ctx* ctx, int64_t nslab, int bps, int nch, int nbin, const double* part_h, const unsigned* part_c,
                      int* scale, unsigned long long* red_c)
{
    dim3 grid((unsigned)((nch * nbin + 7) / 8), (unsigned)nslab);
    hipLaunchKernelGGL(k_det_scales, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned long long*>(part_h), part_c,
                       bps, nch, nbin, scale, red_c);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

int launch_det_reduce(xc_ctx* ctx, int64_t nslab, int bps, int nch, int nbin, const double* part_h, const int* scale, double* red_h)
{
    dim3 grid((unsigned)((nch * nbin + 7) / 8), (unsigned)nslab);
    hipLaunchKernelGGL(k_det_reduce, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned long long*>(part_h), bps, nch, nbin,
                       scale, red_h);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

}  // namespace xc
