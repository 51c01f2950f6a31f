// K3, deterministic sums: the order-free variant of the histogram pass (k_hist<..., DET = 3>, xc_hist_kernel.h) and the exact
// reduction behind it.
//
// The reference's sums come out of np.bincount inside xhistogram (core.py:1284, 1307): the same input gives the same
// bits.  The default K3 adds float64 weights with LDS atomics, so the LAST bits of a sum depend on the order in which
// waves reach the LDS.  `xc_hist_desc.deterministic` / `xc_keff_desc.deterministic` select this path instead:
//
//   bounds  K1 passes    max |dA| (or the caller's), extrema of every supplied integrand: they fix the accumulator windows BEFORE the pass
//   pass    k_hist<DET=3> every weight rounded once to 49 bits and added, as two integer chunks, to the limbs of a fixed-point
//                         superaccumulator per (bin, channel) with ds_add_u64 (xc_binning.h: det_split)
//   reduce  k_det3_reduce the blocks' limbs summed exactly, carried, converted ONCE to float64 (round half to even)
//
// Integer addition is associative and commutative: the result does not depend on the order of arrival, on the block
// geometry or on how many slabs share a launch -- two runs, or a 1-rank and an 8-rank job, give the same bits.  ONE pass over
// the cells (rounds 3-4: two, because a per-bin scale needed the bin's maximum first: 1.9x the default pass).
// Precision: 2^-49 relative per weight, whatever its magnitude inside the window (192 bits below the bound, 96 for dA).
// A bin that saw an infinite weight yields NaN.
#include "xc_internal.h"

namespace xc {

namespace {

#include "xc_hist_kernel.h"

// DET == 3: the one-pass superaccumulator (xc_binning.h).  The Keff layout with everything verified on the host takes the FAST body.
template <typename TQ, int VEC, int NINT, bool GRAD>
int det3_two(xc_ctx* ctx, const HistGeom& g, int64_t nslab, const HistArgs& a)
{
    const bool da2d = a.dA_rank == XC_DA_PLANE || a.dA_rank == XC_DA_SLAB;
    if constexpr (GRAD && NINT == 0 && VEC == 2) {
        if (da2d && a.periodic_x && a.dA_pos_finite && !a.negate && !a.last_closed)
            return a.q_next ? launch_three<TQ, VEC, NINT, GRAD, true, true, true, 3>(ctx, g, nslab, a)
                            : launch_three<TQ, VEC, NINT, GRAD, true, false, true, 3>(ctx, g, nslab, a);
    }
    if constexpr ((GRAD && NINT == 0) || (!GRAD && NINT == 1)) {
        if (a.q_next)
            return da2d ? launch_three<TQ, VEC, NINT, GRAD, true, true, false, 3>(ctx, g, nslab, a)
                        : launch_three<TQ, VEC, NINT, GRAD, false, true, false, 3>(ctx, g, nslab, a);
    }
    if (a.q_next) return fail(ctx, XC_EBADARG, "xc_hist: q_next rides in the Keff layouts only");
    return da2d ? launch_three<TQ, VEC, NINT, GRAD, true, false, false, 3>(ctx, g, nslab, a)
                : launch_three<TQ, VEC, NINT, GRAD, false, false, false, 3>(ctx, g, nslab, a);
}

template <typename TQ, int VEC>
int det3_one(xc_ctx* ctx, int nint, int grad, const HistGeom& g, int64_t nslab, const HistArgs& a)
{
    if (grad) {
        switch (nint) {
            case 0: return det3_two<TQ, VEC, 0, true>(ctx, g, nslab, a);
            case 1: return det3_two<TQ, VEC, 1, true>(ctx, g, nslab, a);
            case 2: return det3_two<TQ, VEC, 2, true>(ctx, g, nslab, a);
        }
    } else {
        switch (nint) {
            case 0: return det3_two<TQ, VEC, 0, false>(ctx, g, nslab, a);
            case 1: return det3_two<TQ, VEC, 1, false>(ctx, g, nslab, a);
            case 2: return det3_two<TQ, VEC, 2, false>(ctx, g, nslab, a);
        }
    }
    return fail(ctx, XC_EBADARG, "xc_hist: nint must be 0..2");
}

// The blocks' canonical limbs summed exactly (32 lanes per (channel, bin), a fixed tree -- integer addition: any order gives the same
// bits), carried, and converted ONCE: the integer (up to ~200 bits) -> float64, round half to even, times 2^(window bottom).
__global__ __launch_bounds__(256)
void k_det3_reduce(const unsigned long long* __restrict__ part_l, const unsigned* __restrict__ part_c, int bps, int nch, int nbin,
                   const int* __restrict__ c0, double* __restrict__ red_h, unsigned long long* __restrict__ red_c)
{
    const int slab = blockIdx.y, tid = threadIdx.x, l = tid & 31;
    const int nvh = nch * nbin, NL = det_total_limbs(nch);
    const int v = blockIdx.x * 8 + (tid >> 5);
    if (v >= nvh) return;
    const int ch = v / nbin, k = v - ch * nbin;
    constexpr int M = kDetLimbsX;
    const int base = kDetLimbsX * ch;
    long long acc[kDetLimbsX] = {0, 0, 0, 0};
    unsigned long long cnt = 0ull; unsigned fl = 0u;
    for (int b = l; b < bps; b += 32) {
        const unsigned long long* p = part_l + ((size_t)slab * bps + b) * NL * nbin + (size_t)base * nbin + k;
#pragma unroll
        for (int i = 0; i < kDetLimbsX; ++i) if (i < M) acc[i] += (long long)p[(size_t)i * nbin];
        const unsigned w = part_c[((size_t)slab * bps + b) * nbin + k];
        cnt += w & 0x0fffffffu; fl |= w >> 28;
    }
    for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
        for (int i = 0; i < kDetLimbsX; ++i) acc[i] += __shfl_xor(acc[i], o);
        cnt += __shfl_xor(cnt, o); fl |= __shfl_xor(fl, o);
    }
    if (l != 0) return;
    if (ch == 0) red_c[(size_t)slab * nbin + k] = cnt;
    double out;
    if ((fl >> ch) & 1u) {
        out = __longlong_as_double(0x7ff8000000000000LL);                  // the bin saw an infinite weight
    } else {
        // carries from the last limb up; then sign + magnitude digits D[0] (any size) , D[1..] < 2^48
        long long carry = 0;
#pragma unroll
        for (int i = kDetLimbsX - 1; i >= 0; --i) if (i < M) {
            long long t = acc[i] + carry; carry = 0;
            if (i > 0) { carry = t >> kDetLimbBits; t -= carry << kDetLimbBits; }
            acc[i] = t;
        }
        const bool neg = acc[0] < 0;
        if (neg) {
            long long borrow = 0;
#pragma unroll
            for (int i = kDetLimbsX - 1; i >= 0; --i) if (i < M) {
                long long t = -acc[i] - borrow; borrow = 0;
                if (i > 0 && t < 0) { t += 1ll << kDetLimbBits; borrow = 1; }
                acc[i] = t;
            }
        }
        // the top (up to) 64 significant bits of the digit string + a sticky bit for everything below them
        int first = -1;
#pragma unroll
        for (int i = 0; i < kDetLimbsX; ++i) if (i < M && first < 0 && acc[i] != 0) first = i;
        if (first < 0) {
            out = 0.0;
        } else {
            unsigned long long top = 0ull; int nb = 0, below = 0; bool sticky = false;     // nb: bits in `top`; below: bits of the string under top's last bit
#pragma unroll
            for (int i = 0; i < kDetLimbsX; ++i) if (i < M) {
                const unsigned long long D = (unsigned long long)acc[i];
                if (i == first) { top = D; nb = 64 - __clzll((long long)D); below = kDetLimbBits * (M - 1 - first); }
                else if (i > first) {
                    if (nb + kDetLimbBits <= 64) { top = (top << kDetLimbBits) | D; nb += kDetLimbBits; below -= kDetLimbBits; }
                    else if (nb < 64) {
                        const int take = 64 - nb, rest = kDetLimbBits - take;
                        top = (top << take) | (D >> rest);
                        sticky = sticky || (D & ((1ull << rest) - 1ull)) != 0ull;
                        nb = 64; below -= take;
                    } else sticky = sticky || D != 0ull;
                }
            }
            // the window: c0 = top_exponent + 1023 + 52 - (53 - P); the string's last bit is worth 2^(top_exponent - S M)
            const int kc0 = c0[(size_t)slab * nch + ch];
            int e = (kc0 - (1023 + 52 - (53 - kDetPrecBits))) - kDetLimbBits * M + below;
            if (nb > 53) {
                const int drop = nb - 53;
                const unsigned long long rem = top & ((1ull << drop) - 1ull), half = 1ull << (drop - 1);
                top >>= drop;
                if (rem > half || (rem == half && (sticky || (top & 1ull)))) ++top;      // (2^53 after the increment is still exact)
                e += drop;
            }
            out = ldexp((double)top, e);
            if (neg) out = -out;
        }
    }
    red_h[(size_t)slab * nvh + v] = out;
}

}  // namespace

int det_limbs_total(int nch) { return det_total_limbs(nch); }

int launch_hist_det3(xc_ctx* ctx, int q_dtype, int nint, int grad, const HistGeom& g, int64_t nslab, const HistArgs& a)
{
    if (g.vec == 4) return fail(ctx, XC_EBADARG, "xc_hist: no four-cell variant with deterministic sums");
    if (!a.part_c || !a.det_c0_out) return fail(ctx, XC_EBADARG, "xc_hist: the one-pass deterministic sums need the count partials and the window constants");
    if (q_dtype == XC_F64) return g.vec == 2 ? det3_one<double, 2>(ctx, nint, grad, g, nslab, a) : det3_one<double, 1>(ctx, nint, grad, g, nslab, a);
    if (q_dtype == XC_F32) return g.vec == 2 ? det3_one<float, 2>(ctx, nint, grad, g, nslab, a) : det3_one<float, 1>(ctx, nint, grad, g, nslab, a);
    return fail(ctx, XC_EBADARG, "xc_hist: q_dtype must be XC_F32 or XC_F64");
}

int launch_det3_reduce(xc_ctx* ctx, int64_t nslab, int bps, int nch, int nbin, const double* part_l, const unsigned* part_c,
                       const int* c0, double* red_h, unsigned long long* red_c)
{
    dim3 grid((unsigned)((nch * nbin + 7) / 8), (unsigned)nslab);
    hipLaunchKernelGGL(k_det3_reduce, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned long long*>(part_l), part_c,
                       bps, nch, nbin, c0, red_h, red_c);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

}  // namespace xc
