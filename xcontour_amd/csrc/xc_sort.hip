// K8 -- exact adiabatic rearrangement: device radix sort of (tracer, dA) pairs, cumulative
// area of the sorted state, sorted profile Q(A) and background-potential-energy integral.
//
// No reference call site (the reference realises "sorting" as histogram CDF + table lookup,
// SURVEY F6); this is SURVEY 8-a9, build-defined and pinned by oracle.sorted_profile:
//   drop NaN / masked cells; stable ascending sort of (q, dA); Acum = cumsum(dA_sorted);
//   Q_exact(A_j) = q_sorted[min(searchsorted(Acum, A_j, 'right'), n-1)].
//
// Every kernel takes the slab from blockIdx.y (or blockIdx.x for the one-block-per-slab kernels): a stack of
// planes is sorted by ONE set of launches (segmented sort: per-slab tile histograms, bases and scans).
//
// float64 tracers (round 3): THREE passes instead of eight.  The sort key of the passes is not the 64-bit pattern but
// the 24-bit RANGE key  d = floor((v - min) * (2^24 - 1) / (max - min))  (min / max from K1; subtract, multiply by a positive
// constant and floor are each monotone, so d never orders two values the wrong way; equal values share a d).  Three
// stable LSD passes over the bytes of d leave the pairs sorted by d with the cells of one d in their original order; a
// RUN of equal d holds 0.4 cells on average for 6.5 M pairs, so `k_fix_runs` finishes the job with a stable odd-even
// transposition sort of every run that is out of order (in LDS, runs of <= 128) and proves the result: one flag is
// read back, and only if some run was longer and out of order (a spike of distinct values narrower than 2^-24 of the
// range) the stack is sorted again by the full eight-pass path.  Ties of any length are already in order.
//
// Hand-written LSD radix sort, 8-bit digits, 64-bit order-preserving keys, f64 payload:
// one block owns one tile of 4096 consecutive elements (4 waves x 1024); stable ranks come from wave
// ballots (8 ballots give the peer mask of a lane's digit) + per-wave digit counters in LDS; the tile is
// reordered by digit in LDS before it is stored; the (digit-major) tile histogram is scanned by two
// small kernels.  The key type is a template parameter: float32 tracers sort 32-bit keys in 4 passes
// (4 B/elem histogram + 24 B/elem scatter), float64 tracers 64-bit keys in 8 passes (8 + 32 B/elem).
#include "xc_internal.h"
#include <type_traits>

#define XC_TRY_(expr) do { int _rc = (expr); if (_rc != XC_OK) return _rc; } while (0)

namespace xc {
namespace {

#ifndef XC_TILE_ROUNDS
#define XC_TILE_ROUNDS 16
#endif
constexpr int TILE_ROUNDS = XC_TILE_ROUNDS;
constexpr int TILE = 64 * TILE_ROUNDS;      // elements per wave
constexpr int BTILE = 4 * TILE;             // elements per block tile
typedef unsigned long long u64;
typedef unsigned int u32;
template <typename K> struct KeyTraits;
template <> struct KeyTraits<u64> {
    static constexpr int passes = 8;
    __device__ static __forceinline__ u64 invalid() { return ~0ull; }
    __device__ static __forceinline__ u64 encode(double v)
    {
        // order-preserving map of IEEE doubles to unsigned integers; -0.0 is folded onto +0.0 so that
        // equal values keep their original order exactly like numpy's stable sort
        const u64 u = (u64)__double_as_longlong(v == 0.0 ? 0.0 : v);
        return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    }
    __device__ static __forceinline__ double decode(u64 k)
    {
        const u64 u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        return __longlong_as_double((long long)u);
    }
};
template <> struct KeyTraits<u32> {            // float32 tracers: the key of the float IS the order of its double
    static constexpr int passes = 4;
    __device__ static __forceinline__ u32 invalid() { return ~0u; }
    __device__ static __forceinline__ u32 encode(double v)
    {
        const float f = (float)v;              // exact: v came from a float (possibly negated)
        const u32 u = (u32)__float_as_int(f == 0.0f ? 0.0f : f);
        return (u >> 31) ? ~u : (u | 0x80000000u);
    }
    __device__ static __forceinline__ double decode(u32 k)
    {
        const u32 u = (k >> 31) ? (k & 0x7fffffffu) : ~k;
        return (double)__int_as_float((int)u);
    }
};

// the (key, payload) pairs of TILE_ROUNDS cells per lane, straight from the tracer / mask / dA: pass 0 of the sort builds its
// pairs with this, so the unsorted pairs are never written and read back (24-32 B per cell).  Phases, not a per-cell
// function: every load of a stream is issued before the first use (clamped addresses, wave-uniform branches only), and a
// per-row dA divides in 32 bits (n < 2^31).
template <typename TQ, typename TM, typename K, int R, bool VALS>
__device__ __forceinline__ void load_pairs(const TQ* __restrict__ q, const TM* __restrict__ mask, const double* __restrict__ dA,
                                           int dA_rank, int64_t nx, int negate, int64_t base, int lane, int64_t n,
                                           K (&key)[R], double (&val)[R])
{
    TQ qv[R];
    TM mv[R];
    unsigned idx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const int64_t i = base + r * 64 + lane; idx[r] = (unsigned)(i < n ? i : n - 1); qv[r] = q[idx[r]]; }
    if (mask) {
#pragma unroll
        for (int r = 0; r < R; ++r) mv[r] = mask[idx[r]];
    }
    if (VALS) {
        if (dA_rank == XC_DA_PLANE) {
#pragma unroll
            for (int r = 0; r < R; ++r) val[r] = dA[idx[r]];
        } else if (dA_rank == XC_DA_ROW) {
            const unsigned unx = (unsigned)nx;
#pragma unroll
            for (int r = 0; r < R; ++r) val[r] = dA[idx[r] / unx];
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) val[r] = 1.0;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double v = negate ? -(double)qv[r] : (double)qv[r];
        const bool ok = (v == v) && (!mask || mv[r] == (TM)1) && (base + r * 64 + lane < n);
        key[r] = ok ? KeyTraits<K>::encode(v) : KeyTraits<K>::invalid();       // dropped cells sort to the end
        if (VALS) val[r] = ok ? val[r] : 0.0;
    }
}
struct PairSrc {                 // where pass 0 finds its input (per-slab strides applied by the kernels)
    const void* q; const void* mask; const double* dA;
    int dA_rank, negate; int64_t nx, mask_stride, dA_stride;
    const double* mm;            // [nslab][4] min, max, robust low, robust high of the tracer (K1 + k_range_bounds): the range-key passes only
    const unsigned* rtab;        // [nslab][2 * RANGE_NB] first range key and number of range keys of every coarse bin
};

// ---- the 24-bit range key (MODE 1 of the passes).  Valid values map to [0, 2^24 - 2] monotonically, dropped cells
// (key == invalid) to 2^24 - 1, so that they gather behind every valid value without sharing a run with the maximum.
// The map is piecewise linear: the value range is cut into RANGE_NB equal coarse bins and every bin gets a share of the 2^24
// range keys proportional to its POPULATION (histogram equalisation: k_range_hist counts, k_range_table divides), so a
// plateau that holds a third of the cells inside a thousandth of the range -- a well-mixed layer, a saturating tanh profile --
// is still resolved to ~2^-30 of the range and its runs of equal range key stay short.  Monotone: x = (v - lo) * S is
// monotone in v, so are b = floor(x) and, inside a bin, x - b (exact) and floor((x - b) * width); bins do not overlap.
constexpr unsigned RANGE_INVALID = 0xFFFFFFu;
constexpr int RANGE_NB = 256;
constexpr int RANGE_SAMPLE = 16;       // k_range_hist looks at one 2048-cell chunk in 16: any positive widths give a monotone map, the
                                        // populations only have to be roughly right for the runs to come out short
// Three zones (round 4).  The 256 equalised coarse bins cover the ROBUST range [rlo, rhi] of the plane -- the 9th smallest of
// the K1 block minima to the 9th largest of the block maxima (k_range_bounds) -- and the cells outside it (a handful: the block
// extrema are extreme order statistics of the plane) get 2^16 keys each, linear over [min, rlo) and (rhi, max].  With the exact
// min / max as the ends of the equalised range (round 3) ONE stray cell -- an unmasked fill value, a spike -- stretched the range,
// the whole field fell into one coarse bin and the sort fell back to eight passes (0.95 ms against 0.38).  Monotone as before:
// the zones are ordered, each map is monotone inside its zone.
constexpr unsigned RANGE_WOUT = 65536u;                                   // keys of each outer zone
constexpr unsigned RANGE_WIN = 16777215u - 2u * RANGE_WOUT;               // keys of the equalised inner zone: [WOUT, WOUT + WIN)
struct RangeMap { double lo, scale, mn, s_lo, hi, s_hi; const unsigned* tab; };      // tab: the slab's table, staged in LDS by the kernel
__device__ __forceinline__ void range_params(const double* __restrict__ mm, int slab, int negate, RangeMap& r)
{
    const double a = mm[4 * slab], b = mm[4 * slab + 1], c = mm[4 * slab + 2], d = mm[4 * slab + 3];    // min, max, robust low, robust high
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    r.mn = negate ? -b : a;
    const double mx = negate ? -a : b;
    r.lo = negate ? -d : c;
    r.hi = negate ? -c : d;
    const double w = r.hi - r.lo, wl = r.lo - r.mn, wh = mx - r.hi;
    r.scale = (w > 0.0 && w < inf) ? (double)RANGE_NB / w : (w == 0.0 ? 1e300 : 0.0);   // constant robust range: strays still leave it (x = +-huge); empty / infinite range: one bin
    r.s_lo = (wl > 0.0 && wl < inf) ? (double)RANGE_WOUT / wl : 0.0;
    r.s_hi = (wh > 0.0 && wh < inf) ? (double)RANGE_WOUT / wh : 0.0;
}
__device__ __forceinline__ int range_bin(double v, double lo, double scale, double& x)
{
    x = (v - lo) * scale;                                               // NaN (inf - inf, 0 * inf) -> bin 0 below
    int b = (int)fmin(fmax(x, 0.0), (double)(RANGE_NB - 1));
    return b;
}
// stage the slab's table in LDS (2 * RANGE_NB words); the caller synchronises before the first range_key
__device__ __forceinline__ RangeMap range_map(const PairSrc& src, int slab, unsigned* s_tab)
{
    RangeMap r;
    range_params(src.mm, slab, src.negate, r);
    const unsigned* g = src.rtab + (size_t)slab * 2 * RANGE_NB;
    for (int i = threadIdx.x; i < 2 * RANGE_NB; i += blockDim.x) s_tab[i] = g[i];
    r.tab = s_tab;
    return r;
}
template <typename K>
__device__ __forceinline__ unsigned range_key(K key, const RangeMap& m)
{
    if (key == KeyTraits<K>::invalid()) return RANGE_INVALID;
    const double v = KeyTraits<K>::decode(key);
    // the zone follows from x = (v - rlo) * scale itself: x < 0 below the robust range, x > 256 above it (a value a rounding
    // away from an end may stay inside: it then shares the end key, which keeps the map monotone); one compare on the hot path.
    // A degenerate robust range (a constant field with strays) has scale = 1e300 (range_params): x is 0 or +-huge.
    const double x = (v - m.lo) * m.scale;                              // NaN (inf - inf, 0 * inf) -> inner bin 0 below
    const double xc = fmin(fmax(x, 0.0), (double)RANGE_NB);
    const int b = (int)fmin(xc, (double)(RANGE_NB - 1));
    const unsigned first = m.tab[2 * b], width = m.tab[2 * b + 1];
    const double f = fmin(xc - (double)b, 1.0) * (double)width;          // (x - b in [0, 1]; the last bin takes x = 256)
    const unsigned off = (unsigned)f;
    unsigned k = first + (off < width ? off : width - 1u);
    // a cell outside the robust range (x NaN: stays in bin 0).  The test is made WAVE-uniform so that it stays a branch: written
    // per lane, the compiler predicates the two outer-zone maps into every key evaluation (+14 instructions per key and pass:
    // measured +30 us on the 6.48 M-pair sort); a handful of waves per plane ever take it.
    if (__ballot(xc != x) != 0ull) {
        if (x < 0.0) k = (unsigned)fmin(fmax((v - m.mn) * m.s_lo, 0.0), (double)(RANGE_WOUT - 1u));
        else if (x > (double)RANGE_NB) k = RANGE_WOUT + RANGE_WIN + (unsigned)fmin(fmax((v - m.hi) * m.s_hi, 0.0), (double)(RANGE_WOUT - 1u));
    }
    return k;
}
template <typename K, int MODE>
__device__ __forceinline__ unsigned digit_of(K key, int shift, const RangeMap& m)
{
    if (MODE == 0) return (unsigned)((key >> shift) & (K)255);
    return (range_key<K>(key, m) >> shift) & 255u;
}

// min, max and the ROBUST range of every plane from the per-block partials of K1 ([nslab][P][2]; a block = a contiguous piece
// of the plane): consecutive blocks are folded into at most 512 groups, dealt round-robin to the eight waves of the workgroup;
// every wave names its TWO smallest group minima and two largest group maxima (two rounds of a shuffle tree with retirement), and
// the T-th smallest / largest of those 16 candidates (T = 9; fewer than 72 groups: an eighth of them, at least 1 = the exact
// extrema) bounds the robust range: up to eight stray-holding groups are trimmed wherever they sit, and a candidate is never
// below the true T-th smallest group minimum, so the handful of cells outside [rlo, rhi] only grows by a few groups' worth when
// the extremes cluster in one wave's share.  (Exact selection by rank counting over all groups: 11-24 us per call; this: ~3.)
// out: [nslab][4] = min, max, rlo, rhi (all-NaN plane: NaN).
__global__ __launch_bounds__(512)
void k_range_bounds(const double* __restrict__ part, int P, double* __restrict__ out, unsigned* __restrict__ rhist, unsigned* __restrict__ tick)
{
    __shared__ double s_c[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (round 5) this kernel runs before the sampled population is counted: it clears the slab's coarse histogram and the arrival
    // tickets of the later kernels itself -- one hipMemsetAsync less in a chain of ~20 dependent launches
    if (tid < RANGE_NB) rhist[(size_t)blockIdx.x * RANGE_NB + tid] = 0u;
    if (tid < 4) tick[(size_t)blockIdx.x * 4 + tid] = 0u;
    const double* mp = part + (size_t)blockIdx.x * P * 2;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    const int per = (P + 511) / 512, ng = (P + per - 1) / per;           // groups of `per` consecutive blocks
    const int g = lane * 8 + wave;                                       // group of this thread: round-robin over the waves
    double a = inf, b = -inf;
    if (g < ng)
        for (int i = g * per; i < (g + 1) * per && i < P; ++i) { a = fmin(a, mp[2 * i]); b = fmax(b, mp[2 * i + 1]); }
    for (int r = 0; r < 2; ++r) {
        double lo = a, hi = b;
        for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
        if (lane == 0) { s_c[0][wave * 2 + r] = lo; s_c[1][wave * 2 + r] = hi; }
        const unsigned long long wa = __ballot(a == lo), wb = __ballot(b == hi);          // retire ONE holder of each extreme
        if (wa && lane == __builtin_ctzll(wa)) a = inf;
        if (wb && lane == __builtin_ctzll(wb)) b = -inf;
    }
    __syncthreads();
    if (tid < 16) {
        const double ca = s_c[0][tid], cb = s_c[1][tid];
        int below = 0, above = 0;                                         // strict rank among the 16 candidates, index as the tie-break
        for (int i = 0; i < 16; ++i) {
            const double x = s_c[0][i], y = s_c[1][i];
            below += (x < ca) || (x == ca && i < tid);
            above += (y > cb) || (y == cb && i < tid);
        }
        const int T = ng >= 72 ? 9 : (ng / 8 > 0 ? ng / 8 : 1);
        double* o = out + (size_t)blockIdx.x * 4;
        if (below == 0) o[0] = ca;
        if (above == 0) o[1] = cb;
        if (below == T - 1) o[2] = ca;
        if (above == T - 1) o[3] = cb;
    }
    __syncthreads();                                                      // (same workgroup: the stores above are visible to thread 0 below)
    if (tid == 0) {
        double* o = out + (size_t)blockIdx.x * 4;
        double lo0 = o[0], hi0 = o[1], c = o[2], d = o[3];
        if (lo0 == inf && hi0 == -inf) { lo0 = hi0 = c = d = __longlong_as_double(0x7ff8000000000000LL); }     // no valid cell
        else {
            if (!(c >= lo0) || c == inf || c == -inf) c = lo0;           // candidates without a valid cell carry +inf / -inf: fall back to the extrema
            if (!(d <= hi0) || d == inf || d == -inf) d = hi0;
            if (!(c <= d)) { c = lo0; d = hi0; }
        }
        o[0] = lo0; o[1] = hi0; o[2] = c; o[3] = d;
    }
}

// population of the RANGE_NB coarse bins (valid cells only, the validity rule of load_pairs); hist zeroed by the caller
template <typename TQ, typename TM>
__global__ __launch_bounds__(256)
void k_range_hist(int64_t n, const PairSrc src, unsigned* __restrict__ hist)
{
    __shared__ unsigned s_h[RANGE_NB];
    for (int i = threadIdx.x; i < RANGE_NB; i += 256) s_h[i] = 0;
    RangeMap rp;
    range_params(src.mm, blockIdx.y, src.negate, rp);
    const double lo = rp.lo, hi = rp.hi, scale = rp.scale;
    const TQ* q = (const TQ*)src.q + (size_t)blockIdx.y * n;
    const TM* mask = src.mask ? (const TM*)src.mask + (size_t)blockIdx.y * src.mask_stride : nullptr;
    __syncthreads();
    constexpr int U = 8;
    const int64_t samp = n > (int64_t)256 * RANGE_SAMPLE * 64 ? RANGE_SAMPLE : 1;             // small planes: every cell
    const int64_t nseg = (n + 256 * samp - 1) / (256 * samp);                                 // the first 256 cells of every 256 * samp
    for (int64_t s0 = (int64_t)blockIdx.x * U; s0 < nseg; s0 += (int64_t)gridDim.x * U) {
        TQ qv[U]; TM mv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = (s0 + u) * 256 * samp + threadIdx.x; qv[u] = q[i < n ? i : n - 1]; }
        if (mask) {
#pragma unroll
            for (int u = 0; u < U; ++u) { const int64_t i = (s0 + u) * 256 * samp + threadIdx.x; mv[u] = mask[i < n ? i : n - 1]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = (s0 + u) * 256 * samp + threadIdx.x;
            const double v = src.negate ? -(double)qv[u] : (double)qv[u];
            if (s0 + u < nseg && i < n && v >= lo && v <= hi && (!mask || mv[u] == (TM)1)) { double x; atomicAdd(&s_h[range_bin(v, lo, scale, x)], 1u); }   // (the robust range only; NaN fails both compares)
        }
    }
    __syncthreads();
    unsigned* h = hist + (size_t)blockIdx.y * RANGE_NB;
    for (int i = threadIdx.x; i < RANGE_NB; i += 256) if (s_h[i]) atomicAdd(&h[i], s_h[i]);
}

// counts -> (first range key, number of range keys) per coarse bin: HALF of the 2^24 - 1 keys are dealt out evenly (a bin the
// sample missed still resolves 2^-23 of the range), the other half in proportion to the sampled counts (rounded down: the
// last key used is at most 2^24 - 2)
__global__ __launch_bounds__(RANGE_NB)
void k_range_table(const unsigned* __restrict__ hist, unsigned* __restrict__ rtab)
{
    __shared__ unsigned s_w[(RANGE_NB + 63) / 64];
    __shared__ unsigned long long s_tot;
    const int b = threadIdx.x, lane = b & 63, wave = b >> 6;
    const unsigned c = hist[(size_t)blockIdx.x * RANGE_NB + b];
    unsigned long long t = c;
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (b == 0) s_tot = 0ull;
    __syncthreads();
    if (lane == 0) atomicAdd(&s_tot, t);
    __syncthreads();
    const unsigned long long tot = s_tot, even = (RANGE_WIN / 2u) / RANGE_NB, budget = (unsigned long long)RANGE_WIN - even * RANGE_NB;
    const unsigned width = (unsigned)even + (tot ? (unsigned)((unsigned long long)c * budget / tot) : 0u);
    unsigned x = width;                                                 // exclusive scan of the widths
    for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    unsigned first = RANGE_WOUT + x - width;                            // behind the lower outer zone
    for (int w = 0; w < wave; ++w) first += s_w[w];
    rtab[((size_t)blockIdx.x * RANGE_NB + b) * 2] = first;
    rtab[((size_t)blockIdx.x * RANGE_NB + b) * 2 + 1] = width;
}

// number of valid cells = position of the first KEY_INVALID in the sorted keys (one thread:
// a per-wave atomic counter while building the keys serialised 100k atomics on one address = 1.1 ms)
template <typename K>
__global__ void k_count_valid(const K* __restrict__ keys, int64_t n, unsigned* __restrict__ nvalid)
{
    if (threadIdx.x != 0) return;
    keys += (size_t)blockIdx.x * n; nvalid += blockIdx.x;
    int64_t lo = 0, hi = n;                        // first index with keys[idx] == KEY_INVALID
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < KeyTraits<K>::invalid()) lo = mid + 1; else hi = mid; }
    *nvalid = (unsigned)lo;
}

// peer mask of lanes holding the same 8-bit digit (only lanes in `valid`)
__device__ __forceinline__ unsigned long long digit_peers(unsigned d, unsigned long long valid)
{
    unsigned long long m = valid;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    return m;
}

// tile = 4 waves x TILE elements (one block); digit-major tile histogram hist[d][tile].
// Counting needs no ranks: one returnless ds_add_u32 per key on per-wave counters (the ballot ranking
// of the scatter costs ~60 VALU instructions per 64 keys and made this kernel ALU-bound); a round
// whose 64 digits are all equal -- sorted or constant data -- is added once by one lane.
template <typename K, bool FIRST = false, typename TQ = double, typename TM = double, int MODE = 0, int TR = XC_TILE_ROUNDS>
__global__ __launch_bounds__(256)
void k_radix_hist(const K* __restrict__ keys, int64_t n, int shift, int ntiles, unsigned* __restrict__ hist, const PairSrc src)
{
    constexpr int TILE_ROUNDS = TR, TILE = 64 * TR, BTILE = 4 * TILE;      // (the tile of THIS instance: small_tiles() below)
    __shared__ unsigned s_cnt[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = blockIdx.x;
    __shared__ unsigned s_rt[MODE == 1 ? 2 * RANGE_NB : 1];
    RangeMap rm = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, nullptr};
    if (MODE == 1) rm = range_map(src, blockIdx.y, s_rt);
    keys += (size_t)blockIdx.y * n; hist += (size_t)blockIdx.y * 256 * ntiles;
    for (int i = lane; i < 256; i += 64) s_cnt[wave][i] = 0;
    if (MODE == 1) __syncthreads();
    if constexpr (FIRST) {
        // pass 0: the keys do not exist yet -- encode them from the tracer (the order inside the tile is irrelevant here)
        const TQ* q = (const TQ*)src.q + (size_t)blockIdx.y * n;
        const TM* mask = src.mask ? (const TM*)src.mask + (size_t)blockIdx.y * src.mask_stride : nullptr;
        const int64_t base = t * BTILE + (int64_t)wave * TILE;
        K kreg[TILE_ROUNDS];
        double dummy[TILE_ROUNDS];
        load_pairs<TQ, TM, K, TILE_ROUNDS, false>(q, mask, nullptr, XC_DA_NONE, src.nx, src.negate, base, lane, n, kreg, dummy);
#pragma unroll
        for (int r = 0; r < TILE_ROUNDS; ++r) {
            const int64_t i = base + r * 64 + lane;
            const bool valid = i < n;
            const unsigned d = digit_of<K, MODE>(kreg[r], shift, rm);
            const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
            if (base + TILE <= n && __ballot(d != d0) == 0ull) { if (lane == 0) atomicAdd(&s_cnt[wave][d0], 64u); }
            else if (valid) atomicAdd(&s_cnt[wave][d], 1u);
        }
        __syncthreads();
        hist[(size_t)threadIdx.x * ntiles + t] = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        return;
    }
    // counting does not care about the order inside the tile: 16-byte loads, KPL keys per lane and load
    constexpr int KPL = 16 / (int)sizeof(K);
    struct alignas(16) Pack { K k[KPL]; };
    const int64_t base = t * BTILE + (int64_t)wave * TILE;
    K kreg[TILE_ROUNDS];                                   // all loads of the wave's part in flight at once
    const bool full = base + TILE <= n;
    if (full) {
        const Pack* kp = (const Pack*)(keys + base);       // workspace is 256-byte aligned, base a multiple of 1024
#pragma unroll
        for (int r = 0; r < TILE_ROUNDS / KPL; ++r) {
            const Pack u = kp[r * 64 + lane];
#pragma unroll
            for (int c = 0; c < KPL; ++c) kreg[KPL * r + c] = u.k[c];
        }
    } else {
#pragma unroll
        for (int r = 0; r < TILE_ROUNDS / KPL; ++r)
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                const int64_t i = base + (int64_t)(r * 64 + lane) * KPL + c;
                kreg[KPL * r + c] = i < n ? keys[i] : (K)0;
            }
    }
#pragma unroll
    for (int r = 0; r < TILE_ROUNDS; ++r) {
        const int64_t i = base + (int64_t)((r / KPL) * 64 + lane) * KPL + (r % KPL);
        const bool valid = full || i < n;
        const unsigned d = digit_of<K, MODE>(kreg[r], shift, rm);
        const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
        if (full && __ballot(d != d0) == 0ull) { if (lane == 0) atomicAdd(&s_cnt[wave][d0], 64u); }
        else if (valid) atomicAdd(&s_cnt[wave][d], 1u);
    }
    __syncthreads();
    const int d = threadIdx.x;
    hist[(size_t)d * ntiles + t] = s_cnt[0][d] + s_cnt[1][d] + s_cnt[2][d] + s_cnt[3][d];
}

// exclusive scan of each digit's row over the tiles (one block per digit); row total out
__global__ __launch_bounds__(1024)
void k_radix_scan_rows(unsigned* __restrict__ hist, int ntiles, unsigned* __restrict__ totals)
{
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_carry;
    unsigned* row = hist + ((size_t)blockIdx.y * 256 + blockIdx.x) * ntiles;
    totals += (size_t)blockIdx.y * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int b = 0; b < ntiles; b += 1024) {
        const int i = b + tid;
        const unsigned v = i < ntiles ? row[i] : 0u;
        unsigned x = v;                                        // inclusive wave scan
        for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        unsigned off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        if (i < ntiles) row[i] = off + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = off + x;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = s_carry;
}

// Scatter of one block tile (4 waves x TILE elements).  The tile is first sorted by digit in LDS
// (stable: wave-major, then round, then lane = element order), then written out position by position:
// consecutive LDS positions with the same digit go to consecutive global addresses, so the stores
// of a wave cover runs of ~BTILE/256 elements instead of 64 unrelated 8-byte targets.
template <typename K, bool FIRST = false, typename TQ = double, typename TM = double, int MODE = 0, int TR = XC_TILE_ROUNDS>
__global__ __launch_bounds__(256)
void k_radix_scatter(const K* __restrict__ kin, const double* __restrict__ vin,
                     K* __restrict__ kout, double* __restrict__ vout, int64_t n, int shift,
                     int ntiles, const unsigned* __restrict__ hist, const unsigned* __restrict__ totals, int inline_scan,
                     const PairSrc src)
{
    constexpr int TILE_ROUNDS = TR, TILE = 64 * TR, BTILE = 4 * TILE;
    extern __shared__ unsigned long long s_dyn[];
    K* s_k = (K*)s_dyn;                                        // [BTILE] staging: keys first, then the payload
    double* s_v = (double*)s_dyn;
    unsigned* s_cnt = (unsigned*)(s_dyn + BTILE);              // [4][256] per-wave digit counts -> start offsets
    unsigned* s_gbase = s_cnt + 4 * 256;                       // [256] global position minus tile-local position
    unsigned* s_wsum = s_gbase + 256;                          // [8]
    unsigned char* s_dig = (unsigned char*)(s_wsum + 8);       // [BTILE] digit of the element at every tile-local position
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t = blockIdx.x;
    __shared__ unsigned s_rt[MODE == 1 ? 2 * RANGE_NB : 1];
    RangeMap rm = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, nullptr};
    if (MODE == 1) { rm = range_map(src, blockIdx.y, s_rt); __syncthreads(); }
    { const size_t so = (size_t)blockIdx.y * n; kin += so; vin += so; kout += so; vout += so; }
    hist += (size_t)blockIdx.y * 256 * ntiles; totals += (size_t)blockIdx.y * 256;
    for (int d = lane; d < 256; d += 64) s_cnt[wave * 256 + d] = 0;
    const int64_t tbase = t * BTILE;
    const int64_t base = tbase + (int64_t)wave * TILE;
    K kreg[TILE_ROUNDS];                                       // the whole part's loads in flight at once
    double vreg[TILE_ROUNDS];
    unsigned short lrank[TILE_ROUNDS];
    unsigned char dreg[TILE_ROUNDS];                           // the digit, computed once (the range key costs ~10 VALU operations)
    if constexpr (FIRST) {
        // pass 0 builds its pairs from the tracer / mask / dA (kin / vin do not exist yet; their slab offset above is harmless)
        const TQ* q = (const TQ*)src.q + (size_t)blockIdx.y * n;
        const TM* mask = src.mask ? (const TM*)src.mask + (size_t)blockIdx.y * src.mask_stride : nullptr;
        const double* dA = src.dA ? src.dA + (size_t)blockIdx.y * src.dA_stride : nullptr;
        load_pairs<TQ, TM, K, TILE_ROUNDS, true>(q, mask, dA, src.dA_rank, src.nx, src.negate, base, lane, n, kreg, vreg);
    } else {
#pragma unroll
        for (int r = 0; r < TILE_ROUNDS; ++r) {
            const int64_t i = base + r * 64 + lane;
            kreg[r] = i < n ? kin[i] : (K)0;
            vreg[r] = i < n ? vin[i] : 0.0;
        }
    }
    // rank of every element among the wave's elements with the same digit
#pragma unroll
    for (int r = 0; r < TILE_ROUNDS; ++r) {
        const int64_t i = base + r * 64 + lane;
        const bool valid = i < n;
        const unsigned d = valid ? digit_of<K, MODE>(kreg[r], shift, rm) : 0u;
        dreg[r] = (unsigned char)d;
        const unsigned long long peers = digit_peers(d, __ballot(valid));
        const unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        unsigned pos = 0;
        if (valid) pos = s_cnt[wave * 256 + d] + rank;          // all peers read the same counter first ...
        if (valid && rank == 0) s_cnt[wave * 256 + d] += (unsigned)__popcll(peers);   // ... then the leader advances it
        lrank[r] = (unsigned short)pos;
    }
    __syncthreads();
    {   // thread d: tile-local start of digit d (exclusive scan over digits), per-wave starts, global base
        const int d = tid;
        const unsigned c0 = s_cnt[d], c1 = s_cnt[256 + d], c2 = s_cnt[512 + d], c3 = s_cnt[768 + d];
        // few tiles per plane (stacks of small planes): the scan over the tiles is done right here on the raw counts,
        // the separate row-scan launch (one 1024-thread block per digit and plane) is skipped
        unsigned gtot, before = 0;
        if (inline_scan) {
            gtot = 0;
            for (int tt = 0; tt < ntiles; ++tt) { const unsigned c = hist[(size_t)d * ntiles + tt]; gtot += c; before += tt < t ? c : 0u; }
        } else { gtot = totals[d]; before = hist[(size_t)d * ntiles + t]; }
        const unsigned tot = c0 + c1 + c2 + c3;
        unsigned x = tot, gx = gtot;                       // two exclusive scans over the digits: tile-local and global
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned y = __shfl_up(x, o), gy = __shfl_up(gx, o);
            if (lane >= o) { x += y; gx += gy; }
        }
        if (lane == 63) { s_wsum[wave] = x; s_wsum[4 + wave] = gx; }
        __syncthreads();
        unsigned start = x - tot, gbase = gx - gtot;
        for (int w = 0; w < wave; ++w) { start += s_wsum[w]; gbase += s_wsum[4 + w]; }
        s_cnt[d] = start; s_cnt[256 + d] = start + c0; s_cnt[512 + d] = start + c0 + c1; s_cnt[768 + d] = start + c0 + c1 + c2;
        s_gbase[d] = gbase + before - start;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TILE_ROUNDS; ++r) {
        const int64_t i = base + r * 64 + lane;
        const unsigned d = dreg[r];
        lrank[r] = (unsigned short)(s_cnt[wave * 256 + d] + lrank[r]);      // tile-local position
        if (i < n) { s_k[lrank[r]] = kreg[r]; s_dig[lrank[r]] = dreg[r]; }
    }
    __syncthreads();
    const int64_t left = n - tbase;
    const int cnt = left < BTILE ? (int)left : BTILE;
    unsigned gpos[TILE_ROUNDS];
#pragma unroll
    for (int r = 0; r < TILE_ROUNDS; ++r) {
        const int p = r * 256 + tid;
        if (p < cnt) {
            const K key = s_k[p];
            gpos[r] = s_gbase[s_dig[p]] + (unsigned)p;
            kout[gpos[r]] = key;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TILE_ROUNDS; ++r) {
        const int64_t i = base + r * 64 + lane;
        if (i < n) s_v[lrank[r]] = vreg[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TILE_ROUNDS; ++r) {
        const int p = r * 256 + tid;
        if (p < cnt) vout[gpos[r]] = s_v[p];
    }
}

// ---- after the three range-key passes: finish every run of equal range key that is out of order, and count the valid
// cells.  IN PLACE; one block per FIX_C consecutive positions [a, b) OWNS the runs whose first cell (head) lies there, to
// their end -- a run belongs to exactly one block; the block's window is [a - 1, a - 1 + FIX_W).
//   1. head[i] / end[i] of the run of every window cell: a prefix-max / suffix-min scan over the head positions.
//   2. Every inversion (a cell whose full key is smaller than its left neighbour's inside one run) marks its run dirty;
//      if that run is owned and longer than FIX_RUN the flag sends the whole stack to the eight-pass path.  Runs without
//      an inversion -- ties of any length -- are never touched.
//   3. Every cell of a dirty owned run counts the cells of its run that sort before it (smaller key, or equal key and
//      earlier position: a stable rank, at most FIX_RUN reads, ~2 on average) and, if its place changes, writes ITSELF
//      (key and payload from its registers) to head + rank.  The writes of a run are a permutation of the run; a
//      neighbouring block reading such a cell meanwhile only derives its range key from it, which the run shares.
//   The step from the last valid key to the first dropped one (always a head) gives nvalid.
constexpr int FIX_C = 1024, FIX_RUN = 128, FIX_NL = 5, FIX_W = FIX_NL * 256;      // window = 1 + FIX_C + 255 cells
template <typename K>
__global__ __launch_bounds__(256)
void k_fix_runs(K* __restrict__ keys, double* __restrict__ vals, int64_t n, unsigned* __restrict__ flag,
                unsigned* __restrict__ nvalid, const PairSrc src)
{
    __shared__ K s_k[FIX_W];
    __shared__ unsigned s_d[FIX_W];
    __shared__ unsigned short s_h[FIX_W], s_e[FIX_W];
    __shared__ unsigned char s_dirty[FIX_W];
    __shared__ int s_wh[4], s_we[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ unsigned s_rt[2 * RANGE_NB];
    const RangeMap rm = range_map(src, blockIdx.y, s_rt);
    keys += (size_t)blockIdx.y * n; vals += (size_t)blockIdx.y * n;
    const int64_t a = (int64_t)blockIdx.x * FIX_C, w0 = a - 1;               // window position i <-> cell w0 + i; owned heads: i in [1, FIX_C]
    K kr[FIX_NL]; double vr[FIX_NL];                                          // every load issued before the first use
#pragma unroll
    for (int c = 0; c < FIX_NL; ++c) {
        int64_t g = w0 + tid + 256 * c;
        g = g < 0 ? 0 : (g < n ? g : n - 1);
        kr[c] = keys[g]; vr[c] = vals[g];
    }
    __syncthreads();                                                           // the range table is in LDS
#pragma unroll
    for (int c = 0; c < FIX_NL; ++c) {
        const int i = tid + 256 * c;
        const int64_t g = w0 + i;
        const bool in = g >= 0 && g < n;
        s_k[i] = in ? kr[c] : (K)0;
        s_d[i] = in ? range_key<K>(kr[c], rm) : 0xFFFFFFF0u + (unsigned)(i & 1);          // no cell: equal to no neighbour
        s_dirty[i] = 0;
    }
    __syncthreads();
    // ---- 1. heads: thread t scans the cells [5t, 5t + 5); last head at or before i (0: the run began before the window),
    //         first head after i (FIX_W: the run leaves the window)
    {
        const int i0 = FIX_NL * tid;
        bool hd[FIX_NL];
        int lastl = -1, firstl = FIX_W;
#pragma unroll
        for (int c = 0; c < FIX_NL; ++c) {
            const int i = i0 + c;
            hd[c] = i > 0 && s_d[i] != s_d[i - 1];
            if (hd[c]) { lastl = i; if (firstl == FIX_W) firstl = i; }
        }
        int pm = lastl, sm = firstl;                                          // inclusive prefix max / suffix min over the lanes
        for (int o = 1; o < 64; o <<= 1) {
            const int x = __shfl_up(pm, o), y = __shfl_down(sm, o);
            if (lane >= o) pm = x > pm ? x : pm;
            if (lane + o < 64) sm = y < sm ? y : sm;
        }
        if (lane == 63) s_wh[wave] = pm;
        if (lane == 0) s_we[wave] = sm;
        __syncthreads();
        int before = __shfl_up(pm, 1), after = __shfl_down(sm, 1);            // exclusive: heads in earlier / later lanes
        if (lane == 0) before = -1;
        if (lane == 63) after = FIX_W;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before = s_wh[w] > before ? s_wh[w] : before;
            if (w > wave) after = s_we[w] < after ? s_we[w] : after;
        }
        int run_h = before < 0 ? 0 : before;
#pragma unroll
        for (int c = 0; c < FIX_NL; ++c) { if (hd[c]) run_h = i0 + c; s_h[i0 + c] = (unsigned short)run_h; }
        int run_e = after;
#pragma unroll
        for (int c = FIX_NL - 1; c >= 0; --c) { s_e[i0 + c] = (unsigned short)run_e; if (hd[c]) run_e = i0 + c; }
    }
    __syncthreads();
    // ---- 2. inversions mark their run; an owned run longer than FIX_RUN cannot be repaired here
    bool bad = false;
#pragma unroll
    for (int c = 0; c < FIX_NL; ++c) {
        const int i = tid + 256 * c;
        if (i == 0) continue;
        const int h = s_h[i];
        if (h == i) {                                                          // a head
            if (s_d[i] == RANGE_INVALID && i <= FIX_C && w0 + i < n) nvalid[blockIdx.y] = (unsigned)(w0 + i);
            continue;
        }
        if (!(s_k[i] < s_k[i - 1]) || h > FIX_C) continue;                    // no inversion, or the run is the right neighbour's
        if (h < 1) {                                                           // the run began before the window: the left neighbour's, who sees this
            if (i >= FIX_RUN) bad = true;                                      // cell only if the run is short -- and this far in, it is not
            continue;
        }
        if ((int)s_e[i] - h > FIX_RUN) bad = true; else s_dirty[h] = 1;
    }
    if (blockIdx.x == 0 && tid == 0 && s_d[1] == RANGE_INVALID) nvalid[blockIdx.y] = 0u;       // only dropped cells
    if (w0 + FIX_C >= n - 1 && tid == 0) {                                     // the block that holds the last cell: no dropped cell at all
        const int il = (int)(n - 1 - w0);
        if (s_d[il] != RANGE_INVALID) nvalid[blockIdx.y] = (unsigned)n;
    }
    if (__syncthreads_or(bad)) {                                               // the stack goes to the eight-pass path: nothing else to do here
        // (pinned host memory.  A plain system-scope STORE, not a read-modify-write: every writer stores the same 1, and an atomic OR on
        // host memory needs PCIe AtomicOps, which pass-through / virtualised hosts may not route -- round-5 advisor)
        if (tid == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // ---- 3. stable rank inside the run; a cell whose place changes writes itself there
#pragma unroll
    for (int c = 0; c < FIX_NL; ++c) {
        const int i = tid + 256 * c;
        const int h = s_h[i];
        if (h < 1 || h > FIX_C || !s_dirty[h] || w0 + i >= n) continue;
        const int e = s_e[i];
        const K k = kr[c];
        int rank = 0;
        for (int j = h; j < e; ++j) { const K kj = s_k[j]; rank += (kj < k) || (kj == k && j < i); }
        if (h + rank != i) { keys[w0 + h + rank] = k; vals[w0 + h + rank] = vr[c]; }
    }
}

// ---- inclusive f64 scan (cumulative area of the sorted state): block sums, their exclusive scan, then
// the block-local scan plus block offset.  Both passes run the same arithmetic, so the sums of pass 1
// are exactly the last values pass 2 produces (read 2x, write 1x; the payload is never re-written).
// A wave owns 512 consecutive values: 4 rounds of coalesced 16-byte accesses, one wave scan per round.
template <bool FINAL>
__global__ __launch_bounds__(256)
void k_scan_local(const double* __restrict__ in, double* __restrict__ out, int64_t n, double* __restrict__ bsum)
{
    __shared__ double s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    in += (size_t)blockIdx.y * n; bsum += (size_t)blockIdx.y * gridDim.x;
    if (FINAL) out += (size_t)blockIdx.y * n;
    const int64_t wbase = (int64_t)blockIdx.x * 2048 + wave * 512;
    double a[4], b[4];
    if (wbase + 512 <= n) {
        const double2* in2 = (const double2*)(in + wbase);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const double2 u = in2[r * 64 + lane]; a[r] = u.x; b[r] = u.y; }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = wbase + (r * 64 + lane) * 2;
            a[r] = i < n ? in[i] : 0.0; b[r] = i + 1 < n ? in[i + 1] : 0.0;
        }
    }
    double carry = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double pair = a[r] + b[r];
        double x = pair;                                       // inclusive wave scan of the pair sums
        for (int o = 1; o < 64; o <<= 1) { const double y = __shfl_up(x, o); if (lane >= o) x += y; }
        const double before = carry + (x - pair);
        a[r] = before + a[r]; b[r] = before + pair;
        carry += __shfl(x, 63);
    }
    if (lane == 63) s_w[wave] = carry;
    __syncthreads();
    double off = 0.0;
    for (int w = 0; w < wave; ++w) off += s_w[w];
    if (FINAL) {
        const double boff = bsum[blockIdx.x];
        if (wbase + 512 <= n) {
            double2* out2 = (double2*)(out + wbase);
#pragma unroll
            for (int r = 0; r < 4; ++r) out2[r * 64 + lane] = make_double2((off + a[r]) + boff, (off + b[r]) + boff);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = wbase + (r * 64 + lane) * 2;
                if (i < n) out[i] = (off + a[r]) + boff;
                if (i + 1 < n) out[i + 1] = (off + b[r]) + boff;
            }
        }
    } else if (tid == 255) bsum[blockIdx.x] = off + carry;
}

__global__ __launch_bounds__(1024)
void k_scan_bsums(double* __restrict__ bsum, int nb)        // exclusive scan in place, one block
{
    __shared__ double s_w[16];
    __shared__ double s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bsum += (size_t)blockIdx.x * nb;
    if (tid == 0) s_carry = 0.0;
    __syncthreads();
    for (int b = 0; b < nb; b += 1024) {
        const int i = b + tid;
        const double v = i < nb ? bsum[i] : 0.0;
        double x = v;
        for (int o = 1; o < 64; o <<= 1) { const double y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        double off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        if (i < nb) bsum[i] = off + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = off + x;
        __syncthreads();
    }
}

template <typename K>
__global__ __launch_bounds__(256)
void k_unkey(const K* __restrict__ keys, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    keys += (size_t)blockIdx.y * n; out += (size_t)blockIdx.y * n;
    if (i < n) out[i] = KeyTraits<K>::decode(keys[i]);
}

// Q_exact(A_j) = q_sorted[min(searchsorted(acum[:nvalid], A_j, 'right'), nvalid-1)]
template <typename K>
__device__ __forceinline__ void profile_body(const K* __restrict__ keys, const double* __restrict__ acum,
                                             const unsigned* __restrict__ nvalid, const double* __restrict__ targets, int J,
                                             double* __restrict__ Q, int64_t ncell, int bx)
{
    const int j = bx * 256 + threadIdx.x;
    if (j >= J) return;
    keys += (size_t)blockIdx.y * ncell; acum += (size_t)blockIdx.y * ncell; Q += (size_t)blockIdx.y * J;
    const int64_t n = nvalid[blockIdx.y];
    if (n == 0) { Q[j] = __longlong_as_double(0x7ff8000000000000LL); return; }
    const double a = targets[j];
    int64_t lo = 0, hi = n;                        // first index with acum[idx] > a
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (acum[mid] <= a) lo = mid + 1; else hi = mid; }
    if (lo > n - 1) lo = n - 1;
    Q[j] = KeyTraits<K>::decode(keys[lo]);
}
template <typename K>
__global__ __launch_bounds__(256)
void k_profile(const K* __restrict__ keys, const double* __restrict__ acum,
               const unsigned* __restrict__ nvalid, const double* __restrict__ targets, int J,
               double* __restrict__ Q, int64_t ncell)
{
    profile_body<K>(keys, acum, nvalid, targets, J, Q, ncell, (int)blockIdx.x);
}

// BPE-like integral: sum_i q_i * z*(A_i - dA_i/2) * dA_i with z* = np.interp(A, tbl, coord)
template <typename K>
__global__ __launch_bounds__(256)
void k_bpe(const K* __restrict__ keys, const double* __restrict__ vals,
           const double* __restrict__ acum, const unsigned* __restrict__ nvalid,
           const double* __restrict__ tbl, const double* __restrict__ coord, int ntbl, double* __restrict__ part,
           int64_t ncell, unsigned* __restrict__ tick, double* __restrict__ out,
           const double* __restrict__ targets, int J, double* __restrict__ Q, int nprof)
{
    // (round 5) the first `nprof` workgroups are the profile Q(A_j) of this plane (k_profile's body): both only read the sorted
    // state, so the J binary searches -- ~19 dependent reads, 7 us as a launch of their own -- run beside the integral (cfg5:
    // -6 us, same-box A/B).  Folding a seam into the LAST-ARRIVING workgroup of the kernel before it does NOT pay: tried on the
    // range table (into k_range_hist) and the block-sum scan (into the first scan pass) -- ticket round trip + agent-scope
    // re-reads cost the 3-4 us the launch boundary costs; 3 / 16 / 64-plane stacks +1..3 us, reverted (profiles/r05_notes.md).
    if ((int)blockIdx.x < nprof) { profile_body<K>(keys, acum, nvalid, targets, J, Q, ncell, (int)blockIdx.x); return; }
    const int bx = (int)blockIdx.x - nprof, nbx = (int)gridDim.x - nprof;
    { const size_t so = (size_t)blockIdx.y * ncell; keys += so; vals += so; acum += so; }
    part += (size_t)blockIdx.y * nbx;
    const int64_t n = nvalid[blockIdx.y];
    // the table goes into LDS when it fits (nz or ny entries): the bracket search is a chain of ~log2(ntbl) dependent reads per
    // cell, ~1 us each from global memory (20 us per launch on the cfg5 stand-in), ~0.1 us from LDS
    constexpr int BPE_TBL = 2048;
    __shared__ double s_tbl[2 * BPE_TBL];
    const bool in_lds = ntbl <= BPE_TBL;
    if (in_lds) {
        for (int i = threadIdx.x; i < ntbl; i += 256) { s_tbl[i] = tbl[i]; s_tbl[BPE_TBL + i] = coord[i]; }
        __syncthreads();
    }
    const bool tinc = tbl[ntbl - 1] > tbl[0];
    double sum = 0.0;
    // BU cells per thread and round: their three loads each are issued before the first bracket search starts (one cell at a time --
    // load, ~log2(ntbl) dependent LDS reads, a division, next load -- was a chain of seven memory round trips per thread on the cfg5
    // planes: 19 us for a kernel that moves 32 MB); the terms are still added in cell order
    constexpr int BU = 4;
    auto walk = [&](auto X, auto F) {
        const int64_t step = (int64_t)nbx * 256;
        for (int64_t i0 = (int64_t)bx * 256 + threadIdx.x; i0 < n; i0 += BU * step) {
            double ac[BU], va[BU]; K ke[BU];
#pragma unroll
            for (int u = 0; u < BU; ++u) {
                const int64_t i = i0 + u * step, ic = i < n ? i : n - 1;
                ac[u] = acum[ic]; va[u] = vals[ic]; ke[u] = keys[ic];
            }
#pragma unroll
            for (int u = 0; u < BU; ++u) {
                if (i0 + u * step >= n) break;
                const double a = ac[u] - 0.5 * va[u];
                double z;
                if (a >= X(ntbl - 1)) z = F(ntbl - 1);
                else if (a <= X(0)) z = F(0);
                else {
                    int lo = 0, hi = ntbl - 1;
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a >= X(mid)) lo = mid; else hi = mid; }
                    z = F(lo) + (F(lo + 1) - F(lo)) * (a - X(lo)) / (X(lo + 1) - X(lo));
                }
                sum += KeyTraits<K>::decode(ke[u]) * z * va[u];
            }
        }
    };
    if (in_lds) walk([&](int k) { return s_tbl[tinc ? k : ntbl - 1 - k]; }, [&](int k) { return s_tbl[BPE_TBL + (tinc ? k : ntbl - 1 - k)]; });
    else walk([&](int k) { return tinc ? tbl[k] : tbl[ntbl - 1 - k]; }, [&](int k) { return tinc ? coord[k] : coord[ntbl - 1 - k]; });
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = sum;
    __syncthreads();
    // (round 5) the block that arrives last sums the plane's partials in block order -- one launch less at the end of the chain.  Every
    // hand-off word is an agent-scope 8-byte atomic on both sides (a partial is ONE store of one lane, the ticket returns the order of
    // arrival): MI355X_MICROARCH.md, valid forms; the sum is taken in a fixed order, whoever arrives last.
    __shared__ unsigned s_last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(part + bx, s[0] + s[1] + s[2] + s[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(tick + (size_t)blockIdx.y * 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nbx - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    const int np = nbx, per = (np + 63) / 64, i0 = (int)threadIdx.x * per;     // one wave: lane l sums its contiguous share, then a fixed xor tree
    double t = 0.0;
    for (int i = i0; i < i0 + per && i < np; ++i) t += __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (threadIdx.x == 0) {
        out[blockIdx.y] = t;
        __hip_atomic_store(tick + (size_t)blockIdx.y * 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch (a stack that is sorted again runs this kernel twice)
    }
}

}  // namespace

// Workspace layout (device), every array with a leading slab dim: keys A/B, vals A/B, hist, totals, nvalid, bsums, bpe parts,
// K1 partials + min/max + the "not sorted" flag of the range-key path
#ifndef XC_BPE_BLOCKS
#define XC_BPE_BLOCKS 256
#endif
constexpr int BPE_BLOCKS = XC_BPE_BLOCKS;
// Tile of the radix passes: 4 waves x 64 lanes x TILE_ROUNDS pairs.  Sixteen rounds per lane keep a large sort's per-tile costs
// (digit scans, the histogram row per tile) small; a stack with few tiles -- the cfg5 stand-in: 3 planes x 110 tiles on 256 CUs --
// fills the chip only with the half tile (measured, r05: cfg5 0.192 -> 0.171 ms, one such plane 0.131 -> 0.108 ms with 8 rounds; 16 planes of
// 256 x 512 -- 512 tiles -- 0.178 -> 0.198 ms, 64 planes 25 % slower: the choice is by the number of tiles, not a constant).
constexpr int TILE_ROUNDS_SMALL = 8, BTILE_SMALL = 4 * 64 * TILE_ROUNDS_SMALL;
#ifndef XC_SMALL_TILES_MAX
#define XC_SMALL_TILES_MAX 400
#endif
static inline bool small_tiles(int64_t n, int64_t nslab) { return nslab * ((n + BTILE - 1) / BTILE) <= XC_SMALL_TILES_MAX; }

size_t sort_workspace_bytes(int64_t n, int64_t nslab)
{
    const int64_t ntiles = (n + BTILE_SMALL - 1) / BTILE_SMALL;        // (the larger of the two tilings)
    const int64_t nb = (n + 2047) / 2048;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t S = (size_t)nslab;
    return 4 * al(S * n * 8) + al(S * 256 * ntiles * 4) + al(S * 256 * 4) + al(S * 4) + al(S * nb * 8) + al(S * BPE_BLOCKS * 8) +
           al(S * kMinmaxBlocks * 2 * 8) + al(S * 4 * 8) + al(S * 4 * 4) + al(S * RANGE_NB * 4) + al(S * RANGE_NB * 8);
}

template <typename TQ, typename K>
static int sort_profile_typed(xc_ctx* ctx, const TQ* q, int q_dtype, const void* mask, int mask_dtype, int mask_per_slab,
                              const double* dA, int dA_rank, int64_t nslab, int64_t ny, int64_t nx, int negate,
                              const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                              void* workspace, double* out_Q, double* out_qsorted, double* out_acum,
                              unsigned* out_nvalid, double* out_bpe)
{
    const int64_t n = ny * nx;
    const bool tsmall = small_tiles(n, nslab);
    const int64_t btile = tsmall ? BTILE_SMALL : BTILE;
    const int64_t ntiles = (n + btile - 1) / btile, ntiles_ws = (n + BTILE_SMALL - 1) / BTILE_SMALL;
    const int nb = (int)((n + 2047) / 2048);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t S = (size_t)nslab;
    char* w = (char*)workspace;
    K* kA = (K*)w; w += al(S * n * 8);                      // (sized for 64-bit keys either way)
    K* kB = (K*)w; w += al(S * n * 8);
    double* vA = (double*)w; w += al(S * n * 8);
    double* vB = (double*)w; w += al(S * n * 8);
    unsigned* hist = (unsigned*)w; w += al(S * 256 * ntiles_ws * 4);
    unsigned* totals = (unsigned*)w; w += al(S * 256 * 4);
    unsigned* nvalid = (unsigned*)w; w += al(S * 4);
    double* bsum = (double*)w; w += al(S * nb * 8);
    double* parts = (double*)w; w += al(S * BPE_BLOCKS * 8);
    double* mmpart = (double*)w; w += al(S * kMinmaxBlocks * 2 * 8);
    double* mm = (double*)w; w += al(S * 4 * 8);
    unsigned* tick = (unsigned*)w; w += al(S * 4 * 4);      // arrival tickets of the kernels that finish in their last block (zeroed by k_range_bounds / a memset)
    unsigned* rhist = (unsigned*)w; w += al(S * RANGE_NB * 4);
    unsigned* rtab = (unsigned*)w;
    if (out_nvalid) nvalid = out_nvalid;                     // the caller's own buffer: no copy at the end
    const unsigned ns = (unsigned)nslab;

    const unsigned gb = (unsigned)((n + 255) / 256);
    // a per-slab dA plane is the PLANE case with a slab stride
    const int krank = dA_rank == XC_DA_SLAB ? XC_DA_PLANE : dA_rank;
    const int64_t dstride = dA_rank == XC_DA_SLAB ? n : 0, mstride = (mask && mask_per_slab) ? n : 0;
    const PairSrc src = {q, mask, dA, krank, negate, nx, mstride, dstride, mm, rtab};
    const unsigned gt = (unsigned)ntiles;
    const size_t sc_lds = (size_t)btile * 8 + (4 * 256 + 256 + 8) * sizeof(unsigned) + btile;
    const int inline_scan = ntiles <= 32 ? 1 : 0;       // measured: the O(ntiles) walk per block costs ~0.14 us per tile, the scan launch ~5 us
    K *kin = kA, *kout = kB;
    double *vin = vA, *vout = vB;
    const bool mf32 = mask && mask_dtype == XC_F32;

    // one LSD pass (histogram, row scan, scatter); MODE 0: byte `shift / 8` of the key, MODE 1: of the 24-bit range key
    auto big_lds = [&](const void* f) { return ensure_big_lds(ctx, f, (int)sc_lds); };
    auto pass_tr = [&](auto mode_tag, auto tr_tag, bool first, int shift) -> int {
        constexpr int MODE = decltype(mode_tag)::value, TR = decltype(tr_tag)::value;
        if (first) {            // reads the tracer itself (the unsorted pairs never touch memory)
            if (mf32) {
                XC_TRY_(big_lds((const void*)k_radix_scatter<K, true, TQ, float, MODE, TR>));
                hipLaunchKernelGGL((k_radix_hist<K, true, TQ, float, MODE, TR>), dim3(gt, ns), dim3(256), 0, ctx->stream, kin, n, shift, (int)ntiles, hist, src);
            } else {
                XC_TRY_(big_lds((const void*)k_radix_scatter<K, true, TQ, double, MODE, TR>));
                hipLaunchKernelGGL((k_radix_hist<K, true, TQ, double, MODE, TR>), dim3(gt, ns), dim3(256), 0, ctx->stream, kin, n, shift, (int)ntiles, hist, src);
            }
        } else {
            XC_TRY_(big_lds((const void*)k_radix_scatter<K, false, double, double, MODE, TR>));
            hipLaunchKernelGGL((k_radix_hist<K, false, double, double, MODE, TR>), dim3(gt, ns), dim3(256), 0, ctx->stream, kin, n, shift, (int)ntiles, hist, src);
        }
        if (!inline_scan) hipLaunchKernelGGL(k_radix_scan_rows, dim3(256, ns), dim3(1024), 0, ctx->stream, hist, (int)ntiles, totals);
        if (first) {
            if (mf32) hipLaunchKernelGGL((k_radix_scatter<K, true, TQ, float, MODE, TR>), dim3(gt, ns), dim3(256), sc_lds, ctx->stream, kin, vin, kout, vout, n, shift,
                                         (int)ntiles, hist, totals, inline_scan, src);
            else hipLaunchKernelGGL((k_radix_scatter<K, true, TQ, double, MODE, TR>), dim3(gt, ns), dim3(256), sc_lds, ctx->stream, kin, vin, kout, vout, n, shift,
                                    (int)ntiles, hist, totals, inline_scan, src);
        } else hipLaunchKernelGGL((k_radix_scatter<K, false, double, double, MODE, TR>), dim3(gt, ns), dim3(256), sc_lds, ctx->stream, kin, vin, kout, vout, n, shift,
                                  (int)ntiles, hist, totals, inline_scan, src);
        XC_HIP(ctx, hipGetLastError());
        K* tk = kin; kin = kout; kout = tk;
        double* tv = vin; vin = vout; vout = tv;
        return XC_OK;
    };
    auto pass = [&](auto mode_tag, bool first, int shift) -> int {
        return tsmall ? pass_tr(mode_tag, std::integral_constant<int, TILE_ROUNDS_SMALL>(), first, shift)
                      : pass_tr(mode_tag, std::integral_constant<int, TILE_ROUNDS>(), first, shift);
    };

    // everything after the sort: cumulative area, profile, BPE, copies of the requested arrays
    auto tail = [&](bool count_valid) -> int {
        if (count_valid) hipLaunchKernelGGL(k_count_valid<K>, dim3(ns), dim3(64), 0, ctx->stream, kin, n, nvalid);
        double* acum = vout;                                   // reuse the idle payload buffer
        hipLaunchKernelGGL(k_scan_local<false>, dim3(nb, ns), dim3(256), 0, ctx->stream, vin, acum, n, bsum);
        hipLaunchKernelGGL(k_scan_bsums, dim3(ns), dim3(1024), 0, ctx->stream, bsum, nb);
        hipLaunchKernelGGL(k_scan_local<true>, dim3(nb, ns), dim3(256), 0, ctx->stream, vin, acum, n, bsum);
        XC_HIP(ctx, hipGetLastError());
        const bool prof = out_Q && J > 0;
        if (prof && !targets) return fail(ctx, XC_EBADARG, "xc_sort_profile: targets is NULL");
        if (out_bpe && (!tbl || !coord || ntbl < 2)) return fail(ctx, XC_EBADARG, "xc_sort_profile: BPE needs tbl/coord");
        const int nprof = prof ? (J + 255) / 256 : 0;
        if (out_bpe)            // (the profile rides in the same launch: see k_bpe)
            hipLaunchKernelGGL(k_bpe<K>, dim3(BPE_BLOCKS + nprof, ns), dim3(256), 0, ctx->stream, kin, vin, acum, nvalid, tbl, coord, ntbl, parts, n, tick, out_bpe,
                               targets, J, out_Q, nprof);
        else if (prof) hipLaunchKernelGGL(k_profile<K>, dim3(nprof, ns), dim3(256), 0, ctx->stream, kin, acum, nvalid, targets, J, out_Q, n);
        if (out_qsorted) hipLaunchKernelGGL(k_unkey<K>, dim3(gb, ns), dim3(256), 0, ctx->stream, kin, n, out_qsorted);
        if (out_acum) XC_HIP(ctx, hipMemcpyAsync(out_acum, acum, S * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        XC_HIP(ctx, hipGetLastError());
        return XC_OK;
    };

    if constexpr (sizeof(K) == 8) {
        if (ctx->knobs.sort_range) {
            // ---- three passes over the 24-bit range key, then the short runs (see the head of this file)
            XC_TRY_(launch_minmax_partial(ctx, q, q_dtype, nslab, n, mmpart));
            // the "not sorted" word lives in pinned host memory and the repair kernel writes it THERE (a system-scope atomic, only when a
            // run fails): no device word to clear before and to copy back after (two of the chain's ~20 dependent launches)
            if (!ctx->pinned_flag) XC_HIP(ctx, hipHostMalloc((void**)&ctx->pinned_flag, 64, hipHostMallocDefault));
            volatile unsigned& h_flag = *ctx->pinned_flag;
            h_flag = 0;
            unsigned* flag = ctx->pinned_flag;
            hipLaunchKernelGGL(k_range_bounds, dim3(ns), dim3(512), 0, ctx->stream, mmpart, minmax_blocks(n, nslab), mm, rhist, tick);
            {
                const int64_t samp = n > (int64_t)256 * RANGE_SAMPLE * 64 ? RANGE_SAMPLE : 1;
                int64_t hb = ((n + 256 * samp - 1) / (256 * samp) + 7) / 8;                  // eight 256-cell segments per block and round
                if (hb > 1024) hb = 1024;
                if (hb < 1) hb = 1;
                if (mf32) hipLaunchKernelGGL((k_range_hist<TQ, float>), dim3((unsigned)hb, ns), dim3(256), 0, ctx->stream, n, src, rhist);
                else hipLaunchKernelGGL((k_range_hist<TQ, double>), dim3((unsigned)hb, ns), dim3(256), 0, ctx->stream, n, src, rhist);
                hipLaunchKernelGGL(k_range_table, dim3(ns), dim3(RANGE_NB), 0, ctx->stream, rhist, rtab);
                XC_HIP(ctx, hipGetLastError());
            }
            for (int p = 0; p < 3; ++p) XC_TRY_(pass(std::integral_constant<int, 1>(), p == 0, 8 * p));
            hipLaunchKernelGGL(k_fix_runs<K>, dim3((unsigned)((n + FIX_C - 1) / FIX_C), ns), dim3(256), 0, ctx->stream, kin, vin, n, flag, nvalid, src);
            XC_HIP(ctx, hipGetLastError());
            // the rest is enqueued as if the repair had sufficed -- it nearly always has -- so that the GPU does not idle through
            // the one host round trip of the sort; a stack that failed the check is sorted again below and the rest redone
            XC_TRY_(tail(false));
            XC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->last_sort_path = h_flag == 0 ? 1 : 2;
            if (h_flag == 0) return XC_OK;
            kin = kA; kout = kB; vin = vA; vout = vB;
        } else ctx->last_sort_path = 0;
    } else ctx->last_sort_path = 0;
    if (ctx->last_sort_path == 0) XC_HIP(ctx, hipMemsetAsync(tick, 0, S * 4 * 4, ctx->stream));       // (the range path's first kernel clears the tickets itself)
    for (int p = 0; p < KeyTraits<K>::passes; ++p) XC_TRY_(pass(std::integral_constant<int, 0>(), p == 0, 8 * p));
    return tail(true);
}

int launch_sort_profile(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype, int mask_per_slab,
                        const double* dA, int dA_rank, int64_t nslab, int64_t ny, int64_t nx, int negate,
                        const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                        void* workspace, double* out_Q, double* out_qsorted, double* out_acum,
                        unsigned* out_nvalid, double* out_bpe)
{
    const int64_t n = ny * nx;
    if (!q || !workspace || n < 1 || n > 0x7fffffff || nslab < 1 || nslab > 65535) return fail(ctx, XC_EBADARG, "xc_sort_profile: bad arguments");
    if (dA_rank < XC_DA_NONE || dA_rank > XC_DA_SLAB) return fail(ctx, XC_EBADARG, "xc_sort_profile: bad dA_rank");
    if (dA_rank != XC_DA_NONE && !dA) return fail(ctx, XC_EBADARG, "xc_sort_profile: dA is NULL");
    if (q_dtype == XC_F64)
        return sort_profile_typed<double, u64>(ctx, (const double*)q, q_dtype, mask, mask_dtype, mask_per_slab, dA, dA_rank, nslab, ny, nx, negate,
                                               targets, J, tbl, coord, ntbl, workspace, out_Q, out_qsorted, out_acum, out_nvalid, out_bpe);
    if (q_dtype == XC_F32)      // the order of floats is the order of their 32-bit keys: 4 passes of 4-byte keys
        return sort_profile_typed<float, u32>(ctx, (const float*)q, q_dtype, mask, mask_dtype, mask_per_slab, dA, dA_rank, nslab, ny, nx, negate,
                                              targets, J, tbl, coord, ntbl, workspace, out_Q, out_qsorted, out_acum, out_nvalid, out_bpe);
    return fail(ctx, XC_EBADARG, "xc_sort_profile: q_dtype must be XC_F32 or XC_F64");
}

}  // namespace xc
