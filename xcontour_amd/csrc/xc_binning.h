// Device helpers of the histogram kernels (xc_hist.hip, K3): exact contour-level / edge arithmetic, np.digitize-exact bin search, LDS atomics,
// cross-lane moves.  Included inside namespace xc { namespace { ... } } of each translation unit.
#pragma once

__device__ __forceinline__ double dnan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ double dinf() { return __longlong_as_double(0x7ff0000000000000LL); }

// Row loads take a wave-uniform row pointer plus a 32-bit per-lane byte offset so that the
// compiler can use the saddr + voffset form (no 64-bit VALU address arithmetic per load).
template <typename T, int VEC> struct RowLoad;
template <> struct RowLoad<double, 2> {
    static __device__ __forceinline__ void ld(const double* p, double (&v)[2]) {
        const double2 t = *reinterpret_cast<const double2*>(p); v[0] = t.x; v[1] = t.y; }
};
template <> struct RowLoad<double, 4> {        // two 16-byte loads per lane: 2 KiB per wave and row
    static __device__ __forceinline__ void ld(const double* p, double (&v)[4]) {
        const double2 t = *reinterpret_cast<const double2*>(p), u = *reinterpret_cast<const double2*>(p + 2);
        v[0] = t.x; v[1] = t.y; v[2] = u.x; v[3] = u.y; }
};
template <> struct RowLoad<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, double (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p); v[0] = (double)t.x; v[1] = (double)t.y; v[2] = (double)t.z; v[3] = (double)t.w; }
    // raw float32 (the E32 variant of K3 keeps rows unconverted while they wait in their buffers)
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
};
template <> struct RowLoad<double, 1> {
    static __device__ __forceinline__ void ld(const double* p, double (&v)[1]) { v[0] = *p; }
};
template <> struct RowLoad<float, 2> {
    static __device__ __forceinline__ void ld(const float* p, double (&v)[2]) {
        const float2 t = *reinterpret_cast<const float2*>(p); v[0] = (double)t.x; v[1] = (double)t.y; }
};
template <> struct RowLoad<float, 1> {
    static __device__ __forceinline__ void ld(const float* p, double (&v)[1]) { v[0] = (double)*p; }
};

// The same loads with the non-temporal (streaming) hint: the tracer is read once per pass and should not push the
// weights -- shared by every slab of a launch -- out of the XCD's L2.
typedef double xc_d2v __attribute__((ext_vector_type(2)));
typedef float  xc_f2v __attribute__((ext_vector_type(2)));
template <typename T, int VEC> struct RowLoadNT;
template <> struct RowLoadNT<double, 2> {
    static __device__ __forceinline__ void ld(const double* p, double (&v)[2]) {
        const xc_d2v t = __builtin_nontemporal_load(reinterpret_cast<const xc_d2v*>(p)); v[0] = t.x; v[1] = t.y; }
};
typedef float xc_f4v __attribute__((ext_vector_type(4)));
template <> struct RowLoadNT<double, 4> {
    static __device__ __forceinline__ void ld(const double* p, double (&v)[4]) {
        const xc_d2v t = __builtin_nontemporal_load(reinterpret_cast<const xc_d2v*>(p));
        const xc_d2v u = __builtin_nontemporal_load(reinterpret_cast<const xc_d2v*>(p + 2));
        v[0] = t.x; v[1] = t.y; v[2] = u.x; v[3] = u.y; }
};
template <> struct RowLoadNT<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, double (&v)[4]) {
        const xc_f4v t = __builtin_nontemporal_load(reinterpret_cast<const xc_f4v*>(p));
        v[0] = (double)t.x; v[1] = (double)t.y; v[2] = (double)t.z; v[3] = (double)t.w; }
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const xc_f4v t = __builtin_nontemporal_load(reinterpret_cast<const xc_f4v*>(p));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
};
template <> struct RowLoadNT<double, 1> {
    static __device__ __forceinline__ void ld(const double* p, double (&v)[1]) { v[0] = __builtin_nontemporal_load(p); }
};
template <> struct RowLoadNT<float, 2> {
    static __device__ __forceinline__ void ld(const float* p, double (&v)[2]) {
        const xc_f2v t = __builtin_nontemporal_load(reinterpret_cast<const xc_f2v*>(p)); v[0] = (double)t.x; v[1] = (double)t.y; }
};
template <> struct RowLoadNT<float, 1> {
    static __device__ __forceinline__ void ld(const float* p, double (&v)[1]) { v[0] = (double)__builtin_nontemporal_load(p); }
};

// ---- contour levels, bit-for-bit the arithmetic of cal_contours (core.py:228-246)
// under np.vectorize: (stop-start) in the tracer dtype, everything else in f64,
// cast to the contour dtype at the end.  __d*_rn / __f*_rn forbid FMA contraction.
__device__ __forceinline__ double level_value(double mn, double mx, int k, int increase,
                                              int q_f32, int ctr_f32, double inv_nm1)
{
    const double start = increase ? mn : mx, stop = increase ? mx : mn;
    const double d = q_f32 ? (double)__fsub_rn((float)stop, (float)start) : __dsub_rn(stop, start);
    const double steps = __dmul_rn(inv_nm1, d);
    double c = __dadd_rn(__dmul_rn(steps, (double)k), start);
    if (ctr_f32) c = (double)(float)c;
    return c;
}

// dummy left edge of _histogram (core.py:1296-1305), in the contour dtype
__device__ __forceinline__ double dummy_edge(double lo, double hi, int N, int ctr_f32)
{
    if (ctr_f32) {
        const float step = __fdiv_rn(__fsub_rn((float)hi, (float)lo), (float)(N - 1));
        return (double)__fsub_rn((float)lo, step);
    }
    const double step = __ddiv_rn(__dsub_rn(hi, lo), (double)(N - 1));
    return __dsub_rn(lo, step);
}

__device__ __forceinline__ double bump_last_edge(double e, int ctr_f32)   // xhistogram's "+1e-8"
{
    return ctr_f32 ? (double)__fadd_rn((float)e, (float)1e-8) : __dadd_rn(e, 1e-8);
}

// np.digitize(v, edges) - 1 restricted to [0, N-1]; -1 when the cell is dropped.
// Rare path: the uniform guess missed its bracket (bin boundary rounding, non-uniform user
// levels), or the value is NaN / out of range / on the closed last edge.
__device__ __noinline__ int find_bin_slow(double v, const double* __restrict__ s_edges, int N, int k,
                                          double e0, double eN, int last_closed)
{
    if (!(v >= e0)) return -1;                       // below range or NaN
    if (last_closed ? !(v <= eN) : !(v < eN)) return -1;
    if (v < s_edges[k]) {
        if (v >= s_edges[k - 1]) return k - 1;       // k >= 1 here because v >= e0
        int lo = 0, hi = k - 1;                      // edges[lo] <= v < edges[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (v >= s_edges[mid]) lo = mid; else hi = mid; }
        return lo;
    }
    if (k < N - 1 && v >= s_edges[k + 1]) {
        if (k + 1 == N - 1 || v < s_edges[k + 2]) return k + 1;
        int lo = k + 2, hi = N;                      // edges[lo] <= v, v < edges[hi] (or v == eN closed)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (v >= s_edges[mid]) lo = mid; else hi = mid; }
        return lo > N - 1 ? N - 1 : lo;
    }
    return k;                                        // v == eN on the closed last edge
}

// Common path: uniform guess, one bracket test against the explicit edges (2 LDS reads).
// A bracket hit proves e0 <= e[k] <= v < e[k+1] <= eN, so no separate range / NaN test is needed.
__device__ __forceinline__ int find_bin(double v, const double* __restrict__ s_edges, int N,
                                        double e0, double eN, double inv, int last_closed)
{
    int k = (int)((v - e0) * inv);                   // NaN -> 0
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(k) : "v"(k), "s"(N - 1));   // clamp to [0, N-1]
    const double lo = s_edges[k], hi = s_edges[k + 1];          // one ds_read2_b64
    const bool hit = (v >= lo) & (v < hi);                      // bitwise: no short-circuit branch between the reads
    if (hit) return k;
    return find_bin_slow(v, s_edges, N, k, e0, eN, last_closed);
}

__device__ __forceinline__ void lds_add(double* p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add(unsigned* p, unsigned v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add(unsigned long long* p, unsigned long long v)      // ds_add_u64: associative, order-free
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// ---- deterministic sums in ONE pass (round 5): a fixed-point superaccumulator per (bin, channel).
// The accumulator is a row of kDetLimbs limbs on a FIXED grid: limb i counts units of 2^(top - S (i + 1)), top = the window's first
// bit, chosen from a rigorous bound of the channel's weights that is known BEFORE the pass (max dA; for the in-kernel squared gradient
// 2 ((max - min) max(rdx, rdy))^2 max dA from K1's extrema and the metrics; max |integrand| max dA) plus kDetTopSlack bits, so that
// the top limb cannot overflow.  A weight's 53-bit significand is cut ONCE to P = S + 1 = 49 bits (its low 4 bits dropped: a function of
// the cell alone) and the resulting integer is split at the limb boundary into a high and a low chunk (each < 2^48), added with two
// ds_add_u64 to the ADJACENT limbs its bits fall into -- whatever its magnitude: the pole row of a lat-lon grid carries squared
// gradients 2^100 times the typical ones and both kinds keep their 49 bits.  Integer addition is associative: the sums do not depend
// on the order of arrival, the block geometry, the launch partition or the number of ranks.  Capacity: a limb of one LDS copy takes at
// most 32767 chunks before a signed 64-bit word could wrap (negative weights are two's complement) -- hist_geometry sizes the blocks
// for it.  LDS layout per (bin, copy): kDetWords words per channel -- the limbs and ONE trash word behind them for a low chunk that
// falls under the window -- then the count / flag word.  The oracle's deterministic_bin_sums / det_chunks restate this rule; the GPU
// reproduces it bit for bit.
constexpr int kDetLimbBits = 48;     // S
constexpr int kDetPrecBits = 49;     // P = S + 1: a P-bit integer at any offset spans exactly two limbs
constexpr int kDetTopSlack = 12;     // the window starts this many bits above the bound (2^23 cells x 2^(48 - 12) < 2^63 in the top limb)
constexpr int kDetTopFloor = -800;   // ... and never below 2^-800: zero and denormal weights then lie under every window by construction
constexpr int kDetLimbsX   = 4;      // limbs per channel: a 192-bit window, 180 bits of it below the bound
constexpr int kDetWords    = kDetLimbsX + 1;
__host__ __device__ constexpr int det_total_limbs(int nch) { return nch * kDetLimbsX; }
// c0 = top + 1023 + 52 - (53 - P): d = c0 - E is the number of bits from the cut weight's last bit up to the window top (E: the
// weight's biased exponent).  Returns jc in [1, kDetLimbsX]: `hi` goes to limb jc - 1, `lo` to word jc (a limb, or the trash word).
// Weights wholly under the window, zeros and denormals (E == 0: d = c0 > S (kDetLimbsX + 1) by the floor on the top) give two zero
// chunks.  A non-finite weight (E == 2047) gives garbage chunks in valid words: the caller flags the bin.
__device__ __forceinline__ int det_split(double w, int c0, unsigned long long& hi, unsigned long long& lo, int& E)
{
    const unsigned bh = (unsigned)__double2hiint(w), bl = (unsigned)__double2loint(w);
    E = (int)((bh >> 20) & 0x7ffu);
    unsigned long long n = (((unsigned long long)((bh & 0xfffffu) | 0x100000u) << 32) | bl) >> (53 - kDetPrecBits);
    const int d = c0 - E;
    if (d > kDetLimbBits * (kDetLimbsX + 1)) n = 0ull;
    int j = (int)((((unsigned)(d - 1) >> 4) * 43691u) >> 17);             // (d - 1) / 48
    asm("v_med3_i32 %0, %1, 1, %2" : "=v"(j) : "v"(j), "v"(kDetLimbsX));  // a weight above its bound (or an infinite one) must not leave the cell
    const int s = kDetLimbBits * (j + 1) - d;                             // 0 .. 47 for a weight inside the window
    hi = n >> (kDetLimbBits - s);
    lo = (n << s) & 0xffffffffffffull;
    return j;
}
__host__ __device__ inline int det_c0_from_bound(double bound)           // bound >= 0: every |w| <= bound
{
    unsigned long long b; __builtin_memcpy(&b, &bound, 8);
    const int E = (int)((b >> 52) & 0x7ffu);
    int top = (E - 1022) + kDetTopSlack;                                  // frexp exponent of the bound + slack
    if (top < kDetTopFloor) top = kDetTopFloor;
    return top + 1023 + 52 - (53 - kDetPrecBits);
}

// copy slot of a lane.  (Rotating the slot by the bin index to spread LDS banks was measured
// 10 % SLOWER on MI355X -- profiles/r01_notes.md -- so the slot is simply lane % ncopy.)
#define XC_ROT(copy, k, ncopy) (copy)

// ---- cross-lane moves that stay off the LDS pipe (the LDS is the busiest unit of this kernel)
// DPP wave shift by one lane; the lane that has no source lane (lane 0 for shr, lane 63 for shl)
// keeps `old`: with old = the halo register the strip-edge neighbour arrives without readlane / select
template <int CTRL>
__device__ __forceinline__ double lane_shift_keep(double v, double old)
{
    const unsigned long long u = __double_as_longlong(v), o = __double_as_longlong(old);
    const unsigned lo = __builtin_amdgcn_update_dpp((int)(o & 0xffffffffu), (int)(u & 0xffffffffu), CTRL, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// value held by lane `idx` (wave-uniform index): two v_readlane_b32
__device__ __forceinline__ double lane_get(double v, int idx)
{
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), idx);
    const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), idx);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
constexpr int DPP_WAVE_SHL1 = 0x130;   // lane i <- lane i+1
constexpr int DPP_WAVE_SHR1 = 0x138;   // lane i <- lane i-1

