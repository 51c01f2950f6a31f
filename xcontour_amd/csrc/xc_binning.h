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
// rint(w * 2^k) (half to even) as a two's-complement 64-bit integer, |w * 2^k| < 2^62 (the deterministic sums,
// xc_hist_det.hip; the oracle's deterministic_bin_sums is np.rint(np.ldexp(w, k)).astype(int64)).  gfx950 has no f64 -> i64
// conversion: split t = hi * 2^32 + lo with hi = floor(t / 2^32) (exact: a power-of-two scaling, a floor and an FMA that
// cancels), 0 <= lo < 2^32, round lo -- the fraction and the parity of t are those of lo, so rint(lo) completes rint(t) --
// and convert the halves; rint(lo) == 2^32 saturates the 32-bit conversion and is carried into the high half.
__device__ __forceinline__ unsigned long long fixed_point(double w, int k)
{
    const double t = ldexp(w, k);
    const double th = floor(t * 0x1p-32);
    const double tl = rint(__builtin_fma(th, -0x1p32, t));
    const unsigned long long lo = (unsigned long long)(unsigned)tl + (tl >= 0x1p32 ? 1ull : 0ull);
    return ((unsigned long long)(long long)(int)th << 32) + lo;
}
__device__ __forceinline__ void lds_max(unsigned long long* p, unsigned long long v)      // ds_max_u64
{
    __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

