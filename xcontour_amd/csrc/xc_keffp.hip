// K3P -- persistent SINGLE-READ Keff kernel (gfx950): min/max -> levels -> weighted histogram with in-kernel
// |grad q|^2 for a whole stack of slabs in ONE launch, the tracer crossing the fabric ONCE.
//
// Replaces, for the fused pipeline (xc_keff_dev; reference call sequence core.py:205-249 -> 412-460 -> 1202-1325), the
// pair "K1 min/max pass, then K3 histogram pass" (or K3 with the next batch's min/max riding along), both of which
// stream the tracer twice: min/max must be known before the first cell can be binned.  Here the slab stays ON CHIP
// between the two steps:
//
//   * the grid is one 1024-thread workgroup per CU, all co-resident; `ngroups` groups of G workgroups each work
//     through their own sequence of slabs (group g: slabs g, g + ngroups, ...), one slab at a time per group, the
//     slab spread over the G x 16 waves of the group;
//   * a wave owns a chunk of <= 14 rows x 124 columns of the slab and holds it -- plus one halo row above and below
//     and one halo lane left and right (strips overlap by 4 columns, so x-neighbours always come from the adjacent
//     lane by DPP and no halo column is ever loaded separately) -- in 64 VGPRs per lane: 256 CUs x 16 waves x
//     16 rows x 1 KiB = 64 MiB of register tile for a 51.9 MB slab;
//   * step A: per-wave min/max of the register tile -> workgroup -> ONE pair per slab by agent-scope 64-bit
//     atomic max on order-preserving keys -> an arrival counter; every workgroup of the group polls that counter
//     (one lane, bounded, sc1 loads), reads the pair and builds the N levels / N+1 edges in LDS with exactly the
//     arithmetic of the two-pass path (xc_binning.h);
//   * step B: the wave walks its rows in registers: bin (nearest-edge guess + ONE exact comparison against the
//     f64 edge in LDS when the levels are equally spaced to a quarter of a bin -- verified per slab -- else the
//     general bracket search), centred differences, weights, three LDS atomics per cell on lane-privatised copies;
//     the weights dA stream in with a two-row lead; as soon as a tile row is dead it is REFILLED with the same row
//     of the group's NEXT slab, so the next slab's loads overlap this slab's arithmetic;
//   * per-workgroup partial histograms go out with plain stores (k_reduce_partials / k_finalize are unchanged).
//
// Every wait on another workgroup is bounded (wall clock); on a timeout the kernel raises an abort flag, every
// workgroup leaves, the unfinished slabs get status 2 and the host re-runs them through the two-pass path.
#include "xc_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace xc {

namespace {

#include "xc_binning.h"

#ifndef XC_PERSIST_NT
#define XC_PERSIST_NT 1
#endif
constexpr bool kNT = XC_PERSIST_NT != 0;
constexpr int PR = kPersistRows;       // rows of a chunk
constexpr int PT = PR + 2;             // tile rows (halo row below and above)
// threads of a workgroup: 768 (one workgroup per CU) or 256 (three) -- always 12 waves per CU, 3 per SIMD, 168 VGPRs.  (A
// workgroup's waves are dealt to the SIMDs from SIMD 0: 6- or 3-wave workgroups pile up on the first SIMDs, a second / fourth
// one then finds no registers there and waits for the first to END -- measured, tools/gpu_persist_check.py --stamps; whole
// multiples of 4 waves fill the SIMDs evenly.)
// tile rows of the NEXT slab that wait in LDS (f64 tracers; the last KL rows of the tile): 8, or 4 when three workgroups share the LDS
constexpr int persist_lds_rows(int nth) { return nth >= 768 ? kPersistLdsRows : kPersistLdsRows / 2; }
constexpr int PCOLS = kPersistCols;    // computed columns of a strip (lanes 1..62, two cells each)
constexpr unsigned long long kTimeoutTicks = 30000000ull;   // 0.3 s of the 100 MHz wall clock

typedef double d2v __attribute__((ext_vector_type(2)));
typedef float  f2v __attribute__((ext_vector_type(2)));
// NT: non-temporal (streaming) hint -- the tracer is read exactly once and should not push the weights out of L2
template <typename TQ, bool NT = false> struct Ld2;
template <bool NT> struct Ld2<double, NT> {
    static __device__ __forceinline__ void ld(const char* row, unsigned voff, double (&o)[2]) {
        const d2v* p = reinterpret_cast<const d2v*>(row + voff);
        const d2v t = NT ? __builtin_nontemporal_load(p) : *p; o[0] = t.x; o[1] = t.y; }
};
template <bool NT> struct Ld2<float, NT> {
    static __device__ __forceinline__ void ld(const char* row, unsigned voff, double (&o)[2]) {
        const f2v* p = reinterpret_cast<const f2v*>(row + voff);
        const f2v t = NT ? __builtin_nontemporal_load(p) : *p; o[0] = (double)t.x; o[1] = (double)t.y; }
};

// order-preserving map double -> uint64 (total order of the finite / infinite values; NaN never gets here)
__device__ __forceinline__ unsigned long long dkey(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k)
{
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

__device__ __forceinline__ double uniform_d(double v)      // a wave-uniform double that the compiler cannot prove uniform -> SGPR pair
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((int)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// FAST: periodic X, dA verified finite and >= 0, half-open last bin (the xhistogram rule) -- the selects for walls,
// fillna and the closed last edge are compiled out.  Otherwise they are runtime (wave-uniform) flags.
template <typename TQ, bool DA2D, bool FAST, int NT>
__global__ __attribute__((amdgpu_flat_work_group_size(NT, NT), amdgpu_waves_per_eu(3, 3)))
void k_keff_persist(const PersistArgs a)
{
    constexpr int NW = NT / 64;                                // waves of a workgroup
    constexpr int KL = persist_lds_rows(NT);
    extern __shared__ __align__(16) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.nbin, ncopy = a.ncopy;
    const int cshift = __builtin_ctz((unsigned)ncopy);
    const int epad = (N + 2) & ~1;
    double*   s_red   = smem;                                  // 64 doubles: wave partials, broadcast slots
    double*   s_edges = smem + 64;                             // N + 1
    double*   s_h     = s_edges + epad;                        // [N * ncopy][2]  (area, intS interleaved)
    unsigned* s_c     = reinterpret_cast<unsigned*>(s_h + (size_t)2 * N * ncopy);   // [N * ncopy]
    int*      s_flag  = reinterpret_cast<int*>(s_c + (size_t)N * ncopy);            // [4]
    // [NW][KL][128] doubles (LDSR only), 16-byte aligned whatever N * ncopy is
    double*   s_tile  = smem + (((reinterpret_cast<char*>(s_flag + 16) - reinterpret_cast<char*>(smem)) + 15) / 16) * 2;
    const int hsz = N * ncopy;
    constexpr bool LDSR = sizeof(TQ) == 8;                     // LDS-DMA moves 16 bytes per lane: float64 rows only

    const int ny = (int)a.ny, nx = (int)a.nx;
    // Workgroup b lands on CU b % cus in arrival slot b / cus (tools/probe/wg_placement.hip, measured on MI355X): with `slots`
    // workgroups per CU the groups of DIFFERENT slots share every CU, so while one group waits for its grid-wide min / max
    // the waves of the others keep the CU busy.  Inside a slot: ngps groups of G = cus / ngps workgroups, as before.
    const int slot = (int)blockIdx.x / a.cus, bslot = (int)blockIdx.x - slot * a.cus;
    const int group = slot * a.ngps + bslot % a.ngps, rank = bslot / a.ngps;
    const int gw = rank * NW + wave;                            // wave index inside the group
    const int strip = gw / a.cps, chunk = gw - strip * a.cps;
    const bool work = strip < a.nstrip;
    const int r0 = work ? chunk * a.rpc : 0;
    const int nrows = work ? ((a.rpc < ny - r0) ? a.rpc : ny - r0) : 0;       // host: cps = ceil(ny / rpc) -> >= 1
    const int x0 = strip * PCOLS;
    const int col0 = x0 - 2 + 2 * lane;                        // first of the lane's two columns (even)
    const bool periodic = FAST ? true : (a.periodic_x != 0);
    int colA = col0;                                           // the column actually loaded (wrapped / clamped, even)
    if (periodic) { if (colA < 0) colA += nx; else if (colA >= nx) colA -= nx; }
    if (colA < 0) colA = 0;
    if (colA > nx - 2) colA = nx - 2;
    const unsigned voff_q = (unsigned)colA * (unsigned)sizeof(TQ), voff_d = (unsigned)colA * 8u;
    bool cv[2];                                                // is cell c of this lane a computed cell of the strip?
#pragma unroll
    for (int c = 0; c < 2; ++c) cv[c] = work && lane >= 1 && lane <= 62 && col0 + c < nx;
    const int copy = lane & (ncopy - 1);
    const size_t rowq = (size_t)nx * sizeof(TQ), rowd = (size_t)nx * 8;
    const size_t slabq = (size_t)ny * rowq, slabd = (size_t)ny * rowd;
    // slab-invariant per-row quantities, lane-distributed (lane i <-> chunk row i): gradient metrics, per-row weights
    const int ym = (r0 + lane < ny) ? r0 + lane : ny - 1;
    const double rdxv = a.rdx[ym], rdyv = a.rdy[ym];
    const double dArv = DA2D ? 0.0 : a.dA[ym];
    const bool wpos = FAST ? true : (a.dA_pos_finite != 0);
    const bool closed = FAST ? false : (a.last_closed != 0);

    for (int i = tid; i < 2 * hsz; i += NT) s_h[i] = 0.0;
    for (int i = tid; i < hsz; i += NT) s_c[i] = 0u;

    double T[PT][2];                                           // the register tile
    // byte offset of tile row t (<-> slab row clamp(r0 - 1 + t)) inside a slab: 32 bits (host: slab < 2 GiB)
    auto qrow_off = [&](int t) -> unsigned {
        int y = r0 - 1 + t;
        y = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
        return (unsigned)y * (unsigned)rowq;
    };
    auto drow_off = [&](int t) -> unsigned {
        int y = r0 - 1 + t;
        y = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
        return (unsigned)y * (unsigned)rowd;
    };

    // diagnostics (xc_dbg_set_stamps): wall-clock stamps of thread 0 at the phase boundaries of every slab
    auto stamp = [&](int sl, int k) { if (a.stamps && tid == 0) a.stamps[((size_t)sl * (a.G * a.ngroups) + blockIdx.x) * 8 + k] = wall_clock64(); };
    // LDS-DMA of tile row t (>= PT - KL) of slab `sl` into this wave's LDS row; lane L's two cells land at +16 L
    auto dma_row = [&](int sl, int t) {
        const char* src = reinterpret_cast<const char*>(a.q) + (size_t)sl * slabq + qrow_off(t < nrows + 1 ? t : nrows + 1) + voff_q;
        double* dst = s_tile + ((size_t)wave * KL + (t - (PT - KL))) * 128;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, kNT ? 2 : 0);
    };
    int s = group;
    if (LDSR && s < a.nslab) {
        const int sl = (s + a.ngroups < a.nslab) ? s + a.ngroups : s;
#pragma unroll
        for (int t = PT - KL; t < PT; ++t) dma_row(sl, t);
    }
    if (s < a.nslab) {
        const char* q0 = reinterpret_cast<const char*>(a.q) + (size_t)s * slabq;
#pragma unroll
        for (int t = 0; t < PT; ++t)
            Ld2<TQ, kNT>::ld(q0 + qrow_off(t < nrows + 1 ? t : nrows + 1), voff_q, T[t]);     // rows past the chunk: duplicates of its last halo row
    }
    int prev = -1;                                             // slab whose histogram still sits in LDS
    bool aborted = false;

    auto flush = [&](int ps) {                                 // LDS copies -> this workgroup's partials; zero the copies
        const size_t pb = (size_t)ps * a.G + rank;
        double* ph = a.part_h + pb * 2 * N;
        for (int i = tid; i < 2 * N; i += NT) {
            const int ch = i / N, b = i - ch * N;
            double sum = 0.0;
            for (int c = 0; c < ncopy; ++c) {
                const int cc = (c + tid) & (ncopy - 1);
                double* p = &s_h[(((size_t)b << cshift) + cc) * 2 + ch];
                sum += *p; *p = 0.0;
            }
            ph[i] = sum;
        }
        unsigned* pc = a.part_c + pb * N;
        for (int b = tid; b < N; b += NT) {
            unsigned sum = 0u;
            for (int c = 0; c < ncopy; ++c) {
                const int cc = (c + tid) & (ncopy - 1);
                unsigned* p = &s_c[((size_t)b << cshift) + cc];
                sum += *p; *p = 0u;
            }
            pc[b] = sum;
        }
    };

    for (; s < a.nslab; s += a.ngroups) {
        // ------------------------------------------------------------ A: min / max of the resident tile -> the group
        stamp(s, 0);
        double mn = dinf(), mx = -dinf();
#pragma unroll
        for (int t = 0; t < PT; ++t) {                         // every tile register holds a real cell of this slab
#pragma unroll
            for (int c = 0; c < 2; ++c) { mn = fmin(mn, T[t][c]); mx = fmax(mx, T[t][c]); }       // NaN-skipping
        }
        for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
        if (lane == 0) { s_red[2 * wave] = mn; s_red[2 * wave + 1] = mx; }
        __syncthreads();                                       // also: every wave is past B of the previous slab
        stamp(s, 1);
        // The group's pair lives in 8 shards of 64 bytes (shard = rank % 8): 256 workgroups on ONE word would serialise
        // at ~11 ns per atomic; a shard sees G / 8 of them.  ~key(min) and key(max) under atomic MAX, zero = "nothing yet".
        SyncShard* rec = a.sync + (size_t)s * 8;
        if (wave == 0) {
            double bmn = (lane < NW) ? s_red[2 * lane] : dinf(), bmx = (lane < NW) ? s_red[2 * lane + 1] : -dinf();
            for (int o = 8; o > 0; o >>= 1) { bmn = fmin(bmn, __shfl_xor(bmn, o)); bmx = fmax(bmx, __shfl_xor(bmx, o)); }
            if (lane == 0) {
                SyncShard* mine = rec + (rank & 7);
                __hip_atomic_fetch_max(&mine->kmn, ~dkey(bmn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_max(&mine->kmx, dkey(bmx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both performed before this workgroup is counted
                __hip_atomic_fetch_add(&mine->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        stamp(s, 2);
        // the previous slab's histogram leaves the LDS while the other workgroups arrive
        if (prev >= 0) flush(prev);
        stamp(s, 3);
        if (wave == 0) {
            const unsigned need = (unsigned)(a.G >> 3);
            const unsigned long long t0 = wall_clock64();
            int ok = 1;
            for (;;) {                                         // lanes 0..7 watch one shard each
                const unsigned c = (lane < 8) ? __hip_atomic_load(&rec[lane].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                if (__ballot(c >= need) == ~0ull) break;
                __builtin_amdgcn_s_sleep(2);
                const unsigned ab = __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane((int)ab) != 0 || wall_clock64() - t0 > kTimeoutTicks) { ok = 0; break; }
            }
            if (ok) {
                unsigned long long kmn = (lane < 8) ? __hip_atomic_load(&rec[lane].kmn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                unsigned long long kmx = (lane < 8) ? __hip_atomic_load(&rec[lane].kmx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                for (int o = 4; o > 0; o >>= 1) {
                    const unsigned long long an = __shfl_xor(kmn, o), ax = __shfl_xor(kmx, o);
                    kmn = an > kmn ? an : kmn; kmx = ax > kmx ? ax : kmx;
                }
                if (lane == 0) { s_red[32] = dunkey(~kmn); s_red[33] = dunkey(kmx); }
            } else if (lane == 0) {
                __hip_atomic_store(a.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) s_flag[0] = ok;
        }
        __syncthreads();
        stamp(s, 4);
        if (!s_flag[0]) { aborted = true; break; }
        double gmn = uniform_d(s_red[32]), gmx = uniform_d(s_red[33]);
        if (gmn == dinf() && gmx == -dinf()) { gmn = dnan(); gmx = dnan(); }      // all-NaN slab
        // ------------------------------------------------------------ levels / edges, exactly as the two-pass prologue;
        // every thread derives the two end edges itself (pure functions of the pair), so ONE barrier closes the phase
        const double c_first = level_value(gmn, gmx, 0, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1);
        const double c_last = level_value(gmn, gmx, N - 1, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1);
        const double lo = a.increase ? c_first : c_last, hi = a.increase ? c_last : c_first;
        const double e0 = uniform_d(dummy_edge(lo, hi, N, a.ctr_f32));
        const double eN = uniform_d(a.right_edge == XC_EDGE_XHISTOGRAM ? bump_last_edge(hi, a.ctr_f32) : hi);
        const double hstep = uniform_d((eN - e0) * a.inv_n);
        int bad = 0;
        double cprev = c_first;                                // level k-1 of this thread's level k (k advances by NT)
        for (int k = tid; k < N; k += NT) {
            const double c = level_value(gmn, gmx, k, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1);
            const int idx = a.increase ? k + 1 : N - k;
            const double e = (idx == N) ? eN : c;
            s_edges[idx] = e;
            if (rank == 0 && a.ctr_out) a.ctr_out[(size_t)s * N + k] = c;
            // equally spaced to a quarter of a bin?  (then ONE comparison against the nearest edge is exact)
            bad |= !(fabs(e - (e0 + (double)idx * hstep)) <= 0.25 * hstep);
            if (k >= 1) {                                                       // 'non monotonic bins', core.py:1233
                cprev = level_value(gmn, gmx, k - 1, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1);
                bad |= (c == cprev) << 1;
            }
        }
        if (tid == 0) s_edges[0] = e0;
        const int flags = __syncthreads_or(bad);
        const bool uni = !(flags & 1);
        // levels that are not equally spaced (float32 contours of a tiny range, infinite extrema): left to the host's
        // two-pass path (status 3); degenerate levels (status 1) make the reference raise
        if (rank == 0 && tid == 0 && a.status) a.status[s] = (flags & 2) ? 1 : (uni ? 0 : 3);
        const double inv = uniform_d((double)N * __builtin_amdgcn_rcp(eN - e0));   // the guess only: exactness comes from the comparison
        stamp(s, 5);

        // ------------------------------------------------------------ B: bin + accumulate from registers, refill behind
        const int sn = s + a.ngroups;
        const bool has_next = sn < a.nslab;
        const char* qnext = reinterpret_cast<const char*>(a.q) + (size_t)(has_next ? sn : s) * slabq;
        const char* dcur = reinterpret_cast<const char*>(a.dA) + (a.dA_rank == XC_DA_SLAB ? (size_t)s * slabd : 0);
        // rows are binned only when the levels passed the spacing test (`uni`); the refills happen regardless
        int nr = nrows, nrc = uni ? nrows : 0;                 // rows loaded / rows computed
        asm volatile("" : "+s"(nr), "+s"(nrc));                           // opaque per slab: the row predicates are NOT hoisted out of the slab loop (SGPR pressure)
        {
            // Every load of the row loop is UNCONDITIONAL (rows past the chunk clamp to its last row, a group without a
            // next slab re-reads row 0 of this one: cache hits): with no branch around a load the compiler keeps exact
            // vmcnt counts and a row never waits for the refill loads issued behind it.
            double dAb[3][2];
            if (DA2D) {
                Ld2<double>::ld(dcur + drow_off(1 < nr ? 1 : nr), voff_d, dAb[0]);
                Ld2<double>::ld(dcur + drow_off(2 < nr ? 2 : nr), voff_d, dAb[1]);
            }
#pragma unroll
            for (int i = 1; i <= PR; ++i) {
                asm volatile("" ::: "memory");                 // keep every row's loads in that row's slot
#ifdef XC_DIAG_NODA
                if (DA2D) { dAb[(i + 1) % 3][0] = 1.0 + i; dAb[(i + 1) % 3][1] = 2.0; }
#else
                if (DA2D) Ld2<double>::ld(dcur + drow_off(i + 2 < nr ? i + 2 : nr), voff_d, dAb[(i + 1) % 3]);
#endif
                if (i <= nrc) {
                    const double rdx = lane_get(rdxv, i - 1), rdy = lane_get(rdyv, i - 1);
                    double dAv[2];
                    if (DA2D) { dAv[0] = dAb[(i - 1) % 3][0]; dAv[1] = dAb[(i - 1) % 3][1]; }
                    else { const double v = lane_get(dArv, i - 1); dAv[0] = v; dAv[1] = v; }
                    const double (&qc)[2] = T[i];
                    int b[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {              // both cells' edge reads in flight together
                        const double q = qc[c];
                        const double tt = __builtin_fma(q - e0, inv, 0.5);
                        int j = (int)tt;                                           // NaN -> 0
                        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(j) : "v"(j), "s"(N));        // clamp to [0, N]
                        const double ej = s_edges[j];
                        b[c] = (q >= ej) ? j : j - 1;                              // NaN -> -1 (dropped)
                        if (!FAST) { if (closed && q == eN) b[c] = N - 1; }
                    }
                    // x-neighbours from the adjacent lanes (halo lanes are part of the wave: no special cases)
                    const double fromL = lane_shift_keep<DPP_WAVE_SHR1>(qc[1], qc[1]);
                    const double fromR = lane_shift_keep<DPP_WAVE_SHL1>(qc[0], qc[0]);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const double q = qc[c];
                        double qW = (c == 0) ? fromL : qc[0];
                        double qE = (c == 0) ? qc[1] : fromR;
                        double gx;
                        if (FAST || periodic) {
                            gx = __dmul_rn(__dsub_rn(qE, qW), rdx);
                        } else {                                                   // walls: one-sided, spacing dx
                            const int col = col0 + c;
                            const bool wl = col == 0, wr = col == nx - 1;
                            if (wl) qW = q;
                            if (wr) qE = q;
                            gx = __dmul_rn(__dsub_rn(qE, qW), rdx);
                            gx = __dmul_rn(gx, (wl || wr) ? 2.0 : 1.0);
                        }
                        const double gy = __dmul_rn(__dsub_rn(T[i + 1][c], T[i - 1][c]), rdy);
                        const double g2 = __dadd_rn(__dmul_rn(gx, gx), __dmul_rn(gy, gy));
                        const double dv = dAv[c];
                        const double p = __dmul_rn(g2, dv);
                        const double w0 = wpos ? dv : ((dv != dv) ? 0.0 : dv);     // fillna(0), core.py:449
                        const double w1 = wpos ? fmax(p, 0.0) : ((p != p) ? 0.0 : p);
                        if (cv[c] && (unsigned)b[c] < (unsigned)N) {
                            const unsigned o = ((unsigned)b[c] << cshift) + (unsigned)copy;
#ifdef XC_DIAG_NOATOMIC                 /* timing experiments only: results are wrong */
                            if (w0 + w1 == -1.25) s_h[2 * o] = w0;
#else
                            lds_add(&s_h[2 * o], w0);
                            lds_add(&s_h[2 * o + 1], w1);
                            lds_add(&s_c[o], 1u);
#endif
                        }
                    }
                }
                // tile row i-1 is dead: the same row of the group's next slab takes its registers
                if (!LDSR || i - 1 < PT - KL)
                    Ld2<TQ, kNT>::ld(qnext + (has_next ? qrow_off(i - 1 < nr + 1 ? i - 1 : nr + 1) : 0u), voff_q, T[i - 1]);
            }
            asm volatile("" ::: "memory");
            if (!LDSR) {
                Ld2<TQ, kNT>::ld(qnext + (has_next ? qrow_off(PR < nr + 1 ? PR : nr + 1) : 0u), voff_q, T[PR]);
                Ld2<TQ, kNT>::ld(qnext + (has_next ? qrow_off(nr + 1) : 0u), voff_q, T[PR + 1]);
            } else {
                // the last KL tile rows of the next slab have been waiting in LDS for a whole slab period: no refill
                // latency at the tail of B.  (vmcnt(0): the youngest vector-memory ops here are weight loads issued two
                // rows ago and register refills issued KL rows ago.)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = PT - KL; t < PT; ++t) {
                    const double2 v = *reinterpret_cast<const double2*>(s_tile + ((size_t)wave * KL + (t - (PT - KL))) * 128 + 2 * lane);
                    T[t][0] = v.x; T[t][1] = v.y;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the rows are in registers before the DMA overwrites them
                const int s2 = (sn + a.ngroups < a.nslab) ? sn + a.ngroups : (has_next ? sn : s);
#pragma unroll
                for (int t = PT - KL; t < PT; ++t) dma_row(s2, t);         // the slab after next starts to arrive
            }
        }
        stamp(s, 6);
        prev = s;
    }
    if (aborted) {
        if (rank == 0 && tid == 0 && a.status)
            for (int t = s; t < a.nslab; t += a.ngroups) a.status[t] = 2;      // not computed: the host re-runs these
        return;
    }
    __syncthreads();
    if (prev >= 0) flush(prev);
}

template <typename TQ, bool DA2D, bool FAST, int NT>
int launch_p4(xc_ctx* ctx, const PersistArgs& a, size_t lds)
{
    auto kern = k_keff_persist<TQ, DA2D, FAST, NT>;
    { const int rc = ensure_big_lds(ctx, reinterpret_cast<const void*>(kern), (int)kLdsBudget + 4096); if (rc != XC_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.G * a.ngroups)), dim3(NT), lds, ctx->stream, a);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

template <typename TQ, bool DA2D, bool FAST>
int launch_p3(xc_ctx* ctx, const PersistArgs& a, size_t lds)
{
    switch (a.slots) {
        case 1: return launch_p4<TQ, DA2D, FAST, kPersistThreads>(ctx, a, lds);
        case 3: return launch_p4<TQ, DA2D, FAST, kPersistThreads / 3>(ctx, a, lds);
    }
    return fail(ctx, XC_EBADARG, "persistent kernel: slots must be 1 or 3");
}

template <typename TQ>
int launch_p2(xc_ctx* ctx, const PersistArgs& a, size_t lds, bool da2d, bool fast)
{
    if (da2d) return fast ? launch_p3<TQ, true, true>(ctx, a, lds) : launch_p3<TQ, true, false>(ctx, a, lds);
    return fast ? launch_p3<TQ, false, true>(ctx, a, lds) : launch_p3<TQ, false, false>(ctx, a, lds);
}

}  // namespace

// Decomposition of a stack of (ny, nx) slabs over the chip; false when the shape does not suit the persistent kernel
// (the caller then takes the two-pass path).
bool persist_geometry(const xc_ctx* ctx, int q_dtype, int64_t nslab, int64_t ny, int64_t nx, int N,
                      const void* q, const double* dA, int dA_rank, PersistGeom* g)
{
    // XC_KEFF_AUTO takes the two-pass path: on MI355X the persistent kernel halves the fabric traffic but pays ~13 us of
    // grid-wide synchronisation and arrival skew per cfg2 slab with ONE slab in flight (a second does not fit on chip),
    // which the streaming path does not have.  XC_KEFF_PERSIST=1 in the environment flips AUTO.
    static const int env = [] { const char* e = getenv("XC_KEFF_PERSIST"); return e ? atoi(e) : 0; }();
    if (ctx->keff_mode == XC_KEFF_TWO_PASS || (ctx->keff_mode == XC_KEFF_AUTO && !env)) return false;
    const int cus = ctx->cus;
    if (cus < 8 || cus % 8 != 0) return false;
    if (nx < 4 || nx % 2 != 0 || ny < 2 || nx > 0x3fffffff || ny > 0x3fffffff) return false;
    if ((double)ny * (double)nx * 8.0 >= 2147483648.0) return false;             // 32-bit row offsets inside a slab
    const size_t esz = q_dtype == XC_F32 ? 4 : 8;
    if (reinterpret_cast<uintptr_t>(q) % (2 * esz) != 0) return false;           // two-cell vector loads
    if ((dA_rank == XC_DA_PLANE || dA_rank == XC_DA_SLAB) && reinterpret_cast<uintptr_t>(dA) % 16 != 0) return false;
    if (ny * nx < 65536) return false;                                           // tiny planes: the streaming path
    const int nstrip = (int)((nx + kPersistCols - 1) / kPersistCols);
    const int cps_min = (int)((ny + kPersistRows - 1) / kPersistRows);
    // Workgroups per CU (slots): 1 or 3, always 12 waves per CU.  Three slots = three groups sharing every CU = the waits of one
    // group hidden behind the work of the others -- possible when a slab is small enough for a group of cus x 4 waves.
    // Default: three slots when a group of that size holds a slab; XC_PERSIST_SLOTS forces a value (experiments).
    static const int env_slots = [] { const char* e = getenv("XC_PERSIST_SLOTS"); return e ? atoi(e) : 0; }();
    int slots = 0, ngps = 0;
    for (int sl = 3; sl >= 1; sl -= 2) {
        if (env_slots > 0 && sl != env_slots) continue;
        const int nw = kPersistThreads / 64 / sl;
        for (int ng = 8; ng >= 1; ng >>= 1) {                                    // groups per slot: as many as fit
            if ((int64_t)ng * sl > nslab && !(ng == 1 && sl == 1)) continue;
            if ((int64_t)nstrip * cps_min <= (int64_t)(cus / ng) * nw) { slots = sl; ngps = ng; break; }
        }
        if (slots) break;
    }
    if (!slots) return false;
    const int nwaves = kPersistThreads / 64 / slots, nthreads = kPersistThreads / slots;
    // LDS of ONE workgroup: edges + (2 doubles + 1 count) per bin and copy + the waiting tile rows; `slots` workgroups share a CU
    int ncopy = kMaxCopies;
    const size_t budget = (kLdsBudget / slots) & ~(size_t)15;
    const size_t tile_lds = (q_dtype == XC_F64) ? (size_t)nwaves * persist_lds_rows(nthreads) * 1024 : 0;
    const size_t fixed = (64 + ((N + 2) & ~1)) * sizeof(double) + 64 + 16 + tile_lds;
    while (ncopy > 1 && fixed + (size_t)N * ncopy * 20 > budget) ncopy >>= 1;
    if (fixed + (size_t)N * ncopy * 20 > budget) return false;
    const int G = cus / ngps;
    const int ngroups = ngps * slots;
    int cps = (int)(((int64_t)G * nwaves) / nstrip);
    int rpc = (int)((ny + cps - 1) / cps);
    if (rpc < 4) rpc = 4;                                                        // halo rows cost 2 loads per chunk
    if (rpc > kPersistRows) rpc = kPersistRows;
    cps = (int)((ny + rpc - 1) / rpc);
    g->slots = slots; g->ngps = ngps;
    g->G = G; g->ngroups = ngroups; g->nstrip = nstrip; g->cps = cps; g->rpc = rpc; g->ncopy = ncopy;
    g->lds = (fixed + (size_t)N * ncopy * 20 + 15) & ~(size_t)15;
    return true;
}

int launch_keff_persist(xc_ctx* ctx, int q_dtype, const PersistArgs& a, const PersistGeom& g)
{
    const bool da2d = a.dA_rank == XC_DA_PLANE || a.dA_rank == XC_DA_SLAB;
    const bool fast = a.periodic_x && a.dA_pos_finite && !a.last_closed;
    if (q_dtype == XC_F64) return launch_p2<double>(ctx, a, g.lds, da2d, fast);
    return launch_p2<float>(ctx, a, g.lds, da2d, fast);
}

}  // namespace xc
