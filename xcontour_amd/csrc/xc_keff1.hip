// K3S -- the SINGLE-READ Keff kernel (gfx950): min/max -> levels -> weighted histogram with the in-kernel |grad q|^2 -> CDF -> Keff
// epilogue for ONE (or two) slabs in ONE launch, the tracer crossing the fabric ONCE.
//
// Replaces, for xc_keff_dev calls of at most kSingleMaxSlabs slabs (the reference's own call pattern: one (time, level) plane per call,
// tests/LWA.py:40-43; core.py:224-225 min / max, then 1307 the histogram, per object), the chain "K1 min/max pass -> K3 histogram
// pass -> k_finalize": three dependent launches that read the tracer twice because the levels need the exact extrema before the first
// cell can be binned.  Here the slab stays ON CHIP between the two steps:
//
//   * the grid is one 512-thread workgroup per CU (8 waves = 2 per SIMD at 230 VGPRs), all co-resident;
//   * a wave owns a chunk of <= 27 rows x 124 columns of the slab and holds it -- plus one halo row above and below and one halo
//     lane left and right (strips overlap by 4 columns, so x-neighbours always come from the adjacent lane by DPP) -- in 116 VGPRs per
//     lane: 256 CUs x 8 waves x 29 rows x 1 KiB = 58 MiB of register tile for a 51.9 MB slab.  Every load of the tile is issued at
//     once; the first XC_SINGLE_DR rows of the weights dA follow when the tile has landed (they would compete with it otherwise);
//   * step A: per-wave min/max of the tile -> workgroup -> ONE pair per slab by agent-scope 64-bit atomic max on order-preserving keys
//     in 8 shards -> an arrival counter per shard; every workgroup polls the 8 counters (one wave, bounded), reads the pair and builds
//     the N levels / N + 1 edges in LDS with exactly the arithmetic of the two-pass path (xc_binning.h);
//   * step B: the wave walks its rows in registers: nearest-edge guess + ONE exact comparison against the f64 edge in LDS (levels
//     equally spaced to a quarter of a bin, verified per slab; otherwise a rolled loop over the chunk with the general search of the
//     two-pass kernel), centred differences, weights, LDS atomics on lane-privatised copies; the rest of dA streams in DR rows ahead;
//   * step C: the workgroup ADDS its sums to the slab's accumulators (agent-scope float64 atomics, zeros not sent) and the kernel ends;
//     k_finalize (xc_misc.hip) reads them in the next launch.  (Round 6 measured the alternative -- tickets, the last workgroup runs
//     the finalize stage at the tail of this kernel -- at 2 us for the ticket + 12 us for the stage against 1.8 us of launch boundary +
//     6 us as a kernel: every agent-scope round trip inside a running kernel costs ~2 us here.  profiles/r06_notes.md.)
//
// Every wait on another workgroup is bounded by the wall clock: on a timeout (a grid that is not co-resident because something else
// holds CUs) the kernel raises an abort flag and every workgroup leaves; k_finalize sees the flag, writes status 2 for every slab of the
// call and NOTHING to their result vectors; the host side (pipeline.KeffPlan.fetch) then repeats the call on the two-pass path.
// The slots and accumulators live in two sets: launch n works in set n % 2 and clears the other one.
#include "xc_internal.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace xc {

namespace {

#include "xc_binning.h"

#ifndef XC_S_BAR
#define XC_S_BAR 3                     // a scheduling barrier every third row of step B (the rows' weight loads stay near their slots)
#endif
#ifndef XC_S_WPE
#define XC_S_WPE 2                     // waves per SIMD (= kSingleThreads / 256)
#endif
#ifndef XC_SINGLE_DR
#define XC_SINGLE_DR 10
#endif
constexpr int NT = kSingleThreads, NW = NT / 64;
constexpr int PR = kSingleRows;        // rows of a chunk
constexpr int PT = PR + 2;             // tile rows (halo row below and above)
constexpr int DR = XC_SINGLE_DR;       // rows of weights in flight in front of the row being binned
constexpr int PCOLS = kSingleCols;
constexpr int CW = 3;                  // LDS cell: area sum, |grad q|^2 dA sum, count

typedef double d2v __attribute__((ext_vector_type(2)));
typedef float  f2v __attribute__((ext_vector_type(2)));
template <typename TQ> struct Ld2;
template <> struct Ld2<double> {
    static __device__ __forceinline__ void ld(const char* row, unsigned voff, double (&o)[2]) {
        const d2v t = *reinterpret_cast<const d2v*>(row + voff); o[0] = t.x; o[1] = t.y; }
};
template <> struct Ld2<float> {
    static __device__ __forceinline__ void ld(const char* row, unsigned voff, double (&o)[2]) {
        const f2v t = *reinterpret_cast<const f2v*>(row + voff); o[0] = (double)t.x; o[1] = (double)t.y; }
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

// FAST: periodic X, dA verified finite and >= 0, half-open last bin (the xhistogram rule) -- the selects for walls, fillna and the
// closed last edge are compiled out.  Otherwise they are runtime (wave-uniform) flags.
template <typename TQ, bool DA2D, bool FAST, bool WCNT>
__global__ __attribute__((amdgpu_flat_work_group_size(NT, NT), amdgpu_waves_per_eu(XC_S_WPE, XC_S_WPE)))
void k_keff_single(const SingleArgs a)
{
    extern __shared__ __align__(16) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.nbin, ncopy = a.ncopy;
    const int cshift = __builtin_ctz((unsigned)ncopy);
    const int epad = (N + 2) & ~1;
    double*   s_red   = smem;                                  // [0 .. 2 NW): wave pairs; [32], [33]: the slab's pair
    int*      s_flag  = reinterpret_cast<int*>(smem + 48);     // [0] the wait succeeded
    double*   s_edges = smem + 64;                             // N + 1
    double*   s_cell  = s_edges + epad;                        // [(N + 1) * ncopy][CW]; bin N is the trash bin
    const int hsz = (N + 1) * ncopy;

    const int ny = (int)a.ny, nx = (int)a.nx;
    const int rank = (int)blockIdx.x;
    const int gw = rank * NW + wave;                           // wave index inside the grid
    // wave -> (strip, chunk).  strip_fast: the waves of a workgroup take ADJACENT strips of the same rows (12 KB contiguous per row and
    // workgroup); otherwise consecutive chunks of one strip (halo rows shared inside the workgroup)
    const int strip = a.strip_fast ? gw % a.nstrip : gw / a.cps;
    const int chunk = a.strip_fast ? gw / a.nstrip : gw - strip * a.cps;
    const bool work = a.strip_fast ? chunk < a.cps : strip < a.nstrip;
    const int r0 = work ? chunk * a.rpc : 0;
    const int nrows = work ? ((a.rpc < ny - r0) ? a.rpc : ny - r0) : 0;       // host: cps = ceil(ny / rpc) -> >= 1
    const int x0 = (work ? strip : 0) * PCOLS;
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
    const unsigned cstride = (unsigned)(CW * 8) << cshift, cbase = (unsigned)copy * (unsigned)(CW * 8);
    const size_t rowq = (size_t)nx * sizeof(TQ), rowd = (size_t)nx * 8;
    const size_t slabq = (size_t)ny * rowq, slabd = (size_t)ny * rowd;
    // slab-invariant per-row quantities, lane-distributed (lane i <-> chunk row i): gradient metrics, per-row weights
    const int ym = (r0 + lane < ny) ? r0 + lane : ny - 1;
    double rdxv = a.rdx[ym], rdyv = a.rdy[ym];
    double dArv = DA2D ? 0.0 : a.dA[ym];
    const bool wpos = FAST ? true : (a.dA_pos_finite != 0);
    const bool closed = FAST ? false : (a.last_closed != 0);
    constexpr bool want_cnt = WCNT;

    double T[PT][2];                                           // the register tile
    double dAb[DR][2];                                         // ring of weight rows
    // byte offset of tile row t (<-> slab row clamp(r0 - 1 + t)) inside a slab: 32 bits (host: slab < 2 GiB)
    auto srow = [&](int t) -> int { int y = r0 - 1 + t; return y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y); };
    auto stamp = [&](int sl, int k) {
        if (a.stamps && tid == 0) a.stamps[((size_t)sl * gridDim.x + blockIdx.x) * kSingleStampSlots + k] = wall_clock64();
    };
    auto load_tile = [&](int sl) {
        const char* q0 = reinterpret_cast<const char*>(a.q) + (size_t)sl * slabq;
        int nrw = nrows;
        asm volatile("" : "+s"(nrw));                          // opaque: the row offsets are recomputed at every use instead of kept (SGPR pressure)
#pragma unroll
        for (int t = 0; t < PT; ++t)                           // rows past the chunk: duplicates of its last halo row
            Ld2<TQ>::ld(q0 + (unsigned)srow(t < nrw + 1 ? t : nrw + 1) * (unsigned)rowq, voff_q, T[t]);
    };
    // the first DR rows of the weights: requested once the tile has LANDED (behind the min/max), not with it -- the tile is what the
    // whole grid waits for, the weights are not needed before step B, and 20 MB of them in front of the slowest workgroup's tile rows
    // moved the end of every workgroup's wait by ~3 us
    auto load_weights = [&](int sl) {
        if (DA2D) {
            const char* d0 = reinterpret_cast<const char*>(a.dA) + (a.dA_rank == XC_DA_SLAB ? (size_t)sl * slabd : 0);
            int nrw = nrows;
            asm volatile("" : "+s"(nrw));
#pragma unroll
            for (int j = 0; j < DR; ++j)                       // weights of chunk rows 1 .. DR
                Ld2<double>::ld(d0 + (unsigned)srow(j + 1 < nrw ? j + 1 : nrw) * (unsigned)rowd, voff_d, dAb[j]);
        }
    };
    stamp(0, 0);

    // the OTHER set of records (the next launch's): cleared here, by the first threads of the grid; the kernel boundary publishes it
    {
        const unsigned gt = (unsigned)blockIdx.x * NT + (unsigned)tid, gn = gridDim.x * NT;
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.other);
        constexpr unsigned nsync = (unsigned)((sizeof(SingleSlot) * kSingleMaxSlabs * kSingleMaxGrid + 64) / 8);      // (the slots, the abort flag)
        for (unsigned w = gt; w < nsync; w += gn) o[w] = 0ull;
        const unsigned nd = (unsigned)a.other_dirty_bins * kSingleMaxSlabs;      // (the launch that dirtied it may have used another N)
        for (unsigned w = gt; w < 2 * nd; w += gn) a.other->acc_h[w] = 0.0;
        for (unsigned w = gt; w < nd; w += gn) a.other->acc_c[w] = 0ull;
    }

    for (int s = 0; s < a.nslab; ++s) {
        // opaque per slab: what the rows derive from these (18 pairs of metrics by v_readlane, ...) is NOT hoisted out of the slab
        // loop and kept alive across it (it was: 240 VGPRs wanted)
        asm volatile("" : "+v"(rdxv), "+v"(rdyv), "+v"(dArv));
        // the whole tile requested at once.  (Loaded HERE, at the top of the slab's iteration, and not ahead of
        // the previous slab's flush: a tile that is alive across the loop's back edge costs a second set of 80 registers in copies.)
        load_tile(s);
        int tz = tid;
        asm volatile("" : "+v"(tz));                           // (the same for what the other phases derive from the thread index: a copy per phase)
        for (int i = tz; i < CW * hsz; i += NT) s_cell[i] = 0.0;
        // ------------------------------------------------------------ A: min / max of the resident tile -> the grid
        double mn = dinf(), mx = -dinf();
#pragma unroll
        for (int t = 0; t < PT; ++t) {                         // every tile register holds a real cell of this slab
#pragma unroll
            for (int c = 0; c < 2; ++c) { mn = fmin(mn, T[t][c]); mx = fmax(mx, T[t][c]); }       // NaN-skipping
        }
        for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
        if (lane == 0) { s_red[2 * wave] = mn; s_red[2 * wave + 1] = mx; }
        asm volatile("" : "+v"(mn), "+v"(mx) :: "memory");      // the weights are requested BEHIND the tile's last use above, not hoisted in front of it
        load_weights(s);
        __syncthreads();
        stamp(s, 1);
        // ALL-GATHER of the workgroups' pairs: every workgroup writes its own 16-byte slot (two write-through stores, nothing to wait
        // for) and one wave polls ALL slots until none is empty (~key(min) and key(max) are never zero; zero = "not there yet") -- one
        // round trip after the last arrival, where counters (atomic max x 2, wait, count, poll the counts, then read the pair) took
        // four.  G <= 256 slots = 4 KB; lane l watches slots l, l + 64, ...
        SingleSlot* slots = a.cur->slot[s];
        if (wave == 0) {
            double bmn = (lane < NW) ? s_red[2 * lane] : dinf(), bmx = (lane < NW) ? s_red[2 * lane + 1] : -dinf();
            for (int o = 8; o > 0; o >>= 1) { bmn = fmin(bmn, __shfl_xor(bmn, o)); bmx = fmax(bmx, __shfl_xor(bmx, o)); }
            if (lane == 0) {
                __hip_atomic_store(&slots[rank].kmn, ~dkey(bmn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slots[rank].kmx, dkey(bmx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            stamp(s, 2);
            const unsigned long long t0 = wall_clock64();
            int ok = 1;
            unsigned long long kmn = 0ull, kmx = 0ull;
            constexpr int SPL = kSingleMaxGrid / 64;           // slots per lane
            for (;;) {
                unsigned long long vn[SPL], vx[SPL];
#pragma unroll
                for (int j = 0; j < SPL; ++j) {                // every load of the sweep in flight together
                    const int i = lane + 64 * j, ic = i < a.G ? i : 0;
                    vn[j] = __hip_atomic_load(&slots[ic].kmn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vx[j] = __hip_atomic_load(&slots[ic].kmx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const unsigned ab = __hip_atomic_load(&a.cur->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool full = true;
                kmn = 0ull; kmx = 0ull;
#pragma unroll
                for (int j = 0; j < SPL; ++j) {
                    full = full && vn[j] != 0ull && vx[j] != 0ull;
                    kmn = vn[j] > kmn ? vn[j] : kmn; kmx = vx[j] > kmx ? vx[j] : kmx;
                }
                // (a workgroup that arrives after the others gave up must give up too: the abort flag is looked at before the slots)
                if (__builtin_amdgcn_readfirstlane((int)ab) != 0) { ok = 0; break; }
                if (__ballot(full) == ~0ull) break;
                if (wall_clock64() - t0 > a.timeout_ticks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (ok) {
                for (int o = 32; o > 0; o >>= 1) {
                    const unsigned long long an = __shfl_xor(kmn, o), ax = __shfl_xor(kmx, o);
                    kmn = an > kmn ? an : kmn; kmx = ax > kmx ? ax : kmx;
                }
                if (lane == 0) { s_red[32] = dunkey(~kmn); s_red[33] = dunkey(kmx); }
            } else if (lane == 0) {
                __hip_atomic_store(&a.cur->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) s_flag[0] = ok;
        }
        __syncthreads();
        stamp(s, 3);
        if (!__builtin_amdgcn_readfirstlane(s_flag[0])) return;       // gave up (the abort flag is up: k_finalize reports status 2); uniform
        double gmn = uniform_d(s_red[32]), gmx = uniform_d(s_red[33]);
        if (gmn == dinf() && gmx == -dinf()) { gmn = dnan(); gmx = dnan(); }      // all-NaN slab
        // ------------------------------------------------------------ levels / edges, exactly as the two-pass prologue;
        // every thread derives the two end edges itself (pure functions of the pair), so ONE barrier closes the phase
        const double c_first = level_value(gmn, gmx, 0, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1);
        const double c_last = level_value(gmn, gmx, N - 1, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1);
        const double lo = a.increase ? c_first : c_last, hi = a.increase ? c_last : c_first;
        const double e0 = uniform_d(dummy_edge(lo, hi, N, a.ctr_f32));
        const double eN = uniform_d(a.right_edge == XC_EDGE_XHISTOGRAM ? bump_last_edge(hi, a.ctr_f32) : hi);
        // (the spacing test and the bin guess take approximate quotients: both are tolerances around decisions that are made exactly)
        const double hstep = uniform_d((eN - e0) * a.inv_n);
        int bad = 0;
        for (int k = tid; k < N; k += NT) {
            const double c = level_value(gmn, gmx, k, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1);
            const int idx = a.increase ? k + 1 : N - k;
            const double e = (idx == N) ? eN : c;
            s_edges[idx] = e;
            if (rank == 0 && a.ctr_out) a.ctr_out[(size_t)s * a.ctr_stride + k] = c;
            if (k >= 1) bad |= (c == level_value(gmn, gmx, k - 1, a.increase, a.q_f32, a.ctr_f32, a.inv_nm1)) << 1;   // 'non monotonic bins', core.py:1233
            // equally spaced to a quarter of a bin?  (then ONE comparison against the nearest edge is exact)
            bad |= !(fabs(e - (e0 + (double)idx * hstep)) <= 0.25 * hstep);
        }
        if (tid == 0) s_edges[0] = e0;
        const bool uni = !__syncthreads_or(bad & 1);           // (__syncthreads_or returns a truth value, not the OR of the words)
        if (rank == 0 && a.status) {                           // (a workgroup-uniform branch: the barrier inside is reached by all of rank 0)
            const int degenerate = __syncthreads_or(bad & 2);
            if (tid == 0) a.status[s] = degenerate ? 1 : 0;
        }
        const double inv = uniform_d((double)N * __builtin_amdgcn_rcp(eN - e0));   // the guess only: exactness comes from the comparison
        stamp(s, 4);

        const char* dcur = reinterpret_cast<const char*>(a.dA) + (a.dA_rank == XC_DA_SLAB ? (size_t)s * slabd : 0);
        // one cell's weights and its three LDS adds; ku: bin, or N (the trash bin)
        auto cell = [&](unsigned ku, double q, double qW, double qE, double qS, double qN, double dv, double rdx, double rdy, int col) {
            double gx;
            if (FAST || periodic) {
                gx = __dmul_rn(__dsub_rn(qE, qW), rdx);
            } else {                                                   // walls: one-sided, spacing dx
                const bool wl = col == 0, wr = col == nx - 1;
                if (wl) qW = q;
                if (wr) qE = q;
                gx = __dmul_rn(__dsub_rn(qE, qW), rdx);
                gx = __dmul_rn(gx, (wl || wr) ? 2.0 : 1.0);
            }
            const double gy = __dmul_rn(__dsub_rn(qN, qS), rdy);
            const double g2 = __dadd_rn(__dmul_rn(gx, gx), __dmul_rn(gy, gy));
            const double p = __dmul_rn(g2, dv);
            const double w0 = wpos ? dv : ((dv != dv) ? 0.0 : dv);     // fillna(0), core.py:449
            const double w1 = wpos ? fmax(p, 0.0) : ((p != p) ? 0.0 : p);
            // byte offset of the cell = ku * (24 ncopy) + copy * 24: ONE full-rate v_mad_u32_u24 (ku <= N, the stride < 2^24)
            double* cp = reinterpret_cast<double*>(reinterpret_cast<char*>(s_cell) + (__umul24(ku, cstride) + cbase));
            lds_add(cp, w0);
            lds_add(cp + 1, w1);
            if (want_cnt) lds_add(reinterpret_cast<unsigned*>(cp + 2), 1u);
        };

        if (uni) {
            // -------------------------------------------------------- B: bin + accumulate from registers
            int nr = nrows;
            asm volatile("" : "+s"(nr));                       // opaque: the row predicates are not hoisted out of the slab loop (SGPR pressure)
            // the bin search of row i + 1 (guess + the LDS read of the edge) is issued BEFORE the LDS adds of row i: the LDS queue of a wave
            // is in order, and a read behind six atomics waits for all of them (measured: 6.6 -> ... us for the 27 rows of a wave)
            int jn[2]; double en[2];
            auto guess = [&](int t) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    int j = (int)__builtin_fma(T[t][c] - e0, inv, 0.5);                        // NaN -> 0
                    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(j) : "v"(j), "s"(N));                // clamp to [0, N]
                    jn[c] = j; en[c] = s_edges[j];
                }
            };
            guess(1);
#pragma unroll
            for (int i = 1; i <= PR; ++i) {
                if (i % XC_S_BAR == 1 || XC_S_BAR == 1) asm volatile("" ::: "memory");   // keep the rows' loads near their slots
                const bool rowok = i <= nr;                    // (a row past the chunk goes to the trash bin: no branch, ONE basic block for the scheduler)
                {
                    const double rdx = lane_get(rdxv, i - 1), rdy = lane_get(rdyv, i - 1);
                    double dAv[2];
                    if (DA2D) { dAv[0] = dAb[(i - 1) % DR][0]; dAv[1] = dAb[(i - 1) % DR][1]; }
                    else { const double v = lane_get(dArv, i - 1); dAv[0] = v; dAv[1] = v; }
                    const double (&qc)[2] = T[i];
                    unsigned b[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const double q = qc[c];
                        int kb = (q >= en[c]) ? jn[c] : jn[c] - 1;                         // NaN -> -1 (dropped); at or beyond the last edge -> N
                        if (!FAST) { if (closed && q == eN) kb = N - 1; }
                        const unsigned ku = (unsigned)kb < (unsigned)N ? (unsigned)kb : (unsigned)N;
                        b[c] = (cv[c] && rowok) ? ku : (unsigned)N;
                    }
                    if (i < PR) guess(i + 1);
                    // x-neighbours from the adjacent lanes (halo lanes are part of the wave: no special cases)
                    const double fromL = lane_shift_keep<DPP_WAVE_SHR1>(qc[1], qc[1]);
                    const double fromR = lane_shift_keep<DPP_WAVE_SHL1>(qc[0], qc[0]);
                    cell(b[0], qc[0], fromL, qc[1], T[i - 1][0], T[i + 1][0], dAv[0], rdx, rdy, col0);
                    cell(b[1], qc[1], qc[0], fromR, T[i - 1][1], T[i + 1][1], dAv[1], rdx, rdy, col0 + 1);
                }
                // the ring slot of row i is free: the weights of row i + DR take it
                if (DA2D && i + DR <= PR)
                    Ld2<double>::ld(dcur + (unsigned)srow(i + DR < nr ? i + DR : nr) * (unsigned)rowd, voff_d, dAb[(i - 1) % DR]);
            }
            asm volatile("" ::: "memory");
        }
        else {
            // -------------------------------------------------------- B': levels that are NOT equally spaced (float32 contours of a tiny
            // range, infinite extrema, an all-NaN slab): the general search of the two-pass kernel, rows re-read from the caches.  Rare.
            const char* q0 = reinterpret_cast<const char*>(a.q) + (size_t)s * slabq;
#pragma nounroll
            for (int i = 1; i <= nrows; ++i) {
                double qS[2], qc[2], qN[2], dAv[2];
                Ld2<TQ>::ld(q0 + (unsigned)srow(i - 1) * (unsigned)rowq, voff_q, qS);
                Ld2<TQ>::ld(q0 + (unsigned)srow(i) * (unsigned)rowq, voff_q, qc);
                Ld2<TQ>::ld(q0 + (unsigned)srow(i + 1) * (unsigned)rowq, voff_q, qN);
                if (DA2D) Ld2<double>::ld(dcur + (unsigned)srow(i) * (unsigned)rowd, voff_d, dAv);
                else { const double v = lane_get(dArv, i - 1); dAv[0] = v; dAv[1] = v; }
                const double rdx = lane_get(rdxv, i - 1), rdy = lane_get(rdyv, i - 1);
                unsigned b[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int kb = find_bin(qc[c], s_edges, N, e0, eN, inv, closed ? 1 : 0);
                    const unsigned ku = (unsigned)kb < (unsigned)N ? (unsigned)kb : (unsigned)N;
                    b[c] = cv[c] ? ku : (unsigned)N;
                }
                const double fromL = lane_shift_keep<DPP_WAVE_SHR1>(qc[1], qc[1]);
                const double fromR = lane_shift_keep<DPP_WAVE_SHL1>(qc[0], qc[0]);
                cell(b[0], qc[0], fromL, qc[1], qS[0], qN[0], dAv[0], rdx, rdy, col0);
                cell(b[1], qc[1], qc[0], fromR, qS[1], qN[1], dAv[1], rdx, rdy, col0 + 1);
            }
        }
        stamp(s, 5);
        __syncthreads();
        stamp(s, 7);
        // ------------------------------------------------------------ C: the workgroup's sums -> the slab's accumulators (k_finalize reads them)
        {
            int tf = tid;
            asm volatile("" : "+v"(tf));
            double* ah = a.cur->acc_h + (size_t)s * 2 * N;
            for (int i = tf; i < 2 * N; i += NT) {
                const int ch = i / N, bn = i - ch * N;
                const double* src = s_cell + (size_t)bn * ncopy * CW + ch;
                double sum = 0.0;
#pragma unroll 8
                for (int c = 0; c < ncopy; ++c) sum += src[(size_t)((c + tf) & (ncopy - 1)) * CW];
                if (sum != 0.0) __hip_atomic_fetch_add(ah + i, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (want_cnt)
                for (int bn = tf; bn < N; bn += NT) {
                    unsigned sum = 0u;
#pragma unroll 8
                    for (int c = 0; c < ncopy; ++c)
                        sum += *reinterpret_cast<const unsigned*>(s_cell + (size_t)(bn * ncopy + ((c + tf) & (ncopy - 1))) * CW + 2);
                    if (sum) __hip_atomic_fetch_add(a.cur->acc_c + (size_t)s * N + bn, (unsigned long long)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        stamp(s, 6);
        if (s + 1 < a.nslab) __syncthreads();                  // (the next slab zeroes the LDS copies)
    }
}

template <typename TQ, bool DA2D, bool FAST, bool WCNT>
int launch_s4(xc_ctx* ctx, const SingleArgs& a, const SingleGeom& g)
{
    auto kern = k_keff_single<TQ, DA2D, FAST, WCNT>;
    { const int rc = ensure_big_lds(ctx, reinterpret_cast<const void*>(kern), (int)kLdsBudget + 4096); if (rc != XC_OK) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)g.G), dim3(NT), g.lds, ctx->stream, a);
    XC_HIP(ctx, hipGetLastError());
    return XC_OK;
}

template <typename TQ, bool DA2D, bool FAST>
int launch_s3(xc_ctx* ctx, const SingleArgs& a, const SingleGeom& g)
{
    return a.want_counts ? launch_s4<TQ, DA2D, FAST, true>(ctx, a, g) : launch_s4<TQ, DA2D, FAST, false>(ctx, a, g);
}

template <typename TQ>
int launch_s2(xc_ctx* ctx, const SingleArgs& a, const SingleGeom& g, bool da2d, bool fast)
{
    if (da2d) return fast ? launch_s3<TQ, true, true>(ctx, a, g) : launch_s3<TQ, true, false>(ctx, a, g);
    return fast ? launch_s3<TQ, false, true>(ctx, a, g) : launch_s3<TQ, false, false>(ctx, a, g);
}

}  // namespace

// Decomposition of a (ny, nx) slab over the chip; false when the shape does not suit the single-read kernel (the caller then takes
// the two-pass path).
bool single_geometry(const xc_ctx* ctx, int q_dtype, int64_t nslab, int64_t ny, int64_t nx, int N, const void* q, const double* dA,
                     int dA_rank, SingleGeom* g)
{
    const int cus = ctx->cus;
    if (nslab < 1 || nslab > kSingleMaxSlabs || N > kSingleMaxBins || N < 2) return false;
    if (cus < 8 || cus > kSingleMaxGrid) return false;
    if (nx < 4 || nx % 2 != 0 || ny < 2 || nx > 0x3fffffff || ny > 0x3fffffff) return false;
    if ((double)ny * (double)nx * 8.0 >= 2147483648.0) return false;             // 32-bit row offsets inside a slab
    const size_t esz = q_dtype == XC_F32 ? 4 : 8;
    if (reinterpret_cast<uintptr_t>(q) % (2 * esz) != 0) return false;           // two-cell vector loads
    if ((dA_rank == XC_DA_PLANE || dA_rank == XC_DA_SLAB) && reinterpret_cast<uintptr_t>(dA) % 16 != 0) return false;
    if (ny * nx < 65536) return false;                                           // tiny planes: three short launches are as good
    const int Gmax = cus & ~7;                                                    // whole shards of the synchronisation records
    const int nstrip = (int)((nx + kSingleCols - 1) / kSingleCols);
    const int cps_min = (int)((ny + kSingleRows - 1) / kSingleRows);
    if ((int64_t)nstrip * cps_min > (int64_t)Gmax * NW) return false;             // the slab does not fit the register tiles of the chip
    int cps = (int)(((int64_t)Gmax * NW) / nstrip);
    int rpc = (int)((ny + cps - 1) / cps);
    if (rpc < 4) rpc = 4;                                                        // halo rows cost 2 loads per chunk
    if (rpc > kSingleRows) rpc = kSingleRows;
    cps = (int)((ny + rpc - 1) / rpc);
    int G = (int)(((int64_t)nstrip * cps + NW - 1) / NW);
    G = (G + 7) & ~7;
    if (G > Gmax) return false;
    // LDS: reductions + edges + one cell per (bin, copy); later the finalize stage's work arrays in the same bytes
    int ncopy = kMaxCopies;
    const size_t fixed = (64 + ((N + 2) & ~1)) * sizeof(double);
    while (ncopy > 2 && fixed + (size_t)(N + 1) * ncopy * CW * 8 > kLdsBudget) ncopy >>= 1;
    if (fixed + (size_t)(N + 1) * ncopy * CW * 8 > kLdsBudget) return false;
    const size_t lds = fixed + (size_t)(N + 1) * ncopy * CW * 8;
    g->G = G; g->nstrip = nstrip; g->cps = cps; g->rpc = rpc; g->ncopy = ncopy;
    g->lds = (lds + 15) & ~(size_t)15;
    return true;
}

int launch_keff_single(xc_ctx* ctx, int q_dtype, const SingleArgs& a, const SingleGeom& g)
{
    const bool da2d = a.dA_rank == XC_DA_PLANE || a.dA_rank == XC_DA_SLAB;
    const bool fast = a.periodic_x && a.dA_pos_finite && !a.last_closed;
    if (q_dtype == XC_F64) return launch_s2<double>(ctx, a, g, da2d, fast);
    return launch_s2<float>(ctx, a, g, da2d, fast);
}

}  // namespace xc
