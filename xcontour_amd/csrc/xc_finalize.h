// K5 / K6: the finalize stage (PDF -> CDF, Keff epilogue) as a device function + its helpers.  Included INSIDE
// `namespace xc { namespace {` of the translation units that run it (xc_misc.hip: the kernel k_finalize; xc_keff1.hip: the tail of
// the single-read Keff kernel).  Needs dnan() in scope.
#pragma once

// =====================================================================================
// K5 + K6  finalize: fixed-order reduction of the per-block partial histograms, PDF ->
// CDF (sequential np.cumsum order), optional lt flip / reversal, optional Keff epilogue.
// One 256-thread block per slab; everything lives in LDS.
// =====================================================================================
// np.interp(x, xp, fp) for ascending xp (numpy's compiled arr_interp, no precomputed slopes),
// split into the bracket search and the evaluation so that several fp can share one search.
// rev: the logical arrays are xp[n-1-i], fp[n-1-i] (the reference's decreasing case,
// core.py:1428-1430).  Returns j = largest index with X(j) <= x, or -1 (left), -2 (right), -3 (NaN).
__device__ __forceinline__ int interp_locate(double x, const double* __restrict__ xp, int n, int rev)
{
    auto X = [&](int i) { return rev ? xp[n - 1 - i] : xp[i]; };
    if (x != x) return -3;
    if (x > X(n - 1)) return -2;
    if (x < X(0)) return -1;
    int lo = 0, hi = n;            // X(lo) <= x, x < X(hi) (virtual)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (x >= X(mid)) lo = mid; else hi = mid; }
    return lo;
}

__device__ __forceinline__ double interp_eval(double x, int j, const double* __restrict__ xp,
                                              const double* __restrict__ fp, int n, int rev)
{
    auto X = [&](int i) { return rev ? xp[n - 1 - i] : xp[i]; };
    auto F = [&](int i) { return rev ? fp[n - 1 - i] : fp[i]; };
    if (j == -3) return x;
    if (j == -2) return F(n - 1);
    if (j == -1) return F(0);
    if (j == n - 1) return F(j);
    const double xj = X(j), fj = F(j);
    if (xj == x) return fj;
    const double slope = __ddiv_rn(__dsub_rn(F(j + 1), fj), __dsub_rn(X(j + 1), xj));
    double r = __dadd_rn(__dmul_rn(slope, __dsub_rn(x, xj)), fj);
    if (r != r) {
        r = __dadd_rn(__dmul_rn(slope, __dsub_rn(x, X(j + 1))), F(j + 1));
        if (r != r && F(j + 1) == fj) r = fj;
    }
    return r;
}

// np.gradient(f, uniform unit spacing, edge_order=1) at index k, f64
__device__ __forceinline__ double grad_f64(const double* f, int k, int N)
{
    if (N == 1) return 0.0;
    if (k == 0) return __dsub_rn(f[1], f[0]);
    if (k == N - 1) return __dsub_rn(f[N - 1], f[N - 2]);
    return __ddiv_rn(__dsub_rn(f[k + 1], f[k - 1]), 2.0);
}
// the same on float32 data (the default contour dtype): arithmetic stays in f32
__device__ __forceinline__ double grad_f32(const double* f, int k, int N)
{
    if (N == 1) return 0.0;
    if (k == 0) return (double)__fsub_rn((float)f[1], (float)f[0]);
    if (k == N - 1) return (double)__fsub_rn((float)f[N - 1], (float)f[N - 2]);
    return (double)__fdiv_rn(__fsub_rn((float)f[k + 1], (float)f[k - 1]), 2.0f);
}

// lane i <- lane i - 1 (DPP wave_shr:1, off the LDS pipe); lane 0 keeps `old`
__device__ __forceinline__ double lane_shr1_keep(double v, double old)
{
    const unsigned long long u = __double_as_longlong(v), o = __double_as_longlong(old);
    const unsigned lo = __builtin_amdgcn_update_dpp((int)(o & 0xffffffffu), (int)(u & 0xffffffffu), 0x138, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(u >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// the value lane `idx` holds (idx wave-uniform)
__device__ __forceinline__ double lane_value(double v, int idx)
{
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), idx);
    const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), idx);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// Stage 2: one workgroup per slab.  (A device function in a header: it ran for a while at the tail of the single-read Keff kernel too, in
// the workgroup that arrived last -- measured 12 us there against 6 us + a 1.8 us launch boundary as a kernel of its own, see
// profiles/r06_notes.md -- and may be wanted inside another kernel again.)
__device__ __forceinline__ void finalize_body(const FinalArgs& a, const int slab, const int tid, const int nthr, double* sm_lds)
{
    const int N = a.nbin, NCH = a.nch;
    auto dstamp = [&](int k) { if (a.dbg && tid == 0 && slab == 0) a.dbg[k] = wall_clock64(); };
    dstamp(0);
    // the single-read Keff kernel gave up waiting for its workgroups (its abort flag, xc_keff1.hip): nothing of this launch is valid
    if (a.abort_flag && *a.abort_flag != 0u) {
        if (tid == 0 && a.status_out) a.status_out[slab] = 2;
        return;
    }
    // thousands of contours: the work arrays live in global memory (same code; __syncthreads orders them)
    double* sm = a.big ? a.big + (size_t)slab * a.big_stride : sm_lds;
    double* s_pdf = sm;                   // [NCH][N]
    double* s_cdf = sm + (size_t)NCH * N; // [NCH][N] in LEVEL order (after optional reversal)
    double* s_x   = s_cdf + (size_t)NCH * N;   // 7*N scratch for the epilogue

    // Everything the epilogue needs from global memory is REQUESTED first, so that its latency runs under the reduction: the A(Yeq)
    // table (eight pairs per thread at a time; it goes into LDS for the look-ups -- these used to be a loop of dependent round trips in
    // the middle of the kernel: 7 of its 20 us per launch) and the contour levels (round 6: another dependent round trip in there).
    constexpr int TB = 8;
    double tb_t[TB], tb_c[TB];
    double* s_ctr = s_x;            double* s_lat = s_x + N;       double* s_lmin = s_x + 2 * N;
    double* s_dS  = s_x + 3 * N;    double* s_dq  = s_x + 4 * N;   double* s_leq = s_x + 5 * N;
    double* s_nk  = s_x + 6 * N;
    double* s_tbl = s_x + 7 * (size_t)N;
    const double ctr_mine = (a.keff && tid < N) ? a.ctr[(size_t)slab * a.vstride + tid] : 0.0;
    const bool tbl_regs = a.keff && a.tbl_in_lds && a.ntbl <= TB * nthr;
    if (tbl_regs) {
#pragma unroll
        for (int u = 0; u < TB; ++u) { const int i = tid + u * nthr; const int ic = i < a.ntbl ? i : a.ntbl - 1; tb_t[u] = a.tbl[ic]; tb_c[u] = a.tbl_coord[ic]; }
    }
    if (a.fuse_reduce) {
        // stage 1 folded in: the per-block partials of this slab summed in block order (fixed: deterministic).  The partials were
        // written by other XCDs a moment ago: every load is a ~1 us round trip to the fabric, so what matters is how many are in
        // flight -- RB blocks per thread and round (cfg2: 20 blocks per slab, five dependent rounds of four loads, two elements
        // one after the other, took 12 of this kernel's 25 us; now one round)
        constexpr int RB = 20;
        const int nvh = NCH * N, bps = a.bps, nel = nvh + (a.counts ? N : 0);
        const double* pp = a.part_h + (size_t)slab * bps * nvh;
        const unsigned* pc = a.part_c + (size_t)slab * bps * N;
        // one element per thread (launch_finalize sizes the workgroup for it): the weighted sums first, the counts behind them
        for (int i = tid; i < nel; i += nthr) {
            if (i < nvh) {
                double sum = 0.0;
                for (int b = 0; b < bps; b += RB) {
                    double v[RB];
#pragma unroll
                    for (int u = 0; u < RB; ++u) v[u] = pp[(size_t)(b + u < bps ? b + u : bps - 1) * nvh + i];
#pragma unroll
                    for (int u = 0; u < RB; ++u) if (b + u < bps) sum = __dadd_rn(sum, v[u]);
                }
                s_pdf[i] = sum;
            } else {
                const int k = i - nvh;
                unsigned long long c = 0;
                for (int b = 0; b < bps; b += RB) {
                    unsigned v[RB];
#pragma unroll
                    for (int u = 0; u < RB; ++u) v[u] = pc[(size_t)(b + u < bps ? b + u : bps - 1) * N + k];
#pragma unroll
                    for (int u = 0; u < RB; ++u) if (b + u < bps) c += v[u];
                }
                a.counts[(size_t)slab * N + (a.reverse ? N - 1 - k : k)] = c;
            }
        }
    } else {
        const double* ph = a.red_h + (size_t)slab * NCH * N;
        for (int i = tid; i < NCH * N; i += nthr)
            s_pdf[i] = ph[i];
        if (a.counts) {
            const unsigned long long* pc = a.red_c + (size_t)slab * N;
            for (int i = tid; i < N; i += nthr)
                a.counts[(size_t)slab * N + (a.reverse ? N - 1 - i : i)] = pc[i];
        }
    }
    if (a.keff) {
        // the levels and the table into LDS now (their loads were the first of the kernel: landed), covered by the SAME barrier as the sums
        for (int k = tid; k < N; k += nthr) s_ctr[k] = k == tid ? ctr_mine : a.ctr[(size_t)slab * a.vstride + k];
        if (a.tbl_in_lds) {
            if (tbl_regs) {
#pragma unroll
                for (int u = 0; u < TB; ++u) { const int i = tid + u * nthr; if (i < a.ntbl) { s_tbl[i] = tb_t[u]; s_tbl[a.ntbl + i] = tb_c[u]; } }
            } else {
                for (int i = tid; i < a.ntbl; i += nthr) { s_tbl[i] = a.tbl[i]; s_tbl[a.ntbl + i] = a.tbl_coord[i]; }
            }
        }
    }
    __syncthreads();
    dstamp(1);
    {
        // np.cumsum order (core.py:1320), strictly left to right, one WAVE per channel.  A chain of N dependent adds is all the
        // arithmetic there is; one thread walking the LDS paid ~85 cycles per element (load, add, store, loop: 7 us for 201
        // bins).  Systolic instead: lane l holds elements 4l .. 4l + 3 of a 256-element chunk, every step is
        // r0 = (r3 of lane l - 1; lane 0: the carry) + x0, r1 = r0 + x1, ... on all lanes -- two DPP moves and four adds.  After
        // step t lanes 0..t hold their final sums (a lane past its step recomputes the same values from a neighbour that no longer
        // changes), so 64 steps finish a chunk, in exactly the order ((p0 + p1) + p2) + ...
        const int wave = tid >> 6, lane = tid & 63, nw = nthr >> 6;
        for (int ch = wave; ch < NCH; ch += nw) {
            const double* p = s_pdf + (size_t)ch * N;
            double* c = s_cdf + (size_t)ch * N;
            double carry = 0.0;
            for (int k0 = 0; k0 < N; k0 += 256) {
                const int k = k0 + 4 * lane;
                double x[4], r[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = k + e < N ? p[k + e] : 0.0;      // (past N: + 0.0, the running sum stays)
                const int nstep = (N - k0 + 3) / 4 < 64 ? (N - k0 + 3) / 4 : 64;    // lanes past the last element only repeat its sum
                for (int t = 0; t < nstep; ++t) {
                    r[0] = __dadd_rn(lane_shr1_keep(r[3], carry), x[0]);
                    r[1] = __dadd_rn(r[0], x[1]); r[2] = __dadd_rn(r[1], x[2]); r[3] = __dadd_rn(r[2], x[3]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < N) c[a.reverse ? N - 1 - k - e : k + e] = r[e];   // level order (core.py:454-455)
                carry = lane_value(r[3], nstep - 1);                  // (the last lane that is final after nstep steps; its r[3] is the running total)
            }
            if (!a.lt)                                                // core.py:1322-1323; every lane revisits its own elements
                for (int k0 = 0; k0 < N; k0 += 256)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = k0 + 4 * lane + e;
                        if (k < N) { const int o = a.reverse ? N - 1 - k : k; c[o] = __dsub_rn(carry, c[o]); }
                    }
        }
    }
    __syncthreads();
    dstamp(2);
    for (int i = tid; i < NCH * N; i += nthr) {
        const int ch = i / N, k = i - ch * N;
        if (a.pdf) a.pdf[(size_t)slab * NCH * N + (size_t)ch * N + (a.reverse ? N - 1 - k : k)] = s_pdf[i];
        if (a.cdf) a.cdf[(size_t)slab * NCH * N + i] = s_cdf[i];
    }
    if (!a.keff) return;

    // ---------------- Keff epilogue, one thread per contour.  Two chains per contour that do not depend on each
    // other -- the look-up A -> Yeq -> Lmin (a bracket search, a division, a cosine) and the gradients dS/dA, dq/dA, Leq^2 (three divisions)
    // -- run side by side in the two halves of the workgroup (one wave per SIMD has nothing else to hide their latencies behind); the
    // quotient that needs both follows.  The arithmetic of every quantity is what it was.
    const double* area = s_cdf;           // channel 0
    const double* intS = s_cdf + N;       // channel 1
    const double* tblp = a.tbl_in_lds ? s_tbl : a.tbl;
    const double* crdp = a.tbl_in_lds ? s_tbl + a.ntbl : a.tbl_coord;
    const int tinc = tblp[a.ntbl - 1] > tblp[0];            // Table.__init__, core.py:1122-1128
    const int half = nthr >= 128 ? ((nthr / 2) & ~63) : 0;
    if (half == 0 || tid < half) {
        for (int k = tid; k < N; k += (half ? half : nthr)) {
            const int jt = interp_locate(area[k], tblp, a.ntbl, !tinc);
            const double le = interp_eval(area[k], jt, tblp, crdp, a.ntbl, !tinc);       // core.py:1136-1174
            const double lm = __dmul_rn(a.lmin_scale, cos(__dmul_rn(le, 0.017453292519943295)));   // utils.py:532
            s_lat[k] = le; s_lmin[k] = lm;
            const size_t o = (size_t)slab * a.vstride + k;
            if (a.o_latEq) a.o_latEq[o] = le;
            if (a.o_Lmin)  a.o_Lmin[o]  = lm;
        }
    }
    if (half == 0 || tid >= half) {
        for (int k = tid - half; k < N; k += nthr - half) {
            const double dA = grad_f64(area, k, N);
            const double dS = __ddiv_rn(grad_f64(intS, k, N), dA);                    // core.py:480-483
            const double dq = __ddiv_rn(a.ctr_f32 ? grad_f32(s_ctr, k, N) : grad_f64(s_ctr, k, N), dA);
            const double leq = __ddiv_rn(dS, __dmul_rn(dq, dq));                      // core.py:635
            s_dS[k] = dS; s_dq[k] = dq; s_leq[k] = leq;
            const size_t o = (size_t)slab * a.vstride + k;
            if (a.o_area)  a.o_area[o]  = area[k];
            if (a.o_intS)  a.o_intS[o]  = intS[k];
            if (a.o_dSdA)  a.o_dSdA[o]  = dS;
            if (a.o_dqdA)  a.o_dqdA[o]  = dq;
            if (a.o_Leq2)  a.o_Leq2[o]  = leq;
        }
    }
    __syncthreads();
    dstamp(3);
    for (int k = tid; k < N; k += nthr) {
        double nk = __ddiv_rn(__ddiv_rn(s_leq[k], s_lmin[k]), s_lmin[k]);         // core.py:963
        if (!(nk < a.nkeff_mask)) nk = dnan();                                    // core.py:964
        s_nk[k] = nk;
        if (a.o_nkeff) a.o_nkeff[(size_t)slab * a.vstride + k] = nk;
    }
    dstamp(4);
    if (a.o_interp && a.npre > 0) {
        __syncthreads();
        // interp_to_coords (core.py:1050-1100): direction from latEq[0] < latEq[-1]
        const int rev = !(s_lat[0] < s_lat[N - 1]);
        const double* vars[9] = {s_ctr, area, intS, s_lat, s_dS, s_dq, s_leq, s_lmin, s_nk};
        for (int p = tid; p < a.npre; p += nthr) {
            const double x = a.preY[p];
            const int j = interp_locate(x, s_lat, N, rev);        // one search shared by the 9 variables
#pragma unroll
            for (int v = 0; v < 9; ++v)
                a.o_interp[((size_t)slab * 9 + v) * a.npre + p] = interp_eval(x, j, s_lat, vars[v], N, rev);
        }
    }
}

