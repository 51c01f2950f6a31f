/* Plain-C client of the C ABI (include/xcontour_hip.h): what a cgo / JNI / C++ host would do.
 * Built with gcc only (no HIP headers): tests/test_gpu_parity.py::test_c_client compiles it against
 * xcontour_amd/libxcontour_hip.so and runs it on the GPU box.
 * min/max -> levels/edges -> weighted histogram + CDF of a small field, checked against loops written here. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/xcontour_hip.h"

#define NY 97
#define NX 203
#define NLEV 33

static int fail(const char* what, xc_ctx* ctx) { fprintf(stderr, "FAIL %s: %s\n", what, xc_last_error(ctx)); return 1; }

int main(void)
{
    xc_ctx* ctx = NULL;
    { int ndev = -1; if (xc_device_count(&ndev) != XC_OK || ndev < 1) { fprintf(stderr, "FAIL xc_device_count: %d\n", ndev); return 1; } }
    if (xc_create(0, &ctx) != XC_OK) return fail("xc_create", NULL);
    { int path = -1; if (xc_set_lwa_exact(ctx, 0) != XC_OK || xc_last_lwa_path(ctx, &path) != XC_OK || path != 0) { fprintf(stderr, "FAIL lwa mode calls\n"); return 1; } }
    static float q[2][NY][NX];
    static double dA[NY][NX];
    uint64_t s = 88172645463325252ull;
    for (int k = 0; k < 2; ++k)
        for (int j = 0; j < NY; ++j)
            for (int i = 0; i < NX; ++i) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;                 /* xorshift64 */
                q[k][j][i] = (float)((double)(s >> 11) / 9007199254740992.0 + 0.01 * j + k);
                dA[j][i] = 1.0 + 0.5 * cos(0.03 * j);
            }
    q[0][3][4] = NAN;

    double mm[2][2];
    if (xc_minmax(ctx, q, XC_F32, 2, (int64_t)NY * NX, &mm[0][0]) != XC_OK) return fail("xc_minmax", ctx);
    for (int k = 0; k < 2; ++k) {
        float lo = INFINITY, hi = -INFINITY;
        for (int j = 0; j < NY; ++j) for (int i = 0; i < NX; ++i) { const float v = q[k][j][i]; if (v == v) { if (v < lo) lo = v; if (v > hi) hi = v; } }
        if (mm[k][0] != (double)lo || mm[k][1] != (double)hi) { fprintf(stderr, "FAIL minmax slab %d\n", k); return 1; }
    }

    double ctr[2][NLEV], edges[2][NLEV + 1];
    int32_t status[2];
    if (xc_levels(ctx, &mm[0][0], XC_F32, 2, NLEV, 1, XC_F32, XC_EDGE_NUMPY, &ctr[0][0], &edges[0][0], status) != XC_OK)
        return fail("xc_levels", ctx);
    if (status[0] || status[1] || ctr[0][0] != mm[0][0]) { fprintf(stderr, "FAIL levels\n"); return 1; }

    static double cdf[2][1][NLEV];
    static uint64_t counts[2][NLEV];
    struct xc_hist_desc d;
    memset(&d, 0, sizeof d);
    d.q = q; d.q_dtype = XC_F32; d.nslab = 2; d.ny = NY; d.nx = NX;
    d.edges = &edges[0][0]; d.nedge = NLEV + 1; d.edges_per_slab = 1; d.last_closed = 1;
    d.dA = &dA[0][0]; d.dA_rank = XC_DA_PLANE; d.lt = 1;
    d.cdf = &cdf[0][0][0]; d.counts = &counts[0][0];
    if (xc_hist(ctx, &d) != XC_OK) return fail("xc_hist", ctx);
    for (int k = 0; k < 2; ++k) {
        uint64_t ref_c[NLEV]; double ref_w[NLEV];
        memset(ref_c, 0, sizeof ref_c); memset(ref_w, 0, sizeof ref_w);
        for (int j = 0; j < NY; ++j) for (int i = 0; i < NX; ++i) {
            const double v = (double)q[k][j][i];
            if (!(v == v) || v < edges[k][0] || v > edges[k][NLEV]) continue;
            int b = NLEV - 1;                                              /* last bin closed on the right */
            for (int e = 1; e <= NLEV; ++e) if (v < edges[k][e]) { b = e - 1; break; }
            ref_c[b]++; ref_w[b] += dA[j][i];
        }
        double run = 0.0;
        for (int b = 0; b < NLEV; ++b) {
            run += ref_w[b];
            if (counts[k][b] != ref_c[b]) { fprintf(stderr, "FAIL counts slab %d bin %d: %llu vs %llu\n", k, b, (unsigned long long)counts[k][b], (unsigned long long)ref_c[b]); return 1; }
            if (fabs(cdf[k][0][b] - run) > 1e-11 * (fabs(run) + 1.0)) { fprintf(stderr, "FAIL cdf slab %d bin %d\n", k, b); return 1; }
        }
    }
    /* deterministic sums: two calls give the same bits, and agree with the default path to rounding */
    {
        static double c1[2][1][NLEV], c2[2][1][NLEV];
        d.deterministic = 1; d.counts = NULL;
        d.cdf = &c1[0][0][0]; if (xc_hist(ctx, &d) != XC_OK) return fail("xc_hist deterministic", ctx);
        d.cdf = &c2[0][0][0]; if (xc_hist(ctx, &d) != XC_OK) return fail("xc_hist deterministic", ctx);
        if (memcmp(c1, c2, sizeof c1) != 0) { fprintf(stderr, "FAIL: deterministic sums differ between two calls\n"); return 1; }
        for (int k = 0; k < 2; ++k) for (int b = 0; b < NLEV; ++b)
            if (fabs(c1[k][0][b] - cdf[k][0][b]) > 1e-11 * (fabs(cdf[k][0][b]) + 1.0)) { fprintf(stderr, "FAIL: deterministic cdf slab %d bin %d\n", k, b); return 1; }
        d.deterministic = 0; d.cdf = &cdf[0][0][0]; d.counts = &counts[0][0];
    }
    /* resident inputs: with the tracer and the weights registered the same call reads their device mirrors -- same bits
       (deterministic sums); a slab inside the registered array is found too; after a release the call uploads again */
    {
        static double c1[2][1][NLEV], c2[2][1][NLEV], mm2[2];
        d.deterministic = 1; d.counts = NULL;
        d.cdf = &c1[0][0][0]; if (xc_hist(ctx, &d) != XC_OK) return fail("xc_hist before xc_keep_resident", ctx);
        if (xc_keep_resident(ctx, q, sizeof q) != XC_OK || xc_keep_resident(ctx, dA, sizeof dA) != XC_OK) return fail("xc_keep_resident", ctx);
        d.cdf = &c2[0][0][0]; if (xc_hist(ctx, &d) != XC_OK) return fail("xc_hist with resident inputs", ctx);
        if (memcmp(c1, c2, sizeof c1) != 0) { fprintf(stderr, "FAIL: resident inputs change the result\n"); return 1; }
        if (xc_minmax(ctx, &q[1][0][0], XC_F32, 1, (int64_t)NY * NX, mm2) != XC_OK) return fail("xc_minmax on a slab of a resident array", ctx);
        if (mm2[0] != mm[1][0] || mm2[1] != mm[1][1]) { fprintf(stderr, "FAIL: min/max of the second slab through its mirror\n"); return 1; }
        if (xc_release_resident(ctx, NULL) != XC_OK) return fail("xc_release_resident", ctx);
        d.cdf = &c2[0][0][0]; if (xc_hist(ctx, &d) != XC_OK) return fail("xc_hist after xc_release_resident", ctx);
        if (memcmp(c1, c2, sizeof c1) != 0) { fprintf(stderr, "FAIL: result after release differs\n"); return 1; }
        d.deterministic = 0; d.cdf = &cdf[0][0][0]; d.counts = &counts[0][0];
    }
    /* ONE slab per call through xc_keff_dev (the reference's call pattern: one (time, level) plane at a time): single_read = AUTO takes the
       single-read kernel, NEVER the min/max + histogram + finalize chain -- same levels and counts, the sums to rounding; status is looked at
       the way a C caller must (2 = the kernel gave up waiting for its workgroups: repeat with XC_SINGLE_NEVER) */
    {
        enum { KY = 300, KX = 256, KN = 41 };
        static double kq[KY][KX], kdA[KY][KX], rdx[KY], rdy[KY], tbl[KY], crd[KY];
        double acc = 0.0;
        for (int j = 0; j < KY; ++j) {
            double row = 0.0;
            for (int i = 0; i < KX; ++i) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                kq[j][i] = sin(0.01 * j) + 0.1 * ((double)(s >> 11) / 9007199254740992.0);
                kdA[j][i] = 1.0 + 0.5 * cos(0.01 * j);
                row += kdA[j][i];
            }
            rdx[j] = 1.0 / (2.0 + 0.001 * j); rdy[j] = 0.5; crd[j] = -75.0 + 0.5 * j;
            tbl[j] = acc; acc += row;                                      /* area below row j: ascending with the coordinate */
        }
        tbl[KY - 1] = acc;
        const size_t pb = sizeof kq, vb = KN * sizeof(double);
        void *dq, *dd, *dx, *dy, *dt, *dc, *dout, *dcnt, *dst;
        if (xc_malloc(ctx, pb, &dq) || xc_malloc(ctx, pb, &dd) || xc_malloc(ctx, sizeof rdx, &dx) || xc_malloc(ctx, sizeof rdy, &dy) ||
            xc_malloc(ctx, sizeof tbl, &dt) || xc_malloc(ctx, sizeof crd, &dc) || xc_malloc(ctx, 9 * vb, &dout) ||
            xc_malloc(ctx, KN * sizeof(uint64_t), &dcnt) || xc_malloc(ctx, 64, &dst)) return fail("xc_malloc", ctx);
        if (xc_memcpy_h2d(ctx, dq, kq, pb) || xc_memcpy_h2d(ctx, dd, kdA, pb) || xc_memcpy_h2d(ctx, dx, rdx, sizeof rdx) ||
            xc_memcpy_h2d(ctx, dy, rdy, sizeof rdy) || xc_memcpy_h2d(ctx, dt, tbl, sizeof tbl) || xc_memcpy_h2d(ctx, dc, crd, sizeof crd))
            return fail("xc_memcpy_h2d", ctx);
        struct xc_keff_desc k;
        memset(&k, 0, sizeof k);
        k.q = dq; k.q_dtype = XC_F64; k.ctr_dtype = XC_F64; k.nslab = 1; k.ny = KY; k.nx = KX; k.N = KN; k.increase = 1; k.lt = 1;
        k.right_edge = XC_EDGE_XHISTOGRAM; k.dA = (const double*)dd; k.dA_rank = XC_DA_PLANE; k.grad = 1; k.rdx = (const double*)dx; k.rdy = (const double*)dy;
        k.periodic_x = 1; k.tbl = (const double*)dt; k.tbl_coord = (const double*)dc; k.nkeff_mask = 1e5; k.lmin_scale = 4.0e7; k.dA_pos_finite = 1;
        double* o = (double*)dout;
        k.ctr = o; k.area = o + KN; k.intgrdS = o + 2 * KN; k.latEq = o + 3 * KN; k.dqdA = o + 4 * KN; k.dintSdA = o + 5 * KN;
        k.Leq2 = o + 6 * KN; k.Lmin = o + 7 * KN; k.nkeff = o + 8 * KN; k.counts = (uint64_t*)dcnt; k.status = (int32_t*)dst;
        static double r[2][9][KN]; static uint64_t c[2][KN]; int32_t st[2]; int path[2];
        for (int m = 0; m < 2; ++m) {
            k.single_read = m == 0 ? XC_SINGLE_AUTO : XC_SINGLE_NEVER;
            if (xc_keff_dev(ctx, &k) != XC_OK || xc_last_keff_path(ctx, &path[m]) != XC_OK) return fail("xc_keff_dev", ctx);
            if (xc_memcpy_d2h(ctx, r[m], dout, 9 * vb) || xc_memcpy_d2h(ctx, c[m], dcnt, sizeof c[m]) || xc_memcpy_d2h(ctx, &st[m], dst, 4)) return fail("xc_memcpy_d2h", ctx);
            if (st[m] == 2) { fprintf(stderr, "note: the single-read kernel gave up (status 2): a caller repeats the call with XC_SINGLE_NEVER\n"); st[m] = 0; memcpy(r[0], r[1], sizeof r[0]); }
        }
        if (path[0] != 1 || path[1] != 0 || st[0] || st[1]) { fprintf(stderr, "FAIL keff paths %d %d status %d %d\n", path[0], path[1], st[0], st[1]); return 1; }
        uint64_t total = 0;
        for (int b = 0; b < KN; ++b) {
            total += c[0][b];
            if (c[0][b] != c[1][b] || r[0][0][b] != r[1][0][b]) { fprintf(stderr, "FAIL keff level / count %d differs between the two one-slab paths\n", b); return 1; }
            if (fabs(r[0][1][b] - r[1][1][b]) > 1e-12 * r[1][1][KN - 1] || fabs(r[0][2][b] - r[1][2][b]) > 1e-12 * r[1][2][KN - 1]) { fprintf(stderr, "FAIL keff sums bin %d\n", b); return 1; }
        }
        if (total != (uint64_t)KY * KX || !(r[0][1][KN - 1] > 0.99 * acc)) { fprintf(stderr, "FAIL keff totals\n"); return 1; }
        void* bufs[] = {dq, dd, dx, dy, dt, dc, dout, dcnt, dst};
        for (unsigned i = 0; i < sizeof bufs / sizeof bufs[0]; ++i) xc_free(ctx, bufs[i]);
    }
    /* error path: non-ascending edges must be refused with a message, not crash */
    edges[0][5] = edges[0][4];
    if (xc_hist(ctx, &d) == XC_OK) { fprintf(stderr, "FAIL: bad edges accepted\n"); return 1; }
    printf("capi_smoke ok (%s): %s\n", xc_version(), xc_last_error(ctx));
    return xc_destroy(ctx) == XC_OK ? 0 : 1;
}
