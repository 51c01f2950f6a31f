/*
 * xcontour_hip.h -- C ABI of libxcontour_hip.so: the MI355X (gfx950) implementation
 * of the contour-coordinate hot path of miniufo/xcontour.
 *
 * The reference (pure Python, /root/reference/xcontour/core.py) has no FFI seam of
 * its own; the seam this library fills is the pair of third-party calls that touch
 * every grid cell,
 *     xhistogram.xarray.histogram(var, bins=[edges], dim=dims, weights=w)   core.py:1284, 1307
 *     (var * dA).sum(dims)                                                  core.py:1376
 * plus the per-slab reductions and O(N) contour-space algebra around them.  Each
 * entry point below names the reference lines it replaces.  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, ints, doubles.  No C++ types cross.
 *   - every function returns an int status (XC_OK == 0, negative on error);
 *     xc_last_error(ctx) gives a message for the calling thread's last failure.
 *   - a "slab" is one (time, level) 2-D field of ny rows (the equivalent dim,
 *     lat/Z) by nx columns (lon/X), row-major with X fastest; slabs of a batch
 *     are contiguous: [nslab][ny][nx].
 *   - functions ending in _dev take DEVICE pointers and only enqueue work on the
 *     context's HIP stream (call xc_sync to wait); the same name without _dev
 *     takes HOST pointers (caller-owned, C-contiguous), stages them through a
 *     context-owned device arena and returns after the results are in the
 *     caller's buffers.
 *   - one context == one HIP device + one stream; calls on one context must be
 *     serialised by the caller; different contexts may be used concurrently.
 */
#ifndef XCONTOUR_HIP_H
#define XCONTOUR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xc_ctx xc_ctx;

/* status codes */
#define XC_OK        0
#define XC_EBADARG  (-1)   /* bad argument (null pointer, size, dtype, unsupported shape) */
#define XC_EEDGES   (-2)   /* bin edges not strictly ascending (reference: 'non monotonic bins', core.py:1233-1251) */
#define XC_EHIP     (-3)   /* HIP runtime error */
#define XC_ENOMEM   (-4)   /* device or host allocation failed */
#define XC_ENODEV   (-5)   /* no usable gfx950 device */

/* element types */
#define XC_F32 0
#define XC_F64 1

/* rank of the area-weight array dA (always passed as float64) */
#define XC_DA_NONE  0      /* weight 1 for every cell                       */
#define XC_DA_ROW   1      /* dA[ny]          one value per row             */
#define XC_DA_PLANE 2      /* dA[ny][nx]      shared by all slabs           */
#define XC_DA_SLAB  3      /* dA[nslab][ny][nx]                             */

/* last-bin rule (see oracle/xcontour_oracle.py header) */
#define XC_EDGE_NUMPY      0   /* last bin closed on the right (np.histogram)          */
#define XC_EDGE_XHISTOGRAM 1   /* last edge += 1e-8 in the edge dtype, bin half-open   */

#define XC_MAX_INTEGRANDS 2

/* X padding of xc_crossing: the `mode` of DataArray.pad / np.pad (core.py:674-676) */
#define XC_PAD_EDGE      0
#define XC_PAD_WRAP      1
#define XC_PAD_NAN       2     /* mode='constant' (xarray fills floats with NaN)      */
#define XC_PAD_REFLECT   3
#define XC_PAD_SYMMETRIC 4

/* ------------------------------------------------------------------ context */
int         xc_create(int device_id, xc_ctx** out);
int         xc_destroy(xc_ctx* ctx);
const char* xc_last_error(xc_ctx* ctx);          /* ctx may be NULL (creation errors) */
const char* xc_version(void);
int         xc_device_count(int* out_count);     /* visible HIP devices (a launcher maps local rank -> device with it) */
int         xc_device_name(xc_ctx* ctx, char* buf, size_t buflen);
int         xc_device_cus(xc_ctx* ctx, int* out_cus);
int         xc_sync(xc_ctx* ctx);
void*       xc_stream(xc_ctx* ctx);              /* the hipStream_t, for interop */
/* Where the host-form entry points of this context spent their time since the last reset, in seconds: out3[0] staging inputs
 * (memcpy into the pinned bounce buffer + enqueue, or the runtime's staged copy of a large block), out3[1] handing results over
 * (enqueue + memcpy out of the pinned buffer), out3[2] waiting for the stream.  Diagnostic (tools/facade_time.py --breakdown). */
int         xc_trace(xc_ctx* ctx, int reset, double* out3);

/* device memory + HIP-event timing on the context's stream */
int xc_malloc(xc_ctx* ctx, size_t bytes, void** out_dptr);
int xc_free(xc_ctx* ctx, void* dptr);
int xc_memcpy_h2d(xc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int xc_memcpy_d2h(xc_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int xc_memset(xc_ctx* ctx, void* dptr, int value, size_t bytes);
/* Resident inputs (no reference call site: the reference keeps everything in host memory).  The reference's Keff sequence hands
 * the SAME tracer to cal_contours and twice to cal_integral_within_contours_hist, the same weights to every call; the host-form
 * entry points upload their inputs on every call (~1 ms per 52 MB cfg2 slab over PCIe).  xc_keep_resident uploads `bytes` at
 * `host_ptr` once into memory the library owns; from then on any host-form entry point whose input lies inside
 * [host_ptr, host_ptr + bytes) -- the array itself or whole slabs of it -- takes it from that mirror on the device instead
 * (xc_minmax / xc_hist read the mirror in place; the others, and xc_memcpy_h2d[_async] from such a source, copy device to device).
 * The caller must not modify (or free) the host array while it is registered; calling xc_keep_resident on the same pointer
 * again refreshes the mirror; xc_release_resident(ctx, host_ptr) drops one entry, xc_release_resident(ctx, NULL) all.
 * Python: Contour2D(..., resident=True). */
int xc_keep_resident(xc_ctx* ctx, const void* host_ptr, size_t bytes);
int xc_release_resident(xc_ctx* ctx, const void* host_ptr);
/* the device address of the mirror of [host_ptr, host_ptr + bytes), or NULL if those bytes are not inside a registered array: a
 * caller that drives the _dev entry points itself (the fused pipeline) points its descriptor at the mirror instead of copying it */
int xc_resident_lookup(xc_ctx* ctx, const void* host_ptr, size_t bytes, void** out_dev);

/* Uploads that overlap compute: xc_memcpy_h2d_async copies on the context's second (copy) stream and returns when the
 * host buffer may be reused (pageable memory: when the copy is done; kernels enqueued earlier on the compute stream run
 * meanwhile).  xc_stream_wait_copies makes everything enqueued on the compute stream AFTER the call wait for the copies
 * issued so far; xc_copies_wait_stream makes later copies wait for the compute work enqueued so far (before a buffer that
 * kernels still read is overwritten). */
int xc_memcpy_h2d_async(xc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int xc_stream_wait_copies(xc_ctx* ctx);
int xc_copies_wait_stream(xc_ctx* ctx);
int xc_event_create(xc_ctx* ctx, void** out_event);
int xc_event_destroy(xc_ctx* ctx, void* event);
int xc_event_record(xc_ctx* ctx, void* event);
int xc_event_elapsed_ms(xc_ctx* ctx, void* start, void* stop, float* out_ms);  /* waits for `stop` */
int xc_event_record_copies(xc_ctx* ctx, void* event);   /* record on the COPY stream: done once the uploads issued so far have landed */
int xc_event_query(xc_ctx* ctx, void* event, int* out_done);   /* 1 when the event has completed, 0 while it has not; never blocks */

/* ------------------------------------------------------------------ K1  min / max
 * Replaces tracer.min(dim=dimVs), tracer.max(dim=dimVs)            core.py:224-225
 * NaN-skipping; an all-NaN slab gives NaN, NaN.  out_minmax: double[nslab][2].    */
int xc_minmax_dev(xc_ctx* ctx, const void* q, int q_dtype,
                  int64_t nslab, int64_t ncell, double* out_minmax);
int xc_minmax(xc_ctx* ctx, const void* q, int q_dtype,
              int64_t nslab, int64_t ncell, double* out_minmax);

/* ------------------------------------------------------------------ levels + edges
 * Replaces cal_contours(int levels)   core.py:222-249  (exact dtype rules: SURVEY 8-a2)
 * and the edge construction of _histogram  core.py:1296-1305.
 * minmax: double[nslab][2] as produced by xc_minmax.
 * ctr:    double[nslab][N]   levels rounded to ctr_dtype, in level order
 *         (ctr[0] == min if increase else max).
 * edges:  double[nslab][N+1] ascending histogram edges (dummy left edge first).
 * status: int32[nslab] 0 ok, 1 = two adjacent levels coincide (reference raises).    */
int xc_levels_dev(xc_ctx* ctx, const double* minmax, int q_dtype, int64_t nslab,
                  int N, int increase, int ctr_dtype, int right_edge,
                  double* ctr, double* edges, int32_t* status);
int xc_levels(xc_ctx* ctx, const double* minmax, int q_dtype, int64_t nslab,
              int N, int increase, int ctr_dtype, int right_edge,
              double* ctr, double* edges, int32_t* status);

/* cal_contours(int levels) (core.py:205-249) as ONE host-form call: xc_minmax + xc_levels without the round trip between them --
 * one upload of the tracer (or none: a resident one), one result hand-over, one synchronisation.  out_ctr double[nslab][N]; out_minmax
 * (double[nslab][2]), out_edges (double[nslab][N+1]) and out_status (int32[nslab]) may be NULL. */
int xc_contours(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ncell, int N, int increase, int ctr_dtype,
                int right_edge, double* out_minmax, double* out_ctr, double* out_edges, int32_t* out_status);

/* ------------------------------------------------------------------ K3+K5  weighted multi-channel histogram + CDF
 * Replaces histogram(var, bins=[edges], dim=dims, weights=w) + cumsum + lt flip
 *          core.py:1307-1323 (and the per-time loop 1259-1294: edges may differ per slab)
 *          and the weights of cal_integral_within_contours_hist   core.py:443-449.
 * One pass over the tracer accumulates nchan = 1 + nint + (grad ? 1 : 0) channels:
 *   channel 0            : dA                      (fillna(0))
 *   channel 1..nint      : integrand_i * dA        (fillna(0); product in f32 if prod_f32)
 *   last (if grad != 0)  : |grad q|^2 * dA computed in-kernel from q (build-defined
 *                          stencil, oracle grad2_sphere), using per-row rdx, rdy.
 * bin k = [edges[k], edges[k+1]); NaN / out-of-range cells dropped; last bin closed
 * iff last_closed.  nbin = nedge - 1.
 * Outputs (any may be NULL): pdf double[nslab][nchan][nbin]; counts uint64[nslab][nbin];
 * cdf double[nslab][nchan][nbin] = cumsum(pdf) (if !lt: cdf[-1]-cdf), reversed along
 * the bin axis iff reverse (decreasing levels, core.py:454-455).                     */
typedef struct xc_hist_desc {
    const void*   q;            int32_t q_dtype;  int32_t dA_pos_finite;  /* caller verified: dA finite and >= 0 (optional speed-up) */
    int64_t       nslab, ny, nx;
    const double* edges;        int64_t nedge;    int32_t edges_per_slab; int32_t last_closed;
    const double* dA;           int32_t dA_rank;  int32_t prod_f32;
    int32_t       nint;         int32_t grad;
    const void*   integrand[XC_MAX_INTEGRANDS];
    int32_t       integrand_dtype[XC_MAX_INTEGRANDS];
    const double* rdx;          /* [ny] 1/(2 dx)  per row (grad only)            */
    const double* rdy;          /* [ny] 1/(y[jn]-y[js]) per row (grad only)      */
    int32_t       periodic_x;   int32_t lt;
    int32_t       reverse;      int32_t negate;       /* negate: bin -q instead of q (right-closed bins on q) */
    double*       pdf;
    uint64_t*     counts;
    double*       cdf;
    int32_t       deterministic;/* != 0: order-free fixed-point sums (see "Deterministic sums" below); 0: float64 LDS atomics */
    int32_t       reserved0;
} xc_hist_desc;
int xc_hist_dev(xc_ctx* ctx, const xc_hist_desc* d);
int xc_hist(xc_ctx* ctx, const xc_hist_desc* d);

/* Deterministic sums (`deterministic != 0` in xc_hist_desc / xc_keff_desc).  The reference's per-bin sums come out of
 * np.bincount inside xhistogram (core.py:1284, 1307): the same input gives the same bits.  The default histogram pass
 * adds float64 weights with LDS atomics (and, in xc_keff_dev launches of FEW slabs -- more than 64 blocks per slab -- the
 * blocks' sums with global float64 atomics into per-slab accumulators), so the last bits of pdf / cdf (area, intgrdS) depend
 * on the order of arrival and differ from run to run (~1e-13 relative).  With `deterministic` every (bin, channel) owns a
 * fixed-point SUPERACCUMULATOR instead: four 48-bit limbs on a fixed grid below a window top that follows from bounds known
 * before the pass (max |dA|; for the in-kernel squared gradient 2 ((max - min) max(rdx, rdy))^2 max |dA| from K1's extrema; for a
 * supplied integrand max |integrand| max |dA|, from one extra min / max pass over it).  A weight is cut once to its leading 49
 * significant bits -- a function of the cell alone -- and added as two integer chunks (ds_add_u64) to the two adjacent limbs its
 * bits fall into, whatever its magnitude; the limbs are summed exactly and converted ONCE to float64 (round half to even).
 * Integer addition is associative: results are bit-identical between runs, launch geometries, slabs per launch and numbers
 * of ranks.  ONE pass over the cells (rounds 3-4 needed two: a per-bin scale came from a maximum pass): measured 1.3x the
 * default pass.  Precision: 2^-49 relative per weight anywhere in the 180 bits below the bound (the pole row of
 * a lat-lon grid carries squared gradients 2^100 times the typical ones: both keep their 49 bits); weights further down
 * lose their last bits, zeros and denormals contribute nothing; a bin that received an infinite weight reports NaN.
 * Levels, edges and counts are the same bits in both modes.  oracle/xcontour_oracle.py: deterministic_bin_sums.
 * Limit: the LDS histogram of this mode holds 5 words per channel and bin -- up to ~1700 contours with two weight channels (the Keff
 * layout), ~850 with four; beyond that the call fails with XC_EBADARG (the default sums take about three times as many). */

/* ------------------------------------------------------------------ K2  row sums for the A(Yeq) table
 * Replaces the degenerate histogram of cal_area_eqCoord_table_hist   core.py:176-193
 * rows[i] = sum_x dA[i,x] * [mask[i,x] == 1]; the prefix / end-point rules
 * are O(ny) host work.  mask may be NULL (all ones).
 * multiply != 0: rows[i] = nansum_x mask[i,x] * dA[i,x]  (the xarray twin, core.py:109-133). */
int xc_rowsum_dev(xc_ctx* ctx, const void* mask, int mask_dtype, const double* dA, int dA_rank,
                  int64_t ny, int64_t nx, int multiply, double* out_rows);
int xc_rowsum(xc_ctx* ctx, const void* mask, int mask_dtype, const double* dA, int dA_rank,
              int64_t ny, int64_t nx, int multiply, double* out_rows);

/* ------------------------------------------------------------------ K4  |grad q|^2 (stand-alone)
 * No reference call site (grdS is an input there); build-defined stencil.
 * out: double[nslab][ny][nx].                                                        */
int xc_grad2_dev(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                 const double* rdx, const double* rdy, int periodic_x, double* out);
int xc_grad2(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
             const double* rdx, const double* rdy, int periodic_x, double* out);

/* ------------------------------------------------------------------ K7  local wave activity / local APE
 * Replaces the python loop of cal_local_wave_activity   core.py:752-791
 *   lwa[j,x] = - sum_y' (q[y',x]-Q[j]) * mask3(j,y',x) * (dA[y',x]/dAmax) * M[y',x]
 * coord: double[ny] equivalent-coordinate values; Q: double[nslab][ny];
 * dA / M: float64, rank XC_DA_ROW or XC_DA_PLANE (M_rank may be XC_DA_NONE -> M = dA,
 * the snapshot text core.py:789).  part: 0 all, 1 upper, 2 lower.
 * variant 0: cal_local_wave_activity (qe = q - Q[j]); 1: cal_local_wave_activity2 core.py:802-905
 * (qe = q[row j] - Q, opposite sign convention).
 * out_lwa: double[nslab][ny][nx].  mask_idx: int32[nmask] rows whose mask3 is returned
 * in out_masks int8[nslab][nmask][ny][nx] (values -1/0/1); nmask may be 0.           */
int xc_lwa_dev(xc_ctx* ctx, const void* q, int q_dtype, const double* Q, const double* coord,
               const double* dA, int dA_rank, double dA_max, const double* M, int M_rank,
               int64_t nslab, int64_t ny, int64_t nx, int increase, int part, int variant,
               const int32_t* mask_idx, int nmask, double* out_lwa, int8_t* out_masks);
int xc_lwa(xc_ctx* ctx, const void* q, int q_dtype, const double* Q, const double* coord,
           const double* dA, int dA_rank, double dA_max, const double* M, int M_rank,
           int64_t nslab, int64_t ny, int64_t nx, int increase, int part, int variant,
           const int32_t* mask_idx, int nmask, double* out_lwa, int8_t* out_masks);

/* Planes of more than 512 rows take an O(ny log ny)-per-column path (variant 0): because the sorted reference state Q is
 * monotone, the targets j a cell contributes to form one interval, so one binary search in Q and four adds into a
 * difference array replace the walk over every (target, row) pair; per-column prefix sums finish.  The premises -- Q finite
 * (no NaN, no infinity), s*Q non-decreasing (s = +1 if increase else -1), the coordinate strictly monotone, no INFINITE tracer cell
 * (found by the interval kernel itself; NaN cells are fine) -- are checked ON THE DEVICE first (a
 * flag the kernels gate themselves on: no host round trip, the _dev form stays asynchronous); if they fail the band walk
 * enqueued behind the interval kernel runs instead.  Same sums in
 * another order: agreement with the band walk ~1e-13 of the column's largest value (tests 1e-9), not bit for bit.
 * xc_set_lwa_exact(ctx, mode): 0 automatic (above), 1 the bit-exact band walk for every plane, 2 the interval kernel for every plane
 * (checked on the device), 3 the same with the premises vouched for by the caller -- it has looked at Q, the coordinate and the
 * tracer (no infinite cell) on the host: one launch, no check (a plane of 256 x 512: 15.8 us for the band walk, see DESIGN.md for the interval kernel).  xc_last_lwa_path: 0 band walk, 1 interval
 * kernel, 2 its premises failed the check (waits for the call when the device decided). */
int xc_set_lwa_exact(xc_ctx* ctx, int exact);
int xc_last_lwa_path(xc_ctx* ctx, int* out_path);

/* ------------------------------------------------------------------ K8  exact adiabatic rearrangement (radix sort)
 * No reference call site (the reference "sorts" by histogram CDF + table lookup);
 * SURVEY 8-a9, pinned by oracle.sorted_profile.  One slab: drop NaN / mask != 1 cells, stable
 * ascending radix sort of (q, dA) pairs, Acum = cumsum(dA_sorted),
 *   Q[j] = q_sorted[min(searchsorted(Acum, targets[j], 'right'), nvalid-1)]   (increase, lt case)
 *   bpe  = sum_i q_sorted[i] * z*(Acum[i] - dA_i/2) * dA_i,  z* = np.interp(., tbl, coord)
 * negate != 0 sorts -q (decreasing tracers; outputs are values of -q).
 * mask may be NULL; dA_rank NONE / ROW / PLANE; any of out_Q (double[J]), out_qsorted
 * (double[ny*nx], invalid cells at the end), out_acum (double[ny*nx]), out_nvalid (uint32),
 * out_bpe (double, needs tbl/coord) may be NULL.                                        */
int xc_sort_profile_dev(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype,
                        const double* dA, int dA_rank, int64_t ny, int64_t nx, int negate,
                        const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                        double* out_Q, double* out_qsorted, double* out_acum, uint32_t* out_nvalid, double* out_bpe);
int xc_sort_profile(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype,
                    const double* dA, int dA_rank, int64_t ny, int64_t nx, int negate,
                    const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                    double* out_Q, double* out_qsorted, double* out_acum, uint32_t* out_nvalid, double* out_bpe);

/* float64 tracers take a three-pass path: a stable LSD sort on a 24-bit RANGE key -- a monotone, piecewise-linear map of the
 * value range onto [0, 2^24 - 2] that gives densely populated parts of the range more keys (sampled histogram equalisation over
 * the plane's ROBUST range, the 9th smallest / largest of its block extrema; the few cells outside it -- a stray fill value -- get
 * 2^16 linear keys on either side instead of stretching the range) -- then a repair pass that puts every run of equal range key into exact order (stable rank inside the run, in
 * LDS) and proves the result sorted; its flag is read back -- the ONE host round trip these calls (also the _dev ones) make;
 * a stack that fails it (more than 128 distinct values inside one range key) is sorted again with the eight key passes.  The result is the same stable sort either way.  xc_last_sort_path: 0 = key passes only (float32
 * tracers, or XC_SORT_RANGE=0 in the environment at xc_create), 1 = range-key path, 2 = range-key path failed the check. */
int xc_last_sort_path(xc_ctx* ctx, int* out_path);

/* The same for a stack of nslab planes in ONE set of launches (segmented sort: every plane keeps its own
 * tile histograms, digit bases and cumulative area).  q: [nslab][ny][nx]; mask: [ny][nx] shared or
 * [nslab][ny][nx] (mask_per_slab); dA_rank NONE / ROW / PLANE (shared) or SLAB ([nslab][ny][nx]);
 * targets / tbl / coord shared.  Outputs: out_Q double[nslab][J], out_qsorted / out_acum
 * double[nslab][ny*nx], out_nvalid uint32[nslab], out_bpe double[nslab]; any may be NULL.        */
int xc_sort_profile_batch_dev(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype, int mask_per_slab,
                              const double* dA, int dA_rank, int64_t nslab, int64_t ny, int64_t nx, int negate,
                              const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                              double* out_Q, double* out_qsorted, double* out_acum, uint32_t* out_nvalid, double* out_bpe);
int xc_sort_profile_batch(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype, int mask_per_slab,
                          const double* dA, int dA_rank, int64_t nslab, int64_t ny, int64_t nx, int negate,
                          const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                          double* out_Q, double* out_qsorted, double* out_acum, uint32_t* out_nvalid, double* out_bpe);

/* ------------------------------------------------------------------ K9  box-counting contour crossing
 * Replaces Contour2D.cal_contour_crossing (core.py:640-693) and the numba kernel
 * _contour_crossing (core.py:1490-1566), for ALL contours of a slab in one pass.
 * The slab is padded on the X side by pad_x columns (mode pad_mode; pass 0 when the grid
 * has no 'X' dim, core.py:673-679); coarse shape Jn = round(ny/stride),
 * In = round((nx+pad_x)/stride) (round half to even, core.py:1510-1511); box (j, i),
 * j < Jn-1, covers corner rows j*s..j*s+s and corner columns i*s..i*s+s; contour c crosses
 * it iff some non-NaN corner <= c and some non-NaN corner > c (core.py:1531-1557).
 *   out_len[slab][k] = nansum over crossed boxes of sqrt(areaPad[j][i]) * stride
 *                      (area taken at the COARSE indices, core.py:1560; f32 area: f32 sqrt)
 *   out_cnt[slab][k] = number of crossed boxes (exact)
 * full_width == 0 scans box columns i < min(Jn, In) - 1 -- the reference's loop bound is
 * range(Jn-1) (core.py:1521), cut at In-1 where it would leave the array; full_width != 0
 * scans all In-1 columns.  contours: double[ncont] or double[nslab][ncont]
 * (contours_per_slab), ASCENDING, no NaN (the host entry point checks; callers with
 * other orders sort and un-permute, as xcontour_amd/core.py does).  area: [ny][nx] or
 * [nslab][ny][nx] (area_per_slab) of area_dtype.  Either output may be NULL.           */
int xc_crossing_dev(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                    int pad_x, int pad_mode, const double* contours, int ncont, int contours_per_slab,
                    const void* area, int area_dtype, int area_per_slab, int stride, int full_width,
                    double* out_len, uint64_t* out_cnt);
int xc_crossing(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                int pad_x, int pad_mode, const double* contours, int ncont, int contours_per_slab,
                const void* area, int area_dtype, int area_per_slab, int stride, int full_width,
                double* out_len, uint64_t* out_cnt);

/* ------------------------------------------------------------------ fused, batched Keff pipeline
 * The reference's call sequence SURVEY 3.1 steps 2-10 for a batch of slabs resident
 * in HBM: min/max -> levels/edges -> one histogram pass (dA, |grad q|^2 dA or grdS dA)
 * -> CDF -> table lookup -> d/dA -> Leq2 -> Lmin -> nkeff [-> interpolation to preY].
 * Replaces core.py:205-249, 412-460, 1136-1174, 463-488, 619-637, 945-966, 1050-1100
 * and utils.py:518-534.  Three kernel launches per batch, no host round trip.
 * All pointers are DEVICE pointers.  Outputs are double[nslab][N] unless noted and any
 * of them may be NULL except ctr/area.                                               */
typedef struct xc_keff_desc {
    const void*   q;            int32_t q_dtype;  int32_t ctr_dtype;
    int64_t       nslab, ny, nx;
    int32_t       N;            int32_t increase; int32_t lt; int32_t right_edge;
    const double* dA;           int32_t dA_rank;  int32_t grad;       /* grad: 1 in-kernel, 0 use grdS */
    const void*   grdS;         int32_t grdS_dtype; int32_t prod_f32;
    const double* rdx;          const double* rdy;  int32_t periodic_x; int32_t npre;
    const double* tbl;          /* double[ny] area table A(Yeq), ascending-coordinate order */
    const double* tbl_coord;    /* double[ny] ascending coordinate values                   */
    const double* preY;         /* double[npre] prescribed equivalent coordinates (or NULL) */
    double        nkeff_mask;   /* values >= this become NaN (reference default 1e5)        */
    double        lmin_scale;   /* 2*pi*R: Lmin = lmin_scale*cos(deg2rad(latEq))            */
    double*       ctr;          double* area;   double* intgrdS; double* latEq;
    double*       dqdA;         double* dintSdA; double* Leq2;   double* Lmin;  double* nkeff;
    uint64_t*     counts;       /* uint64[nslab][N]; NULL: not wanted -- the histogram pass then skips the count adds (a third of its LDS atomics) */
    double*       interp;       /* double[nslab][9][npre]: ctr, area, intgrdS, latEq, dintSdA, dqdA, Leq2, Lmin, nkeff on preY */
    int32_t*      status;       /* int32[nslab] 0 ok, 1 degenerate levels ('non monotonic bins', core.py:1233), 2 the single-read kernel gave up
                                   waiting for its workgroups (see single_read): nothing was computed for the slab */
    const void*   q_next;       /* optional: the batch the NEXT xc_keff_dev call will process (same dtype and
                                   shape, may equal q).  Its per-slab min/max partials are accumulated inside this
                                   call's histogram pass (+8 B/cell of loads, no extra kernel) and the next call on
                                   that pointer skips its K1 pass -- provided the batch is provably unchanged: the
                                   cached partials are dropped when xc_memcpy_h2d / xc_memset / xc_synth_dev / xc_free
                                   touch the batch, and whenever `q_gen` differs from the value given with q_next (a
                                   caller that writes the batch through its own pointer bumps q_gen).  NULL: off. */
    int32_t       dA_pos_finite;/* caller verified that every dA value is finite and >= 0: the kernel then skips the
                                   fillna(0) selects on the dA channel (optional speed-up; 0 is always safe) */
    int32_t       q_gen;        /* generation of the tracer buffers (see q_next): any change invalidates chained min/max */
    int32_t       deterministic;/* != 0: order-free fixed-point sums (below); q_next rides in their second pass */
    int32_t       out_stride;   /* doubles between consecutive slabs in each of the nine vector outputs (ctr ... nkeff); 0 = N
                                   (nine dense [nslab][N] arrays).  9 * N with ctr = base, area = base + N, ... lays the
                                   results out slab-major, [nslab][9][N] -- the block a rank hands to the one gather
                                   with no repacking pass.  counts / interp / status keep their dense layout. */
    double        dA_max;       /* deterministic sums only: the largest finite |dA| value, if the caller knows it (a static metric: computed once
                                   on the host); <= 0 or NaN: the library takes it from the device array with one extra min / max pass over dA per call */
    int32_t       single_read;  /* calls of ONE slab (the reference's own pattern: one (time, level) plane per call, tests/LWA.py:40-43;
                                   core.py:224-225 then 1307 per object): XC_SINGLE_AUTO (0) takes the single-read kernel -- min / max, levels and
                                   histogram in ONE launch with the slab held in registers between them, the tracer read once, then k_finalize --
                                   when grad = 1, the sums are not `deterministic` and the slab fits the chip's register tiles (3600 x 1801 does);
                                   XC_SINGLE_NEVER (1) keeps the min/max pass + histogram pass + finalize chain; XC_SINGLE_FORCE (2) takes the
                                   kernel for calls of TWO slabs too (measured slower there than the chain, 64.7 against 59.3 us: the two slabs
                                   run one after the other; kept for tests).  That kernel waits for ALL its workgroups to be resident at once;
                                   every wait is bounded (XC_KEFF_SINGLE_TIMEOUT_US, default 50 ms): when something else holds compute units
                                   for longer, status[slab] = 2 comes back, NOTHING is written to the slab's result vectors, and the caller
                                   repeats the call with XC_SINGLE_NEVER (pipeline.KeffPlan.fetch does). */
    int32_t       reserved0;
} xc_keff_desc;
#define XC_SINGLE_AUTO  0
#define XC_SINGLE_NEVER 1
#define XC_SINGLE_FORCE 2
int xc_keff_dev(xc_ctx* ctx, const xc_keff_desc* d);
/* which path the last xc_keff_dev call took: 0 the min/max + histogram + finalize chain (two reads of the tracer), 1 the single-read kernel */
int xc_last_keff_path(xc_ctx* ctx, int* out_path);
/* diagnostics of the single-read kernel: wall-clock stamps (100 MHz) of every workgroup at its phase boundaries, [2][CUs][slots] uint64 on
 * the device (enable = 0 frees them) */
int xc_dbg_single_stamps(xc_ctx* ctx, int enable, void** out_dev, int* out_slots);

/* K5 / K6 alone -- the Keff epilogue without the cell-touching passes (SURVEY 8b `xc_keff_epilogue`): from the per-bin sums of
 * dA and |grad q|^2 dA (the PDFs xc_hist returns, ascending-VALUE bin order, bin 0 = the dummy bin) to the nine Keff vectors.
 * Replaces, per slab: cumsum + `cdf[-1] - cdf` when not lt + reversal to level order (core.py:1320-1323, 454-455),
 * Table.lookup_coordinates (1136-1174), cal_gradient_wrt_area (463-488), cal_sqared_equivalent_length (619-637),
 * latitude_lengths_at (utils.py:518-534), cal_normalized_Keff (945-966) and, with npre > 0, interp_to_dataset (1050-1100).
 *   pdf   double[nslab][2][N]   channel 0: sum of dA per bin, channel 1: sum of integrand * dA per bin
 *   ctr   double[nslab][N]      the contour levels in LEVEL order (ctr_dtype: XC_F32 when they are float32 values: d/dk then
 *                               runs in float32 like np.gradient on a float32 array)
 *   tbl / tbl_coord  double[ntbl]  A(Yeq) table, ascending-coordinate order;  preY double[npre] or NULL
 *   outputs double[nslab][N] each (any may be NULL); interp double[nslab][9][npre] or NULL.
 * The _dev form takes device pointers and only enqueues; the host form stages through the context's arena. */
int xc_keff_epilogue_dev(xc_ctx* ctx, const double* pdf, const double* ctr, int ctr_dtype, int64_t nslab, int N,
                         int increase, int lt, const double* tbl, const double* tbl_coord, int ntbl,
                         const double* preY, int npre, double nkeff_mask, double lmin_scale,
                         double* area, double* intgrdS, double* latEq, double* dqdA, double* dintSdA,
                         double* Leq2, double* Lmin, double* nkeff, double* interp);
int xc_keff_epilogue(xc_ctx* ctx, const double* pdf, const double* ctr, int ctr_dtype, int64_t nslab, int N,
                     int increase, int lt, const double* tbl, const double* tbl_coord, int ntbl,
                     const double* preY, int npre, double nkeff_mask, double lmin_scale,
                     double* area, double* intgrdS, double* latEq, double* dqdA, double* dintSdA,
                     double* Leq2, double* Lmin, double* nkeff, double* interp);

/* time of the dominant kernel (the histogram pass) of the last xc_keff_dev / xc_hist_dev
 * call, from HIP events recorded on the context's stream around that launch only.
 * Enable with xc_set_kernel_timing(ctx, 1); xc_last_hist_ms waits for the launch.     */
int xc_set_kernel_timing(xc_ctx* ctx, int enable);
int xc_last_hist_ms(xc_ctx* ctx, float* out_ms);
/* One-shot: record the caller's events (from xc_event_create) immediately before and after
 * the NEXT histogram launch instead of the context's own pair -- lets a benchmark time every
 * launch of a timed region without synchronising inside it.                              */
int xc_set_hist_events(xc_ctx* ctx, void* start_event, void* stop_event);

/* ------------------------------------------------------------------ X1  the one collective (RCCL over xGMI)
 * No reference call site (the reference has no multi-process path).  Independent slabs are
 * partitioned over one process per GPU; at the end of a job every rank contributes its block of
 * per-slab result vectors to ONE all-gather.  Rank 0 creates the 128-byte id, the launcher
 * distributes it (bench.py: torch.distributed store), every rank calls xc_comm_init.
 * xc_comm_allgather_dev enqueues ncclAllGather on the context's stream: `send` holds
 * bytes_per_rank bytes, `recv` nranks * bytes_per_rank (device pointers).                 */
int xc_comm_unique_id(xc_ctx* ctx, void* out_id128);
int xc_comm_init(xc_ctx* ctx, int nranks, int rank, const void* id128);
/* The same in two steps, for callers that put a deadline on the first contact with RCCL: ncclCommInitRank blocks until every rank has
 * joined, so it runs in a helper thread -- xc_comm_create touches NO context (device: the HIP device of the rank; err: the failure text),
 * xc_comm_attach hands the finished communicator to a context from the thread that owns it, xc_comm_release disposes of one that was
 * never attached (its creator stopped waiting). */
int xc_comm_create(int device, int nranks, int rank, const void* id128, void** out_comm, char* err, size_t errlen);
int xc_comm_attach(xc_ctx* ctx, void* comm, int nranks, int rank);
int xc_comm_release(void* comm);
/* What RCCL itself reports about the context's communicator (ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion, the
 * file librccl was loaded from): the evidence that a multi-GPU job's collective really spanned N ranks on N devices.
 * comm_count = 0: no RCCL communicator on this context (the gather runs on another carrier). */
typedef struct xc_comm_info_t {
    int32_t comm_count, comm_rank, comm_device, ctx_device;
    int32_t rccl_version, reserved0;
    char    rccl_path[256];
} xc_comm_info_t;
int xc_comm_info(xc_ctx* ctx, xc_comm_info_t* out);
int xc_comm_allgather_dev(xc_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank);
int xc_comm_finalize(xc_ctx* ctx);
/* Gather to ONE root (north_star: "RCCL gather"; SURVEY 8e "or gather-to-root"): every rank's `bytes` at `send` arrive at
 * recv + r * rank_stride on rank `root` (grouped ncclSend / ncclRecv; the root's own block is a device-to-device copy).  Moves
 * 1 / nranks of the all-gather's bytes.  Enqueued on the context's COMM stream (below), not on the compute stream. */
int xc_comm_gather_dev(xc_ctx* ctx, const void* send, size_t bytes, void* recv, size_t rank_stride, int root);
/* Give up on a communicator whose collective does not finish (ncclCommAbort): the first contact with RCCL on a new node runs
 * under a deadline (bench.py), and a job must be able to move on to the next carrier instead of hanging. */
int xc_comm_abort(xc_ctx* ctx);

/* The COMM stream: a third HIP stream of the context (beside compute and upload), created on first use.  A rank's result
 * block leaves on it -- an RCCL send or a device-to-device push -- while the next launch set computes.
 *   xc_comm_wait_compute  everything enqueued on the comm stream AFTER the call waits for the compute work enqueued so far
 *   xc_compute_wait_comm  later compute-stream work (and therefore xc_sync) waits for the comm work enqueued so far
 *   xc_comm_memcpy_d2d    device-to-device copy on the comm stream (dst may be another process's memory opened by xc_ipc_open)
 *   xc_streams_idle       *out_idle = 1 once both streams have drained, 0 while they have not; never blocks */
int xc_comm_wait_compute(xc_ctx* ctx);
int xc_compute_wait_comm(xc_ctx* ctx);
int xc_comm_memcpy_d2d(xc_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
int xc_streams_idle(xc_ctx* ctx, int* out_idle);

/* HIP IPC carrier of the same gather (no RCCL involved; no reference call site): the root exports the base pointer of its
 * receive buffer (an xc_malloc allocation) as a 64-byte handle, the rendezvous hands it to every rank, every rank opens it
 * and pushes its block with xc_comm_memcpy_d2d -- a peer write over xGMI between GPUs, a plain copy between ranks that share
 * one GPU.  Needs dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment on this driver stack (see bench.py).
 *   xc_ipc_export  handle of the ALLOCATION that starts at dptr        xc_ipc_open  map it (returns its base address)
 *   xc_ipc_close   unmap (drains the comm stream first)                xc_device_can_access_peer  hipDeviceCanAccessPeer */
int xc_ipc_export(xc_ctx* ctx, const void* dptr, void* out_handle64);
int xc_ipc_open(xc_ctx* ctx, const void* handle64, void** out_dptr);
int xc_ipc_close(xc_ctx* ctx, void* dptr);
int xc_device_can_access_peer(int device, int peer, int* out_can);

/* ------------------------------------------------------------------ host-only helpers of the facade (no context, no device)
 * d(var)/d(area) along the contour index -- cal_gradient_wrt_area (core.py:463-488): np.gradient of both arguments against their
 * contour coordinate (uniform spacing, edge_order 1) and the quotient, numpy's arithmetic operation for operation and dtype for dtype
 * (float32 stays float32 until it meets a float64).  var: nlead rows of n values, var_row_stride ELEMENTS apart (a row of a larger result
 * array is fine), area: area_rows = nlead or 1 rows (one area profile for every row), area_row_stride apart; the two coordinates: [n] each;
 * out: [nlead][n], dense, in the WIDER of the two dtypes.
 * Returns XC_OK, XC_EBADARG, or 1 when a coordinate is not equally spaced (or is float64 under a float32 array): nothing was written,
 * the caller takes np.gradient's general branch. */
int xc_host_gradient_wrt_area(const void* var, int var_dtype, const void* var_coord, int var_coord_dtype,
                              const void* area, int area_dtype, const void* area_coord, int area_coord_dtype,
                              int64_t nlead, int64_t n, int64_t area_rows, int64_t var_row_stride, int64_t area_row_stride, void* out);

/* Histogram edges from contour levels, the way _histogram prepares them (core.py:1296-1305: a dummy first edge one mean step below the
 * lowest level) plus xhistogram's `last edge + 1e-8` when right_edge = XC_EDGE_XHISTOGRAM -- both evaluated in the levels' OWN dtype --
 * for a stack: levels levels_dtype[nslab][N] in level order -> out_edges double[nslab][N + 1] ascending; *out_increasing = 1 when the
 * levels ascend (the direction of slab 0; a slab running the other way is an error unless it holds a NaN).  Errors, with the
 * reference's texts in xc_last_error(NULL): XC_EEDGES 'non monotonic bins' (two adjacent levels coincide, core.py:1233-1251),
 * XC_EBADARG 'need at least two contour levels' / 'not every time or level is increasing/decreasing'. */
int xc_host_edges_from_levels(const void* levels, int levels_dtype, int64_t nslab, int64_t N, int right_edge,
                              double* out_edges, int* out_increasing);

/* ------------------------------------------------------------------ synthetic slabs (bench / tests)
 * PV-like tracer q = sin(phi) + 0.25 sum_k a_k cos(k lambda + theta_k) cos^2(phi) + 0.02 eps
 * generated on device from a counter-based RNG.  variant 0: PV-like,
 * 1: pure noise, 2: sin(phi) only.  out: q_dtype[nslab][ny][nx]; slab s uses seed+s. */
int xc_synth_dev(xc_ctx* ctx, void* out, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                 const double* lat_deg, const double* lon_deg, uint64_t seed, int variant);

#ifdef __cplusplus
}
#endif
#endif /* XCONTOUR_HIP_H */
