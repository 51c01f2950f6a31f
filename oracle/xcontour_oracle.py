# -*- coding: utf-8 -*-
"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT a product path, NOT a fallback.

A numpy-only restatement of the contour-coordinate hot path of
miniufo/xcontour @ 2024_10_08 (`xcontour/core.py`, `xcontour/utils.py`),
operating on plain ndarrays.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this file, and only as the checker.
`xcontour_amd` never imports it and fails loudly without its HIP library.

PARITY UNPINNED
---------------
The reference cannot be imported in the build container (needs xarray,
xhistogram, numba, skimage, xgcm -- none installed, no network) and its own
tests contain no assertions / golden numbers (all 11 files under
/root/reference/tests are plot scripts; .travis.yml is `script: true`).
The arithmetic of the path lives in un-vendored third-party packages:

  * xhistogram (README.md:23 says 0.3.0; setup.py:40-45 unpinned) -- call
    sites core.py:1284, 1307.  Published algorithm: np.digitize against the
    explicit edges, out-of-range and NaN dropped, np.bincount(weights) in
    float64.  Restated here as `weighted_histogram` (digitize + bincount),
    cross-checked against `np.histogram` (numpy IS installed).
    Last-bin rule (the one place where xhistogram differs from np.histogram):
    xhistogram >= 0.1.2 (`core._bincount_2d_vectorized`: "Add small increment to
    the last bin edge to make the final bin right-edge inclusive") does
    `bins = [np.concatenate((b[:-1], b[-1:] + 1e-8)) for b in bins]`, i.e. the
    bump is evaluated IN THE EDGE DTYPE, then np.digitize with half-open bins and
    everything outside [first, last) thrown away.  Restated from the published
    source (not available offline; the reference's own docstrings / demo scripts
    do not contradict it).  Consequences on the reference's bundled barotropic
    field: float32 contours of magnitude 1e-4 DO absorb the bump (count 131 072:
    the max cell is kept), float32 latitudes (89.46) do NOT (the last row drops
    out of the A(Yeq) table).  `right_edge='xhistogram'` (DEFAULT = what the
    reference runs) and `right_edge='numpy'` (closed last bin, np.histogram) are
    both implemented.
  * xarray (README 0.15.1): min/max/where/sum/cumsum/differentiate/fillna/
    apply_ufunc(vectorize=True) -- restated with the numpy calls xarray
    dispatches to (np.nanmin, np.nansum, np.cumsum, np.gradient, np.vectorize).
  * numpy: np.interp, np.gradient, np.insert -- called directly.

What pins this file: (i) line-by-line citations below, (ii) the survey's
independently derived known-answer vectors on the reference's bundled
`Data/barotropic_vorticity.nc` (asserted in
tests/test_oracle_golden.py), (iii) identities the reference's own demo
scripts rely on (hist API == conditional API away from edges,
tests/test_hist.py:132-167; table end point == total area, core.py:133-140).

All `core.py:` / `utils.py:` citations are into /root/reference/xcontour/.
"""
import numpy as np

Rearth = 6371200.0   # utils.py:19


# ---------------------------------------------------------------------------
# a2  contour levels                                        core.py:205-266
# ---------------------------------------------------------------------------
def cal_contours(q, levels, increase=True, dtype=np.float32):
    """Per-slab contour levels.  `q` is (..., ny, nx); returns (..., N).

    int `levels` (core.py:222-249): N equally spaced levels from the slab's
    NaN-skipping min to max (max to min if not `increase`).  The arithmetic
    is what `xr.apply_ufunc(mylinspace, ..., vectorize=True,
    output_dtypes=[dtype])` evaluates: `stop-start` in the tracer dtype,
    `1.0/np.int64(N-1)` in f64, product / `*arange` / `+start` in f64, cast
    to `dtype` at the end.
    array `levels` (core.py:251-264): `tracer_min - tracer_min + levs`
    broadcast to every slab (NaN for an all-NaN slab), cast to `dtype`.
    """
    q = np.asarray(q)
    lead = q.shape[:-2]
    flat = q.reshape((-1,) + q.shape[-2:])
    if isinstance(levels, (int, np.integer)):
        N = int(levels)
        out = np.empty((flat.shape[0], N), dtype=dtype)
        inv = np.float64(1.0) / np.int64(N - 1)            # core.py:229-230
        for s in range(flat.shape[0]):
            mmin = _nanmin(flat[s])                          # core.py:224
            mmax = _nanmax(flat[s])                          # core.py:225
            start, stop = (mmin, mmax) if increase else (mmax, mmin)
            steps = inv * (stop - start)                     # f64 * T -> f64
            ctr = steps * np.arange(N) + np.float64(start)   # core.py:232
            out[s] = ctr.astype(dtype)                       # core.py:246
        return out.reshape(lead + (N,))
    levs = np.asarray(levels)
    out = np.empty((flat.shape[0], levs.size), dtype=dtype)
    for s in range(flat.shape[0]):
        mmin = _nanmin(flat[s])
        out[s] = (mmin - mmin + levs).astype(dtype)          # core.py:254
    return out.reshape(lead + (levs.size,))


def _nanmin(a):
    """xarray `.min()` on floats: NaN-skipping; all-NaN slab -> NaN."""
    a = np.asarray(a)
    if a.dtype.kind != 'f':
        return a.min()
    m = ~np.isnan(a)
    return a[m].min() if m.any() else a.dtype.type(np.nan)


def _nanmax(a):
    a = np.asarray(a)
    if a.dtype.kind != 'f':
        return a.max()
    m = ~np.isnan(a)
    return a[m].max() if m.any() else a.dtype.type(np.nan)


# ---------------------------------------------------------------------------
# a3  _histogram                                            core.py:1202-1325
# ---------------------------------------------------------------------------
def hist_edges(b):
    """Ascending N+1 edges from N levels `b` (in b's dtype)  core.py:1296-1305.

    Returns (edges, bincrease).  A dummy left edge one `step` below the lowest
    level is prepended so that #bins == #levels.  Raises like the reference if
    any two adjacent levels coincide (core.py:1233-1234).
    """
    b = np.asarray(b)
    if b.ndim != 1:
        raise Exception('bins should be numpy.array or xarray.DataArray')
    if not np.diff(b).all():
        raise Exception('non monotonic bins')
    bincrease = bool(b[0] < b[-1])
    n1 = len(b) - 1
    if bincrease:
        step = (b[-1] - b[0]) / n1
        edges = np.insert(b, 0, b[0] - step)
    else:
        step = (b[0] - b[-1]) / n1
        edges = np.insert(b[::-1], 0, b[-1] - step)
    return edges, bincrease


def weighted_histogram(x, edges, weights=None, right_edge='xhistogram', deterministic=False, det_top=None, det_limbs=4):
    """N-bin histogram of `x` over ascending `edges` (xhistogram semantics).

    bin k = [edges[k], edges[k+1]); NaN and out-of-range values dropped;
    weights summed in float64 by np.bincount (what xhistogram dispatches to).
    right_edge='xhistogram' : last edge replaced by `edges[-1] + 1e-8` evaluated in
                              the edge dtype, bin stays half-open (default: the
                              reference calls xhistogram, core.py:1284, 1307).
    right_edge='numpy'      : last bin closed on the right (np.histogram rule).
    Returns (sums_f64[N], counts_int64[N]).
    """
    x = np.asarray(x).ravel()
    edges = np.asarray(edges)
    nb = len(edges) - 1
    if right_edge == 'xhistogram':
        edges = np.concatenate((edges[:-1], edges[-1:] + 1e-8))
    elif right_edge != 'numpy':
        raise Exception('right_edge should be "numpy" or "xhistogram"')
    idx = np.digitize(x, edges)              # 0: below, nb+1: >= last / NaN
    if right_edge == 'numpy':
        idx = np.where(x == edges[-1], nb, idx)
    counts = np.bincount(idx, minlength=nb + 2)[1:nb + 1].astype(np.int64)
    if weights is None:
        return counts.astype(np.float64), counts
    w = np.asarray(weights, dtype=np.float64).ravel()
    if deterministic:
        return deterministic_bin_sums(idx, w, nb, det_top, det_limbs), counts
    sums = np.bincount(idx, weights=w, minlength=nb + 2)[1:nb + 1]
    return sums, counts


# ---- the order-free summation rule of the `deterministic` mode (round 5: ONE pass, a fixed-point superaccumulator)
DET_LIMB_BITS = 48        # S: the accumulator of a (bin, channel) is a row of limbs, limb i counting units of 2^(top - S (i + 1))
DET_PREC_BITS = 49        # P = S + 1: a weight rounded to P significant bits spans exactly two adjacent limbs
DET_TOP_SLACK = 12        # the window starts 12 bits above the bound of the weights: the top limb never overflows
DET_TOP_FLOOR = -800      # ... and never below 2^-800: zeros and denormals lie under every window


def det_window_top(bound):
    """exponent e with |w| < 2^e for every weight |w| <= bound (the exponent field of the bound, as frexp gives it for a normal
    number), + DET_TOP_SLACK, not below DET_TOP_FLOOR"""
    E = (int(np.float64(bound).view(np.uint64)) >> 52) & 0x7ff
    return max(E - 1022 + DET_TOP_SLACK, DET_TOP_FLOOR)


def det_chunks(w, top, nlimb):
    """The integer a weight contributes to its accumulator, in units of 2^(top - S nlimb) (the window's last bit).
    w = (-1)^sg m 2^x with m a 53-bit integer; n = m cut to its leading DET_PREC_BITS bits (ONE truncation per cell, a
    function of the cell alone); n 2^x is cut at the limb boundaries into a high and a low chunk; a chunk that lies below
    the window's last limb is dropped whole.  Zero, denormal and non-finite weights contribute nothing here."""
    S, P = DET_LIMB_BITS, DET_PREC_BITS
    b = int(np.float64(w).view(np.uint64))
    E = (b >> 52) & 0x7ff
    if E == 0 or E == 0x7ff:
        return 0
    m = (b & ((1 << 52) - 1)) | (1 << 52)
    n = m >> (53 - P)                           # the low 4 bits of the significand are dropped
    d = top + 1023 + 52 - (53 - P) - E          # bits from the rounded value's last bit up to the window top (>= P)
    j = (d - 1) // S                            # limb that holds the last bit (limb 0 is the most significant)
    sh = S * (j + 1) - d
    hi, lo = n >> (S - sh), (n << sh) & ((1 << S) - 1)
    t = 0
    if j - 1 < nlimb:
        t += hi << (S * (nlimb - j))
    if j < nlimb:
        t += lo << (S * (nlimb - 1 - j))
    return -t if (b >> 63) else t


def deterministic_bin_sums(idx, w, nb, top=None, nlimb=4):
    """Order-free per-bin sums (BUILD-DEFINED: the `deterministic` mode of the HIP histogram pass, include/xcontour_hip.h
    "Deterministic sums"; the reference itself sums with np.bincount, core.py:1284, 1307: the same input, the same bits).
    Every weight is cut ONCE to its leading 49 significant bits and becomes an exact integer multiple of the accumulator's last bit
    (`det_chunks`); the integers of a bin are added exactly and the total is converted ONCE to float64 (round half to even).
    Integer addition is associative: the result does not depend on the order of the cells, the block geometry, the launch
    partition or the number of ranks -- the GPU must reproduce it BIT FOR BIT.
    `top`: the window's first bit, det_window_top(bound) for a rigorous bound of |w| known BEFORE the pass (the GPU: max dA; for
    the in-kernel squared gradient 2 ((max - min) max(rdx, rdy))^2 max dA; max |integrand| max dA) -- default: from max |w|
    itself.  The window holds 48 `nlimb` bits (the GPU: 4 limbs, 192 bits): weights more than 2^-119 below the bound start
    losing their last bits, like in any finite accumulator -- the pole row of a lat-lon grid carries squared gradients 2^100
    times the typical ones and both kinds keep all their 49 bits.
    A bin that holds an infinite weight yields NaN.  idx: bin of every cell in 1..nb (np.digitize convention; everything
    else is dropped)."""
    import math
    idx = np.asarray(idx); w = np.asarray(w, dtype=np.float64)
    ok = (idx >= 1) & (idx <= nb)
    b, v = idx[ok] - 1, w[ok]
    out = np.zeros(nb, dtype=np.float64)
    fin = np.isfinite(v)
    if top is None:
        mx = float(np.max(np.abs(v[fin]))) if fin.any() else 0.0
        top = det_window_top(mx if mx > 0 else 1.0)
    S, P = DET_LIMB_BITS, DET_PREC_BITS
    ebot = top - S * nlimb
    bad = np.bincount(b[~fin], minlength=nb) > 0                  # a bin that saw an infinite weight reports NaN
    # det_chunks for every cell at once (uint64 arithmetic; the per-bin sums of the 48-bit chunks are taken in two 24-bit
    # halves with np.bincount, whose float64 accumulators hold integers below 2^53 exactly)
    u = v.view(np.uint64)
    E = ((u >> np.uint64(52)) & np.uint64(0x7ff)).astype(np.int64)
    live = (E != 0) & (E != 0x7ff)
    m = (u & np.uint64((1 << 52) - 1)) | np.uint64(1 << 52)
    n = m >> np.uint64(53 - P)
    d = top + 1023 + 52 - (53 - P) - E
    j = (d - 1) // S
    sh = (S * (j + 1) - d).astype(np.uint64)
    hi = n >> (np.uint64(S) - sh)
    lo = (n & ((np.uint64(1) << (np.uint64(S) - sh)) - np.uint64(1))) << sh
    neg = (u >> np.uint64(63)) == 1
    tot = [0] * nb
    m24 = np.uint64((1 << 24) - 1)
    for L in range(nlimb):
        c = np.where(live & (j - 1 == L), hi, np.uint64(0)) + np.where(live & (j == L), lo, np.uint64(0))     # a cell feeds a limb at most once
        for sgn, sel in ((1, ~neg), (-1, neg)):
            if not sel.any():
                continue
            cl = np.bincount(b[sel], weights=(c[sel] & m24).astype(np.float64), minlength=nb)
            ch = np.bincount(b[sel], weights=(c[sel] >> np.uint64(24)).astype(np.float64), minlength=nb)
            for k in range(nb):
                if cl[k] or ch[k]:
                    tot[k] += sgn * ((int(ch[k]) << 24) + int(cl[k])) << (S * (nlimb - 1 - L))
    for k in range(nb):
        out[k] = np.nan if bad[k] else (math.ldexp(float(tot[k]), ebot) if tot[k] else 0.0)     # float(int): correctly rounded, half to even
    return out


def histogram_cdf(q, b, weights, lt, right_edge='xhistogram', deterministic=False, det_top=None, det_limbs=4):
    """`_histogram(var, bins, dim, weights, lt)` for one slab (core.py:1296-1325).

    Result is in ASCENDING-VALUE order (position i <-> i-th smallest level),
    exactly as `_histogram` returns its data; the caller (a4) reverses it when
    the levels were decreasing.  Returns (cdf, pdf, counts, bincrease).
    """
    edges, bincrease = hist_edges(b)
    q = np.asarray(q)
    w = np.broadcast_to(np.asarray(weights), q.shape)
    pdf, counts = weighted_histogram(q, edges, w, right_edge, deterministic, det_top, det_limbs)
    cdf = np.cumsum(pdf)                                     # core.py:1320
    if not lt:
        cdf = cdf[-1] - cdf                                  # core.py:1322-1323
    return cdf, pdf, counts, bincrease


# ---------------------------------------------------------------------------
# a4  conditional integrals                                 core.py:363-460
# ---------------------------------------------------------------------------
def _weights(dA, integrand, shape):
    """`wei = (integrand*dA or dA).fillna(0)`  core.py:443-449 (result dtype of
    the product is numpy's promotion of the two operands)."""
    dA = np.asarray(dA)
    if dA.ndim == 1:                  # per-row metric, broadcast along X
        dA = dA[:, None]
    wei = dA if integrand is None else np.asarray(integrand) * dA
    wei = np.broadcast_to(wei, shape)
    return np.where(np.isnan(wei), wei.dtype.type(0), wei)


def cal_integral_within_contours_hist(q, ctr, dA, integrand=None, lt=False,
                                      right_edge='xhistogram', return_counts=False, deterministic=False, det_top=None, det_limbs=4):
    """core.py:412-460 for one slab: out[k] <-> ctr[k] whatever the direction."""
    q = np.asarray(q)
    wei = _weights(dA, integrand, q.shape)
    cdf, pdf, counts, binc = histogram_cdf(q, ctr, wei, lt, right_edge, deterministic, det_top, det_limbs)
    if not binc:                                             # core.py:454-455
        cdf, pdf, counts = cdf[::-1], pdf[::-1], counts[::-1]
    if return_counts:
        return cdf, counts
    return cdf


def cal_integral_within_contours(q, ctr, dA, integrand=None, lt=False):
    """The xarray conditional-integration twin (core.py:363-409), intended
    semantics: NaN-skipping sum over the two plane dims of
    `integrand.where(q < c_k) * dA`, strict comparison at every k."""
    q = np.asarray(q)
    ctr = np.asarray(ctr)
    dA2 = np.asarray(dA)
    if dA2.ndim == 1:
        dA2 = dA2[:, None]
    if integrand is None:
        integrand = q - q + 1                                # core.py:396
    f = np.asarray(integrand) * dA2
    out = np.empty(ctr.shape[-1], dtype=np.float64)
    for k in range(ctr.shape[-1]):
        cond = (q < ctr[k]) if lt else (q > ctr[k])          # core.py:398-401
        out[k] = np.nansum(np.where(cond, f, np.nan), dtype=np.float64)
    return out


# ---------------------------------------------------------------------------
# a5  area <-> equivalent-coordinate table                  core.py:73-203
# ---------------------------------------------------------------------------
def cal_area_eqCoord_table_hist(mask, dA, coord, increase=True, lt=False,
                                right_edge='xhistogram'):
    """core.py:150-203.  Returns (tbl, coord_ascending)."""
    mask = np.asarray(mask)
    coord = np.asarray(coord)
    ctrVar = np.broadcast_to(coord[:, None], mask.shape)     # core.py:176
    ctrVar = np.where(mask == 1, ctrVar, np.nan)             # core.py:178
    yIncre = not (coord[-1] < coord[0])                      # core.py:180-182
    ylt = lt if (increase == yIncre) else (not lt)           # core.py:184-188
    dA2 = np.asarray(dA)
    if dA2.ndim == 1:
        dA2 = dA2[:, None]
    w = np.broadcast_to(dA2, mask.shape)                     # no fillna here
    cdf, _, _, _ = histogram_cdf(ctrVar, coord, w, ylt, right_edge)
    cs = coord if yIncre else coord[::-1]                    # core.py:195-198
    return cdf, cs.copy()


def cal_area_eqCoord_table(mask, dA, coord, increase=True, lt=False):
    """core.py:73-147 (xarray twin), intended semantics.  Keeps the original
    coordinate order; end point overwritten with the total masked area."""
    mask = np.asarray(mask, dtype=np.float64)
    coord = np.asarray(coord)
    dA2 = np.asarray(dA)
    if dA2.ndim == 1:
        dA2 = dA2[:, None]
    eqDimIncre = coord[-1] > coord[0]
    same = (eqDimIncre == increase)
    less = (same if lt else (not same))      # core.py:103-128: '<' or '>' case
    J = len(coord)
    tbl = np.empty(J, dtype=np.float64)
    cv = coord[:, None]
    for j in range(J):
        cond = (cv < coord[j]) if less else (cv > coord[j])
        tbl[j] = abs(np.nansum(np.where(cond, mask, np.nan) * dA2))
    maxArea = abs(np.nansum(mask * dA2))                     # core.py:133
    if tbl[-1] > tbl[0]:                                     # core.py:136-140
        tbl[-1] = maxArea
    else:
        tbl[0] = maxArea
    return tbl, coord.copy()


# ---------------------------------------------------------------------------
# a6  Table.lookup_coordinates / _interp1d                  core.py:1103-1174, 1405-1434
# ---------------------------------------------------------------------------
def interp1d(x, xf, yf, inc=True):
    if inc:
        return np.interp(x, xf, yf)                          # core.py:1427
    return np.interp(x, xf[::-1], yf[::-1])                  # core.py:1430


def table_increasing(tbl):
    """Table.__init__ direction flag (core.py:1122-1128)."""
    return bool(tbl[-1] > tbl[0])


def lookup_coordinates(values, tbl, coord):
    """Table.lookup_coordinates (core.py:1136-1174); output dtype = table dtype."""
    tbl = np.asarray(tbl)
    return interp1d(np.asarray(values), tbl, np.asarray(coord),
                    table_increasing(tbl)).astype(tbl.dtype)


# ---------------------------------------------------------------------------
# a7  O(N) epilogue                                         core.py:463-488, 619-637, 945-966
# ---------------------------------------------------------------------------
def differentiate_contour(var):
    """`DataArray.differentiate('contour')`: np.gradient w.r.t. the contour
    coordinate 0..N-1 (float32 by default, core.py:248 / 1255), edge_order=1."""
    var = np.asarray(var)
    k = np.arange(var.shape[-1], dtype=np.float32)
    return np.gradient(var, k, axis=-1, edge_order=1)


def cal_gradient_wrt_area(var, area):
    with np.errstate(divide='ignore', invalid='ignore'):
        return differentiate_contour(var) / differentiate_contour(area)   # core.py:480-483


def cal_sqared_equivalent_length(dgrdSdA, dqdA):
    with np.errstate(divide='ignore', invalid='ignore'):
        return dgrdSdA / dqdA ** 2                           # core.py:635


def cal_normalized_Keff(Leq2, Lmin, mask=1e5):
    with np.errstate(divide='ignore', invalid='ignore'):
        nkeff = Leq2 / Lmin / Lmin                           # core.py:963
        return np.where(nkeff < mask, nkeff, np.nan)         # core.py:964


def equivalent_latitudes(areas, Rearth=Rearth):
    areas = np.asarray(areas)
    ratio = areas / 2.0 / np.pi / Rearth / Rearth - 1.0      # utils.py:506
    ratio = np.where(ratio < -1, -1.0, ratio)
    ratio = np.where(ratio > 1, 1.0, ratio)
    return np.rad2deg(np.arcsin(ratio)).astype(areas.dtype)  # utils.py:512


def latitude_lengths_at(lats, Rearth=Rearth):
    lats = np.asarray(lats)
    return (2.0 * np.pi * Rearth * np.cos(np.deg2rad(lats))).astype(lats.dtype)  # utils.py:532


# ---------------------------------------------------------------------------
# a8  interpolation to prescribed equivalent coordinates    core.py:1050-1100
# ---------------------------------------------------------------------------
def interp_to_coords(predef, eqCoords, var):
    """One slab: direction taken from eqCoords[0] < eqCoords[-1] (core.py:1085)."""
    eqCoords = np.asarray(eqCoords)
    inc = bool(eqCoords[0] < eqCoords[-1])
    return interp1d(np.asarray(predef), eqCoords, np.asarray(var), inc)


# ---------------------------------------------------------------------------
# f3  contours at prescribed equivalent coordinates         core.py:269-360
# ---------------------------------------------------------------------------
def cal_contours_at(q, predef, tbl, tbl_coord, dA, increase=True, lt=False,
                    dtype=np.float32, hist=True, right_edge='xhistogram'):
    """One slab.  `cal_contours_at_hist` (core.py:316-360, hist=True) and its
    conditional-integration twin `cal_contours_at` (core.py:269-313, hist=False):
    N = predef.size equally spaced levels (302/349), their enclosed areas
    (303/350), equivalent coordinates from the A(Yeq) table (304/351), then q(Y)
    by np.interp of the levels from those coordinates onto `predef` (306/352;
    direction from dimEq[0] < dimEq[-1], core.py:1080-1088).  np.vectorize'd
    np.interp returns float64 whatever `dtype` is.  Returns (qIntp, ctr)."""
    predef = np.asarray(predef)
    if len(predef.shape) != 1:
        raise Exception('predef should be a 1D array')            # core.py:294, 341
    N = predef.size
    ctr = cal_contours(q, N, increase, dtype)
    if hist:
        area = cal_integral_within_contours_hist(q, ctr, dA, None, lt, right_edge)
    else:
        area = cal_integral_within_contours(q, ctr, dA, None, lt)
    dimEq = lookup_coordinates(area, tbl, tbl_coord)
    return interp_to_coords(predef, dimEq, ctr), ctr


# ---------------------------------------------------------------------------
# a10  local wave activity / local APE                      core.py:696-799, 908-942
# ---------------------------------------------------------------------------
def cal_local_wave_activity(q, Q, coord, dA, increase=True, part='all',
                            mask_idx=None, metric=None):
    """q (ny,nx); Q (J=ny,); coord (ny,).  Returns lwa (J,nx) [, contours, masks].

    `metric=None` follows the snapshot text (core.py:789): M = dA.
    `metric=<array>` is the legacy length metric (commented core.py:787-788).
    """
    q = np.asarray(q)
    Q = np.asarray(Q)
    coord = np.asarray(coord)
    dA2 = np.asarray(dA)
    if dA2.ndim == 1:
        dA2 = dA2[:, None]
    wei = dA2 / np.nanmax(dA2)                               # core.py:723-724
    M = dA2 if metric is None else np.asarray(metric)
    if M.ndim == 1:
        M = M[:, None]
    part = part.lower()
    if part not in ['all', 'upper', 'lower']:                # core.py:732-733
        raise Exception("invalid part, should be in ['all', 'upper', 'lower']")
    coord_incre = not (coord[-1] < coord[0])                 # core.py:736-738
    J = len(coord)
    returnmask = mask_idx is not None
    if returnmask and max(mask_idx) >= J:                    # core.py:747-748
        raise Exception('indices in mask_idx out of boundary')
    mask_idx = list(mask_idx) if returnmask else []
    lwa = np.empty((J, q.shape[1]), dtype=np.float64)
    contours, masks = [], []
    for j in range(J):                                       # core.py:752
        qe = q - Q[j]                                        # core.py:754
        m = (coord >= coord[j]) if coord_incre else (coord <= coord[j])
        m = m[:, None]
        if increase:                                         # core.py:759-766
            mask1 = np.where(qe > 0, -1, 0)
            mask2 = np.where(m, 0, mask1)
            mask3 = np.where(np.logical_and(qe < 0, m), 1, mask2)
        else:
            mask1 = np.where(qe < 0, -1, 0)
            mask2 = np.where(m, 0, mask1)
            mask3 = np.where(np.logical_and(qe > 0, m), 1, mask2)
        if j in mask_idx:
            contours.append(Q[j])
            masks.append(mask3.astype(np.int64))
        if part == 'all':                                    # core.py:773-784
            mf = mask3.astype(np.float64)
        else:
            pos = (part == 'upper') == bool(increase)
            mf = np.where(mask3 > 0 if pos else mask3 < 0, mask3, np.nan)
        lwa[j] = -np.nansum(qe * mf * wei * M, axis=0)       # core.py:789
    if returnmask:
        return lwa, contours, masks
    return lwa


def cal_local_wave_activity2(q, Q, coord, dA, increase=True, part='all', mask_idx=None, metric=None):
    """core.py:802-905: qe = q[row j] - Q (dims (x, eq)), masks with the opposite sign
    convention, same part selection and weights.  Returns lwa (J, nx) [, contours, masks (ny, nx)]."""
    q = np.asarray(q)
    Q = np.asarray(Q)
    coord = np.asarray(coord)
    dA2 = np.asarray(dA)
    if dA2.ndim == 1:
        dA2 = dA2[:, None]
    wei = np.broadcast_to(dA2 / np.nanmax(dA2), q.shape)
    M = dA2 if metric is None else np.asarray(metric)
    if M.ndim == 1:
        M = M[:, None]
    M = np.broadcast_to(M, q.shape)
    part = part.lower()
    if part not in ['all', 'upper', 'lower']:
        raise Exception("invalid part, should be in ['all', 'upper', 'lower']")
    coord_incre = not (coord[-1] < coord[0])
    J = len(coord)
    returnmask = mask_idx is not None
    if returnmask and max(mask_idx) >= J:
        raise Exception('indices in mask_idx out of boundary')
    mask_idx = list(mask_idx) if returnmask else []
    lwa = np.empty((J, q.shape[1]), dtype=np.float64)
    contours, masks = [], []
    for j in range(J):
        qe = q[j][None, :] - Q[:, None]                      # (eq, x) layout of core.py:860
        m = (coord >= coord[j]) if coord_incre else (coord <= coord[j])
        m = m[:, None]
        if not increase:                                     # core.py:865-872
            mask1 = np.where(qe > 0, -1, 0)
            mask2 = np.where(m, 0, mask1)
            mask3 = np.where(np.logical_and(qe < 0, m), 1, mask2)
        else:
            mask1 = np.where(qe < 0, -1, 0)
            mask2 = np.where(m, 0, mask1)
            mask3 = np.where(np.logical_and(qe > 0, m), 1, mask2)
        if j in mask_idx:
            contours.append(Q[j])
            masks.append(mask3.astype(np.int64))
        if part == 'all':
            mf = mask3.astype(np.float64)
        else:
            pos = (part == 'upper') == bool(increase)
            mf = np.where(mask3 > 0 if pos else mask3 < 0, mask3, np.nan)
        lwa[j] = -np.nansum(qe * mf * wei * M, axis=0)       # core.py:895
    if returnmask:
        return lwa, contours, masks
    return lwa


# ---------------------------------------------------------------------------
# Build-defined pieces (no reference call site: SURVEY F6, F7)
# ---------------------------------------------------------------------------
def cell_area(lat, lon, Rearth=Rearth, to_poles=True):
    """2-D f64 cell areas R^2 |sin(phi_n) - sin(phi_s)| dlambda (the `rA`
    formula of utils.py:179-208) with mid-point cell edges.  BUILD-DEFINED
    input helper (the reference builds metrics with xgcm, off the hot path).
    `to_poles=True`: the two end cells reach the poles (sum == 4 pi R^2 on a
    global grid); False: end edges extrapolated half a spacing and clipped to
    +-90 like utils.py:186-190."""
    lat = np.asarray(lat, dtype=np.float64)
    lon = np.asarray(lon, dtype=np.float64)
    mid = 0.5 * (lat[1:] + lat[:-1])
    first = lat[0] - 0.5 * (lat[1] - lat[0])
    last = lat[-1] + 0.5 * (lat[-1] - lat[-2])
    if to_poles:
        first = np.sign(lat[0]) * 90.0
        last = np.sign(lat[-1]) * 90.0
    lo = np.clip(np.concatenate(([first], mid)), -90.0, 90.0)
    hi = np.clip(np.concatenate((mid, [last])), -90.0, 90.0)
    dlam = np.deg2rad(abs(lon[1] - lon[0]))
    band = Rearth * Rearth * np.abs(np.sin(np.deg2rad(hi)) - np.sin(np.deg2rad(lo))) * dlam
    return np.repeat(band[:, None], len(lon), axis=1)


def grad2_sphere(q, lat, lon, Rearth=Rearth):
    """|grad q|^2 on a regular lat-lon grid, float64.

    BUILD-DEFINED (the reference takes grdS as an input computed by the
    external GeoApps/xinvert packages): X periodic centred
    difference, Y centred interior / one-sided edges on the (possibly
    non-uniform) latitude index, metrics dx = R cos(phi) dlambda,
    dy = R dphi.  NaN neighbours propagate NaN (-> weight 0 via fillna).
    The arithmetic below is the normative order of operations for the kernel:
        gx = (q[j,i+1] - q[j,i-1]) * rdx[j]       rdx = 1 / (2 R cos(phi_j) dlambda)
        gy = (q[jn,i]  - q[js,i])  * rdy[j]       rdy = 1 / (R (phi_jn - phi_js))
        g2 = gx*gx + gy*gy
    """
    q = np.asarray(q, dtype=np.float64)
    rdx, rdy = grad_metrics(lat, lon, Rearth)
    ny = q.shape[0]
    gx = (np.roll(q, -1, axis=1) - np.roll(q, 1, axis=1)) * rdx[:, None]
    jn = np.minimum(np.arange(ny) + 1, ny - 1)
    js = np.maximum(np.arange(ny) - 1, 0)
    gy = (q[jn, :] - q[js, :]) * rdy[:, None]
    return gx * gx + gy * gy


def grad_metrics(lat, lon, Rearth=Rearth):
    """Per-row reciprocal metrics used by `grad2_sphere` (f64)."""
    lat = np.asarray(lat, dtype=np.float64)
    lon = np.asarray(lon, dtype=np.float64)
    ny = len(lat)
    phi = np.deg2rad(lat)
    dlam = np.deg2rad(lon[1] - lon[0])
    with np.errstate(divide='ignore'):
        rdx = 1.0 / (2.0 * Rearth * np.cos(phi) * dlam)
    jn = np.minimum(np.arange(ny) + 1, ny - 1)
    js = np.maximum(np.arange(ny) - 1, 0)
    rdy = 1.0 / (Rearth * (phi[jn] - phi[js]))
    return rdx, rdy


def sorted_profile(q, dA, tbl_targets, mask=None):
    """Exact adiabatic rearrangement (a9, build-defined; increasing / lt case).

    Stable ascending sort of the valid cells by q carrying dA;
    Acum = cumsum(dA_sorted); Q_exact(target) = q_sorted[searchsorted(Acum,
    target, 'right')] clipped to the last cell.  Returns (Q_exact, q_sorted,
    Acum).
    """
    q = np.asarray(q)
    dA2 = np.asarray(dA, dtype=np.float64)
    if dA2.ndim == 1:
        dA2 = dA2[:, None]
    w = np.broadcast_to(dA2, q.shape).ravel()
    x = q.ravel()
    ok = ~np.isnan(x)
    if mask is not None:
        ok &= (np.asarray(mask).ravel() == 1)
    x, w = x[ok], w[ok]
    order = np.argsort(x, kind='stable')
    xs, ws = x[order], w[order]
    acum = np.cumsum(ws)
    if len(xs) == 0:                             # nothing valid (an all-NaN / fully masked plane): no state to look up
        return np.full(np.shape(tbl_targets), np.nan), xs, acum
    idx = np.searchsorted(acum, np.asarray(tbl_targets, dtype=np.float64), side='right')
    idx = np.minimum(idx, len(xs) - 1)
    return xs[idx], xs, acum


def sorted_profile_brackets(acum, targets, rtol=1e-11):
    """Tie rule of `Q_exact` (part of its DEFINITION): the cumulative areas of a parallel scan differ
    from np.cumsum's by summation order (<= 1e-12 relative), and table values coincide with Acum values in exact
    arithmetic wherever the sorted order follows the rows.  A target within `rtol * Acum[-1]` of an Acum value may
    therefore resolve to either neighbouring cell.  Returns (lo, hi): every sorted index in [lo[j], hi[j]] is a
    valid answer for target j; lo == hi except at such ties."""
    acum = np.asarray(acum, dtype=np.float64)
    t = np.asarray(targets, dtype=np.float64)
    tol = rtol * acum[-1]
    n = len(acum)
    lo = np.minimum(np.searchsorted(acum, t - tol, side='right'), n - 1)
    hi = np.minimum(np.searchsorted(acum, t + tol, side='right'), n - 1)
    return lo, hi


def bpe_integral(q, dA, tbl, coord, mask=None):
    """Background-potential-energy-like integral of the exactly sorted state (a9,
    build-defined): sum_i q_i * z*(A_i - dA_i/2) * dA_i over the sorted cells, with
    z* = np.interp(A, tbl, coord) (reversed if the table decreases, like _interp1d)."""
    _, xs, acum = sorted_profile(q, dA, [0.0], mask)
    ws = np.diff(np.concatenate(([0.0], acum)))
    tbl = np.asarray(tbl, dtype=np.float64)
    coord = np.asarray(coord, dtype=np.float64)
    z = interp1d(acum - 0.5 * ws, tbl, coord, table_increasing(tbl))
    return float(np.sum(xs * z * ws))


# ---------------------------------------------------------------------------
# Box-counting contour crossing (core.py:640-693 driver, 1490-1566 kernel)
# ---------------------------------------------------------------------------
PAD_MODES = ('edge', 'wrap', 'constant', 'reflect', 'symmetric')


def pad_x(a, npad, mode='edge'):
    """DataArray.pad({X: (0, npad)}, mode=mode) on the last axis (core.py:674-676);
    xarray's 'constant' default fill for floats is NaN."""
    a = np.asarray(a)
    width = [(0, 0)] * (a.ndim - 1) + [(0, int(npad))]
    if mode == 'constant':
        return np.pad(a, width, mode='constant', constant_values=np.nan)
    return np.pad(a, width, mode=mode)


def crossing_shape(jo, io, stride):
    """Coarse shape of core.py:1510-1511: np.round (half to even) of the padded sizes."""
    return int(np.round(jo / stride)), int(np.round(io / stride))


def contour_crossing_literal(dataPad, contour, areaPad, stride=1):
    """The per-(slab, contour) kernel, loop for loop (core.py:1490-1566) -- small inputs only.

    Box (j, i) covers the fine cells [j*s, j*s+s) x [i*s, i*s+s); each cell contributes its four
    corners; `le` = some non-NaN corner <= contour, `gt` = some non-NaN corner > contour; a box
    with both counts sqrt(areaPad[j, i]) * stride -- areaPad is indexed with the COARSE indices
    (1560) -- and the column loop runs over range(Jn-1), not In-1 (1521): both kept.  Where
    the reference would index outside the padded array (Jn > In; numba does not bounds-check)
    the column range is cut at In-1."""
    dataPad = np.asarray(dataPad)
    areaPad = np.asarray(areaPad)
    Jn, In = crossing_shape(dataPad.shape[0], dataPad.shape[1], stride)
    re = np.zeros((Jn, In))
    ncross = 0
    for j in range(0, Jn - 1):
        for i in range(0, min(Jn, In) - 1):
            le = gt = False
            for jj in range(j * stride, j * stride + stride):
                for ii in range(i * stride, i * stride + stride):
                    for v in (dataPad[jj, ii], dataPad[jj, ii + 1], dataPad[jj + 1, ii], dataPad[jj + 1, ii + 1]):
                        if not np.isnan(v):
                            if v <= contour:
                                le = True
                            else:
                                gt = True
            if le and gt:
                ncross += 1
                with np.errstate(invalid='ignore'):
                    # numba types float32 * int64 as float64: f32 area -> sqrt rounded in f32, product in f64
                    re[j, i] = np.float64(np.sqrt(areaPad[j, i])) * stride
    return float(np.nansum(re)), ncross


def _box_minmax(dataPad, stride, nbj, nbi):
    """NaN-skipping min / max of the (s+1) x (s+1) corner block of every box."""
    mn = np.full((nbj, nbi), np.nan)
    mx = np.full((nbj, nbi), np.nan)
    for dr in range(stride + 1):
        for dc in range(stride + 1):
            blk = dataPad[dr:dr + (nbj - 1) * stride + 1:stride, dc:dc + (nbi - 1) * stride + 1:stride].astype(np.float64)
            mn = np.fmin(mn, blk)
            mx = np.fmax(mx, blk)
    return mn, mx


def contour_crossing(dataPad, contours, areaPad, stride=1, full_width=False):
    """All contours of one slab at once: a box is crossed by c iff min <= c < max over its
    non-NaN corners.  Returns (lengths f64 (N,), box counts int64 (N,)); lengths[k] is what the
    reference's `_contour_crossing(dataPad, contours[k], areaPad, stride)` returns.
    `full_width=True` scans all In-1 box columns (the evident intent) instead of Jn-1."""
    dataPad = np.asarray(dataPad)
    areaPad = np.asarray(areaPad)
    Jn, In = crossing_shape(dataPad.shape[0], dataPad.shape[1], stride)
    nbj, nbi = Jn - 1, (In - 1) if full_width else (min(Jn, In) - 1)
    contours = np.asarray(contours, dtype=np.float64).ravel()
    if nbj < 1 or nbi < 1:
        return np.zeros(len(contours)), np.zeros(len(contours), dtype=np.int64)
    mn, mx = _box_minmax(dataPad, stride, nbj, nbi)
    with np.errstate(invalid='ignore'):
        w = np.sqrt(areaPad[:nbj, :nbi]).astype(np.float64) * stride
    lengths = np.zeros(len(contours))
    counts = np.zeros(len(contours), dtype=np.int64)
    for k, c in enumerate(contours):
        cross = (mn <= c) & (mx > c)
        counts[k] = np.count_nonzero(cross)
        lengths[k] = np.nansum(np.where(cross, w, 0.0))
    return lengths, counts


def cal_contour_crossing(tracer, ctr, dA, stride=1, mode='edge', has_x=True, dtype=np.float32,
                         full_width=False):
    """Contour2D.cal_contour_crossing (core.py:640-693) for one slab: pad X by max(stride)
    (only when the grid has an 'X' dim, 673-679), then every stride on the SAME padded arrays;
    output cast to `dtype` (output_dtypes=[self.dtype], 689).  Returns a list when `stride` is
    iterable."""
    strides = list(stride) if np.iterable(stride) else [stride]
    npad = max(strides) if has_x else 0
    dataPad = pad_x(tracer, npad, mode) if has_x else np.asarray(tracer)
    areaPad = pad_x(np.broadcast_to(dA, np.shape(tracer)), npad, mode) if has_x else np.broadcast_to(dA, np.shape(tracer))
    res = [contour_crossing(dataPad, ctr, areaPad, s, full_width)[0].astype(dtype) for s in strides]
    return res if np.iterable(stride) else res[0]


# ---------------------------------------------------------------------------
# The reference's Keff call sequence (tests/test_Keff_atmos.py:75-92)
# ---------------------------------------------------------------------------
def keff_pipeline(q, dA, lat, N, grdS=None, lon=None, mask=None, increase=True,
                  lt=True, dtype=np.float32, preLats=None, right_edge='xhistogram',
                  nkeff_mask=1e5, deterministic=False):
    """One slab, hist API, steps 1-10 of SURVEY 3.1.  Returns a dict of
    ndarrays on the contour dim (+ '<name>_eq' on preLats if given)."""
    q = np.asarray(q)
    if mask is None:
        mask = np.ones(q.shape, dtype=q.dtype)
    supplied = grdS is not None
    if grdS is None:
        grdS = grad2_sphere(q, lat, lon)
    tbl, cs = cal_area_eqCoord_table_hist(mask, dA, lat, increase, lt, right_edge)
    ctr = cal_contours(q, N, increase, dtype)
    top0 = top1 = None
    if deterministic:
        # the accumulator windows of the deterministic mode come from bounds known BEFORE the pass (deterministic_bin_sums):
        # max |dA|; for the in-kernel squared gradient 2 ((max - min) max(rdx, rdy))^2 max dA; for a supplied one max |grdS| max dA
        dA64 = np.asarray(dA, dtype=np.float64)
        fin = np.abs(dA64[np.isfinite(dA64)])
        dmax = np.float64(fin.max()) if fin.size else np.float64(0.0)
        top0 = det_window_top(dmax)
        if supplied:
            g64 = np.asarray(grdS, dtype=np.float64)
            gmax = max(abs(np.float64(np.nanmin(g64))), abs(np.float64(np.nanmax(g64))))
            top1 = det_window_top(gmax * dmax)
        else:
            mn, mx = np.nanmin(q), np.nanmax(q)                      # in the tracer dtype, like cal_contours (core.py:224-225)
            rng = np.float64(abs(mx - mn))
            rdx, rdy = grad_metrics(lat, lon)
            rdm = np.float64(max(np.max(np.abs(rdx)), np.max(np.abs(rdy))))
            B = rng * rdm
            top1 = det_window_top(((B * B) * np.float64(2.0)) * dmax)
    area, counts = cal_integral_within_contours_hist(q, ctr, dA, None, lt, right_edge, return_counts=True,
                                                     deterministic=deterministic, det_top=top0)
    intgrdS = cal_integral_within_contours_hist(q, ctr, dA, grdS, lt, right_edge, deterministic=deterministic, det_top=top1)
    latEq = lookup_coordinates(area, tbl, cs)
    Lmin = latitude_lengths_at(latEq)
    dintSdA = cal_gradient_wrt_area(intgrdS, area)
    dqdA = cal_gradient_wrt_area(ctr, area)
    Leq2 = cal_sqared_equivalent_length(dintSdA, dqdA)
    nkeff = cal_normalized_Keff(Leq2, Lmin, nkeff_mask)
    out = dict(ctr=ctr, counts=counts, area=area, intgrdS=intgrdS, tbl=tbl,
               tbl_coord=cs, latEq=latEq, Lmin=Lmin, dintSdA=dintSdA,
               dqdA=dqdA, Leq2=Leq2, nkeff=nkeff)
    if preLats is not None:
        for name in ('ctr', 'area', 'intgrdS', 'latEq', 'dintSdA', 'dqdA',
                     'Leq2', 'Lmin', 'nkeff'):
            out[name + '_eq'] = interp_to_coords(preLats, latEq, out[name])
    return out
