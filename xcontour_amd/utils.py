# -*- coding: utf-8 -*-
"""
Host-side helpers of xcontour_amd: O(ny) / O(N) numpy algebra on coordinate and
contour vectors, mirroring /root/reference/xcontour/utils.py.  Nothing here
touches a full (ny, nx) slab -- that work is done by the HIP kernels.
"""
import numpy as np

# Radius of the Earth (m)                                   reference utils.py:19
Rearth = 6371200.0


def equivalent_latitudes(areas, Rearth=Rearth):
    """Equivalent latitude from contour-enclosed area on a full sphere:
    2*pi*a^2*[sin(latEq) + 1] = area            (reference utils.py:491-515)."""
    vals, wrap = _unwrap(areas)
    ratio = vals / 2.0 / np.pi / Rearth / Rearth - 1.0
    ratio = np.where(ratio < -1, -1.0, ratio)
    ratio = np.where(ratio > 1, 1.0, ratio)
    return wrap(np.rad2deg(np.arcsin(ratio)).astype(vals.dtype))


def latitude_lengths_at(lats, Rearth=Rearth):
    """Minimum possible contour length 2*pi*a*cos(lat)   (reference utils.py:518-534)."""
    vals, wrap = _unwrap(lats)
    return wrap((2.0 * np.pi * Rearth * np.cos(np.deg2rad(vals))).astype(vals.dtype))


def _unwrap(x):
    """(ndarray, rewrap) for ndarray / labelled-array inputs."""
    if hasattr(x, 'values') and hasattr(x, 'dims'):
        return np.asarray(x.values), (lambda v: x.copy(data=v))
    return np.asarray(x), (lambda v: v)


def cell_area(lat, lon, Rearth=Rearth, to_poles=True):
    """2-D float64 cell areas R^2 |sin(phi_n) - sin(phi_s)| dlambda with mid-point cell
    edges (the `rA` formula of reference utils.py:179-208; the reference builds its
    metrics with xgcm, which is off the hot path and not reproduced here).
    `to_poles`: the two end cells reach the poles so that the sum is 4 pi R^2."""
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


def grad_metrics(lat, lon, Rearth=Rearth):
    """Per-row reciprocal metrics of the in-kernel |grad q|^2 stencil on a regular
    lat-lon grid: rdx[j] = 1/(2 R cos(phi_j) dlambda), rdy[j] = 1/(R (phi_jn - phi_js))
    with jn/js = j+-1 clamped to the grid (one-sided at the first/last row)."""
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


def cartesian_metrics(y, dx):
    """The same for a Cartesian plane (e.g. X-Z sections): uniform `dx`, coordinate `y`."""
    y = np.asarray(y, dtype=np.float64)
    ny = len(y)
    jn = np.minimum(np.arange(ny) + 1, ny - 1)
    js = np.maximum(np.arange(ny) - 1, 0)
    return np.full(ny, 1.0 / (2.0 * dx)), 1.0 / (y[jn] - y[js])


def last_row_included(coord, right_edge='xhistogram'):
    """Does the row on the LAST (largest) coordinate value enter the A(Yeq) table?

    The reference histograms the coordinate field against its own values (core.py:176-193), so
    the cells of the last row sit exactly on the last bin edge.  np.histogram closes that edge
    (`right_edge='numpy'`: always included).  xhistogram instead replaces the last edge by
    `edge + 1e-8` evaluated in the edge dtype and keeps the bin half-open (`_histogram`,
    core.py:1307, calls xhistogram): the row is kept only if the bump is representable, i.e.
    dropped for every float32 coordinate of magnitude >= 0.25 (float32 latitudes!), kept for
    float64 coordinates up to ~1e8."""
    if right_edge == 'numpy':
        return True
    if right_edge != 'xhistogram':
        raise Exception('right_edge should be "numpy" or "xhistogram"')
    c = np.asarray(coord)
    hi = c.max()
    return bool((np.array([hi]) + 1e-8)[0] > hi)


def table_from_rowsums(rows_asc, ylt, include_last=True):
    """A(Yeq) table from per-row sums taken in ASCENDING-coordinate order.

    The reference histograms the coordinate field against its own values
    (core.py:176-193); every cell sits on an edge, so the histogram is
    pdf = [0, r_0, ..., r_{J-3}, r_{J-2} (+ r_{J-1} if `include_last`)] (dummy bin first; the last
    row sits on the last edge: see `last_row_included`), then `cumsum` and, if not `ylt`,
    `cdf[-1] - cdf` (core.py:1320-1323)."""
    r = np.asarray(rows_asc, dtype=np.float64)
    J = len(r)
    pdf = np.zeros(J, dtype=np.float64)
    if J >= 2:
        pdf[1:J] = r[0:J - 1]
        if include_last:
            pdf[J - 1] = r[J - 2] + r[J - 1]
    elif include_last:
        pdf[0] = r[0]
    cdf = np.cumsum(pdf)
    if not ylt:
        cdf = cdf[-1] - cdf
    return cdf
