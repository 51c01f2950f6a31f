"""-m gpu: K7 local wave activity (band walk, interval kernel and its premises, leading dims of Q) and K9 box-counting crossing.
(Regrouped in round 5 from the per-round files of rounds 2-4; nothing dropped.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import xcontour_oracle as O
from test_gpu_parity import rel, RTOL, TIGHT, LMIN_FLOOR, _baro_da
from gpu_common import GOLD, NINE, ROOT, bits, check_nine, check_nine_det, _clean_env

pytestmark = pytest.mark.gpu


def test_lwa_Q_leading_dims_in_any_order(ctx):
    """q is (time, level, lat, lon); Q given as (level, time, lat), as (level, lat) and as (lat,): slab s of q must meet
    ITS row of Q (the reference relies on xarray's by-name broadcasting, core.py:754)"""
    import xcontour_amd as xa
    rng = np.random.default_rng(8)
    nt, nl, ny, nx = 2, 3, 21, 30
    lat = np.linspace(-70, 70, ny); lon = np.arange(nx) * 12.0
    q = rng.standard_normal((nt, nl, ny, nx)) + np.linspace(-2, 2, ny)[None, None, :, None]
    Q = np.sort(rng.standard_normal((nt, nl, ny)), axis=-1) + np.arange(nl)[None, :, None] * 0.3 + np.arange(nt)[:, None, None] * 0.7
    dAv = O.cell_area(lat, lon)
    c = {'time': np.arange(float(nt)), 'level': np.arange(float(nl)), 'lat': lat, 'lon': lon}
    tr = xa.DataArray(q, ('time', 'level', 'lat', 'lon'), c, 'pv')
    cm = xa.Contour2D(tr, xa.DataArray(dAv, ('lat', 'lon'), {'lat': lat, 'lon': lon}, 'dA'),
                      dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True)
    Qa = xa.DataArray(np.transpose(Q, (1, 0, 2)).copy(), ('level', 'time', 'lat'), c, 'pv')      # swapped leading dims
    Qb = xa.DataArray(Q, ('time', 'level', 'lat'), c, 'pv')
    la, lb_ = cm.cal_local_wave_activity(tr, Qa), cm.cal_local_wave_activity(tr, Qb)
    assert la.dims == tr.dims and np.array_equal(la.values, lb_.values)
    for t in range(nt):
        for l in range(nl):
            assert np.array_equal(la.values[t, l], O.cal_local_wave_activity(q[t, l], Q[t, l], lat, dAv, True, 'all'))
    Qc = xa.DataArray(Q[0], ('level', 'lat'), {'level': c['level'], 'lat': lat}, 'pv')          # a subset of the dims: broadcast over time
    lc = cm.cal_local_wave_activity(tr, Qc)
    for t in range(nt):
        for l in range(nl):
            assert np.array_equal(lc.values[t, l], O.cal_local_wave_activity(q[t, l], Q[0, l], lat, dAv, True, 'all'))
    Qd = xa.DataArray(Q[0, 0], ('lat',), {'lat': lat}, 'pv')
    ld = cm.cal_local_wave_activity(tr, Qd, mask_idx=[3, 9])
    assert np.array_equal(ld[0].values[1, 2], O.cal_local_wave_activity(q[1, 2], Q[0, 0], lat, dAv, True, 'all'))
    with pytest.raises(Exception, match='does not have'):
        cm.cal_local_wave_activity(tr, xa.DataArray(Q[:, 0], ('member', 'lat'), {'lat': lat}, 'pv'))


def _lwa_case(rng, ny, nx, dt, increase, coord_up, with_nan):
    lat = np.linspace(-80, 80, ny) if coord_up else np.linspace(80, -80, ny)
    prof = np.sin(np.deg2rad(np.linspace(-80, 80, ny)))
    if not increase:
        prof = -prof
    q = (prof[:, None] + 0.3 * np.sin(np.linspace(0, 12, nx))[None, :] * np.cos(np.deg2rad(lat))[:, None]
         + 0.05 * rng.standard_normal((ny, nx))).astype(dt)
    Q = np.sort(q.astype(np.float64).mean(axis=1))
    if not increase:
        Q = Q[::-1].copy()
    # ties: some cells exactly ON reference levels
    idx = rng.integers(0, ny * nx, 500)
    q.ravel()[idx] = Q[rng.integers(0, ny, 500)].astype(dt)
    if with_nan:
        q[rng.integers(0, ny, 40), rng.integers(0, nx, 40)] = np.nan
        q[5:9, 10:30] = np.nan
    dA = np.abs(np.cos(np.deg2rad(lat)))[:, None] * np.ones((1, nx)) * 1e9 + 1e7 * rng.random((ny, nx))
    return lat, q, Q, dA


@pytest.mark.parametrize('dt,increase,coord_up,part,mkind', [
    (np.float64, True, True, 'all', 'row'), (np.float32, True, False, 'upper', 'plane'), (np.float64, False, True, 'lower', None),
    (np.float32, False, False, 'all', 'row'), (np.float64, True, True, 'upper', None), (np.float64, False, False, 'upper', 'plane')])
def test_lwa_interval_kernel_matches_the_oracle(ctx, dt, increase, coord_up, part, mkind):
    """planes of more than 512 rows: one binary search in the (monotone) reference state per cell + difference arrays + prefix
    sums instead of the band walk -- against the oracle's literal python loop (core.py:752-791) for both directions of the
    tracer and of the coordinate, every `part`, the three metric forms, float32 / float64 tracers, NaN cells, ties with
    levels: <= 1e-11 of the plane's largest value (summation order; the bit-exact band walk is `exact=True`)"""
    rng = np.random.default_rng(int(increase) * 4 + int(coord_up) * 2 + (mkind is not None))
    ny, nx = 600, 334                                       # 334: a ragged last column group
    lat, q, Q, dA = _lwa_case(rng, ny, nx, dt, increase, coord_up, True)
    M = None if mkind is None else (np.abs(np.gradient(np.deg2rad(lat))) * 6.371e6 if mkind == 'row' else 1.0 + rng.random((ny, nx)))
    pcode = {'all': 0, 'upper': 1, 'lower': 2}[part]
    got, _ = ctx.lwa(q[None], Q[None], lat, dA, float(dA.max()), M=M, increase=increase, part=pcode)
    assert ctx.last_lwa_path() == 1
    ref = O.cal_local_wave_activity(q, Q, lat, dA, increase, part, metric=M)
    scale = np.abs(ref).max()
    assert scale > 0 and np.abs(got[0] - ref).max() <= 1e-11 * scale
    ex, _ = ctx.lwa(q[None], Q[None], lat, dA, float(dA.max()), M=M, increase=increase, part=pcode, exact=True)
    assert ctx.last_lwa_path() == 0 and np.array_equal(ex[0], ref)              # the band walk: numpy's own summation order


def test_lwa_interval_kernel_premises_and_stacks(ctx):
    """a reference state that is NOT monotone (or holds a NaN), or a coordinate with a repeated value, sends the call to the
    bit-exact band walk (path 2); a stack of slabs with per-slab Q; masks for mask_idx stay exact on the fast path; planes of
    up to 512 rows never take it"""
    rng = np.random.default_rng(9)
    ny, nx = 520, 128
    lat, q, Q, dA = _lwa_case(rng, ny, nx, np.float64, True, True, False)
    Qbad = Q.copy(); Qbad[100], Qbad[101] = Q[101], Q[100]
    for Qx, c in ((Qbad, lat), (np.where(np.arange(ny) == 7, np.nan, Q), lat), (Q, np.where(np.arange(ny) == 300, lat[299], lat))):
        got, _ = ctx.lwa(q[None], Qx[None], c, dA, float(dA.max()))
        assert ctx.last_lwa_path() == 2
        assert np.array_equal(got[0], O.cal_local_wave_activity(q, Qx, c, dA, True, 'all'), equal_nan=True)
    S = 3
    qs = np.stack([q * (1 + 0.1 * s) for s in range(S)])
    Qs = np.stack([Q * (1 + 0.1 * s) for s in range(S)])
    got, masks = ctx.lwa(qs, Qs, lat, dA, float(dA.max()), mask_idx=[3, 400])
    assert ctx.last_lwa_path() == 1
    for s in range(S):
        ref, _, mref = O.cal_local_wave_activity(qs[s], Qs[s], lat, dA, True, 'all', mask_idx=[3, 400])
        assert np.abs(got[s] - ref).max() <= 1e-11 * np.abs(ref).max()
        assert np.array_equal(masks[s, 0], mref[0]) and np.array_equal(masks[s, 1], mref[1])
    small = _lwa_case(rng, 256, 128, np.float64, True, True, False)
    sref = O.cal_local_wave_activity(small[1], small[2], small[0], small[3], True, 'all')
    g2, _ = ctx.lwa(small[1][None], small[2][None], small[0], small[3], float(small[3].max()))
    assert ctx.last_lwa_path() == 0 and np.array_equal(g2[0], sref)
    # exact=False: the premises are checked on the host and vouched for -- ONE launch of the interval kernel, any plane size
    g3, _ = ctx.lwa(small[1][None], small[2][None], small[0], small[3], float(small[3].max()), exact=False)
    assert ctx.last_lwa_path() == 1 and np.abs(g3[0] - sref).max() <= 1e-11 * np.abs(sref).max() and not np.array_equal(g3[0], sref)
    g4, _ = ctx.lwa(q[None], Qbad[None], lat, dA, float(dA.max()), exact=False)            # not monotone: the host check sends it to the band walk
    assert ctx.last_lwa_path() == 0 and np.array_equal(g4[0], O.cal_local_wave_activity(q, Qbad, lat, dA, True, 'all'))


def test_lwa_interval_kernel_refuses_infinities(ctx):
    """round-4 advisor: an INFINITE reference level passed the premise check (only NaN failed it) and the interval kernel then
    computed inf * 0 = NaN where the reference sums to 0; an infinite tracer CELL turned every row behind it into inf - inf.  Now the
    device check wants Q finite, and a cell the interval kernel finds infinite stamps the flag: the gated band walk behind it runs
    and the call returns the reference's own answer (path 2).  exact=False checks the same on the host."""
    rng = np.random.default_rng(21)
    ny, nx = 600, 96
    lat, q, Q, dA = _lwa_case(rng, ny, nx, np.float64, True, True, False)
    Qinf = Q.copy(); Qinf[-1] = np.inf                                     # still monotone, no NaN
    with np.errstate(invalid='ignore'):
        ref = O.cal_local_wave_activity(q, Qinf, lat, dA, True, 'all')
    got, _ = ctx.lwa(q[None], Qinf[None], lat, dA, float(dA.max()))
    assert ctx.last_lwa_path() == 2 and np.array_equal(got[0], ref, equal_nan=True)
    qinf = q.copy(); qinf[300, 17] = -np.inf; qinf[10, 60] = np.inf         # two infinite cells, far from each other
    with np.errstate(invalid='ignore'):
        ref = O.cal_local_wave_activity(qinf, Q, lat, dA, True, 'all')
    got, _ = ctx.lwa(qinf[None], Q[None], lat, dA, float(dA.max()))
    assert ctx.last_lwa_path() == 2 and np.array_equal(got[0], ref, equal_nan=True)
    assert np.isfinite(got[0][:, :17]).all()                               # columns without an infinite cell are untouched by it
    g2, _ = ctx.lwa(qinf[None], Q[None], lat, dA, float(dA.max()), exact=False)       # vouching needs a look at the tracer too
    assert ctx.last_lwa_path() == 0 and np.array_equal(g2[0], ref, equal_nan=True)
    g3, _ = ctx.lwa(q[None], Qinf[None], lat, dA, float(dA.max()), exact=False)
    assert ctx.last_lwa_path() == 0


def test_deterministic_facade_keeps_lwa_on_the_band_walk(ctx):
    """round-4 advisor: Contour2D(deterministic=True) promises run-to-run identical bits; planes of more than 512 rows used to take the
    interval kernel (LDS atomics in arrival order) all the same.  Now `exact` defaults to True there: the oracle's own bits."""
    import xcontour_amd as xa
    rng = np.random.default_rng(5)
    ny, nx = 560, 64
    lat, q, Q, dA = _lwa_case(rng, ny, nx, np.float64, True, True, False)
    lon = np.arange(nx) * 1.0
    c = {'lat': lat, 'lon': lon}
    tr = xa.DataArray(q, ('lat', 'lon'), c, 'pv')
    cm = xa.Contour2D(tr, xa.DataArray(dA, ('lat', 'lon'), c, 'dA'), dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True,
                      deterministic=True)
    lwa = cm.cal_local_wave_activity(tr, xa.DataArray(Q, ('lat',), {'lat': lat}, 'Q'))
    assert cm.ctx.last_lwa_path() == 0
    assert np.array_equal(lwa.values, O.cal_local_wave_activity(q, Q, lat, dA, True, 'all'))
    cm.close()


def test_crossing_uncrossed_interior_levels_are_exact_zeros(ctx):
    """two regions of the plane separated by NaN columns hold values in [0, 0.3] and [0.7, 1]; no box has corners in
    both, so the levels in between are crossed by nothing and must come out as the exact 0 of the reference's per-contour
    loop (the difference-array accumulation used to leave ~1e-16-of-the-mass residues there), and nothing is negative"""
    rng = np.random.default_rng(8)
    ny, nx = 96, 260
    q = np.empty((2, ny, nx))
    q[:, :, :128] = 0.3 * rng.random((2, ny, 128))
    q[:, :, 128:132] = np.nan
    q[:, :, 132:] = 0.7 + 0.3 * rng.random((2, ny, nx - 132))
    area = 1e9 * (1 + rng.random((ny, nx)))
    ctr = np.linspace(0.0, 1.0, 101)
    lens, cnts = ctx.crossing(q, ctr, area, stride=1, full_width=True)
    mid = (ctr > 0.305) & (ctr < 0.695)
    assert (cnts[:, mid] == 0).all() and (cnts[:, (ctr > 0.05) & (ctr < 0.25)] > 0).all() and (cnts[:, (ctr > 0.75) & (ctr < 0.95)] > 0).all()
    assert (lens[:, mid] == 0.0).all()
    assert (lens >= 0).all()
    for s in range(2):
        ol, oc = O.contour_crossing(q[s], ctr, area, 1, True)
        assert np.array_equal(cnts[s].astype(np.int64), oc) and rel(lens[s], ol) < 1e-13


@pytest.mark.parametrize('ny,nx,dt,da_row', [(2100, 23, np.float64, False), (4500, 9, np.float32, True), (530, 3, np.float64, False)])
def test_lwa_interval_kernel_tall_planes_and_odd_reference_states(ctx, ny, nx, dt, da_row):
    """k_lwa_fast (round 5: persistent over column groups, a bucket table in front of the bracket search): planes so tall that a
    workgroup holds TWO columns (2100 rows) or ONE (4500) instead of four, a plane narrower than one group (3 columns), a
    row-rank weight, a stack of two slabs (one reference state each) -- and reference states that stress the table: levels crowded
    into the last bucket, one constant level (no bucket width: the full binary search), a range of 1e150."""
    rng = np.random.default_rng(ny)
    lat, q, Q, dA = _lwa_case(rng, ny, nx, dt, True, True, True)
    dAu = dA[:, 0].copy() if da_row else dA
    Qc = Q.copy(); Qc[ny // 2:] = Q[ny // 2] + (Q[-1] - Q[ny // 2]) * (1 - 1e-9 * (ny - 1 - np.arange(ny // 2, ny)))     # half of the levels within 1e-6 of the maximum
    cases = [Q, np.sort(Qc), np.full(ny, float(np.nanmean(q))), Q * 1e150 / np.abs(Q).max()]
    for k, Qx in enumerate(cases):
        qx = q if k < 3 else (q.astype(np.float64) * 1e150 / np.abs(Q).max()).astype(np.float64)
        qs = np.stack([qx, qx[:, ::-1] * 0.5]); Qs = np.stack([Qx, Qx * 0.5])
        got, _ = ctx.lwa(qs, Qs, lat, dAu, float(dAu.max()), exact=False)
        assert ctx.last_lwa_path() == 1, k
        for s in range(2):
            ref = O.cal_local_wave_activity(qs[s], Qs[s], lat, dA if not da_row else dAu[:, None] * np.ones((1, nx)), True, 'all')
            scale = np.abs(ref).max()
            assert np.isfinite(scale) and np.abs(got[s] - ref).max() <= 1e-11 * max(scale, 1e-300), (k, s)


def _lwa_rows(q, Q, coord, dA, rows, increase=True):
    """core.py:752-791 for the target rows `rows` only (the oracle's own loop body, one j at a time: a full 1801-row plane has 1801
    of them at ~0.2 s each)"""
    q64 = q.astype(np.float64)
    wei = dA / np.nanmax(dA)
    cinc = not (coord[-1] < coord[0])
    out = np.empty((len(rows), q.shape[1]))
    for i, j in enumerate(rows):
        qe = q64 - Q[j]                                                               # core.py:754
        m = ((coord >= coord[j]) if cinc else (coord <= coord[j]))[:, None]
        if increase:                                                                   # core.py:759-766
            mask3 = np.where(np.logical_and(qe < 0, m), 1, np.where(m, 0, np.where(qe > 0, -1, 0)))
        else:
            mask3 = np.where(np.logical_and(qe > 0, m), 1, np.where(m, 0, np.where(qe < 0, -1, 0)))
        out[i] = -np.nansum(qe * mask3.astype(np.float64) * wei * dA, axis=0)         # core.py:789 (M = dA)
    return out


def test_lwa_interval_kernel_at_the_full_cfg2_size(ctx):
    """K7F on ONE 1801 x 3600 float64 plane with J = 1801 target rows -- the size its 0.08 ms are quoted on (the
    tall-plane tests use nx = 3-23) -- against the reference's loop body (core.py:752-791, restated row by row) on 16 sampled target
    rows x all 3600 columns, 1e-12 of the plane's largest value; NaN cells and cells exactly on reference levels included."""
    ny, nx = 1801, 3600
    rng = np.random.default_rng(2026)
    lat, q, Q, dA = _lwa_case(rng, ny, nx, np.float64, True, True, True)
    got, _ = ctx.lwa(q[None], Q[None], lat, dA, float(dA.max()), exact=False)
    assert ctx.last_lwa_path() == 1
    rows = sorted(set([0, 1, 5, ny // 7, ny // 3, ny // 2 - 1, ny // 2, ny // 2 + 1, 1000, 1234, 1500, ny - 300, ny - 17, ny - 3, ny - 2, ny - 1]))
    assert len(rows) == 16
    ref = _lwa_rows(q, Q, lat, dA, rows)
    scale = np.abs(ref).max()
    assert np.isfinite(scale) and scale > 0
    err = np.abs(got[0][rows] - ref).max() / scale
    assert err <= 1e-12, err
    # and every row is finite and non-negative up to rounding where the plane is (LWA >= 0 for a sorted reference state)
    assert np.isfinite(got[0]).all() and got[0].min() > -1e-9 * scale
