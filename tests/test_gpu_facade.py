"""-m gpu: the reference-shaped facade (Contour2D / Table) -- contours_at twins, ocean sequence with a land mask, table order, batched / lazy / resident inputs,
the xarray branch through a test double, INTEGRATION.md's stub run as printed, no device-memory leak over object cycles.
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


@pytest.mark.parametrize('rule', ['xhistogram', 'numpy'])
@pytest.mark.parametrize('increase', [True, False])
@pytest.mark.parametrize('lt', [True, False])
def test_contours_at_both_twins(ctx, baro, rule, increase, lt):
    """core.py:269-360: contours at prescribed equivalent latitudes, histogram and conditional-integration twins,
    against the committed golden vectors (== the oracle, tests/test_oracle_golden.py) and the oracle run live"""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    g = np.load(os.path.join(GOLD, 'baro_contours_at.npz'))
    pre = g['predef']
    mask = xa.DataArray(np.ones_like(q), tr.dims, tr.coords, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'},
                      increase=increase, lt=lt, right_edge=rule)
    table = cm.cal_area_eqCoord_table_hist(mask)
    o_tbl, o_cs = O.cal_area_eqCoord_table_hist(np.ones_like(q), dA.values, lat, increase, lt, rule)
    for hist, fn in ((True, cm.cal_contours_at_hist), (False, cm.cal_contours_at)):
        got = fn(pre, table)
        want = g['%s_inc%d_lt%d_%s' % (rule, increase, lt, 'hist' if hist else 'cond')]
        live, _ = O.cal_contours_at(q, pre, o_tbl, o_cs, dA.values, increase, lt, np.float32, hist, rule)
        assert np.array_equal(want, live)
        assert got.dims == ('contour',) and got.name == 'absolute_vorticity' and got.shape == pre.shape
        assert got.coords['contour'].dtype == np.float32 and got.coords['contour'][-1] == len(pre) - 1   # core.py:311, 358
        assert rel(got.values, want) < 1e-9
        # labelled predef (the reference accepts a DataArray with its own dim name, core.py:297-299)
        got2 = fn(xa.DataArray(pre, ('latitude',), {'latitude': pre}), table)
        assert rel(got2.values, got.values) < 1e-12 and got2.dims == ('contour',)      # LDS atomics: sums vary in the last bits run to run
    with pytest.raises(Exception, match='predef should be a 1D array'):
        cm.cal_contours_at_hist(np.zeros((3, 3)), table)


def test_contours_at_leading_dims(ctx, baro):
    """a (time, lat, lon) stack: one q(Y) profile per time, each equal to the single-slab oracle"""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    st = np.stack([q, q * 1.5 + 1e-5, q[:, ::-1].copy()])
    c3 = dict(tr.coords); c3['time'] = np.arange(3.0)
    cm = xa.Contour2D(xa.DataArray(st, ('time',) + tr.dims, c3, 'absolute_vorticity'), dA,
                      dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    table = cm.cal_area_eqCoord_table_hist(xa.DataArray(np.ones_like(q), tr.dims, tr.coords, 'mask'))
    pre = np.linspace(-80, 80, 33)
    got = cm.cal_contours_at_hist(pre, table)
    assert got.dims == ('time', 'contour') and got.shape == (3, 33)
    tbl, cs = O.cal_area_eqCoord_table_hist(np.ones_like(q), dA.values, lat, True, True)
    for k in range(3):
        want, _ = O.cal_contours_at(st[k], pre, tbl, cs, dA.values, True, True, np.float32, True)
        assert rel(got.values[k], want) < 1e-9


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_keff_ocean_call_sequence_with_land_mask(ctx, dt):
    """tests/test_Keff_ocean.py computeKeff: an ocean domain -- tracer NaN over land (`where(tracer != 0)`), `maskC` (0 / 1) for
    the A(Yeq) table, 401 contours, a user-supplied Lmin (zonal sum of mask * dx interpolated to Yeq), nkeff mask 2e7,
    interp_to_dataset -- step by step against the oracle, and the fused pipeline on the same input"""
    import xcontour_amd as xa
    rng = np.random.default_rng(31)
    ny, nx, N = 146, 360, 401
    lat = np.linspace(-70, 75, ny); lon = np.arange(nx) * 1.0
    land = np.zeros((ny, nx), bool)
    land[40:90, 60:130] = True; land[95:140, 200:300] = True; land[:6, :] = True      # two continents and a polar cap
    land |= rng.random((ny, nx)) < 0.01                                               # islands
    maskC = (~land).astype(np.float64)
    q = (np.tanh(np.deg2rad(lat) * 2)[:, None] * 10 + 15 + 1.5 * np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat))[:, None]
         + 0.3 * rng.standard_normal((ny, nx))).astype(dt)
    q[land] = np.nan
    c = {'latitude': lat, 'longitude': lon}
    dAv = O.cell_area(lat, lon)
    g2 = O.grad2_sphere(np.where(land, np.nan, q), lat, lon)                           # NaN next to the coasts, like a masked fd.grad
    tr = xa.DataArray(q, ('latitude', 'longitude'), c, 'PTRACER04')
    dA = xa.DataArray(dAv, ('latitude', 'longitude'), c, 'rA')
    grdS = xa.DataArray(g2, tr.dims, tr.coords, 'grdS')
    mask = xa.DataArray(maskC, tr.dims, tr.coords, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True, check_mono=False)
    table = cm.cal_area_eqCoord_table_hist(mask)
    ctr = cm.cal_contours(N)
    area = cm.cal_integral_within_contours_hist(ctr).rename('intArea')
    intgrdS = cm.cal_integral_within_contours_hist(ctr, integrand=grdS).rename('intgrdS')
    Yeq = table.lookup_coordinates(area).rename('Yeq')
    dx = O.Rearth * np.cos(np.deg2rad(lat)) * np.deg2rad(1.0)
    preLmin = (maskC * dx[:, None]).sum(1)                                             # (mask * dxF).sum('longitude')
    Lmin = xa.DataArray(np.interp(Yeq.values, lat, preLmin), Yeq.dims, Yeq.coords, 'Lmin')   # .interp(latitude=Yeq)
    dgrdSdA = cm.cal_gradient_wrt_area(intgrdS, area)
    dqdA = cm.cal_gradient_wrt_area(ctr, area)
    Leq2 = cm.cal_sqared_equivalent_length(dgrdSdA, dqdA)
    nkeff = cm.cal_normalized_Keff(Leq2, Lmin, mask=2e7)
    # the oracle, same sequence
    o_tbl, o_cs = O.cal_area_eqCoord_table_hist(maskC, dAv, lat, True, True)
    o_ctr = O.cal_contours(q, N, True, np.float32)
    o_area, o_cnt = O.cal_integral_within_contours_hist(q, o_ctr, dAv, None, True, return_counts=True)
    o_S = O.cal_integral_within_contours_hist(q, o_ctr, dAv, g2, True)
    o_Yeq = O.lookup_coordinates(o_area, o_tbl, o_cs)
    o_Lmin = np.interp(o_Yeq, lat, preLmin)
    o_dS = O.cal_gradient_wrt_area(o_S, o_area); o_dq = O.cal_gradient_wrt_area(o_ctr, o_area)
    o_Leq2 = O.cal_sqared_equivalent_length(o_dS, o_dq)
    o_nk = O.cal_normalized_Keff(o_Leq2, o_Lmin, 2e7)
    assert int((~land).sum()) - o_cnt.sum() in (0, 1)        # every ocean cell, no land cell (float32 contours: the max cell may sit above ctr[-1])
    assert abs(o_tbl[-1] / (maskC * dAv).sum() - 1) < 1e-12                            # the table ends at the ocean area
    assert rel(table._table.values, o_tbl) < 1e-13
    assert np.array_equal(ctr.values, o_ctr)
    assert rel(area.values, o_area) < TIGHT and rel(intgrdS.values, o_S) < TIGHT
    assert rel(Yeq.values, o_Yeq) < 1e-9
    assert rel(dqdA.values, o_dq) < 1e-8 and rel(dgrdSdA.values, o_dS) < 1e-8
    assert rel(Leq2.values, o_Leq2) < RTOL and rel(nkeff.values, o_nk) < RTOL
    preY = np.linspace(-70, 75, N)
    interp = cm.interp_to_dataset(preY, Yeq, [ctr, area, Yeq, intgrdS, dgrdSdA, dqdA, Leq2, Lmin, nkeff]).rename({'new': 'latitude'})
    assert rel(interp['nkeff'].values, O.interp_to_coords(preY, o_Yeq, o_nk)) < RTOL
    assert rel(interp['intArea'].values, O.interp_to_coords(preY, o_Yeq, o_area)) < 1e-9
    # the fused pipeline on the same field: levels, counts, area, intgrdS (in-kernel gradient == the supplied grdS), Yeq
    ds = cm.keff(N, table, lat=lat, lon=lon, periodic_x=True)
    assert np.array_equal(ds['ctr'].values, o_ctr.astype(np.float64))
    assert rel(ds['area'].values, o_area) < TIGHT and rel(ds['intgrdS'].values, o_S) < TIGHT
    assert rel(ds['latEq'].values, o_Yeq) < 1e-9


def test_keff_table_length_and_order(ctx, baro):
    """xc_keff_dev reads ny table entries in ascending-coordinate order: a table of another length must be refused
    and a table kept in DESCENDING coordinate order (cal_area_eqCoord_table keeps the input order) must be flipped"""
    import xcontour_amd as xa
    from xcontour_amd.pipeline import KeffPlan
    tr, dA, q, lat, lon = _baro_da(xa, baro, flip=True)                # latitude runs north -> south
    mask = xa.DataArray(np.ones_like(q), tr.dims, tr.coords, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=False, lt=False)
    t_desc = cm.cal_area_eqCoord_table(mask)                           # keeps the descending coordinate
    assert t_desc._coord[0] > t_desc._coord[-1]
    ds = cm.keff(121, t_desc, lat=lat, lon=lon)
    ctr = cm.cal_contours(121)
    area = cm.cal_integral_within_contours_hist(ctr)
    want = t_desc.lookup_coordinates(area)                             # host np.interp on the same table
    assert rel(ds['area'].values, area.values) < 1e-13
    assert rel(ds['latEq'].values, want.values) < 1e-9
    o_tbl, _ = O.cal_area_eqCoord_table(np.ones_like(q), dA.values, lat, False, False)
    assert rel(t_desc._table.values, o_tbl) < 1e-13
    short = xa.Table(xa.DataArray(t_desc._table.values[:-1], ('latitude',), {'latitude': lat[:-1]}, 'AeqCTbl'), 'latitude')
    with pytest.raises(Exception, match='table has'):
        cm.keff(121, short, lat=lat, lon=lon)
    with pytest.raises(Exception, match='length ny'):
        KeffPlan(ctx, 1, 256, 512, 11, np.float32, np.float32, dA=None, lat=lat, lon=lon, tbl=np.arange(255.), tbl_coord=lat[:255])
    with pytest.raises(Exception, match='monotonic'):
        KeffPlan(ctx, 1, 4, 8, 11, np.float32, np.float32, dA=None, rdx=np.ones(4), rdy=np.ones(4), tbl=np.arange(4.),
                 tbl_coord=np.array([0., 2., 1., 3.]))


def test_facade_methods_stream_large_stacks_in_batches(ctx, baro):
    """a stack larger than a deliberately small staging cap goes through the device in batches of whole slabs -- histogram
    integrals, crossing, LWA, sorted profile, the fused keff() with its double-buffered uploads -- with the results of one
    big launch (counts / levels / exact kernels bit for bit, float sums to rounding)"""
    import xcontour_amd as xa
    q0, lat, lon = baro
    S = 7
    rng = np.random.default_rng(4)
    q = np.stack([q0 * (1 + 0.1 * s) + 1e-6 * rng.standard_normal(q0.shape).astype(np.float32) for s in range(S)])
    c = {'time': np.arange(S), 'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(q, ('time', 'latitude', 'longitude'), c, 'absolute_vorticity')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'rA')
    mask = xa.DataArray(np.ones_like(q0), ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    table = cm.cal_area_eqCoord_table_hist(mask)
    ctr = cm.cal_contours(41)
    dy = np.gradient(np.deg2rad(lat.astype(np.float64))) * O.Rearth
    Qeq = xa.DataArray(np.sort(q.mean(axis=2), axis=1), ('time', 'latitude'), {'time': np.arange(S), 'latitude': lat}, 'absolute_vorticity')

    def everything(cap_keff):
        return dict(area=cm.cal_integral_within_contours_hist(ctr).values,
                    cross=cm.cal_contour_crossing(ctr, stride=[1, 2]),
                    lwa=cm.cal_local_wave_activity(tr, Qeq, metric=dy).values,
                    prof=cm.cal_sorted_profile(table).values,
                    keff=cm.keff(41, table, lat=lat, lon=lon, max_batch_bytes=cap_keff))

    cap = cm.ctx.max_batch_bytes
    big = everything(8 << 30)
    try:
        cm.ctx.max_batch_bytes = 3 * q0.nbytes + 1000                 # room for one or two slabs' worth of staged bytes
        assert len(cm.ctx._batches(S, q0.nbytes)) >= 3
        small = everything(5 * q0.nbytes)                              # keff: two device halves of two slabs each, four batches
    finally:
        cm.ctx.max_batch_bytes = cap
    assert rel(small['area'], big['area']) < 1e-12
    for a, b in zip(small['cross'], big['cross']):
        assert rel(a.values, b.values) < 1e-12
    assert np.array_equal(small['lwa'], big['lwa'])                    # sequential sums: bit-identical
    assert np.array_equal(small['prof'], big['prof'])
    for k in ('ctr', 'area', 'intgrdS', 'latEq', 'nkeff'):
        a, b = small['keff'][k].values, big['keff'][k].values
        assert a.shape == (S, 41)
        assert np.array_equal(a, b) if k == 'ctr' else rel(a, b) < 1e-9, k
    cm.close()


def test_keff_double_buffered_batches_with_supplied_grdS_and_per_slab_dA(ctx, baro):
    """the multi-batch path of Contour2D.keff (two device halves, uploads on the copy stream) with everything that travels per
    slab -- the tracer, a supplied squared gradient, time-varying weights -- against the single-batch result and the oracle"""
    import xcontour_amd as xa
    q0, lat, lon = baro
    S = 5
    rng = np.random.default_rng(14)
    q = np.stack([q0 * (1 + 0.05 * s) for s in range(S)])
    g = (rng.random(q.shape) * 1e-16).astype(np.float32)
    dA0 = O.cell_area(lat, lon)
    dA3 = np.stack([dA0 * (1 + 0.01 * s) for s in range(S)])
    c3 = {'time': np.arange(S), 'latitude': lat, 'longitude': lon}
    c2 = {'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(q, ('time', 'latitude', 'longitude'), c3, 'absolute_vorticity')
    gs = xa.DataArray(g, ('time', 'latitude', 'longitude'), c3, 'grdS')
    mask = xa.DataArray(np.ones_like(q0), ('latitude', 'longitude'), c2, 'mask')
    for dA in (xa.DataArray(dA0, ('latitude', 'longitude'), c2, 'rA'), xa.DataArray(dA3, ('time', 'latitude', 'longitude'), c3, 'rA')):
        cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
        table = xa.Contour2D(tr, xa.DataArray(dA0, ('latitude', 'longitude'), c2, 'rA'), dims={'X': 'longitude', 'Y': 'latitude'},
                             dimEq={'Y': 'latitude'}, increase=True, lt=True).cal_area_eqCoord_table_hist(mask)
        one = cm.keff(31, table, grdS=gs)
        per = q0.nbytes + g[0].nbytes + (dA0.nbytes if dA.values.ndim == 3 else 0)
        many = cm.keff(31, table, grdS=gs, max_batch_bytes=2 * per * 2 + 100)          # two slabs per half: batches 2 + 2 + 1
        for k in ('ctr', 'area', 'intgrdS', 'latEq', 'nkeff'):
            a, b = many[k].values, one[k].values
            assert np.array_equal(a, b) if k == 'ctr' else rel(a, b) < 1e-9, k
        for s in (0, S - 1):
            w = dA.values if dA.values.ndim == 2 else dA.values[s]
            r = O.keff_pipeline(q[s], w, lat, 31, grdS=g[s], increase=True, lt=True, dtype=np.float32)
            assert np.array_equal(many['ctr'].values[s], r['ctr'].astype(np.float64))
            assert rel(many['area'].values[s], r['area']) < TIGHT and rel(many['intgrdS'].values[s], r['intgrdS']) < 1e-6
        cm.close()


def test_resident_inputs_give_the_same_results(ctx, baro):
    """Contour2D(resident=True): tracer and weights are uploaded once and the host-form calls copy from the device mirror --
    same bits as without, for the reference's Keff call sequence on a stack, also when the stack goes through in batches of
    whole slabs (slices of the registered array) and after touch() following an in-place change"""
    import xcontour_amd as xa
    q0, lat, lon = baro
    S = 4
    q = np.stack([q0 * (1 + 0.1 * s) for s in range(S)])
    c3 = {'time': np.arange(S), 'latitude': lat, 'longitude': lon}
    c2 = {'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(q, ('time', 'latitude', 'longitude'), c3, 'absolute_vorticity')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), c2, 'rA')
    g = xa.DataArray(np.random.default_rng(2).random(q.shape).astype(np.float32), ('time', 'latitude', 'longitude'), c3, 'grdS')
    mask = xa.DataArray(np.ones_like(q0), ('latitude', 'longitude'), c2, 'mask')
    kw = dict(dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True, deterministic=True)

    def sequence(cm):
        table = cm.cal_area_eqCoord_table_hist(mask)
        ctr = cm.cal_contours(61)
        area = cm.cal_integral_within_contours_hist(ctr)
        intS = cm.cal_integral_within_contours_hist(ctr, integrand=g)
        ds = cm.keff(61, table, lat=lat, lon=lon)                    # the fused pipeline uploads through xc_memcpy_h2d_async: mirror-aware too
        return [table.lookup_coordinates(area).values, ctr.values, area.values, intS.values, ds['area'].values, ds['nkeff'].values]

    plain = xa.Contour2D(tr, dA, **kw)
    ref = sequence(plain)
    res = xa.Contour2D(tr, dA, resident=True, **kw)
    n0 = len(res.ctx._resident)
    got = sequence(res)
    assert len(res.ctx._resident) == n0 + 4                         # the tracer stack, the float64 weights, the (time-invariant) mask and (round 5) the last integrand, once each
    for a, b in zip(got, ref):
        assert np.array_equal(bits(a), bits(b))
    old = res.ctx.max_batch_bytes
    try:
        res.ctx.max_batch_bytes = 2 * q0.nbytes + 100                # two slabs per batch: slices of the registered stack
        for a, b in zip(sequence(res), ref):
            assert np.array_equal(bits(a), bits(b))
    finally:
        res.ctx.max_batch_bytes = old
    q[1] = np.roll(q[1], 9, axis=0)                                 # in place (rows meet other weights): the mirror is stale until touch()
    res.touch()
    again = sequence(res)
    fresh = sequence(xa.Contour2D(tr, dA, **kw))
    for a, b in zip(again, fresh):
        assert np.array_equal(bits(a), bits(b))
    assert not np.array_equal(bits(again[2]), bits(ref[2]))
    res.close(); plain.close()
    assert len(res.ctx._resident) == n0


def test_resident_keff_reuses_its_plan_by_identity_and_forgets_on_touch(ctx, baro):
    """round 6: a resident object that gets the SAME argument objects again does not rebuild the key of its plan from their bytes
    (core.Contour2D.keff, `_keff_last`).  Same results call after call; another table object with other contents -> its own plan; the
    contents of the table changed IN PLACE + touch() -> the new results; more configurations than plans are kept (2) and back to the
    first: the evicted plan is rebuilt, metrics included"""
    import xcontour_amd as xa
    q0, lat, lon = baro
    q = np.stack([q0 * (1 + 0.1 * s) for s in range(2)])
    c3 = {'time': np.arange(2), 'latitude': lat, 'longitude': lon}; c2 = {'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(q, ('time', 'latitude', 'longitude'), c3, 'absolute_vorticity')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), c2, 'rA')
    mask = xa.DataArray(np.ones_like(q0), ('latitude', 'longitude'), c2, 'mask')
    kw = dict(dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True, deterministic=True)
    res = xa.Contour2D(tr, dA, resident=True, **kw)
    plain = xa.Contour2D(tr, dA, **kw)
    names = ('ctr', 'area', 'intgrdS', 'latEq', 'nkeff')

    def vec(ds):
        return [ds[n].values.copy() for n in names]

    def same(a, b):
        return all(np.array_equal(bits(x), bits(y)) for x, y in zip(a, b))
    table = res.cal_area_eqCoord_table_hist(mask)
    r1 = vec(res.keff(41, table, lat=lat, lon=lon))
    assert res.__dict__.get('_keff_last') is not None
    r2 = vec(res.keff(41, table, lat=lat, lon=lon))                  # the identity path
    ref = vec(plain.keff(41, table, lat=lat, lon=lon))
    assert same(r1, ref) and same(r2, ref)
    tv = table._table.values
    t2 = xa.Table(xa.DataArray(tv * 2.0, ('latitude',), {'latitude': lat}, 'AeqCTbl'), 'latitude')       # another object, other contents
    r3 = vec(res.keff(41, t2, lat=lat, lon=lon))
    assert same(r3, vec(plain.keff(41, t2, lat=lat, lon=lon))) and not same(r3, ref)
    assert same(vec(res.keff(41, table, lat=lat, lon=lon)), ref)
    tv *= 0.5                                                        # in place, same objects: touch() is the contract
    res.touch()
    r4 = vec(res.keff(41, table, lat=lat, lon=lon))
    assert same(r4, vec(plain.keff(41, table, lat=lat, lon=lon))) and not same(r4, ref)
    for N in (21, 31, 51):                                           # three more configurations: the N = 41 plan is evicted ...
        assert same(vec(res.keff(N, table, lat=lat, lon=lon)), vec(plain.keff(N, table, lat=lat, lon=lon)))
    res.keff(41, table, lat=lat, lon=lon)
    for N in (21, 31):
        res.keff(N, table, lat=lat, lon=lon)
    last = res.__dict__['_keff_last']
    assert last[1] in res.__dict__['_keff_plans']
    res.__dict__['_keff_plans'].pop(last[1]).free()                  # ... and the plan of the LAST call too: same objects, no plan
    assert same(vec(res.keff(31, table, lat=lat, lon=lon)), vec(plain.keff(31, table, lat=lat, lon=lon)))
    res.lt = False; plain.lt = False                                 # an attribute of the object changed between two calls with the same arguments
    r5 = vec(res.keff(31, table, lat=lat, lon=lon))
    assert same(r5, vec(plain.keff(31, table, lat=lat, lon=lon)))
    res.close(); plain.close()


def test_integration_md_stub_runs_as_printed(baro):
    """the two python blocks INTEGRATION.md shows a maintainer of the reference (ctypes binding of xc_hist and xc_crossing) are
    executed verbatim against the built library: histogram_cdf == the oracle's _histogram restatement (counts-exact levels,
    sums to rounding), contour_crossing == the oracle's box counting"""
    import re
    from xcontour_amd import _native as nat
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    stub = next(b for b in blocks if 'class _HistDesc' in b)
    cross = next(b for b in blocks if 'def contour_crossing' in b)
    ns = {}
    exec(stub.replace("C.CDLL('libxcontour_hip.so')", 'C.CDLL(%r)' % nat.LIB_PATH), ns)
    exec(cross, ns)
    q, lat, lon = baro
    dA = O.cell_area(lat, lon)
    for inc in (True, False):
        ctr = O.cal_contours(q, 61, inc, np.float32)
        for lt in (True, False):
            got = ns['histogram_cdf'](q[None], ctr, dA, lt)
            want = O.histogram_cdf(q, ctr, dA, lt)                 # ascending-value order like _histogram's return
            want = want[0] if isinstance(want, tuple) else want
            assert rel(got[0], np.asarray(want, dtype=np.float64)) < TIGHT, (inc, lt)
    levels = np.sort(np.linspace(float(np.nanmin(q)), float(np.nanmax(q)), 9)[1:-1]).astype(np.float64)
    got = ns['contour_crossing'](q[None].astype(np.float64), levels[None], dA, 2, 2, 'edge')      # stride 2, padded by max_stride = 2 columns
    want, _ = O.contour_crossing(O.pad_x(q.astype(np.float64), 2, 'edge'), levels, O.pad_x(dA, 2, 'edge'), 2)
    assert rel(got[0], np.asarray(want)) < 1e-12


def test_facade_cycles_do_not_leak_device_memory(ctx):
    """create / use / close Contour2D objects over and over (resident or not, deterministic or not, every operator): the
    free device memory settles -- plans, resident mirrors and work buffers are returned"""
    import ctypes as C
    import xcontour_amd as xa
    hip = C.CDLL('libamdhip64.so')

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    ny, nx, S = 91, 180, 4
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 2.0
    rng = np.random.default_rng(0)
    c3 = {'t': np.arange(S), 'lat': lat, 'lon': lon}; c2 = {'lat': lat, 'lon': lon}
    dA = xa.DataArray(O.cell_area(lat, lon), ('lat', 'lon'), c2, 'dA')
    mask = xa.DataArray(np.ones((ny, nx)), ('lat', 'lon'), c2, 'mask')
    base = None
    for it in range(45):
        q = np.sin(np.deg2rad(lat))[None, :, None] + 0.05 * rng.standard_normal((S, ny, nx))
        tr = xa.DataArray(q, ('t', 'lat', 'lon'), c3, 'pv')
        cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True,
                          resident=(it % 2 == 0), deterministic=(it % 3 == 0))
        table = cm.cal_area_eqCoord_table_hist(mask)
        ctr = cm.cal_contours(31)
        cm.cal_integral_within_contours_hist(ctr)
        ds = cm.keff(31, table, preY=lat, lat=lat, lon=lon)
        if it % 5 == 0:
            cm.cal_local_wave_activity(tr, ds['ctr_eq'].rename({'new': 'lat'}))
            cm.cal_sorted_profile(table)
            cm.cal_contour_crossing(ctr, stride=[1, 2])
        cm.close()
        del cm
        if it == 14:
            base = free_bytes()
    assert base - free_bytes() < (1 << 20), 'device memory keeps shrinking: %d bytes since iteration 14' % (base - free_bytes())


def test_resident_memo_follows_reassignment_and_is_private(ctx, baro):
    """Contour2D(resident=True): `c.tracer = other` / `c.dA = other` must not be served the OLD mirror (round-3 advisor), and
    what is registered is a private copy: a second, non-resident object that hands the SAME ndarray to the library after an
    in-place change gets the new values, not the first object's mirror"""
    import xcontour_amd as xa
    q0, lat, lon = baro
    c2 = {'latitude': lat, 'longitude': lon}
    q = np.ascontiguousarray(q0.astype(np.float64))
    tr = xa.DataArray(q, ('latitude', 'longitude'), c2, 'absolute_vorticity')
    dAv = O.cell_area(lat, lon)
    dA = xa.DataArray(dAv, ('latitude', 'longitude'), c2, 'rA')
    kw = dict(dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True, deterministic=True)

    def area(cm):
        return cm.cal_integral_within_contours_hist(cm.cal_contours(41)).values

    res = xa.Contour2D(tr, dA, resident=True, **kw)
    a0 = area(res)
    # 1. reassign the tracer and the weights
    q2 = np.ascontiguousarray(np.roll(q, 17, axis=0) * 1.5)
    res.tracer = xa.DataArray(q2, ('latitude', 'longitude'), c2, 'absolute_vorticity')
    a1 = area(res)
    ref1 = area(xa.Contour2D(res.tracer, dA, **kw))
    assert np.array_equal(bits(a1), bits(ref1)) and not np.array_equal(bits(a1), bits(a0))
    res.dA = xa.DataArray(dAv * 2.0, ('latitude', 'longitude'), c2, 'rA')
    a2 = area(res)
    assert np.array_equal(bits(a2), bits(area(xa.Contour2D(res.tracer, res.dA, **kw))))
    assert np.array_equal(bits(a2), bits(2.0 * a1))
    # 2. the registered host memory is not the caller's array
    assert all(not np.shares_memory(arr, q2) for arr in res.ctx._resident.values())
    q2[:] = np.roll(q2, 5, axis=0)                                   # in place, no touch(): res keeps its (documented) old mirror ...
    other = xa.Contour2D(res.tracer, res.dA, **kw)                  # ... but a non-resident object must see the new values
    b = area(other)
    res.touch()
    assert np.array_equal(bits(b), bits(area(res))) and not np.array_equal(bits(b), bits(a2))
    res.close(); other.close()


def test_facade_call_sequences_through_the_xarray_branch():
    """the Keff (tests/test_hist.py), contour-mean and LWA (tests/test_LWA.py) call sequences of the reference with xarray
    objects in: every result is an xarray object with the reference's dims and names ('AeqCTbl', 'd...dA', 'Leq2', 'nkeff',
    'LWA', 'LAPE', 'cm...'), values bit-identical to the same calls on the in-house DataArray, the fused pipeline against
    the oracle (subprocess: the double must be importable BEFORE the package is)"""
    env = dict(_clean_env(), PYTHONPATH=os.path.join(ROOT, 'tests', 'fake_xarray'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'xarray_branch_script.py'), 'gpu'], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok gpu'), r.stderr[-3000:]


class _Budget(object):
    """a lazy (time, level, lat, lon) source that refuses to hand out more than `limit` bytes at once"""

    def __init__(self, a, limit):
        self.a, self.shape, self.dtype, self.limit, self.peak, self.reads = a, a.shape, a.dtype, limit, 0, 0

    def __getitem__(self, k):
        r = self.a[k]
        self.peak = max(self.peak, r.nbytes)
        self.reads += 1
        if r.nbytes > self.limit:
            raise MemoryError('asked for %d bytes at once, budget %d' % (r.nbytes, self.limit))
        return r


def test_lazy_stack_goes_through_in_batches(baro, tmp_path):
    """the reference's histogram API is lazy (dask='allowed', core.py:242, 258): a lazy tracer stack -- here a source that
    RAISES when more than two slabs are requested at once, and a multi-record .nc opened with lazy=True -- is pulled through
    cal_contours / the histogram integrals / contour means / keff / LWA / crossing batch by batch under max_batch_bytes; results equal the eager run bit for bit"""
    import xcontour_amd as xa
    from xcontour_amd import ncio
    q0, lat, lon = baro
    T, Z = 3, 2
    rng = np.random.default_rng(8)
    q = np.stack([q0 * (1 + 0.05 * k) for k in range(T * Z)]).reshape(T, Z, *q0.shape).astype(np.float32)
    g = rng.random(q.shape).astype(np.float32)
    c4 = {'time': np.arange(T), 'level': np.arange(Z), 'latitude': lat, 'longitude': lon}
    c2 = {'latitude': lat, 'longitude': lon}
    d4 = ('time', 'level', 'latitude', 'longitude')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), c2, 'rA')
    mask = xa.DataArray(np.ones_like(q0), ('latitude', 'longitude'), c2, 'mask')
    kw = dict(dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True, deterministic=True)
    L = np.load(os.path.join(ROOT, 'tests', 'golden', 'baro_lwa_N121.npz'))

    def run(tr, grd):
        cm = xa.Contour2D(tr, dA, **kw)
        table = cm.cal_area_eqCoord_table_hist(mask)
        ctr = cm.cal_contours(41)
        area = cm.cal_integral_within_contours_hist(ctr)
        intS = cm.cal_integral_within_contours_hist(ctr, integrand=grd)
        strict = cm.cal_integral_within_contours(ctr)
        mean = cm.cal_contour_mean_hist(ctr, grd, grd)
        ds = cm.keff(41, table, lat=lat, lon=lon, max_batch_bytes=2 * q0.nbytes + 64)
        Q = xa.DataArray(L['Q'], ('latitude',), {'latitude': lat}, 'Q')
        lwa = cm.cal_local_wave_activity(tr, Q, metric=L['dy'])
        cross = cm.cal_contour_crossing(ctr, stride=2)
        cm.close()
        return [ctr.values, area.values, intS.values, strict.values, mean.values, ds['nkeff'].values, ds['area'].values, lwa.values, cross.values]

    eager = run(xa.DataArray(q, d4, c4, 'pv'), xa.DataArray(g, d4, c4, 'grdS'))
    from xcontour_amd import _native as nat
    ctx = nat.default_context(0)                                     # the facade's own context
    old = ctx.max_batch_bytes
    try:
        ctx.max_batch_bytes = 2 * q0.nbytes + 64                     # at most two slabs per batch, one when an integrand rides along
        src, gsrc = _Budget(q, 2 * q0.nbytes), _Budget(g, 2 * q0.nbytes)
        lazy = run(xa.DataArray(src, d4, c4, 'pv'), xa.DataArray(gsrc, d4, c4, 'grdS'))
        assert 0 < src.peak <= 2 * q0.nbytes and src.reads >= 3 * 6 and gsrc.peak <= 2 * q0.nbytes
        for a, b in zip(lazy, eager):
            assert np.array_equal(bits(a), bits(b))
        # the same from a file: a classic NetCDF stack opened lazily
        from scipy.io import netcdf_file
        path = str(tmp_path / 'stack.nc')
        with netcdf_file(path, 'w', version=2) as f:
            f.createDimension('time', None); f.createDimension('level', Z); f.createDimension('latitude', len(lat)); f.createDimension('longitude', len(lon))
            for n_, v_ in (('latitude', lat), ('longitude', lon)):
                w = f.createVariable(n_, 'f4', (n_,)); w[:] = v_
            w = f.createVariable('pv', 'f4', d4); w[:] = q
            w = f.createVariable('grdS', 'f4', d4); w[:] = g
        ds = ncio.open_dataset(path, lazy=True)
        assert isinstance(ds.pv.data, ncio.LazyVariable) and ds.pv.dims == d4
        filed = run(ds.pv, ds.grdS)
        for a, b in zip(filed, eager):
            assert np.array_equal(bits(a), bits(b))
        assert ds.pv.data.rows_read >= T and isinstance(ds.pv.data, ncio.LazyVariable)      # still lazy afterwards
    finally:
        ctx.max_batch_bytes = old
