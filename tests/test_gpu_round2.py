"""-m gpu, round 2: the gaps the round-1 review named -- cal_contours_at(_hist), the cfg4 workload shape
(BASELINE configs[3]) through chained launch sets, per-slab (time-varying) dA through the fused pipeline and the
façade, stale-chain protection, table order / length checks, leading-dim order of Q in the LWA family, contexts on
every visible device."""
import os

import numpy as np
import pytest

import xcontour_oracle as O
from test_gpu_parity import rel, RTOL, TIGHT, LMIN_FLOOR, _baro_da

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NINE = ('ctr', 'area', 'intgrdS', 'latEq', 'dqdA', 'dintSdA', 'Leq2', 'Lmin', 'nkeff')


def check_nine(out, s, r, with_eq=False):
    """all nine result vectors of slab `s` against the oracle's dict `r`"""
    assert np.array_equal(out['counts'][s].astype(np.int64), r['counts'])
    assert np.array_equal(out['ctr'][s], r['ctr'].astype(np.float64))
    assert rel(out['area'][s], r['area']) < TIGHT and rel(out['intgrdS'][s], r['intgrdS']) < TIGHT
    for k in ('latEq', 'dqdA', 'dintSdA', 'Leq2'):
        assert rel(out[k][s], r[k]) < RTOL, k
    assert rel(out['Lmin'][s], r['Lmin'], LMIN_FLOOR) < RTOL
    ok = r['Lmin'] > LMIN_FLOOR
    assert rel(out['nkeff'][s][ok], r['nkeff'][ok]) < RTOL
    if with_eq:
        for k in ('ctr', 'area', 'intgrdS', 'latEq'):
            assert rel(out[k + '_eq'][s], r[k + '_eq']) < RTOL, k


# ---------------------------------------------------------------- f3: cal_contours_at / cal_contours_at_hist
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


def test_levels_against_the_reference_notebook_printout(ctx):
    """a2 against a REFERENCE-HELD known answer: the contours the reference itself printed in
    notebooks/1.Keff_atmos.ipynb cell 3 (PV.nc, 15 x 241 x 480 float32, N = 121; tests/golden/nb1_ctr_printout.json).
    PV.nc is not bundled: a stand-in stack with exactly the printed minima and the (few-ulp) maxima that the printout
    admits goes through K1 + the level kernel on the device -- all 36 printed values must come back bit for bit."""
    import xcontour_amd as xa
    from test_oracle_golden import nb1_rows, nb1_max_candidates
    N, rows = nb1_rows()
    lvl = lambda mn, mx: O.cal_contours(np.array([[mn, mx]], np.float32), N, True, np.float32)
    rng = np.random.default_rng(11)
    lat = np.linspace(-90, 90, 241).astype(np.float32); lon = (np.arange(480) * 0.75).astype(np.float32)
    keys = sorted(rows)
    st = np.empty((len(keys), 241, 480), np.float32)
    for i, k in enumerate(keys):
        first, last = rows[k]
        mx = nb1_max_candidates(first, last, N, lvl)[0]
        pl = rng.uniform(first[0], mx, (241, 480)).astype(np.float32)
        pl = np.clip(pl, first[0], mx)
        pl[rng.integers(241), rng.integers(480)] = first[0]
        pl[5, 7] = mx
        pl[100, 3] = np.nan                                    # xarray's min / max skip NaN (core.py:224-225)
        st[i] = pl
    c = {'level': np.arange(len(keys)), 'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(st, ('level', 'latitude', 'longitude'), c, 'pv')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'rA')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    ctr = cm.cal_contours(N)
    assert ctr.dims == ('level', 'contour') and ctr.values.dtype == np.float32
    for i, k in enumerate(keys):
        first, last = rows[k]
        assert np.array_equal(ctr.values[i, :3], first) and np.array_equal(ctr.values[i, -3:], last), k


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


@pytest.mark.parametrize('increase', [True, False])
@pytest.mark.parametrize('lt', [True, False])
@pytest.mark.parametrize('cd', [np.float32, np.float64])
def test_keff_epilogue_alone(ctx, baro, increase, lt, cd):
    """xc_keff_epilogue (K5 / K6 without the cell-touching passes): PDFs from numpy's own histogram of the barotropic field go
    in, the nine Keff vectors and their interpolation to the latitudes come out -- against the oracle's step-by-step sequence"""
    q, lat, lon = baro
    dA = O.cell_area(lat, lon)
    N = 121
    r = O.keff_pipeline(q, dA, lat, N, lon=lon, increase=increase, lt=lt, dtype=cd, preLats=lat.astype(np.float64))
    ctr = r['ctr']
    g2 = O.grad2_sphere(q, lat, lon)
    e, _ = O.hist_edges(ctr)                                         # ascending-value edges (core.py:1296-1305)
    w1 = np.where(np.isnan(g2 * dA), 0.0, g2 * dA)
    pdf = np.stack([O.weighted_histogram(q, e, dA)[0], O.weighted_histogram(q, e, w1)[0]])[None]      # np.digitize + np.bincount
    out = ctx.keff_epilogue(pdf, ctr[None].astype(np.float64), r['tbl'], r['tbl_coord'], increase=increase, lt=lt,
                            ctr_dtype=cd, preY=lat.astype(np.float64))
    assert rel(out['area'][0], r['area']) < TIGHT and rel(out['intgrdS'][0], r['intgrdS']) < TIGHT
    for k in ('latEq', 'dqdA', 'dintSdA', 'Leq2'):
        assert rel(out[k][0], r[k]) < RTOL, k
    assert rel(out['Lmin'][0], r['Lmin'], LMIN_FLOOR) < RTOL
    okm = r['Lmin'] > LMIN_FLOOR
    assert rel(out['nkeff'][0][okm], r['nkeff'][okm]) < RTOL
    names = ('ctr', 'area', 'intgrdS', 'latEq', 'dintSdA', 'dqdA', 'Leq2', 'Lmin', 'nkeff')
    for v in (0, 1, 3):
        assert rel(out['interp'][0, v], r[names[v] + '_eq']) < RTOL, names[v]
    with pytest.raises(Exception):
        ctx.keff_epilogue(pdf, ctr[None].astype(np.float64), r['tbl'][:1], r['tbl_coord'][:1])      # a table needs >= 2 entries


# ---------------------------------------------------------------- BASELINE configs[3]: 1440x721 f64 slabs, per-slab levels, chained
def test_cfg4_shape_chained_launch_sets(ctx):
    """74 slabs of 721x1440 f64 generated on device (seed + slab id), N = 201, processed in two chained launch sets of
    37 (the NEXT set's min/max rides in this set's histogram pass), twice over so that every set also runs on chained
    partials: counts against the oracle on ALL slabs, all nine vectors on 6 of them, bit-identical to unchained"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    ny, nx, N, S, chunk = 721, 1440, 201, 74, 37
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.25
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
    preY = np.linspace(-90, 90, 91)
    plan = KeffPlan(ctx, S, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, preY=preY, nslots=3)
    plan.synth(lat, lon, 4242, 0)
    plan.run(0)                                   # unchained, one launch set
    plan.run(1, chunk, chain=True)                # set 0: K1, set 1: partials from set 0's pass
    plan.run(2, chunk, chain=True)                # both sets on chained partials (set 1's pass carried set 0's)
    ref, a, b = plan.fetch(slot=0), plan.fetch(slot=1), plan.fetch(slot=2)
    for o in (a, b):
        assert np.array_equal(o['ctr'], ref['ctr']) and np.array_equal(o['counts'], ref['counts'])
        assert rel(o['area'], ref['area']) < 1e-13 and rel(o['intgrdS'], ref['intgrdS']) < 1e-13
    q = plan.download_q()
    assert len({q[s].tobytes()[:64] for s in range(S)}) == S                  # distinct slabs
    for s in range(S):
        ctr = O.cal_contours(q[s], N, True, np.float64)
        assert np.array_equal(b['ctr'][s], ctr)                               # per-slab levels
        _, cnt = O.cal_integral_within_contours_hist(q[s], ctr, dA, None, True, return_counts=True)
        assert np.array_equal(b['counts'][s].astype(np.int64), cnt), s
    for s in (0, 1, 36, 37, 55, 73):
        r = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float64, preLats=preY)
        check_nine(b, s, r, with_eq=True)
    plan.free()


@pytest.mark.parametrize('nx', [1440, 1040])
def test_float32_tracers_four_cells_per_lane(ctx, nx):
    """float32 tracers with nx % 4 == 0 and nx >= 1024 take the four-cells-per-lane histogram variant (256-column strips; 1040
    columns leave a ragged last strip of 16) in the Keff FAST layout: chained and unchained launch sets, NaNs, both contour
    dtypes -- counts against the oracle on every slab, all nine vectors on three"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    rng = np.random.default_rng(nx)
    ny, N, S = 181, 121, 10
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * (360.0 / nx)
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
    q = (np.sin(np.deg2rad(lat))[None, :, None] * (1 + 0.2 * rng.random((S, 1, 1))) + 0.05 * rng.standard_normal((S, ny, nx))).astype(np.float32)
    q[2, 40:44, 100:300] = np.nan; q[7, :, 5] = np.nan
    for cd in (np.float32, np.float64):
        plan = KeffPlan(ctx, S, ny, nx, N, np.float32, cd, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True, nslots=2)
        plan.set_q(q)
        plan.run(0)                                   # unchained
        plan.run(1, 5, chain=True)                    # two chained launch sets of 5
        ref, b = plan.fetch(slot=0), plan.fetch(slot=1)
        assert np.array_equal(b['ctr'], ref['ctr']) and np.array_equal(b['counts'], ref['counts'])
        assert rel(b['area'], ref['area']) < 1e-13 and rel(b['intgrdS'], ref['intgrdS']) < 1e-13
        for s in range(S):
            ctr = O.cal_contours(q[s], N, True, cd)
            _, cnt = O.cal_integral_within_contours_hist(q[s], ctr, dA, None, True, return_counts=True)
            assert np.array_equal(b['counts'][s].astype(np.int64), cnt), s
        for s in (0, 2, 7):
            check_nine(b, s, O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=True, lt=True, dtype=cd))
        plan.free()
    # the reference's own workflow: a SUPPLIED float32 squared gradient (its notebooks pass grdSpv) -- same variant, other layout
    g2 = np.stack([O.grad2_sphere(q[s], lat, lon) for s in range(S)]).astype(np.float32)
    g2[3, 10, 20:40] = np.nan                                             # fillna(0) on the product, core.py:449
    plan = KeffPlan(ctx, S, ny, nx, N, np.float32, np.float32, dA=dA, tbl=tbl, tbl_coord=lat, increase=True, lt=True,
                    grdS_dtype=np.float32, nslots=2)
    plan.set_q(q); plan.set_grdS(g2)
    plan.run(0); plan.run(1, 5, chain=True)
    ref, b = plan.fetch(slot=0), plan.fetch(slot=1)
    assert np.array_equal(b['counts'], ref['counts']) and rel(b['intgrdS'], ref['intgrdS']) < 1e-13
    for s in (0, 3, 9):
        r = O.keff_pipeline(q[s], dA, lat, N, grdS=g2[s], increase=True, lt=True, dtype=np.float32)
        check_nine(b, s, r)
    plan.free()


# ---------------------------------------------------------------- the chained min/max cache must not go stale
def test_chain_then_new_batch_is_not_stale(ctx):
    """run(chain=True) leaves min/max partials keyed on the batch pointer; set_q / synth / a raw upload / touch() of a
    NEW batch behind the same pointer must drop them -- the natural time loop `run(chain); set_q(next); run()`"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N, S = 91, 180, 51, 4
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 2.0
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(dA.sum(1), True)
    rng = np.random.default_rng(5)
    mk = lambda k: (np.sin(np.deg2rad(lat))[None, :, None] * (1 + k) + 0.1 * rng.standard_normal((S, ny, nx)))
    plan = KeffPlan(ctx, S, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, nslots=2)

    def fresh(q):
        p2 = KeffPlan(ctx, S, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                      increase=True, lt=True)
        p2.set_q(q); p2.run(); o = p2.fetch(); p2.free()
        return o

    q0, q1, q2, q3 = mk(0), mk(1), mk(2), mk(3)
    plan.set_q(q0); plan.run(0, chain=True)                     # leaves partials of q0 (the batch "again")
    # (1) set_q
    plan.set_q(q1); plan.run(1)
    o, f = plan.fetch(slot=1), fresh(q1)
    assert np.array_equal(o['ctr'], f['ctr']) and np.array_equal(o['counts'], f['counts'])
    for s in range(S):
        assert np.array_equal(o['ctr'][s], O.cal_contours(q1[s], N, True, np.float64))
    # (2) raw upload through the buffer (xc_memcpy_h2d overlap test), no touch()
    plan.run(0, chain=True)
    plan.q_buf.upload(q2); plan.run(1)
    o, f = plan.fetch(slot=1), fresh(q2)
    assert np.array_equal(o['ctr'], f['ctr']) and np.array_equal(o['counts'], f['counts'])
    # (3) on-device generator
    plan.run(0, chain=True)
    plan.synth(lat, lon, 123, 0); plan.run(1)
    o = plan.fetch(slot=1)
    qs = plan.download_q()
    for s in range(S):
        assert np.array_equal(o['ctr'][s], O.cal_contours(qs[s], N, True, np.float64))
    # (4) a write the library cannot see (here: a second context's copy engine) + touch()
    plan.run(0, chain=True)
    other = type(ctx)(0)
    q3c = np.ascontiguousarray(q3)
    other._check(other.lib.xc_memcpy_h2d(other.handle, plan.q_buf.ptr, q3c.ctypes.data, q3c.nbytes))
    other.close()
    plan.touch(); plan.run(1)
    o, f = plan.fetch(slot=1), fresh(q3)
    assert np.array_equal(o['ctr'], f['ctr']) and np.array_equal(o['counts'], f['counts'])
    # (5) an unchanged batch still uses the chained partials and is bit-identical
    plan.run(0, chain=True); plan.run(1, chain=True)
    assert np.array_equal(plan.fetch(slot=1)['ctr'], f['ctr'])
    # (6) freeing the plan and re-allocating (address reuse) must not inherit anything
    plan.run(0, chain=True)
    plan.free()
    p3 = KeffPlan(ctx, S, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                  increase=True, lt=True)
    p3.q_buf.upload(q1); p3.run()
    o = p3.fetch(); p3.free()
    f = fresh(q1)
    assert np.array_equal(o['ctr'], f['ctr']) and np.array_equal(o['counts'], f['counts'])


# ---------------------------------------------------------------- time-varying weights (core.py:1271-1274)
@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_per_slab_dA_through_plan_and_facade(ctx, dt):
    """dA with a leading (time) dim: KeffPlan (XC_DA_SLAB), chained launch sets with a slab offset into dA, and
    Contour2D.keff incl. batching (max_batch_bytes) -- every slab against the oracle with ITS weights"""
    import xcontour_amd as xa
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N, S = 73, 144, 61, 5
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 2.5
    base = cell_area(lat, lon)
    rng = np.random.default_rng(17)
    dA = base[None] * (1.0 + 0.3 * rng.random((S, ny, nx)))               # e.g. a layer thickness that evolves in time
    q = (np.sin(np.deg2rad(lat))[None, :, None] * (1 + 0.2 * np.arange(S))[:, None, None]
         + 0.1 * rng.standard_normal((S, ny, nx))).astype(dt)
    tbl = table_from_rowsums(base.sum(1), True)
    plan = KeffPlan(ctx, S, ny, nx, N, dt, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, nslots=2)
    plan.set_q(q)
    plan.run(0)
    plan.run(1, 2, chain=True)                                             # sets of 2, 2, 1 slabs: dA pointer advances per set
    rs = []
    for s in range(S):
        r = O.keff_pipeline(q[s], dA[s], lat, N, lon=lon, increase=True, lt=True, dtype=np.float64)
        r['tbl'] = tbl                                                     # the table belongs to the (static) mask metric
        r['latEq'] = O.lookup_coordinates(r['area'], tbl, lat)
        rs.append(r)
    for slot in (0, 1):
        out = plan.fetch(slot=slot)
        for s in range(S):
            assert np.array_equal(out['counts'][s].astype(np.int64), rs[s]['counts'])
            assert rel(out['area'][s], rs[s]['area']) < TIGHT and rel(out['intgrdS'][s], rs[s]['intgrdS']) < TIGHT
            assert rel(out['latEq'][s], rs[s]['latEq']) < 1e-9
    plan.free()
    # façade
    c = {'time': np.arange(float(S)), 'lat': lat, 'lon': lon}
    tr = xa.DataArray(q, ('time', 'lat', 'lon'), c, 'pv')
    dAl = xa.DataArray(dA, ('time', 'lat', 'lon'), c, 'dA')
    cm = xa.Contour2D(tr, dAl, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, dtype=np.float64)
    table = xa.Table(xa.DataArray(tbl, ('lat',), {'lat': lat}, 'AeqCTbl'), 'lat')
    one = cm.keff(N, table, lat=lat, lon=lon)
    two = cm.keff(N, table, lat=lat, lon=lon, max_batch_bytes=2 * ny * nx * (q.itemsize + 8))   # batches of 2, 2, 1
    for ds in (one, two):
        assert ds['area'].dims == ('time', 'contour')
        for s in range(S):
            assert rel(ds['area'].values[s], rs[s]['area']) < TIGHT
            assert rel(ds['intgrdS'].values[s], rs[s]['intgrdS']) < TIGHT
            assert rel(ds['latEq'].values[s], rs[s]['latEq']) < 1e-9
    # the separate-call API already took per-slab weights; it must agree
    ctr = cm.cal_contours(N)
    area = cm.cal_integral_within_contours_hist(ctr)
    assert rel(area.values, one['area'].values) < 1e-13


# ---------------------------------------------------------------- table length / order
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


# ---------------------------------------------------------------- Q's leading dims follow the TRACER's order
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


# ---------------------------------------------------------------- one context per device, several devices per process
def test_contexts_on_every_visible_device():
    """INTEGRATION.md threading contract: one context per GPU, driven from threads of ONE process.  The >64 KB LDS
    opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-device property of a kernel: every device's FIRST
    launch of the big-LDS kernels must work.  Skips on a single-GPU box."""
    import threading
    from xcontour_amd import _native as nat
    ctxs = []
    for d in range(16):
        try:
            ctxs.append(nat.Context(d))
        except nat.XContourHipError:
            break
    try:
        if len(ctxs) < 2:
            pytest.skip('needs at least two visible GPUs (this box has %d)' % len(ctxs))
        rng = np.random.default_rng(9)
        q = rng.standard_normal((3, 120, 200))
        ed = np.linspace(-4, 4, 202)
        dA = rng.random((120, 200)) + 0.5
        _, cnt = zip(*[O.weighted_histogram(q[s], ed, dA, 'numpy') for s in range(3)])
        out, errs = [None] * len(ctxs), []

        def work(i):
            try:
                out[i] = ctxs[i].hist(q, ed, dA=dA, want=('counts', 'cdf'))     # ~100 KB of LDS histogram copies
            except Exception as e:            # pragma: no cover
                errs.append((i, e))
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for o in out:
            for s in range(3):
                assert np.array_equal(o['counts'][s].astype(np.int64), cnt[s])
    finally:
        for c in ctxs:
            c.close()
