"""-m gpu: the HIP path (through the C ABI) against the oracle on the same inputs.
Bar: bit-exact for bin counts / masks / levels; float64 sums within 1e-6 relative as
BASELINE.json states (most checks use a far tighter bound, written next to each)."""
import os

import numpy as np
import pytest

import xcontour_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
RTOL = 1e-6          # north_star tolerance for float64 area / Keff / LWA values
TIGHT = 1e-11        # what the kernels actually achieve on sums (summation order only)


def rel(a, b, floor=1e-300):
    """max |a-b| / max(|b|, floor); NaN / inf patterns must be identical.  `floor` is an
    absolute scale for quantities that legitimately pass through zero (Lmin = 2 pi R cos(lat)
    at the pole is ~1e-9 m on a 4e7 m scale: its relative error is meaningless there)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = np.isfinite(b)
    assert np.array_equal(a[~m & ~np.isnan(b)], b[~m & ~np.isnan(b)])       # infs identical
    if not m.any():
        return 0.0
    scale = np.maximum(np.abs(b[m]), floor)
    return float(np.max(np.abs(a[m] - b[m]) / scale))


LMIN_FLOOR = 2 * np.pi * 6371200.0 * 1e-6    # Lmin compared on the scale of 1e-6 of the equator length


# ---------------------------------------------------------------- K1 / levels
@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_minmax_and_levels_bit_exact(ctx, baro, dt):
    q = baro[0].astype(dt)
    rng = np.random.default_rng(3)
    x = np.stack([q, q[::-1] * 2, rng.standard_normal(q.shape).astype(dt)])
    x[2, 5, 7] = np.nan
    mm = ctx.minmax(x)
    for s in range(3):
        assert mm[s, 0] == np.nanmin(x[s]) and mm[s, 1] == np.nanmax(x[s])
    for N in (2, 20, 121, 201, 251):
        for inc in (True, False):
            for cd in (np.float32, np.float64):
                for re_, rn in ((0, 'numpy'), (1, 'xhistogram')):
                    ctr, edges, st = ctx.levels(mm, dt, N, inc, cd, re_)
                    for s in range(3):
                        ref = O.cal_contours(x[s], N, inc, cd)
                        assert np.array_equal(ctr[s], ref.astype(np.float64))
                        e, _ = O.hist_edges(ref)
                        if rn == 'xhistogram':
                            e = np.concatenate((e[:-1], e[-1:] + 1e-8))
                        assert np.array_equal(edges[s], e.astype(np.float64))
                        assert st[s] == 0


def test_minmax_all_nan_and_odd_sizes(ctx):
    x = np.full((2, 1, 77), np.nan)
    x[1, 0, 3] = 2.5
    mm = ctx.minmax(x)
    assert np.isnan(mm[0]).all() and mm[1, 0] == 2.5 and mm[1, 1] == 2.5
    for n in (1, 2, 3, 5, 1023, 4097):
        y = np.arange(n, dtype=np.float32)[None] - 7
        mm = ctx.minmax(y)
        assert mm[0, 0] == -7 and mm[0, 1] == n - 8


def test_constant_field_raises_like_reference(ctx):
    import xcontour_amd as xa
    q = xa.DataArray(np.ones((8, 16)), ('lat', 'lon'), {'lat': np.arange(8.), 'lon': np.arange(16.)}, 'q')
    cm = xa.Contour2D(q, np.ones(8), {'X': 'lon', 'Y': 'lat'}, {'Y': 'lat'}, lt=True)
    ctr = cm.cal_contours(5)
    with pytest.raises(Exception, match='non monotonic bins'):
        cm.cal_integral_within_contours_hist(ctr)


# ---------------------------------------------------------------- K3 histogram
@pytest.mark.parametrize('nx', [512, 131, 130, 64, 2, 1])
@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_hist_random_shapes(ctx, nx, dt):
    """ragged strips, odd nx (VEC=1), NaNs, per-slab edges, all weight ranks"""
    rng = np.random.default_rng(nx)
    ny, S = 37, 3
    x = rng.standard_normal((S, ny, nx)).astype(dt)
    x[0, 3, 0] = np.nan
    x[1, :, nx // 2] = np.nan
    ed = np.stack([np.sort(rng.uniform(-2.5, 2.5, 41)) for _ in range(S)])
    for dA in (None, rng.random(ny), rng.random((ny, nx)), rng.random((S, ny, nx))):
        out = ctx.hist(x, ed, dA=dA)
        for s in range(S):
            w = np.ones((ny, nx)) if dA is None else (dA[:, None] if dA.ndim == 1 else (dA if dA.ndim == 2 else dA[s]))
            p, c = O.weighted_histogram(x[s], ed[s], np.broadcast_to(w, (ny, nx)), 'numpy')   # ctx.hist: explicit edges, closed last bin
            assert np.array_equal(out['counts'][s].astype(np.int64), c)
            assert np.allclose(out['pdf'][s, 0], p, rtol=1e-12, atol=1e-13)
            assert np.allclose(out['cdf'][s, 0], np.cumsum(p), rtol=1e-12, atol=1e-13)


def test_hist_ties_on_edges_and_last_bin(ctx):
    """cells exactly on edges; closed vs half-open last bin; out-of-range both sides"""
    ed = np.array([0., 1., 2., 3., 4.])
    vals = np.array([-1., 0., 0.5, 1., 1., 2., 3., 3.999, 4., 4., 4.0000001, 5., np.nan, np.inf, -np.inf, 2.])
    x = np.tile(vals, (4, 8))[None]                  # (1, 4, 128)
    w = np.arange(x.size, dtype=np.float64).reshape(x.shape[1:]) + 1
    for closed in (True, False):
        out = ctx.hist(x, ed, dA=w, last_closed=closed)
        e = ed if closed else ed.copy()
        idx = np.digitize(x.ravel(), e)
        if closed:
            idx = np.where(x.ravel() == e[-1], 4, idx)
        c = np.bincount(idx, minlength=6)[1:5]
        p = np.bincount(idx, weights=w.ravel(), minlength=6)[1:5]
        assert np.array_equal(out['counts'][0].astype(np.int64), c)
        assert np.allclose(out['pdf'][0, 0], p, rtol=1e-13)


def test_hist_nonuniform_levels_binary_search(ctx):
    """strongly non-uniform edges defeat the uniform guess: the fix-up search must still be exact"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 64, 256))
    ed = np.concatenate(([-4.0], -3 + np.cumsum(np.abs(rng.standard_normal(300)) ** 3 * 0.02)))
    out = ctx.hist(x, ed, dA=None)
    for s in range(2):
        _, c = O.weighted_histogram(x[s], ed, None, 'numpy')
        assert np.array_equal(out['counts'][s].astype(np.int64), c)


def test_hist_many_bins_reduced_copies(ctx):
    """N large enough that the LDS copy count drops below 16"""
    rng = np.random.default_rng(6)
    x = rng.standard_normal((1, 50, 300))
    w = rng.random((50, 300))
    for nb in (401, 1001, 3000):
        ed = np.linspace(-3, 3, nb + 1)
        out = ctx.hist(x, ed, dA=w, integrands=[x], want=('pdf', 'counts'))
        p, c = O.weighted_histogram(x[0], ed, w, 'numpy')
        p1, _ = O.weighted_histogram(x[0], ed, w * x[0], 'numpy')
        assert np.array_equal(out['counts'][0].astype(np.int64), c)
        assert np.allclose(out['pdf'][0, 0], p, rtol=1e-12, atol=1e-14)
        assert np.allclose(out['pdf'][0, 1], p1, rtol=1e-11, atol=1e-13)


def test_hist_rejects_bad_edges(ctx):
    from xcontour_amd import _native as nat
    x = np.zeros((1, 4, 4))
    with pytest.raises(nat.XContourHipError, match='non monotonic bins') as e:
        ctx.hist(x, np.array([0., 1., 1., 2.]))
    assert e.value.code == nat.XC_EEDGES
    with pytest.raises(nat.XContourHipError):
        ctx.hist(x, np.array([0., 2., 1., 3.]))


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_grad_in_kernel_matches_standalone_and_oracle(ctx, baro, dt):
    q, lat, lon = baro
    q = q.astype(dt)
    dA = O.cell_area(lat, lon)
    rdx, rdy = O.grad_metrics(lat, lon)
    g2 = O.grad2_sphere(q, lat, lon)
    gg = ctx.grad2(q[None], rdx, rdy, True)
    assert np.array_equal(gg[0], g2)                              # same order of operations: bit-exact
    ctr = O.cal_contours(q, 201, True, np.float32)
    edges, _ = O.hist_edges(ctr)
    out = ctx.hist(q[None], edges.astype(np.float64), dA=dA, grad=(rdx, rdy, True))
    w = np.where(np.isnan(g2 * dA), 0, g2 * dA)
    p1, c = O.weighted_histogram(q, edges, w, 'numpy')
    assert np.array_equal(out['counts'][0].astype(np.int64), c)
    assert rel(out['pdf'][0, 1], p1) < TIGHT
    # non-periodic walls (X-Z planes): one-sided differences
    q2 = q[:50, :300].copy()
    rdx2, rdy2 = np.full(50, 1 / 4.0), 1.0 / (np.minimum(np.arange(50) + 1, 49) - np.maximum(np.arange(50) - 1, 0))
    gx = np.gradient(q2.astype(np.float64), 2.0, axis=1, edge_order=1)
    gy = np.gradient(q2.astype(np.float64), 1.0, axis=0, edge_order=1)
    gg2 = ctx.grad2(q2[None], rdx2, rdy2, False)
    assert rel(gg2[0], gx * gx + gy * gy) < 1e-12
    out2 = ctx.hist(q2[None], edges.astype(np.float64), dA=None, grad=(rdx2, rdy2, False))
    p2, _ = O.weighted_histogram(q2, edges, gg2[0], 'numpy')
    assert rel(out2['pdf'][0, 1], p2) < TIGHT


# ---------------------------------------------------------------- facade: the reference's own call sequences
def _baro_da(xa, baro, flip=False):
    q, lat, lon = baro
    if flip:
        q, lat = q[::-1].copy(), lat[::-1].copy()
    tr = xa.DataArray(q, ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'absolute_vorticity')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'rA')
    return tr, dA, q, lat, lon


@pytest.mark.parametrize('increase', [True, False])
@pytest.mark.parametrize('lt', [True, False])
@pytest.mark.parametrize('flip', [False, True])
@pytest.mark.parametrize('rule', ['xhistogram', 'numpy'])
def test_keff_call_sequence_8_cases(ctx, baro, increase, lt, flip, rule):
    """tests/test_hist.py computeKeff_hist / computeKeff for all (increase, lt) x coordinate direction, under both
    last-bin rules (the fixture's latitudes are float32: under 'xhistogram' the last row leaves the table)"""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro, flip)
    N = 251
    g2 = O.grad2_sphere(q, lat, lon)
    grdS = xa.DataArray(g2, tr.dims, tr.coords, 'grdS')
    mask = xa.DataArray(np.ones_like(q), tr.dims, tr.coords, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'},
                      increase=increase, lt=lt, right_edge=rule)
    table = cm.cal_area_eqCoord_table_hist(mask)
    ctr = cm.cal_contours(N)
    area = cm.cal_integral_within_contours_hist(ctr).rename('intArea')
    intgrdS = cm.cal_integral_within_contours_hist(ctr, integrand=grdS).rename('intgrdS')
    Yeq = table.lookup_coordinates(area).rename('Yeq')
    Lmin = xa.latitude_lengths_at(Yeq).rename('Lmin')
    dgrdSdA = cm.cal_gradient_wrt_area(intgrdS, area)
    dqdA = cm.cal_gradient_wrt_area(ctr, area)
    Leq2 = cm.cal_sqared_equivalent_length(dgrdSdA, dqdA)
    nkeff = cm.cal_normalized_Keff(Leq2, Lmin, mask=2e7)
    # oracle, same sequence
    o_tbl, o_cs = O.cal_area_eqCoord_table_hist(np.ones_like(q), dA.values, lat, increase, lt, rule)
    o_ctr = O.cal_contours(q, N, increase, np.float32)
    o_area = O.cal_integral_within_contours_hist(q, o_ctr, dA.values, None, lt, rule)
    o_S = O.cal_integral_within_contours_hist(q, o_ctr, dA.values, g2, lt, rule)
    o_Yeq = O.lookup_coordinates(o_area, o_tbl, o_cs)
    o_Lmin = O.latitude_lengths_at(o_Yeq)
    o_dS = O.cal_gradient_wrt_area(o_S, o_area)
    o_dq = O.cal_gradient_wrt_area(o_ctr, o_area)
    o_Leq2 = O.cal_sqared_equivalent_length(o_dS, o_dq)
    o_nk = O.cal_normalized_Keff(o_Leq2, o_Lmin, 2e7)
    assert rel(table._table.values, o_tbl) < 1e-13 and np.array_equal(table._coord, o_cs)
    assert ctr.values.dtype == np.float32 and np.array_equal(ctr.values, o_ctr)
    assert rel(area.values, o_area) < TIGHT and rel(intgrdS.values, o_S) < TIGHT
    assert rel(Yeq.values, o_Yeq) < 1e-9
    assert rel(dqdA.values, o_dq) < 1e-8 and rel(dgrdSdA.values, o_dS) < 1e-8
    assert rel(Leq2.values, o_Leq2) < RTOL and rel(nkeff.values, o_nk) < RTOL
    assert area.name == 'intArea' and dqdA.name == 'dabsolute_vorticitydA' and Leq2.name == 'Leq2' and nkeff.name == 'nkeff'
    assert area.dims == ('contour',) and area.coords['contour'].dtype == np.float32
    # the xarray-style twin (strict comparisons) on the GPU vs the oracle's twin
    t2 = cm.cal_area_eqCoord_table(mask)
    o_t2, _ = O.cal_area_eqCoord_table(np.ones_like(q), dA.values, lat, increase, lt)
    assert rel(t2._table.values, o_t2) < 1e-13
    a2 = cm.cal_integral_within_contours(ctr)
    o_a2 = O.cal_integral_within_contours(q, o_ctr, dA.values, None, lt)
    assert rel(a2.values, o_a2) < TIGHT
    s2 = cm.cal_integral_within_contours(ctr, integrand=grdS)
    assert rel(s2.values, O.cal_integral_within_contours(q, o_ctr, dA.values, g2, lt)) < TIGHT
    # interpolation to prescribed latitudes
    preY = np.linspace(-90, 90, N)
    ds = cm.interp_to_dataset(preY, Yeq, [ctr, area, Yeq, nkeff])
    assert rel(ds['intArea'].values, O.interp_to_coords(preY, o_Yeq, o_area)) < 1e-9
    assert ds['nkeff'].dims == ('new',)


@pytest.mark.parametrize('rule,sfx', [('xhistogram', ''), ('numpy', '_numpy')])
def test_golden_keff_fixtures(ctx, baro, rule, sfx):
    """the committed golden vectors (tests/golden/make_golden.py) through the fused pipeline, both last-bin rules.
    The reference's bundled field has float32 latitudes and float32 contours of magnitude 1e-4: under 'xhistogram'
    (the default) the max cell IS counted (131 072) and the last row leaves the table (end = 0.99994 x total)."""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    assert lat.dtype == np.float32
    mask = xa.DataArray(np.ones_like(q), tr.dims, tr.coords, 'mask')
    kw = {} if rule == 'xhistogram' else {'right_edge': rule}           # 'xhistogram' must be the default
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'},
                      increase=True, lt=True, **kw)
    table = cm.cal_area_eqCoord_table_hist(mask)
    tot = dA.values.sum()
    end = table._table.values[-1] / tot
    assert abs(end - (0.99994 if rule == 'xhistogram' else 1.0)) < (1e-5 if rule == 'xhistogram' else 1e-14)
    for N in (121, 201):
        g = np.load(os.path.join(GOLD, 'baro_keff_N%d%s.npz' % (N, sfx)))
        assert rel(table._table.values, g['tbl']) < 1e-13
        ds = cm.keff(N, table, preY=lat, lat=lat, lon=lon)
        assert np.array_equal(ds['ctr'].values, g['ctr'].astype(np.float64))
        for k in ('area', 'intgrdS'):
            assert rel(ds[k].values, g[k]) < TIGHT, k
        for k in ('latEq', 'dqdA', 'dintSdA', 'Leq2', 'nkeff'):
            assert rel(ds[k].values, g[k]) < RTOL, k
        assert rel(ds['Lmin'].values, g['Lmin'], LMIN_FLOOR) < RTOL
        for k in ('ctr', 'area', 'latEq', 'nkeff', 'Leq2'):
            assert rel(ds[k + '_eq'].values, g[k + '_eq']) < RTOL, k
        # the façade's separate calls (reference call sequence) land on the same golden numbers
        ctr = cm.cal_contours(N)
        area = cm.cal_integral_within_contours_hist(ctr)
        assert rel(area.values, g['area']) < TIGHT
        assert rel(table.lookup_coordinates(area).values, g['latEq']) < 1e-9
    # the headline differences between the rules, N = 201 above, N = 121 here
    g = np.load(os.path.join(GOLD, 'baro_keff_N121%s.npz' % sfx))
    ds = cm.keff(121, table, lat=lat, lon=lon)
    e, _ = O.hist_edges(g['ctr'])
    cnt = cm.ctx.hist(q[None], _rule_edges(e, rule), dA=dA.values, last_closed=(rule == 'numpy'), want=('counts',))['counts'][0]
    assert int(cnt.sum()) == (131072 if rule == 'xhistogram' else 131071) and np.array_equal(cnt.astype(np.int64), g['counts'])
    assert abs(ds['latEq'].values[-1] - (89.4631 if rule == 'xhistogram' else 89.4399)) < 1e-4
    assert abs(ds['nkeff'].values[-1] - (104.963 if rule == 'xhistogram' else 95.8586)) < 1e-2


def _rule_edges(e, rule):
    """explicit f64 edges for ctx.hist from level-dtype edges: xhistogram bumps the last one in the edge dtype"""
    e = np.asarray(e)
    if rule == 'xhistogram':
        e = np.concatenate((e[:-1], e[-1:] + 1e-8))
    return e.astype(np.float64)


@pytest.mark.parametrize('increase,lt,cd,re_,latdt', [
    (True, True, np.float64, 'numpy', np.float64), (False, True, np.float32, 'numpy', np.float32),
    (True, False, np.float32, 'xhistogram', np.float64), (False, False, np.float64, 'numpy', np.float64),
    (True, True, np.float64, 'xhistogram', np.float32), (False, False, np.float32, 'xhistogram', np.float32),
    (False, True, np.float64, 'xhistogram', np.float32), (True, False, np.float64, 'xhistogram', np.float32)])
def test_fused_pipeline_batch_vs_oracle(ctx, increase, lt, cd, re_, latdt):
    """xc_keff_dev on a batch of synthetic slabs with per-slab levels, counts bit-exact; the A(Yeq) table comes from
    the product's own path (K2 row sums + the last-row rule) and float32 latitudes drop its last row under 'xhistogram'"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    ny, nx, N, S = 181, 360, 101, 5
    lat = np.linspace(-90, 90, ny).astype(latdt); lon = np.arange(nx) * 1.0
    dA = cell_area(lat, lon)
    ylt = lt if increase else (not lt)
    keep = last_row_included(lat, re_)
    assert keep == (not (re_ == 'xhistogram' and latdt == np.float32))
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), ylt, keep)
    o_tbl, _ = O.cal_area_eqCoord_table_hist(np.ones((ny, nx)), dA, lat, increase, lt, re_)
    assert rel(tbl, o_tbl) < 1e-13
    plan = KeffPlan(ctx, S, ny, nx, N, np.float64, cd, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    preY=lat, increase=increase, lt=lt, right_edge=re_)
    plan.synth(lat, lon, 77, 0)
    plan.run()
    out = plan.fetch()
    q = plan.download_q()
    for s in range(S):
        r = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=increase, lt=lt, dtype=cd, preLats=lat, right_edge=re_)
        assert np.array_equal(out['counts'][s].astype(np.int64), r['counts'])
        assert np.array_equal(out['ctr'][s], r['ctr'].astype(np.float64))
        assert rel(out['area'][s], r['area']) < TIGHT and rel(out['intgrdS'][s], r['intgrdS']) < TIGHT
        for k in ('latEq', 'dqdA', 'dintSdA', 'Leq2'):
            assert rel(out[k][s], r[k]) < RTOL, k
        assert rel(out['Lmin'][s], r['Lmin'], LMIN_FLOOR) < RTOL
        ok = r['Lmin'] > LMIN_FLOOR                                    # nkeff = Leq2/Lmin^2 away from the pole
        assert rel(out['nkeff'][s][ok], r['nkeff'][ok]) < RTOL
        for k in ('ctr', 'area'):
            assert rel(out[k + '_eq'][s], r[k + '_eq']) < RTOL, k
    plan.free()


def test_pipeline_supplied_grdS_f32(ctx, baro):
    """PV.nc-like usage: float32 tracer, supplied float32 grdS, float32 dA -> f32 products (core.py:444)"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import table_from_rowsums
    q, lat, lon = baro
    dA32 = O.cell_area(lat, lon).astype(np.float32)
    g32 = O.grad2_sphere(q, lat, lon).astype(np.float32)
    tbl = table_from_rowsums(dA32.astype(np.float64).sum(1), True)
    plan = KeffPlan(ctx, 1, 256, 512, 121, np.float32, np.float32, dA=dA32, tbl=tbl, tbl_coord=lat, increase=True,
                    lt=True, grdS_dtype=np.float32, prod_f32=True)
    plan.set_q(q); plan.set_grdS(g32); plan.run()
    out = plan.fetch()
    ctr = O.cal_contours(q, 121, True, np.float32)
    S = O.cal_integral_within_contours_hist(q, ctr, dA32, g32, True)
    assert rel(out['intgrdS'][0], S) < TIGHT
    plan.free()


# ---------------------------------------------------------------- K7 local wave activity
def test_lwa_golden_and_call_sequence(ctx, baro):
    """tests/test_LWA.py:35-78: sorted state, then LWA with mask_idx=[37,125,170,213]"""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    g = np.load(os.path.join(GOLD, 'baro_lwa_N121.npz'))
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    mask = xa.DataArray(np.ones_like(q), tr.dims, tr.coords, 'mask')
    ctr = cm.cal_contours(121)
    table = cm.cal_area_eqCoord_table_hist(mask)
    area = cm.cal_integral_within_contours_hist(ctr).rename('intArea')
    latEq = table.lookup_coordinates(area).rename('latEq')
    ds_latEq = cm.interp_to_dataset(xa.DataArray(lat, ('latitude',), {'latitude': lat}), latEq, [ctr, area, latEq])
    Q = ds_latEq['absolute_vorticity']
    assert rel(Q.values, g['Q']) < 1e-9
    Qg = xa.DataArray(g['Q'], ('latitude',), {'latitude': lat}, 'absolute_vorticity')
    lwa, ctrs, masks = cm.cal_local_wave_activity(tr, Qg, mask_idx=[37, 125, 170, 213], part='all', metric=g['dy'])
    assert lwa.name == 'LWA' and lwa.dims == tr.dims
    assert np.array_equal(lwa.values, g['lwa_dy'])                 # same summation order: bit-exact
    assert abs(lwa.values.max() - 28.921) < 1e-3
    for i in range(4):
        assert np.array_equal(masks[i].values, g['masks'][i])
        assert ctrs[i].values == g['Q'][[37, 125, 170, 213][i]]
    assert np.array_equal(cm.cal_local_wave_activity(tr, Qg).values, g['lwa_dA'])          # snapshot metric
    assert np.array_equal(cm.cal_local_wave_activity(tr, Qg, part='upper', metric=g['dy']).values, g['lwa_upper'])
    assert np.array_equal(cm.cal_local_APE(tr, Qg, part='lower', metric=g['dy']).values, g['lwa_lower'])
    with pytest.raises(Exception, match='invalid part'):
        cm.cal_local_wave_activity(tr, Qg, part='middle')
    with pytest.raises(Exception, match='out of boundary'):
        cm.cal_local_wave_activity(tr, Qg, mask_idx=[256])


@pytest.mark.parametrize('increase', [True, False])
@pytest.mark.parametrize('flip', [False, True])
def test_lwa_directions_and_nans(ctx, increase, flip):
    rng = np.random.default_rng(11)
    ny, nx = 45, 70
    coord = np.linspace(-200, 0, ny) if not flip else np.linspace(0, -200, ny)
    q = rng.standard_normal((2, ny, nx)) + np.linspace(0, 3, ny)[None, :, None]
    q[0, 4, 5] = np.nan
    Q = np.sort(rng.standard_normal((2, ny)), axis=1)
    dA = rng.random((ny, nx)) + 0.5
    for part, pc in (('all', 0), ('upper', 1), ('lower', 2)):
        out, _ = ctx.lwa(q, Q, coord, dA, dA.max(), M=None, increase=increase, part=pc)
        for s in range(2):
            ref = O.cal_local_wave_activity(q[s], Q[s], coord, dA, increase, part)
            assert np.array_equal(out[s], ref)


# ---------------------------------------------------------------- full-size properties (BASELINE cfg2)
def test_cfg2_full_size_properties(ctx):
    """BASELINE cfg2 at full size (3600x1801 f64, 2-D f64 dA, 201 contours): ALL result vectors of the fused pipeline
    against the oracle's Keff call sequence on the same slabs (0.4 s of numpy per slab), counts bit-exact, plus the
    size-independent invariants"""
    from xcontour_amd.pipeline import KeffPlan, OUT_NAMES
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    ny, nx, N = 1801, 3600, 201
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.1
    dA = cell_area(lat, lon)
    rows = ctx.rowsum(None, dA, ny, nx)
    assert abs(rows.sum() / (4 * np.pi * O.Rearth ** 2) - 1) < 1e-12       # sphere area
    tbl = table_from_rowsums(rows, True, last_row_included(lat))
    assert tbl[0] == 0 and abs(tbl[-1] / rows.sum() - 1) < 1e-13           # f64 latitudes: end point = total area
    preY = np.linspace(-90, 90, 181)
    plan = KeffPlan(ctx, 2, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, preY=preY)
    plan.synth(lat, lon, 20241008, 0)
    plan.run()
    out = plan.fetch()
    q = plan.download_q()
    for s in range(2):
        assert int(out['counts'][s].sum()) == ny * nx                       # xhistogram rule: every cell in one bin
        assert out['area'][s, 0] == 0 and (np.diff(out['area'][s]) >= 0).all()
        assert abs(out['area'][s, -1] / rows.sum() - 1) < 1e-12             # cdf[-1] == total weight
        assert (np.diff(out['intgrdS'][s]) >= 0).all() and (np.diff(out['latEq'][s]) >= 0).all()
        r = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float64, preLats=preY)
        assert rel(tbl, r['tbl']) < 1e-13
        assert np.array_equal(out['counts'][s].astype(np.int64), r['counts'])   # bit-exact counts at full size
        assert np.array_equal(out['ctr'][s], r['ctr'])                           # bit-exact levels
        assert rel(out['area'][s], r['area']) < TIGHT and rel(out['intgrdS'][s], r['intgrdS']) < TIGHT
        for k in ('latEq', 'dqdA', 'dintSdA', 'Leq2'):
            assert rel(out[k][s], r[k]) < RTOL, k
        assert rel(out['Lmin'][s], r['Lmin'], LMIN_FLOOR) < RTOL
        ok = r['Lmin'] > LMIN_FLOOR                                            # nkeff = Leq2 / Lmin^2 away from the pole
        assert rel(out['nkeff'][s][ok], r['nkeff'][ok]) < RTOL
        assert np.array_equal(np.isnan(out['nkeff'][s]), np.isnan(r['nkeff']))
        for k in OUT_NAMES:
            if k in ('Lmin', 'nkeff'):
                continue
            assert rel(out[k + '_eq'][s], r[k + '_eq']) < RTOL, k + '_eq'
    # idempotence: a second run of the same plan gives bit-identical counts and 1e-13-close sums
    plan.run()
    out2 = plan.fetch()
    assert np.array_equal(out['counts'], out2['counts'])
    assert rel(out2['area'], out['area']) < 1e-13
    plan.free()


def test_chained_minmax_is_bit_identical(ctx):
    """xc_keff_desc.q_next: min/max of the next launch set accumulated inside the histogram pass"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N, S = 181, 360, 101, 6
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 1.0
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(dA.sum(1), True)
    plan = KeffPlan(ctx, S, ny, nx, N, np.float64, np.float32, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, nslots=4)
    plan.synth(lat, lon, 5, 0)
    plan.run(0)                          # stand-alone K1
    ref = plan.fetch(slot=0)
    for group in (None, 2, 3):
        plan.run(1, group, chain=True)   # first sets use K1, later ones the chained partials
        plan.run(2, group, chain=True)   # every set now runs on partials produced by a histogram pass
        for slot in (1, 2):
            out = plan.fetch(slot=slot)
            assert np.array_equal(out['ctr'], ref['ctr'])
            assert np.array_equal(out['counts'], ref['counts'])
            assert rel(out['area'], ref['area']) < 1e-13
    # a different batch behind the same pointer without chaining must not reuse anything
    plan.synth(lat, lon, 99, 0)
    plan.run(3)
    out = plan.fetch(slot=3)
    q = plan.download_q()
    assert np.array_equal(out['ctr'][0], O.cal_contours(q[0], N, True, np.float32).astype(np.float64))
    plan.free()


# ---------------------------------------------------------------- K8 exact adiabatic sort
@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_radix_sort_profile_vs_oracle(ctx, baro, dt):
    q, lat, lon = baro
    q = q.astype(dt)
    dA = O.cell_area(lat, lon)
    tbl, cs = O.cal_area_eqCoord_table_hist(np.ones_like(q), dA, lat, True, True)
    out = ctx.sort_profile(q, dA=dA, targets=tbl, tbl=tbl, coord=cs, want_sorted=True, want_acum=True)
    Qo, xs, acum = O.sorted_profile(q, dA, tbl)
    n = out['nvalid']
    assert n == q.size
    assert np.array_equal(out['q_sorted'][:n], xs.astype(np.float64))          # the sort itself: exact
    assert rel(out['acum'][:n], acum) < 1e-12                                  # parallel scan vs np.cumsum
    # Q_exact, strictly: q_sorted[searchsorted(Acum, target, 'right')]; where a target ties with an Acum value to
    # within 1e-11 of the total (the table IS a cumulative sum of the same areas) either neighbour is valid
    # (oracle.sorted_profile_brackets) -- no other deviation is tolerated
    lo, hi = O.sorted_profile_brackets(acum, tbl)
    assert (out['Q'] >= xs[lo]).all() and (out['Q'] <= xs[hi]).all()
    assert np.isin(out['Q'], xs).all()
    untied = lo == hi
    assert untied.sum() >= 5 and np.array_equal(out['Q'][untied], Qo[untied])      # away from ties: identical to the oracle
    assert abs(out['bpe'] / O.bpe_integral(q, dA, tbl, cs) - 1) < 1e-10
    # relation to the reference's histogram profile: within one contour spacing
    g = np.load(os.path.join(GOLD, 'baro_keff_N121.npz'))
    step = (q.max() - q.min()) / 120
    assert np.max(np.abs(out['Q'][1:-1] - g['ctr_eq'][1:-1])) < 1.5 * step


def test_radix_sort_nan_mask_negatives_and_ties(ctx):
    rng = np.random.default_rng(21)
    ny, nx = 61, 130                                   # not a multiple of the 1024-element wave tile
    q = np.round(rng.standard_normal((ny, nx)) * 3, 1)        # many ties, negatives, zeros
    q[3, 4] = np.nan; q[10, :] = np.nan; q[20, 5] = -np.inf; q[21, 6] = np.inf
    mask = np.ones((ny, nx)); mask[:, 7] = 0
    w = rng.random(ny) + 0.1
    out = ctx.sort_profile(q, dA=w, mask=mask, targets=np.linspace(0, w.sum() * nx, 50), want_sorted=True, want_acum=True)
    Qo, xs, acum = O.sorted_profile(q, w, np.linspace(0, w.sum() * nx, 50), mask)
    n = out['nvalid']
    assert n == len(xs)
    assert np.array_equal(out['q_sorted'][:n], xs)
    # stability: equal keys keep their original order, so the payload sequence matches numpy's stable sort
    assert rel(out['acum'][:n], acum) < 1e-12
    lo, hi = O.sorted_profile_brackets(acum, np.linspace(0, w.sum() * nx, 50))
    assert (out['Q'] >= xs[lo]).all() and (out['Q'] <= xs[hi]).all()
    assert np.array_equal(out['Q'][lo == hi], Qo[lo == hi]) and (lo == hi).sum() >= 40
    # no weights, no mask
    out2 = ctx.sort_profile(q, want_sorted=True)
    ok = ~np.isnan(q)
    assert out2['nvalid'] == ok.sum() and np.array_equal(out2['q_sorted'][:ok.sum()], np.sort(q[ok]))


def test_radix_sort_full_size_sortedness(ctx):
    """cfg2-sized slab: sortedness + permutation invariants (size-independent properties)"""
    rng = np.random.default_rng(3)
    q = rng.standard_normal((1801, 3600))
    out = ctx.sort_profile(q, want_sorted=True, want_acum=True)
    s = out['q_sorted']
    assert out['nvalid'] == q.size
    assert (np.diff(s) >= 0).all()
    assert abs(s.sum() - q.sum()) < 1e-6 and s[0] == q.min() and s[-1] == q.max()
    assert out['acum'][-1] == q.size                     # unit weights: cumulative count


@pytest.mark.parametrize('increase', [True, False])
@pytest.mark.parametrize('lt', [True, False])
def test_facade_sorted_profile_brackets_histogram_profile(ctx, baro, increase, lt):
    """Contour2D.cal_sorted_profile (exact) vs the reference's N-contour profile (SURVEY a9 relation)"""
    import xcontour_amd as xa
    q, lat, lon = baro
    if not increase:
        q = -q                                            # tracer decreasing with latitude
    c = {'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(q, ('latitude', 'longitude'), c, 'pv')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), c, 'rA')
    mask = xa.DataArray(np.ones_like(q), ('latitude', 'longitude'), c, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'},
                      increase=increase, lt=lt)
    table = cm.cal_area_eqCoord_table_hist(mask)
    Q, qs = cm.cal_sorted_profile(table, return_sorted=True)
    assert Q.dims == ('latitude',) and len(qs) == q.size
    # histogram route of the reference (tests/test_LWA.py:60-72) with many contours
    N = 401
    ctr = cm.cal_contours(N)
    area = cm.cal_integral_within_contours_hist(ctr)
    latEq = table.lookup_coordinates(area)
    Qh = cm.interp_to_coords(lat, latEq, ctr)
    step = (q.max() - q.min()) / (N - 1)
    assert np.max(np.abs(Q.values[2:-2] - Qh.values[2:-2])) < 2.5 * step
    d = np.diff(Q.values)
    assert (d >= 0).all() if increase else (d <= 0).all()


# ---------------------------------------------------------------- BASELINE cfg5 stand-in: X-Z plane, LAPE
def test_cfg5_xz_plane_lape_and_bpe(ctx):
    """tests/test_LAPE.py call sequence on a synthetic X-Z section (internalwave.nc is missing):
    non-periodic X, decreasing Z coordinate, topography mask, increase=False, lt=False."""
    import xcontour_amd as xa
    nz, nxx = 100, 4480                                                # nz = 100 x nx = 4480
    Z = -(np.arange(nz) + 0.5) * 2.0                                   # 0 ... -200 m, decreasing
    X = (np.arange(nxx) + 0.5) * 20.0
    xx, zz = np.meshgrid(X, Z)
    T = 20 + 5 * np.tanh((zz + 60 + 15 * np.sin(2 * np.pi * xx / 30000.0)) / 20.0)
    depth = 200 - 80 * np.exp(-((X - 60000) / 15000.0) ** 2)            # a ridge
    maskC = (zz > -depth[None, :]).astype(np.float64)
    b = 2e-4 * (np.where(maskC == 1, T, np.nan) - 20) * 9.81            # buoyancy, NaN in topography
    c = {'Z': Z, 'XC': X}
    bb = xa.DataArray(b, ('Z', 'XC'), c, 'buoyancy')
    yA = xa.DataArray(np.full((nz, nxx), 40.0), ('Z', 'XC'), c, 'yA')
    mk = xa.DataArray(maskC, ('Z', 'XC'), c, 'maskC')
    cm = xa.Contour2D(bb, yA, dims={'X': 'XC', 'Z': 'Z'}, dimEq={'Z': 'Z'}, increase=False, lt=False)
    N = 121
    ctr = cm.cal_contours(N)
    table = cm.cal_area_eqCoord_table_hist(mk)
    area = cm.cal_integral_within_contours_hist(ctr)
    ZEq = table.lookup_coordinates(area)
    ds = cm.interp_to_dataset(xa.DataArray(Z.astype(np.float32), ('Z',), {'Z': Z}), ZEq.rename('ZEq'), [ctr, area])
    # oracle, same sequence
    o_ctr = O.cal_contours(b, N, False, np.float32)
    o_tbl, o_cs = O.cal_area_eqCoord_table_hist(maskC, yA.values, Z, False, False)
    o_area = O.cal_integral_within_contours_hist(b, o_ctr, yA.values, None, False)
    o_ZEq = O.lookup_coordinates(o_area, o_tbl, o_cs)
    o_Q = O.interp_to_coords(Z.astype(np.float32), o_ZEq, o_ctr)
    assert np.array_equal(ctr.values, o_ctr)
    assert rel(table._table.values, o_tbl) < 1e-13 and np.array_equal(table._coord, o_cs)
    assert rel(area.values, o_area) < TIGHT and rel(ZEq.values, o_ZEq) < 1e-9
    assert rel(ds['buoyancy'].values, o_Q) < 1e-9
    lape, ctrs, masks = cm.cal_local_APE(bb, ds['buoyancy'], mask_idx=[8, 28, 51, 81])
    o_lape, o_c, o_m = O.cal_local_wave_activity(b, o_Q, Z, yA.values, False, 'all', [8, 28, 51, 81])
    assert lape.name == 'LAPE' and np.array_equal(lape.values, o_lape)
    assert all(np.array_equal(masks[i].values, o_m[i]) for i in range(4))
    # in-kernel gradient on a non-periodic Cartesian plane + exact sort + BPE integral
    from xcontour_amd.utils import cartesian_metrics
    rdx, rdy = cartesian_metrics(Z, 20.0)
    g = cm.cal_squared_gradient(rdx=rdx, rdy=rdy, periodic_x=False)
    gx = np.gradient(b, 20.0, axis=1, edge_order=1); gz = np.gradient(b, Z, axis=0, edge_order=1)
    ok = np.isfinite(g.values) & np.isfinite(gx * gx + gz * gz)
    assert rel(g.values[ok], (gx * gx + gz * gz)[ok]) < 1e-9
    Qx = cm.cal_sorted_profile(table, mask=mk)
    assert (np.diff(Qx.values[np.isfinite(Qx.values)]) >= 0).all()      # buoyancy increases with ascending Z
    out = ctx.sort_profile(b, dA=yA.values, mask=maskC, tbl=o_tbl, coord=o_cs, negate=False)
    assert abs(out['bpe'] / O.bpe_integral(b, yA.values, o_tbl, o_cs, maskC) - 1) < 1e-10


@pytest.mark.parametrize('increase', [True, False])
def test_lwa2_variant(ctx, baro, increase):
    """cal_local_wave_activity2 (core.py:802-905; tests/test_LWA.py:79): SURVEY 8(f2)"""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    g = np.load(os.path.join(GOLD, 'baro_lwa_N121.npz'))
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=increase, lt=True)
    Qv = g['Q'] if increase else g['Q'][::-1].copy()
    Qg = xa.DataArray(Qv, ('latitude',), {'latitude': lat}, 'absolute_vorticity')
    for part in ('all', 'upper', 'lower'):
        out = cm.cal_local_wave_activity2(tr, Qg, part=part, metric=g['dy'])
        ref = O.cal_local_wave_activity2(q, Qv, lat, dA.values, increase, part, metric=g['dy'])
        assert np.array_equal(out.values, ref), part
    out, ctrs, masks = cm.cal_local_wave_activity2(tr, Qg, mask_idx=[37, 125], metric=g['dy'])
    ref, _, rm = O.cal_local_wave_activity2(q, Qv, lat, dA.values, increase, 'all', [37, 125], metric=g['dy'])
    assert all(np.array_equal(masks[i].values, rm[i]) for i in range(2))


def test_contour_mean_family(ctx, baro):
    """cal_contour_weigh_mean(_hist) / cal_contour_mean(_hist) (core.py:491-616): SURVEY 8(f1)"""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    g2 = O.grad2_sphere(q, lat, lon)
    grdm = xa.DataArray(np.sqrt(g2), tr.dims, tr.coords, 'grdm')
    integ = xa.DataArray(np.cos(np.deg2rad(lat))[:, None] * np.ones_like(q, dtype=np.float64), tr.dims, tr.coords, 'coslat')
    ctr = cm.cal_contours(61)
    o_ctr = O.cal_contours(q, 61, True, np.float32)
    o_area = O.cal_integral_within_contours_hist(q, o_ctr, dA.values, None, True)
    for hist in (True, False):
        fn_w = cm.cal_contour_weigh_mean_hist if hist else cm.cal_contour_weigh_mean
        fn_m = cm.cal_contour_mean_hist if hist else cm.cal_contour_mean
        integral = O.cal_integral_within_contours_hist if hist else O.cal_integral_within_contours
        oa = o_area if hist else O.cal_integral_within_contours(q, o_ctr, dA.values, None, True)
        lw = fn_w(ctr, integ)
        o_lw = O.cal_gradient_wrt_area(integral(q, o_ctr, dA.values, integ.values, True), oa)
        assert lw.name == 'lwmcoslat' and rel(lw.values, o_lw) < 1e-7
        mean = fn_m(ctr, integ, grdm)
        up = O.cal_gradient_wrt_area(integral(q, o_ctr, dA.values, integ.values * grdm.values, True), oa)
        lo = O.cal_gradient_wrt_area(integral(q, o_ctr, dA.values, grdm.values, True), oa)
        with np.errstate(divide='ignore', invalid='ignore'):
            assert mean.name == 'cmcoslat' and rel(mean.values, up / lo) < 1e-6
    # the histogram variant runs area + both integrals as channels of ONE K3 pass
    calls = []
    real = type(cm.ctx).hist
    try:
        type(cm.ctx).hist = lambda self, *a, **k: (calls.append(len(k.get('integrands', ()))), real(self, *a, **k))[1]
        fused = cm.cal_contour_mean_hist(ctr, integ, grdm)
        assert calls == [2]
        calls.clear()
        lw_u = cm.cal_contour_weigh_mean_hist(ctr, xa.DataArray(integ.values * grdm.values, tr.dims, tr.coords, None))
        lw_l = cm.cal_contour_weigh_mean_hist(ctr, grdm)
        assert len(calls) == 4
    finally:
        type(cm.ctx).hist = real
    with np.errstate(divide='ignore', invalid='ignore'):
        assert rel(fused.values, lw_u.values / lw_l.values) < 1e-9


# ---------------------------------------------------------------- BASELINE cfg1 stand-in: PV-like (level, lat, lon) f32 stack
def test_cfg1_multilevel_facade(ctx):
    """notebooks/1.Keff_atmos.ipynb / tests/test_Keff_atmos.py on a PV.nc-shaped stand-in (PV.nc is missing):
    15 levels x 241 x 480 float32, contour sets that differ per level (the thing xhistogram cannot do,
    core.py:1259-1294), supplied float32 grdS, results with a leading 'level' dim."""
    import xcontour_amd as xa
    nl, ny, nx, N = 15, 241, 480, 121
    lat = np.linspace(-90, 90, ny).astype(np.float32); lon = (np.arange(nx) * 0.75).astype(np.float32)
    lev = np.linspace(265, 850, nl).astype(np.float32)
    rng = np.random.default_rng(8)
    phi = np.deg2rad(lat.astype(np.float64))[None, :, None]; lam = np.deg2rad(lon.astype(np.float64))[None, None, :]
    amp = (1 + np.arange(nl) / nl)[:, None, None]
    pv = (amp * (np.sin(phi) + 0.2 * np.cos(3 * lam + amp) * np.cos(phi) ** 2) * 5e-6
          + 1e-8 * rng.standard_normal((nl, ny, nx))).astype(np.float32)
    pv[3, 100:104, 50:60] = np.nan
    dA2 = O.cell_area(lat, lon)
    g = np.stack([O.grad2_sphere(pv[l], lat, lon) for l in range(nl)]).astype(np.float32)
    c3 = {'level': lev, 'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(pv, ('level', 'latitude', 'longitude'), c3, 'pv')
    grdS = xa.DataArray(g, ('level', 'latitude', 'longitude'), c3, 'grdSpv')
    dA = xa.DataArray(dA2, ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'rA')
    mask = xa.DataArray(np.ones((ny, nx), np.float32), ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    ctr = cm.cal_contours(N)
    assert ctr.dims == ('level', 'contour') and ctr.values.shape == (nl, N) and ctr.values.dtype == np.float32
    table = cm.cal_area_eqCoord_table_hist(mask)
    area = cm.cal_integral_within_contours_hist(ctr).rename('intArea')
    intgrdS = cm.cal_integral_within_contours_hist(ctr, integrand=grdS).rename('intgrdS')
    latEq = table.lookup_coordinates(area).rename('latEq')
    Lmin = xa.latitude_lengths_at(latEq).rename('Lmin')
    dintSdA = cm.cal_gradient_wrt_area(intgrdS, area).rename('dintSdA')
    dqdA = cm.cal_gradient_wrt_area(ctr, area).rename('dqdA')
    Leq2 = cm.cal_sqared_equivalent_length(dintSdA, dqdA).rename('Leq2')
    nkeff = cm.cal_normalized_Keff(Leq2, Lmin).rename('nkeff')
    preLats = np.linspace(-90, 90, 181).astype(np.float32)
    ds = cm.interp_to_dataset(preLats, latEq, [ctr, area, nkeff])
    assert area.dims == ('level', 'contour') and ds['nkeff'].dims == ('level', 'new') and ds['nkeff'].shape == (nl, 181)
    a2 = cm.cal_integral_within_contours(ctr)                      # notebook 1 cell 5: the conditional API
    for l in range(nl):
        r = O.keff_pipeline(pv[l], dA2, lat, N, grdS=g[l], increase=True, lt=True, dtype=np.float32, preLats=preLats)
        assert np.array_equal(ctr.values[l], r['ctr'])
        assert rel(area.values[l], r['area']) < TIGHT and rel(intgrdS.values[l], r['intgrdS']) < 1e-6   # f32 products
        assert rel(latEq.values[l], r['latEq']) < 1e-9
        ok = r['Lmin'] > LMIN_FLOOR
        assert rel(nkeff.values[l][ok], r['nkeff'][ok]) < RTOL
        assert rel(ds['intArea'].values[l], r['area_eq']) < 1e-9
        assert rel(a2.values[l], O.cal_integral_within_contours(pv[l], r['ctr'], dA2, None, True)) < TIGHT
    # the fused pipeline on the whole stack (per-level contours, f32 tracer + f32 grdS)
    dsk = cm.keff(N, table, grdS=grdS, preY=preLats)
    assert dsk['nkeff'].dims == ('level', 'contour') and dsk['area_eq'].dims == ('level', 'new')
    assert np.array_equal(dsk['ctr'].values, ctr.values.astype(np.float64))
    assert rel(dsk['area'].values, area.values) < 1e-13


def test_native_rccl_single_rank(ctx):
    """xc_comm_*: RCCL communicator of one rank (all this box has); the N > 1 logic is covered on gloo"""
    uid = ctx.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(1, 0, uid)
    x = np.arange(1000, dtype=np.float64)
    a = ctx.to_device(x)
    b = ctx.alloc(x.nbytes)
    ctx.comm_allgather(a.ptr, b.ptr, x.nbytes)
    ctx.sync()
    assert np.array_equal(b.download((1000,), np.float64), x)
    ctx.comm_finalize()
    a.free(); b.free()


def test_misaligned_device_pointers_fall_back(ctx):
    """a slab that starts 8 bytes into a device buffer: the 16-byte vector loads must not be used"""
    import ctypes as C
    from xcontour_amd import _native as nat
    rng = np.random.default_rng(4)
    ny, nx = 33, 130
    x = rng.standard_normal((1, ny, nx))
    ed = np.linspace(-3, 3, 42)
    buf = ctx.alloc(x.nbytes + 64)
    ctx._check(ctx.lib.xc_memcpy_h2d(ctx.handle, buf.ptr + 8, x.ctypes.data, x.nbytes))
    de = ctx.to_device(ed)
    dc = ctx.alloc(41 * 8)
    d = nat.HistDesc()
    d.q, d.q_dtype, d.nslab, d.ny, d.nx = buf.ptr + 8, nat.XC_F64, 1, ny, nx
    d.edges, d.nedge, d.last_closed, d.dA_rank, d.lt = de.ptr, 42, 1, nat.XC_DA_NONE, 1
    d.counts = dc.ptr
    ctx._check(ctx.lib.xc_hist_dev(ctx.handle, C.byref(d)))
    cnt = dc.download((41,), np.uint64)
    _, c = O.weighted_histogram(x[0], ed, None, 'numpy')
    assert np.array_equal(cnt.astype(np.int64), c)


# ---------------------------------------------------------------- K9 box-counting contour crossing
@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('stride,mode', [(1, 'edge'), (2, 'wrap'), (3, 'constant'), (2, 'reflect'), (5, 'symmetric'),
                                         (7, 'wrap'), (13, 'edge'), (31, 'constant'), (63, 'reflect'), (64, 'edge')])
def test_crossing_random_vs_oracle(ctx, dt, stride, mode):
    """xc_crossing against the oracle: box counts exact, lengths to summation order; NaN cells,
    NaN / negative areas, per-slab contours, padding modes, Jn < In and Jn > In shapes."""
    rng = np.random.default_rng(stride * 10 + len(mode))
    for (ny, nx) in [(37, 301), (300, 41), (64, 64)]:
        q = rng.standard_normal((3, ny, nx)).astype(dt)
        q[rng.random(q.shape) < 0.05] = np.nan
        q[1, :4, :] = np.nan                                          # boxes with no valid corner
        area = (rng.random((ny, nx)) * 9 + 1).astype(dt)
        area[2, 3] = np.nan; area[5, 1] = -1.0; area[7, 2] = np.inf      # skipped / skipped / an infinite length at the crossed levels
        cs = np.sort(rng.standard_normal((3, 17)), axis=1)
        cs[0, 3] = cs[0, 4]                                           # duplicated level
        pad = stride + 2                                              # max(stride list) > stride
        for full in (False, True):
            lens, cnts = ctx.crossing(q, cs, area, stride=stride, pad_x=pad, pad_mode=mode, full_width=full)
            for s in range(3):
                ol, oc = O.contour_crossing(O.pad_x(q[s], pad, mode), cs[s], O.pad_x(area, pad, mode), stride, full)
                assert np.array_equal(cnts[s].astype(np.int64), oc)
                assert rel(lens[s], ol) < 1e-13
    # shared contours, per-slab area, no padding (grids without an 'X' dim)
    q = rng.standard_normal((2, 50, 70)).astype(dt)
    a3 = rng.random((2, 50, 70)) + 0.5
    c1 = np.linspace(-2, 2, 9)
    lens, cnts = ctx.crossing(q, c1, a3, stride=stride, pad_x=0)
    for s in range(2):
        ol, oc = O.contour_crossing(q[s], c1, a3[s], stride)
        assert np.array_equal(cnts[s].astype(np.int64), oc) and rel(lens[s], ol) < 1e-13


@pytest.mark.parametrize('adt', [np.float32, np.float64])
def test_crossing_stack_sharing_one_area_plane(ctx, adt):
    """>= 8 slabs with a shared area plane take the square roots once (k_cross_weights); equally spaced levels go through the
    difference-array accumulation, other levels through the scan: counts exact, lengths to summation order, per level"""
    rng = np.random.default_rng(77)
    ny, nx, S = 61, 200, 9
    lat = np.linspace(-1, 1, ny)
    q = lat[None, :, None] + 0.15 * rng.standard_normal((S, ny, nx))
    q[rng.random(q.shape) < 0.02] = np.nan
    area = (rng.random((ny, nx)) * 9 + 1).astype(adt)
    area[2, 3] = np.nan; area[5, 1] = -1.0; area[7, 2] = np.inf; area[9, 9] = 0.0
    for levels in (np.linspace(-1.2, 1.2, 33), np.sort(rng.uniform(-1.2, 1.2, 33))):
        for stride, mode in ((1, 'wrap'), (1, 'constant'), (2, 'edge'), (7, 'constant')):
            lens, cnts = ctx.crossing(q, levels, area, stride=stride, pad_x=stride + 1, pad_mode=mode, full_width=True)
            for s in range(S):
                ol, oc = O.contour_crossing(O.pad_x(q[s], stride + 1, mode), levels, O.pad_x(area, stride + 1, mode), stride, True)
                assert np.array_equal(cnts[s].astype(np.int64), oc)
                fin = np.isfinite(ol)
                assert np.array_equal(np.isinf(lens[s]), np.isinf(ol))
                assert np.all(np.abs(lens[s][fin] - ol[fin]) <= 1e-12 * np.maximum(ol[fin], 1e-300) + 0.0)       # per level, and 0 stays exactly 0


def test_crossing_values_on_and_next_to_levels(ctx):
    """equally spaced levels take the arithmetic route (count_below_uniform): corner values exactly ON a level, one ulp either
    side, and levels that are only nearly equally spaced (float32 contours of a 300 +- 1 field: ~1e-3 of a spacing off) must give
    the oracle's counts exactly"""
    rng = np.random.default_rng(5)
    ny, nx, N = 90, 130, 41
    for lev in (np.linspace(-1.0, 1.0, N),                                        # exact in the mean, rounded per level
                np.linspace(299.0, 301.0, N).astype(np.float32).astype(np.float64),   # float32 contours far from zero
                np.linspace(-2e-4, 3e-4, N).astype(np.float32).astype(np.float64)):   # the barotropic magnitude
        span = lev[-1] - lev[0]
        q = lev[0] + span * (np.linspace(-0.05, 1.05, ny)[:, None] + 0.04 * rng.standard_normal((ny, nx)))
        k = rng.integers(0, N, size=(ny, nx))
        on = rng.random((ny, nx))
        q = np.where(on < 0.15, lev[k], q)                                        # exactly on a level
        q = np.where((on >= 0.15) & (on < 0.25), np.nextafter(lev[k], np.inf), q)
        q = np.where((on >= 0.25) & (on < 0.35), np.nextafter(lev[k], -np.inf), q)
        q[3, 4] = np.inf; q[5, 6] = -np.inf; q[7, 8] = np.nan
        area = rng.random((ny, nx)) + 0.5
        for dt in (np.float64, np.float32):
            qq = q.astype(dt)[None]
            lens, cnts = ctx.crossing(qq, lev, area, stride=1, pad_x=1, pad_mode='wrap', full_width=True)
            ol, oc = O.contour_crossing(O.pad_x(qq[0], 1, 'wrap'), lev, O.pad_x(area, 1, 'wrap'), 1, True)
            assert np.array_equal(cnts[0].astype(np.int64), oc)
            assert rel(lens[0], ol) < 1e-13


def test_crossing_literal_loops_small(ctx):
    """the pure-python loop restatement of core.py:1490-1566, contour by contour"""
    rng = np.random.default_rng(5)
    q = rng.standard_normal((1, 11, 23)).astype(np.float32)
    q[0, 4, 5] = np.nan
    area = (rng.random((11, 23)) + 1).astype(np.float32)              # f32 area: sqrt in f32
    cs = np.linspace(-1.5, 1.5, 6)
    for stride in (1, 2, 3):
        lens, cnts = ctx.crossing(q, cs, area, stride=stride, pad_x=3, pad_mode='edge')
        for k, c in enumerate(cs):
            l, n = O.contour_crossing_literal(O.pad_x(q[0], 3, 'edge'), c, O.pad_x(area, 3, 'edge'), stride)
            assert n == int(cnts[0, k]) and abs(lens[0, k] - l) <= 1e-13 * max(l, 1.0)


def test_crossing_rejects_bad_input(ctx):
    from xcontour_amd._native import XContourHipError
    q = np.zeros((1, 8, 8)); a = np.ones((8, 8))
    with pytest.raises(XContourHipError):
        ctx.crossing(q, np.array([1.0, 0.0]), a)                      # not ascending
    with pytest.raises(XContourHipError):
        ctx.crossing(q, np.array([0.0, np.nan]), a)
    with pytest.raises(XContourHipError):
        ctx.crossing(q, np.array([0.0]), a, stride=0)
    lens, cnts = ctx.crossing(q, np.array([0.0]), a, stride=8)        # Jn = 1: no boxes
    assert lens[0, 0] == 0.0 and cnts[0, 0] == 0


def test_facade_contour_crossing_golden(ctx, baro):
    """Contour2D.cal_contour_crossing on the barotropic field: golden fixture + oracle, decreasing levels
    (façade sorts / un-permutes), list of strides sharing one padding, box-count dimension sanity."""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    G = np.load(os.path.join(GOLD, 'baro_crossing_N41.npz'))
    for increase in (True, False):
        cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=increase, lt=True)
        ctr = cm.cal_contours(41)
        res = cm.cal_contour_crossing(ctr, stride=[1, 2, 4], mode='wrap')
        assert isinstance(res, list) and len(res) == 3
        for r, s in zip(res, (1, 2, 4)):
            assert r.dims == ('contour',) and r.dtype == np.float32
            want = O.cal_contour_crossing(q, ctr.values, dA.values, [1, 2, 4], 'wrap')[(1, 2, 4).index(s)]
            assert rel(r.values, want) < 1e-6
            if increase:
                assert rel(r.values, G['len_s%d' % s]) < 1e-6
    single = cm.cal_contour_crossing(ctr, stride=2, mode='wrap')
    assert np.array_equal(single.values, res[1].values)
    # counts: halving the resolution of a smooth contour roughly halves the number of crossed boxes
    lens1, c1 = ctx.crossing(q[None], G['ctr'].astype(np.float64), dA.values, 1, 4, 'wrap')
    lens4, c4 = ctx.crossing(q[None], G['ctr'].astype(np.float64), dA.values, 4, 4, 'wrap')
    mid = slice(10, 30)
    ratio = c1[0, mid].astype(float) / np.maximum(c4[0, mid].astype(float), 1)
    assert np.all(ratio > 2.0) and np.all(ratio < 8.0)
    assert np.array_equal(c1[0].astype(np.int64), G['cnt_sorted_s1'])


def test_more_slabs_than_one_launch_takes(ctx):
    """70 000 tiny slabs: the binding splits at the library's 65 535-slabs-per-launch limit"""
    rng = np.random.default_rng(0)
    S, ny, nx = 70000, 8, 16
    q = rng.standard_normal((S, ny, nx)).astype(np.float32)
    mm = ctx.minmax(q)
    assert np.array_equal(mm[:, 0], q.reshape(S, -1).min(1)) and np.array_equal(mm[:, 1], q.reshape(S, -1).max(1))
    ed = np.linspace(-3, 3, 12)
    out = ctx.hist(q, ed, dA=np.ones((ny, nx)), want=('counts',))
    assert out['counts'].shape == (S, 11)
    for s in (0, 65534, 65535, S - 1):
        assert np.array_equal(out['counts'][s].astype(np.int64), np.histogram(q[s], bins=ed)[0])


def test_keff_plan_reuse_is_not_stale(ctx, baro):
    """Contour2D.keff keeps its device plan between calls: a changed tracer must give new results, a changed dA
    a new plan, an unchanged call the same values."""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    kw = dict(dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    cm = xa.Contour2D(tr, dA, **kw)
    table = cm.cal_area_eqCoord_table_hist(xa.DataArray(np.ones_like(q), tr.dims, tr.coords, 'mask'))
    a = cm.keff(61, table, lat=lat, lon=lon)
    b = cm.keff(61, table, lat=lat, lon=lon)
    assert len(cm._keff_plans) == 1 and np.array_equal(a['ctr'].values, b['ctr'].values)
    for k in ('area', 'intgrdS'):                       # LDS float atomics: the summation order varies run to run
        assert rel(a[k].values, b[k].values) < 1e-13
    cm.tracer = xa.DataArray(q[::-1].copy() * 2, tr.dims, tr.coords, 'absolute_vorticity')     # same shapes: plan reused
    c = cm.keff(61, table, lat=lat, lon=lon)
    assert len(cm._keff_plans) == 1 and not np.array_equal(a['ctr'].values, c['ctr'].values)
    fresh = xa.Contour2D(cm.tracer, dA, **kw).keff(61, table, lat=lat, lon=lon)
    assert np.array_equal(c['ctr'].values, fresh['ctr'].values)
    for k in ('area', 'intgrdS'):
        assert rel(c[k].values, fresh[k].values) < 1e-13
    cm.dA = xa.DataArray(dA.values * 2.0, dA.dims, dA.coords, 'rA')                             # other metric: new plan
    d = cm.keff(61, table, lat=lat, lon=lon)
    assert len(cm._keff_plans) == 2 and rel(d['area'].values, 2.0 * c['area'].values) < 1e-14
    cm.close()
    assert '_keff_plans' not in cm.__dict__
    # a stack larger than max_batch_bytes goes through in batches of whole slabs (5 slabs, 2 per batch)
    st = np.stack([q * (1 + 0.1 * k) for k in range(5)])
    c3 = dict(tr.coords); c3['level'] = np.arange(5.0)
    cm5 = xa.Contour2D(xa.DataArray(st, ('level',) + tr.dims, c3, 'absolute_vorticity'), dA, **kw)
    one = cm5.keff(61, table, lat=lat, lon=lon)
    two = cm5.keff(61, table, lat=lat, lon=lon, max_batch_bytes=2 * q.nbytes)
    assert one['ctr'].shape == (5, 61) and np.array_equal(one['ctr'].values, two['ctr'].values)
    assert rel(two['area'].values, one['area'].values) < 1e-13 and rel(two['nkeff'].values, one['nkeff'].values) < 1e-8


def test_fractal_call_sequence(ctx, baro):
    """tests/test_fractal.py:30-75: `analysis.cal_contour_crossing(ctr, stride=[1, 2, 4, 8, 16, 32], mode='edge')`
    with N = 121 on the barotropic field (strides 8 / 16 / 32 run the run-time-stride kernel)."""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    G = np.load(os.path.join(GOLD, 'baro_fractal_N121.npz'))
    analysis = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    ctr = analysis.cal_contours(121)
    assert np.array_equal(ctr.values, G['ctr'])
    strides = [1, 2, 4, 8, 16, 32]
    bclens = analysis.cal_contour_crossing(ctr, stride=strides, mode='edge')
    for b, s in zip(bclens, strides):
        assert b.dtype == np.float32 and rel(b.values, G['bclens%d' % s]) < 1e-6
    # box-counting: fewer, larger boxes are crossed as the stride grows, and the measured length shrinks slowly
    tot = [float(b.values.sum()) for b in bclens]
    assert all(t > 0 for t in tot) and tot[0] > tot[-1]


def test_sorted_profile_over_a_stack(ctx, baro):
    """cal_sorted_profile loops the exact sort over the leading dims like every other method"""
    import xcontour_amd as xa
    tr, dA, q, lat, lon = _baro_da(xa, baro)
    kw = dict(dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    table = xa.Contour2D(tr, dA, **kw).cal_area_eqCoord_table_hist(xa.DataArray(np.ones_like(q), tr.dims, tr.coords, 'mask'))
    st = np.stack([q, q * 1.5 + 1e-5, q[:, ::-1].copy()])
    c3 = dict(tr.coords); c3['time'] = np.arange(3.0)
    cm = xa.Contour2D(xa.DataArray(st, ('time',) + tr.dims, c3, 'absolute_vorticity'), dA, **kw)
    Q, sorted_ = cm.cal_sorted_profile(table, return_sorted=True)
    assert Q.dims == ('time', 'latitude') and Q.shape == (3, 256) and len(sorted_) == 3
    for k in range(3):
        one = xa.Contour2D(xa.DataArray(st[k], tr.dims, tr.coords, 'absolute_vorticity'), dA, **kw).cal_sorted_profile(table)
        assert np.array_equal(Q.values[k], one.values)
        assert np.array_equal(sorted_[k], np.sort(st[k].ravel().astype(np.float64), kind='stable'))
    assert np.array_equal(Q.values[0], Q.values[2])               # a zonal flip changes no area


@pytest.mark.parametrize('variant', [0, 1])
def test_lwa_band_skipping_is_invisible(ctx, variant):
    """K7 never loads rows whose extrema rule out a contribution: fields built to stress that test -- a sharp
    front (narrow bands), a non-monotone Q, rows of NaN, +-inf cells, constant rows, f32 and f64, both variants,
    every (increase, part), sizes that are not multiples of the 8-row batch or the 64-lane band search."""
    rng = np.random.default_rng(3 + variant)
    for (ny, nx, dt) in [(131, 70, np.float64), (67, 130, np.float32), (9, 5, np.float64)]:
        coord = np.linspace(-80, 80, ny)
        front = np.tanh((coord[:, None] - 10 * np.sin(np.linspace(0, 6.28, nx))[None, :]) / 4.0)
        q = (front + 0.01 * rng.standard_normal((ny, nx))).astype(dt)[None]
        q = np.concatenate([q, q[:, ::-1]], axis=0)                      # second slab: reversed in y
        q[0, ny // 3, :] = np.nan                                        # a row of NaN
        q[0, ny // 2, 2] = np.inf; q[1, 1, 1] = -np.inf
        q[1, ny // 4, :] = 0.25                                          # a constant row
        Q = np.stack([np.sort(q[0, :, 0].astype(np.float64)), rng.standard_normal(ny)])   # sorted / non-monotone
        Q[0][np.isnan(Q[0])] = 0.0
        dA = (rng.random((ny, nx)) + 0.5)
        for increase in (True, False):
            for part, pc in (('all', 0), ('upper', 1), ('lower', 2)):
                out, _ = ctx.lwa(q, Q, coord, dA, dA.max(), M=None, increase=increase, part=pc, variant=variant)
                fn = O.cal_local_wave_activity2 if variant else O.cal_local_wave_activity
                for s in range(2):
                    with np.errstate(invalid='ignore'):                   # inf * 0 inside the oracle's products
                        ref = fn(q[s], Q[s], coord, dA, increase, part)
                    assert np.array_equal(out[s], ref, equal_nan=True), (ny, nx, increase, part, s)


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_batched_sort_equals_per_slab_oracle(ctx, dt):
    """xc_sort_profile_batch: a stack in one set of launches (segmented sort) -- per-slab masks, per-slab dA, NaNs,
    a slab without a single valid cell, targets beyond the total area; every slab against the oracle."""
    rng = np.random.default_rng(21)
    S, ny, nx = 5, 37, 131                                   # 4847 cells: two tiles, the second one ragged
    q = rng.standard_normal((S, ny, nx)).astype(dt)
    q[rng.random(q.shape) < 0.03] = np.nan
    q[1, :, :7] = 0.5                                        # ties
    mask = (rng.random((S, ny, nx)) < 0.9).astype(np.float64)
    mask[3] = 0.0                                            # nothing valid in slab 3
    dA = rng.random((S, ny, nx)) + 0.1
    tbl = np.linspace(0, dA[0].sum() * 1.2, 23)
    coord = np.linspace(-1.0, 1.0, 23)
    r = ctx.sort_profile(q, dA=dA, mask=mask, targets=tbl, tbl=tbl, coord=coord, want_sorted=True, want_acum=True)
    assert r['Q'].shape == (S, 23) and r['nvalid'].shape == (S,)
    for s in range(S):
        ok = ~np.isnan(q[s]) & (mask[s] == 1)
        assert int(r['nvalid'][s]) == int(ok.sum())
        if ok.sum() == 0:
            assert np.isnan(r['Q'][s]).all()
            continue
        Qo, xs, ac = O.sorted_profile(q[s], dA[s], tbl, mask[s])
        n = len(xs)
        assert np.array_equal(r['q_sorted'][s][:n], xs.astype(np.float64))
        assert rel(r['acum'][s][:n], ac) < 1e-12
        lo, hi = O.sorted_profile_brackets(ac, tbl)                                       # the documented tie rule
        assert (r['Q'][s] >= xs[lo]).all() and (r['Q'][s] <= xs[hi]).all()
        assert np.array_equal(r['Q'][s][lo == hi], Qo[lo == hi].astype(np.float64))
        assert abs(r['bpe'][s] - O.bpe_integral(q[s], dA[s], tbl, coord, mask[s])) <= 1e-11 * abs(dA[s].sum())
    # shared mask / shared dA variants agree with the per-slab call
    r2 = ctx.sort_profile(q, dA=dA[0], mask=mask[0], targets=tbl)
    for s in range(S):
        one = ctx.sort_profile(q[s], dA=dA[0], mask=mask[0], targets=tbl)
        assert np.array_equal(r2['Q'][s], one['Q'], equal_nan=True) and int(r2['nvalid'][s]) == one['nvalid']


def test_readme_example_runs(ctx):
    """the python block of README.md, verbatim"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = re.search(r"```python\n(.*?)```", open(os.path.join(root, 'README.md')).read(), re.S).group(1)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        ns = {}
        exec(code, ns)
    finally:
        os.chdir(cwd)
    assert ns['lwa'].shape == (256, 512) and ns['Qx'].shape == (256,) and len(ns['bc']) == 3


def test_c_client(ctx, tmp_path):
    """tests/capi_smoke.c: a plain C program (gcc, no HIP headers) drives the C ABI the way a cgo / JNI host would"""
    import subprocess
    from xcontour_amd import _native as nat
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'capi_smoke')
    libdir = os.path.dirname(nat.LIB_PATH)
    subprocess.run(['gcc', '-O1', '-std=c11', '-o', exe, os.path.join(root, 'tests', 'capi_smoke.c'),
                    '-L' + libdir, '-lxcontour_hip', '-Wl,-rpath,' + libdir, '-lm'], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'capi_smoke ok' in r.stdout, r.stdout + r.stderr


def test_two_contexts_from_two_threads(ctx):
    """one context per thread, driven concurrently (ctypes releases the GIL): same results as one after the other"""
    import threading
    from xcontour_amd import _native as nat
    rng = np.random.default_rng(9)
    q = rng.standard_normal((4, 200, 300))
    ed = np.linspace(-4, 4, 42)
    dA = rng.random((200, 300)) + 0.5
    ref = ctx.hist(q, ed, dA=dA, want=('counts', 'cdf'))
    out, errs = [None, None], []

    def work(i):
        try:
            c = nat.Context(0)
            for _ in range(5):
                out[i] = c.hist(q, ed, dA=dA, want=('counts', 'cdf'))
                c.minmax(q)
            c.close()
        except Exception as e:            # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for o in out:
        assert np.array_equal(o['counts'], ref['counts']) and rel(o['cdf'], ref['cdf']) < 1e-13


def test_errors_are_reported_not_fatal(ctx):
    """out-of-memory and bad arguments come back as XContourHipError with a message; the context stays usable"""
    from xcontour_amd._native import XContourHipError
    with pytest.raises(XContourHipError) as e:
        ctx.alloc(1 << 46)                                   # 64 TiB
    assert str(e.value)
    q = np.zeros((1, 8, 8))
    with pytest.raises(XContourHipError):
        ctx.hist(q, np.array([0.0, 1.0]), dA=np.ones((8, 9)))            # dA shape mismatch
    with pytest.raises(XContourHipError):
        ctx.crossing(q, np.array([0.0]), np.ones((8, 8)), pad_mode='mirror')
    out = ctx.hist(q, np.array([-1.0, 1.0]), dA=np.ones((8, 8)), want=('counts',))   # still alive
    assert int(out['counts'][0, 0]) == 64


def test_keff_thousands_of_contours(ctx, baro):
    """N = 2000 / 4500: the epilogue's work arrays no longer fit the LDS and move to global memory
    (the histogram pass itself takes up to ~5000 levels with two weight channels)"""
    q, lat, lon = baro
    dA = O.cell_area(lat, lon)
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import table_from_rowsums, last_row_included
    tbl = table_from_rowsums(ctx.rowsum(None, dA, 256, 512), True, last_row_included(lat))
    for N in (2000, 4500):
        plan = KeffPlan(ctx, 1, 256, 512, N, np.float32, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                        increase=True, lt=True)
        plan.set_q(q[None])
        plan.run()
        r = plan.fetch()
        plan.free()
        o = O.keff_pipeline(q, dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float64)
        assert np.array_equal(r['ctr'][0], o['ctr'])
        assert rel(r['area'][0], o['area']) < 1e-12 and rel(r['intgrdS'][0], o['intgrdS']) < 1e-11
        assert rel(r['latEq'][0], o['latEq']) < 1e-9


def test_differential_fuzz_short(ctx):
    """8 s of tools/gpu_fuzz.py (random shapes / dtypes / flags / NaNs; HIP path vs oracle for hist, the fused pipeline,
    crossing, LWA, sort and the facade call sequence); the long runs are logged in profiles/r01_notes.md"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('gpu_fuzz', os.path.join(root, 'tools', 'gpu_fuzz.py'))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    n, checked = fz.run(8.0, seed=2024)
    assert sum(n.values()) > 200 and all(v > 0 for v in checked.values()) and len(checked) >= 6
