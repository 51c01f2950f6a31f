"""Pin the oracle: SURVEY.md 8(c) known-answer vectors on the reference's bundled
Data/barotropic_vorticity.nc, numpy's own histogram, and the committed golden npz."""
import hashlib
import os

import numpy as np
import pytest

import xcontour_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_fixture_bytes(baro):
    q, lat, lon = baro
    assert q.dtype == np.float32 and q.shape == (256, 512)
    assert hashlib.sha256(q.tobytes()).hexdigest().startswith('c8d30d7acd84c77a')
    assert hashlib.sha256(lat.tobytes()).hexdigest().startswith('49888387fcbea5e6')
    assert q.min() == np.float32(-1.5368119e-4) and q.max() == np.float32(1.7907852e-4)


@pytest.mark.parametrize('N,first5,last5,sha', [
    (121, [0, 7058, 3160, 2646, 2468], [190, 222, 188, 194, 161], 'fe188fcc7af47e9d'),
    (201, [0, 5396, 2364, 1886, 1670], [122, 110, 108, 122, 81], '5c785dbfe747de26'),
])
def test_survey_known_counts(baro, N, first5, last5, sha):
    q, lat, lon = baro
    dA = O.cell_area(lat, lon)
    ctr = O.cal_contours(q, N, True, np.float32)
    assert ctr.dtype == np.float32 and ctr[0] == q.min()
    assert ctr[-1] == np.float32(0.0001790785) and ctr[-1] != q.max()
    _, cnt = O.cal_integral_within_contours_hist(q, ctr, dA, None, True, 'numpy', return_counts=True)
    assert cnt.sum() == 131071                                               # numpy rule: max cell excluded
    assert list(cnt[:5]) == first5 and list(cnt[-5:]) == last5
    assert hashlib.sha256(cnt.astype(np.int64).tobytes()).hexdigest().startswith(sha)
    # numpy's own histogram has the same semantics
    e, _ = O.hist_edges(ctr)
    h, _ = np.histogram(q.ravel(), bins=e)
    assert np.array_equal(h, cnt)
    # the xhistogram (+1e-8 in the f32 edge dtype: 1.79e-4 absorbs it) rule keeps the max cell -- the DEFAULT
    _, c2 = O.cal_integral_within_contours_hist(q, ctr, dA, None, True, return_counts=True)
    assert c2.sum() == 131072 and np.array_equal(c2[:-1], cnt[:-1]) and c2[-1] == cnt[-1] + 1


def test_survey_known_keff_and_lwa(baro):
    q, lat, lon = baro
    dA = O.cell_area(lat, lon)
    assert abs(dA.sum() / (4 * np.pi * O.Rearth ** 2) - 1) < 1e-14
    r = O.keff_pipeline(q, dA, lat, 121, lon=lon, preLats=lat, right_edge='numpy')     # SURVEY's numbers: numpy rule
    assert abs(r['area'][-1] / 5.1009325369688875e14 - 1) < 1e-12
    assert abs(r['area'][-1] / dA.sum() - 0.9999936) < 1e-7
    nk = r['nkeff']
    assert np.isfinite(nk).all()
    assert abs(np.nanmin(nk) - 0.967) < 1e-3 and abs(np.nanmedian(nk) - 2.59) < 1e-2 and abs(np.nanmax(nk) - 168.2) < 0.1
    assert abs(r['latEq'][1] + 79.78) < 1e-2 and abs(r['latEq'][-2] - 86.64) < 1e-2
    assert (np.diff(r['area']) > 0).all()
    Q = r['ctr_eq']
    assert (np.diff(Q) >= 0).all()
    dy = np.gradient(np.deg2rad(lat.astype(np.float64))) * O.Rearth
    lwa = O.cal_local_wave_activity(q, Q, lat, dA, True, 'all', metric=dy)
    assert lwa.min() >= -1e-12 and abs(lwa.max() - 28.921) < 1e-3
    assert abs(lat[np.argmax(lwa.mean(1))] - 29.82) < 0.01
    for j, v in zip((37, 125, 170, 213), (0.285, 3.224, 9.943, 7.817)):
        assert abs(lwa[j].mean() - v) < 1e-3
    r2 = O.keff_pipeline(q, dA, lat, 201, lon=lon, right_edge='numpy')
    nk = r2['nkeff']
    assert abs(np.nanmin(nk) - 0.902) < 1e-3 and abs(np.nanmedian(nk) - 2.35) < 1e-2 and abs(np.nanmax(nk) - 98.1) < 0.1


def test_hist_vs_strict_twin(baro):
    """tests/test_hist.py:132-167 overlay: the two APIs agree except on the closed last bin."""
    q, lat, lon = baro
    dA = O.cell_area(lat, lon)
    g2 = O.grad2_sphere(q, lat, lon)
    for N in (121, 201):
        ctr = O.cal_contours(q, N, True, np.float32)
        for integ in (None, g2):
            a = O.cal_integral_within_contours_hist(q, ctr, dA, integ, True, 'numpy')
            b = O.cal_integral_within_contours(q, ctr, dA, integ, True)
            assert np.max(np.abs(a[1:] - b[1:]) / b[1:]) < 1e-12
            # default (xhistogram) rule: the last level additionally holds the cells in [ctr[-1], ctr[-1] + 1e-8)
            ax = O.cal_integral_within_contours_hist(q, ctr, dA, integ, True)
            assert np.array_equal(ax[:-1], a[:-1]) and ax[-1] > a[-1]


def test_histogram_semantics_small():
    """SURVEY A3."""
    x = np.array([np.nan, -1, 0, .5, 1, 2.999, 3, 3 + 5e-9, 3.0001])
    s, c = O.weighted_histogram(x, np.array([0., 1, 2, 3]), right_edge='numpy')
    assert list(c) == [2, 1, 2]          # last bin closed: 2.999, 3
    s, c = O.weighted_histogram(x, np.array([0., 1, 2, 3]), right_edge='xhistogram')
    assert list(c) == [2, 1, 3]          # [2, 3+1e-8): 2.999, 3, 3+5e-9
    # float32 edges of magnitude >= 0.25 absorb the 1e-8: the last bin is then half-open
    e32 = np.array([0., 1, 2, 3], dtype=np.float32)
    s, c = O.weighted_histogram(x.astype(np.float32), e32, right_edge='xhistogram')
    assert list(c) == [2, 1, 1]


def test_table_last_row_rule_f32_vs_f64_coordinates(baro):
    """under the xhistogram rule the last row of the A(Yeq) table drops out iff
    `coord[-1] + 1e-8 == coord[-1]` in the coordinate dtype -- true for the float32 latitudes of the
    reference's own barotropic_vorticity.nc, false for float64 ones."""
    q, lat, lon = baro
    assert lat.dtype == np.float32 and (lat[-1:] + 1e-8)[0] == lat[-1]
    dA = O.cell_area(lat, lon)
    tot = dA.sum()
    t_x, _ = O.cal_area_eqCoord_table_hist(np.ones_like(q), dA, lat, True, True)              # default rule
    t_n, _ = O.cal_area_eqCoord_table_hist(np.ones_like(q), dA, lat, True, True, 'numpy')
    t_64, _ = O.cal_area_eqCoord_table_hist(np.ones_like(q), dA, lat.astype(np.float64), True, True)
    assert abs(t_n[-1] / tot - 1) < 1e-14 and abs(t_64[-1] / tot - 1) < 1e-14
    assert abs(t_x[-1] / tot - 0.99994) < 1e-5 and np.array_equal(t_x[:-1], t_n[:-1])
    assert abs(t_x[-1] - (tot - dA[-1].sum())) < 1e-3 * dA[-1].sum()
    r = O.keff_pipeline(q, dA, lat, 121, lon=lon)                                              # default rule
    assert r['counts'].sum() == 131072 and abs(r['latEq'][-1] - 89.4631) < 1e-4 and abs(r['nkeff'][-1] - 104.963) < 1e-2
    r = O.keff_pipeline(q, dA, lat, 121, lon=lon, right_edge='numpy')
    assert r['counts'].sum() == 131071 and abs(r['latEq'][-1] - 89.4399) < 1e-4 and abs(r['nkeff'][-1] - 95.8586) < 1e-2


def test_contours_at_oracle_golden(baro):
    """f3 (core.py:269-360): committed vectors == oracle; q(Y) is monotone and brackets the tracer range"""
    q, lat, lon = baro
    dA = O.cell_area(lat, lon)
    g = np.load(os.path.join(GOLD, 'baro_contours_at.npz'))
    pre = g['predef']
    for rule in ('xhistogram', 'numpy'):
        for inc in (True, False):
            for lt in (True, False):
                tbl, cs = O.cal_area_eqCoord_table_hist(np.ones_like(q), dA, lat, inc, lt, rule)
                for hist in (True, False):
                    qi, ctr = O.cal_contours_at(q, pre, tbl, cs, dA, inc, lt, np.float32, hist, rule)
                    assert qi.dtype == np.float64 and qi.shape == pre.shape
                    assert np.array_equal(qi, g['%s_inc%d_lt%d_%s' % (rule, inc, lt, 'hist' if hist else 'cond')])
                    assert qi.min() >= q.min() and qi.max() <= q.max()
    with pytest.raises(Exception, match='predef should be a 1D array'):
        O.cal_contours_at(q, np.zeros((2, 2)), tbl, cs, dA)


def test_table_rules():
    """the degenerate histogram of the coordinate field."""
    J, nx = 7, 5
    rng = np.random.default_rng(1)
    dA = rng.random((J, nx)) + 0.1
    mask = np.ones((J, nx)); mask[2, 1] = 0; mask[5, :] = 0
    r = np.where(mask == 1, dA, 0).sum(1)
    for coord in (np.linspace(-60, 60, J), np.linspace(60, -60, J)):
        ra = r if coord[-1] > coord[0] else r[::-1]
        for inc in (True, False):
            for lt in (True, False):
                tbl, cs = O.cal_area_eqCoord_table_hist(mask, dA, coord, inc, lt)
                ylt = lt if (inc == (coord[-1] > coord[0])) else (not lt)
                assert (np.diff(cs) > 0).all()
                if ylt:
                    exp = np.concatenate(([0], np.cumsum(ra)[:-1])); exp[-1] = ra.sum()
                else:
                    exp = np.array([ra[j:].sum() for j in range(J)]); exp[-1] = 0
                assert np.allclose(tbl, exp, rtol=1e-13, atol=1e-13)


def test_golden_files_match_oracle(baro):
    q, lat, lon = baro
    dA = O.cell_area(lat, lon)
    for rule, sfx in (('xhistogram', ''), ('numpy', '_numpy')):
        for N in (121, 201):
            g = np.load(os.path.join(GOLD, 'baro_keff_N%d%s.npz' % (N, sfx)))
            r = O.keff_pipeline(q, dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float32, preLats=lat,
                                right_edge=rule)
            for k in g.files:
                assert np.array_equal(g[k], r[k], equal_nan=True), k


def test_crossing_oracle_vectorised_equals_literal_loops():
    """K9 oracle: the all-contours min/max formulation against the loop-for-loop restatement of
    core.py:1490-1566 (small shapes, every pad mode, NaN cells and areas, Jn > In)."""
    rng = np.random.default_rng(1)
    for (ny, nx, s, mode) in [(9, 14, 1, 'edge'), (12, 20, 2, 'wrap'), (13, 17, 3, 'constant'), (20, 9, 2, 'edge'),
                              (7, 30, 1, 'reflect'), (10, 12, 2, 'symmetric')]:
        q = rng.standard_normal((ny, nx))
        q[rng.random((ny, nx)) < 0.1] = np.nan
        a = rng.random((ny, nx)) * 10
        a[0, 0] = np.nan
        dp, ap = O.pad_x(q, s + 1, mode), O.pad_x(a, s + 1, mode)
        cs = np.linspace(-2, 2, 7)
        L, C = O.contour_crossing(dp, cs, ap, s)
        for k, c in enumerate(cs):
            l, n = O.contour_crossing_literal(dp, c, ap, s)
            assert n == C[k] and abs(l - L[k]) <= 1e-13 * max(abs(l), 1.0)
    # coarse shape uses round-half-even (np.round): 7 / 2 = 3.5 -> 4, 5 / 2 = 2.5 -> 2
    assert O.crossing_shape(7, 5, 2) == (4, 2)


def test_crossing_golden_fixture():
    G = np.load(os.path.join(GOLD, 'baro_crossing_N41.npz'))
    q = np.load(os.path.join(GOLD, 'baro_q.npy'))
    lat, lon = np.load(os.path.join(GOLD, 'baro_lat.npy')), np.load(os.path.join(GOLD, 'baro_lon.npy'))
    dA = O.cell_area(lat, lon)
    res = O.cal_contour_crossing(q, G['ctr'], dA, [1, 2, 4], 'wrap')
    for r, s in zip(res, (1, 2, 4)):
        assert r.dtype == np.float32 and np.array_equal(r, G['len_s%d' % s])
    # the reference scans only Jn-1 = 255 of the In-1 = 515 box columns (core.py:1521): the literal
    # result is smaller than the full-width one
    full = O.cal_contour_crossing(q, G['ctr'], dA, 1, 'wrap', full_width=True)
    assert np.all(full >= res[0]) and full.sum() > 1.5 * res[0].sum()
    assert int(G['cnt_s1'][0]) == 15 and int(G['cnt_s1'][1]) == 291


def test_fractal_golden_fixture():
    """tests/test_fractal.py's cal_contour_crossing call (N = 121, strides 1..32, mode='edge') through the oracle"""
    G = np.load(os.path.join(GOLD, 'baro_fractal_N121.npz'))
    q = np.load(os.path.join(GOLD, 'baro_q.npy'))
    dA = O.cell_area(np.load(os.path.join(GOLD, 'baro_lat.npy')), np.load(os.path.join(GOLD, 'baro_lon.npy')))
    ctr = O.cal_contours(q, 121, True, np.float32)
    assert np.array_equal(ctr, G['ctr'])
    for s, r in zip(G['strides'], O.cal_contour_crossing(q, ctr, dA, [int(t) for t in G['strides']], 'edge')):
        assert np.array_equal(r, G['bclens%d' % s])


def nb1_rows():
    """the 36 contour values the reference itself printed in notebooks/1.Keff_atmos.ipynb (cell 3)"""
    import json
    d = json.load(open(os.path.join(GOLD, 'nb1_ctr_printout.json')))
    return d['N'], {int(k): (np.array(v[0], np.float32), np.array(v[1], np.float32)) for k, v in d['rows'].items()}


def nb1_max_candidates(first, last, N, levels_fn, span=40):
    """PV.nc is not bundled, so a slab's max is unknown -- but ctr[0] IS its min (core.py:229-239) and the max lies within
    a few ulp of the printed ctr[-1]: return every float32 max for which `levels_fn(min, max)` reproduces all six
    printed values of the row bit for bit (float32)."""
    mn, c = first[0], last[2]
    for _ in range(span):
        c = np.nextafter(c, np.float32(-1))
    found = []
    for _ in range(2 * span + 1):
        ctr = levels_fn(mn, c)
        if np.array_equal(ctr[:3], first) and np.array_equal(ctr[-3:], last):
            found.append(c)
        c = np.nextafter(c, np.float32(1))
    return found


def test_levels_against_the_reference_notebook_printout():
    """A REFERENCE-HELD known answer for a2 (core.py:205-266): the level arithmetic of the oracle (the tracer-dtype
    difference, the float64 product / sum, the float32 cast) reproduces the reference's own printed contours.  The
    second and third value from each end are fully determined by (min, max); a wrong dtype rule (e.g. all-float32 or
    np.linspace) fails this test for at least one row -- checked below."""
    N, rows = nb1_rows()
    for k, (first, last) in rows.items():
        found = nb1_max_candidates(first, last, N, lambda mn, mx: O.cal_contours(np.array([[mn, mx]], np.float32), N, True, np.float32))
        assert 1 <= len(found) <= 3, (k, found)

    def all_f32(mn, mx):                      # the tempting wrong rule: everything in the tracer dtype
        st = np.float32(1.0 / (N - 1)) * (mx - mn)
        return (st * np.arange(N, dtype=np.float32) + mn).astype(np.float32)
    assert any(len(nb1_max_candidates(f, l, N, all_f32)) == 0 for f, l in rows.values())


def test_deterministic_bin_sums_rule():
    """the build-defined order-free summation rule (oracle.deterministic_bin_sums; the GPU reproduces it bit for bit in
    the -m gpu tests): independent of the order of the cells, within a few ulp of np.bincount on benign weights,
    exact on integers, NaN for a bin with an infinite weight"""
    rng = np.random.default_rng(31)
    x = rng.standard_normal(50000)
    w = rng.standard_normal(50000) * 10.0 ** rng.integers(-6, 6, 50000)
    e = np.linspace(-3, 3, 41)
    a, c = O.weighted_histogram(x, e, w, 'numpy', deterministic=True)
    for _ in range(3):
        p = rng.permutation(len(x))
        b, c2 = O.weighted_histogram(x[p], e, w[p], 'numpy', deterministic=True)
        assert np.array_equal(a.view(np.int64), b.view(np.int64)) and np.array_equal(c, c2)
    f, _ = O.weighted_histogram(x, e, w, 'numpy')
    scale = np.array([np.abs(w[(x >= e[k]) & (x < e[k + 1])]).sum() for k in range(40)])
    assert (np.abs(a - f) <= 2e-14 * scale).all()
    wi = rng.integers(-1000, 1000, 50000).astype(np.float64)
    ai, _ = O.weighted_histogram(x, e, wi, 'numpy', deterministic=True)
    assert np.array_equal(ai, np.array([wi[(x >= e[k]) & (x < e[k + 1])].sum() for k in range(40)]))
    w2 = np.abs(w); w2[np.argmin(np.abs(x))] = np.inf
    an, _ = O.weighted_histogram(x, e, w2, 'numpy', deterministic=True)
    assert np.isnan(an).sum() == 1 and np.isfinite(np.delete(an, np.argmax(np.isnan(an)))).all()
    # round 5, the one-pass rule: the vectorised sums ARE the scalar definition (det_chunks) cell by cell, zero / denormal
    # weights contribute nothing, a 2^100 spread inside one bin keeps both ends exactly, and a narrow window drops what lies under it
    import math
    from fractions import Fraction
    ws = w.copy(); ws[:3] = (0.0, 5e-324, -1e-310)
    top = O.det_window_top(np.abs(ws).max())
    av, _ = O.weighted_histogram(x, e, ws, 'numpy', deterministic=True)
    idx = np.digitize(x, e)
    for k in (1, 7, 20, 40):
        T = sum(O.det_chunks(v, top, 4) for v in ws[idx == k])
        assert av[k - 1] == math.ldexp(float(T), top - 4 * O.DET_LIMB_BITS)
    big = np.array([2.0 ** 100, 3.0, 2.0 ** -20, -(2.0 ** 100)])
    assert O.deterministic_bin_sums(np.ones(4, dtype=int), big, 1)[0] == 3.0 + 2.0 ** -20        # exact: float64 summation in this order gives 0
    odd = np.array([1.0 + 15 * 2.0 ** -52] * 3)                                                     # 53-bit weights are cut to 49 bits first
    assert O.deterministic_bin_sums(np.ones(3, dtype=int), odd, 1)[0] == 3.0
    vals = rng.random(1000) * 10.0 ** rng.integers(-3, 3, 1000)
    exact = sum(Fraction(float(np.ldexp(np.floor(np.ldexp(m, 49)), ex - 49))) for m, ex in (np.frexp(v) for v in vals))
    assert O.deterministic_bin_sums(np.ones(1000, dtype=int), vals, 1)[0] == float(exact)          # = the exact sum of the rounded weights, rounded once
    lost = O.deterministic_bin_sums(np.ones(2, dtype=int), np.array([1.0, 2.0 ** -60]), 1, top=O.det_window_top(1.0), nlimb=1)
    assert lost[0] == 1.0                                                                          # one limb: 2^-60 lies under the window
