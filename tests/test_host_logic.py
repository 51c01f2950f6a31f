"""Host-side logic that needs no GPU: labelled arrays, edge construction, table rules,
O(N) contour-space algebra of the facade against the oracle, slab sharding."""
import os

import numpy as np
import pytest

import xcontour_oracle as O
import xcontour_amd as xa
from xcontour_amd import core
from xcontour_amd.utils import table_from_rowsums, cell_area, grad_metrics
from xcontour_amd.pipeline import shard_slabs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_public_api_surface():
    """reference xcontour/__init__.py:2-6 (hot-path subset) + method names of core.py"""
    for n in ('Contour2D', 'Table', 'equivalent_latitudes', 'latitude_lengths_at'):
        assert hasattr(xa, n)
    for m in ('cal_area_eqCoord_table', 'cal_area_eqCoord_table_hist', 'cal_contours', 'cal_contours_at',
              'cal_contours_at_hist', 'cal_integral_within_contours', 'cal_integral_within_contours_hist',
              'cal_gradient_wrt_area', 'cal_contour_weigh_mean', 'cal_contour_weigh_mean_hist',
              'cal_contour_mean', 'cal_contour_mean_hist', 'cal_sqared_equivalent_length',
              'cal_local_wave_activity', 'cal_local_APE', 'cal_normalized_Keff', 'interp_to_dataset',
              'interp_to_coords'):
        assert callable(getattr(xa.Contour2D, m)), m
    assert callable(xa.Table.lookup_coordinates) and callable(xa.Table.lookup_values)


def test_constructor_errors_like_reference():
    q = xa.DataArray(np.zeros((4, 6)), ('lat', 'lon'), {'lat': np.arange(4.), 'lon': np.arange(6.)}, 'q')
    with pytest.raises(Exception, match='dimEq should be one dimension'):
        xa.Contour2D(q, np.ones(4), {'X': 'lon', 'Y': 'lat'}, {'Y': 'lat', 'X': 'lon'})
    with pytest.raises(Exception, match='dims should be a 2D plane'):
        xa.Contour2D(q, np.ones(4), {'X': 'lon'}, {'Y': 'lat'})
    cm = xa.Contour2D(q, np.ones(4), {'X': 'lon', 'Y': 'lat'}, {'Y': 'lat'})
    assert cm.dimVs == ['lon', 'lat'] and cm.dimEqV == 'lat' and cm.lt is False and cm.dtype == np.float32


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_edges_from_levels_match_oracle(dt):
    rng = np.random.default_rng(0)
    for inc in (True, False):
        b = np.sort(rng.standard_normal((3, 17)).astype(dt), axis=1)
        if not inc:
            b = b[:, ::-1]
        e, binc, closed = core._edges_from_levels(b, 'numpy')
        assert binc == inc and closed
        for s in range(3):
            eo, bo = O.hist_edges(b[s])
            assert bo == inc and np.array_equal(e[s], eo.astype(np.float64))
        e2, _, closed2 = core._edges_from_levels(b, 'xhistogram')
        assert not closed2
        eo, _ = O.hist_edges(b[0])
        assert e2[0, -1] == np.float64((eo[-1:] + 1e-8)[0])
    with pytest.raises(Exception, match='non monotonic bins'):
        core._edges_from_levels(np.array([[0., 1., 1., 2.]]), 'numpy')


def test_edges_from_levels_in_the_library_equal_the_numpy_statement():
    """core._edges_from_levels hands C-contiguous float32 / float64 level stacks to xc_host_edges_from_levels (host-only entry point of the
    library) and everything else to its numpy statement of core.py:1296-1305 + xhistogram's 1e-8: same edges bit for bit, same direction
    flag, same exceptions with the reference's texts -- over sorted / reversed stacks with coinciding levels, a slab running the other
    way, NaN slabs, infinite ends, one level only"""
    rng = np.random.default_rng(5)
    n_ok = n_err = 0
    for it in range(600):
        S = int(rng.integers(2, 5)); N = int(rng.integers(2, 12))
        dt = (np.float32, np.float64)[it % 2]
        inc = rng.random() < 0.5
        b = np.sort(rng.standard_normal((S, N)) * 10.0 ** int(rng.integers(-6, 6)), axis=1)
        b = np.ascontiguousarray(b if inc else b[:, ::-1]).astype(dt)
        r = rng.random()
        if r < 0.1:
            b[rng.integers(0, S), 1] = b[0, 0]
        if r > 0.9:
            b[1] = b[1, ::-1]
        if 0.8 < r < 0.9:
            b[1, rng.integers(0, N)] = np.nan
        if 0.7 < r < 0.8:
            b[0, -1] = np.inf if inc else -np.inf
        for re in ('numpy', 'xhistogram'):
            res = []
            for arr in (b, np.asfortranarray(b)):                    # C-contiguous: the library; Fortran order: numpy
                try:
                    with np.errstate(all='ignore'):
                        e, bi, lc = core._edges_from_levels(arr, re)
                    res.append(('ok', e.tobytes(), e.dtype, bi, lc))
                except Exception as ex:
                    res.append(('err', str(ex)))
            assert res[0] == res[1], (it, re, b)
            n_ok += res[0][0] == 'ok'; n_err += res[0][0] == 'err'
    assert n_ok > 600 and n_err > 100
    with pytest.raises(Exception, match='need at least two contour levels'):
        core._edges_from_levels(np.array([[1.0]]), 'numpy')


def test_table_from_rowsums_matches_histogram_semantics():
    """table_from_rowsums + last_row_included == the oracle's degenerate histogram (core.py:150-203 -> 1296-1325) for
    both last-bin rules, float32 / float64 / small-magnitude float32 coordinates, both coordinate directions"""
    from xcontour_amd.utils import last_row_included
    rng = np.random.default_rng(2)
    J, nx = 9, 4
    dA = rng.random((J, nx)) + 0.1
    mask = np.ones((J, nx)); mask[3, 1] = 0
    rows = np.where(mask == 1, dA, 0).sum(1)
    for coord in (np.linspace(-80, 80, J), np.linspace(-80, 80, J).astype(np.float32), np.linspace(80, -80, J).astype(np.float32),
                  np.linspace(-0.2, 0.0, J).astype(np.float32), np.linspace(-1e9, 1e9, J), np.arange(J)):
        asc = rows if coord[-1] > coord[0] else rows[::-1]
        for rule in ('xhistogram', 'numpy'):
            keep = last_row_included(coord, rule)
            if rule == 'numpy':
                assert keep
            else:      # f32 of magnitude >= 0.25 and f64 beyond ~1e8 cannot represent the +1e-8 bump
                assert keep == (not ((coord.dtype == np.float32 and abs(coord).max() > 0.25) or abs(coord).max() > 1e8))
            for inc in (True, False):
                for lt in (True, False):
                    tbl, cs = O.cal_area_eqCoord_table_hist(mask, dA, coord, inc, lt, rule)
                    ylt = lt if (inc == bool(coord[-1] > coord[0])) else (not lt)
                    assert np.allclose(table_from_rowsums(asc, ylt, keep), tbl, rtol=1e-14, atol=1e-14), (coord.dtype, rule, inc, lt)
    with pytest.raises(Exception, match='right_edge'):
        last_row_included(np.arange(3.), 'closed')


def test_contour_space_algebra_matches_oracle(baro):
    """cal_gradient_wrt_area / Leq2 / nkeff / lookup / interp are host numpy in both"""
    g = np.load(__import__('os').path.join(__import__('os').path.dirname(__file__), 'golden', 'baro_keff_N121.npz'))
    q, lat, lon = baro
    tr = xa.DataArray(q, ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'absolute_vorticity')
    cm = xa.Contour2D(tr, cell_area(lat, lon), {'X': 'longitude', 'Y': 'latitude'}, {'Y': 'latitude'}, increase=True, lt=True)
    k = np.linspace(0, 120, 121, dtype=np.float32)
    mk = lambda v, n: xa.DataArray(v, ('contour',), {'contour': k}, n)
    ctr, area, S = mk(g['ctr'], 'absolute_vorticity'), mk(g['area'], 'intArea'), mk(g['intgrdS'], 'intgrdS')
    tbl = xa.Table(xa.DataArray(g['tbl'], ('latitude',), {'latitude': g['tbl_coord']}, 'AeqCTbl'), 'latitude')
    latEq = tbl.lookup_coordinates(area)
    assert np.array_equal(latEq.values, g['latEq']) and latEq.dims == ('contour',)
    dqdA = cm.cal_gradient_wrt_area(ctr, area)
    dS = cm.cal_gradient_wrt_area(S, area)
    assert dqdA.name == 'dabsolute_vorticitydA' and dS.name == 'dintgrdSdA'
    assert np.array_equal(dqdA.values, g['dqdA']) and np.array_equal(dS.values, g['dintSdA'])
    Leq2 = cm.cal_sqared_equivalent_length(dS, dqdA)
    Lmin = xa.latitude_lengths_at(latEq)
    nk = cm.cal_normalized_Keff(Leq2, Lmin)
    assert np.array_equal(Leq2.values, g['Leq2']) and np.array_equal(Lmin.values, g['Lmin'])
    assert np.array_equal(nk.values, g['nkeff'], equal_nan=True) and nk.name == 'nkeff'
    ds = cm.interp_to_dataset(lat, latEq.rename('latEq'), [ctr, area, nk])
    assert np.array_equal(ds['absolute_vorticity'].values, g['ctr_eq'])
    assert np.array_equal(ds['nkeff'].values, g['nkeff_eq'], equal_nan=True)
    assert ds['intArea'].dims == ('new',)
    # lookup_values: the inverse direction (restated; the snapshot's attribute is undefined)
    v = tbl.lookup_values(np.array([-30., 0., 45.]))
    assert np.allclose(v, np.interp([-30., 0., 45.], g['tbl_coord'], g['tbl']))
    # decreasing tables reverse (core.py:1428-1430)
    t2 = xa.Table(xa.DataArray(g['tbl'][::-1].copy(), ('latitude',), {'latitude': g['tbl_coord']}, 'AeqCTbl'), 'latitude')
    assert np.array_equal(t2.lookup_coordinates(area).values, O.lookup_coordinates(g['area'], g['tbl'][::-1], g['tbl_coord']))
    # equivalent_latitudes (utils.py:491-515)
    assert np.array_equal(xa.equivalent_latitudes(g['area']), O.equivalent_latitudes(g['area']))


def test_table_direction_check():
    t = xa.DataArray(np.array([[0., 1., 2.], [2., 1., 0.]]), ('time', 'lat'), {'lat': np.arange(3.)}, 'AeqCTbl')
    with pytest.raises(Exception, match='not every time or level is increasing/decreasing'):
        xa.Table(t, 'lat')


def test_labeled_array_basics():
    a = xa.DataArray(np.arange(24.).reshape(2, 3, 4), ('time', 'lat', 'lon'),
                     {'time': np.arange(2), 'lat': np.arange(3.), 'lon': np.arange(4.)}, 'x')
    assert a.rename('y').name == 'y' and a.rename({'lat': 'Y'}).dims == ('time', 'Y', 'lon')
    # Dataset.rename, as tests/test_Keff_ocean.py:76 uses it on the merged result
    from xcontour_amd.labeled import Dataset
    b = xa.DataArray(np.arange(3.0), ('new',), {'new': np.arange(3.0)}, 'nkeff')
    r = Dataset([('nkeff', b)]).rename({'new': 'latitude'})
    assert r['nkeff'].dims == ('latitude',) and list(r['nkeff'].coords) == ['latitude'] and r.nkeff.name == 'nkeff'
    assert Dataset([('nkeff', b)]).rename({'nkeff': 'K'})['K'].name == 'K'
    assert a.isel({'time': 1}).shape == (3, 4) and a['lat'].values.tolist() == [0., 1., 2.]
    assert a.transpose('lon', 'lat', 'time').shape == (4, 3, 2)
    assert a.isel({'lat': slice(None, None, -1)}).coords['lat'].tolist() == [2., 1., 0.]
    assert xa.DataArray(np.zeros((1, 3)), ('a', 'b')).squeeze().dims == ('b',)


def test_shard_slabs_partition():
    for S in (1, 7, 8, 18944, 37):
        for G in (1, 2, 4, 8):
            parts = [shard_slabs(S, r, G) for r in range(G)]
            flat = [i for lo, hi in parts for i in range(lo, hi)]
            assert flat == list(range(S))                       # contiguous, disjoint, complete
            assert max(hi - lo for lo, hi in parts) == -(-S // G)


def test_metrics_helpers():
    lat = np.linspace(-90, 90, 181); lon = np.arange(360.)
    dA = cell_area(lat, lon)
    assert abs(dA.sum() / (4 * np.pi * xa.Rearth ** 2) - 1) < 1e-13
    assert np.array_equal(dA, O.cell_area(lat, lon))
    rdx, rdy = grad_metrics(lat, lon)
    ox, oy = O.grad_metrics(lat, lon)
    assert np.array_equal(rdx, ox) and np.array_equal(rdy, oy)


def test_cfg4_launch_sets_tile_the_blocks_of_every_rank_count():
    """bench.cfg4_launch_set: the launch-set size for a rank's block of the 18 944-slab stack tiles the block (every set chains
    its min/max) and fills whole rounds of workgroups, for the rank counts the driver runs; odd blocks fall back to 256"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for world, want in ((1, 512), (2, 592), (4, 592), (8, 592)):
        n = 18944 // world
        d = m.cfg4_launch_set(n)
        assert d == want and n % d == 0
        rounds = 3 * d / 256.0
        assert rounds / np.ceil(rounds) > 0.98
    assert m.cfg4_launch_set(11) == 256 and m.cfg4_launch_set(3157) == 256       # no divisor that fills the rounds: ragged sets of 256


def test_xarray_branch_of_labeled_runs_with_a_test_double():
    """xarray is the reference's in/out type (core.py:8-13) and is absent from both boxes: tests/fake_xarray/xarray.py is a
    60-line double that makes `labeled.is_xarray` true, so unwrap (coords filter included) / wrap / merge run for real
    (CPU part here; the facade's call sequences through it: tests/test_gpu_facade.py)"""
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'tests', 'fake_xarray'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'xarray_branch_script.py'), 'cpu'], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip() == 'ok cpu', r.stderr[-2000:]


def test_fast_uniform_gradient_is_np_gradient_bit_for_bit():
    """core._gradient_edge1 (round 6: np.gradient spends 15 us per call on dispatching at the reference's sizes) must BE
    np.gradient(f, x, axis, edge_order=1) on a coordinate with constant spacing -- values AND dtype, float32 / float64 in both places --
    and hand anything else (uneven spacing, integer data) to np.gradient itself (reference core.py:463-488)"""
    rng = np.random.default_rng(0)
    for fd in (np.float32, np.float64):
        for xd in (np.float32, np.float64):
            for shape, axis in (((15, 121), 1), ((121,), 0), ((3, 5, 201), 2), ((201, 4), 0), ((2, 2), 1), ((4, 3), 1)):
                f = (rng.standard_normal(shape) * 10.0 ** int(rng.integers(-8, 8))).astype(fd)
                n = shape[axis]
                x = np.linspace(0, n - 1, n, dtype=xd)
                a, b = np.gradient(f, x, axis=axis, edge_order=1), core._gradient_edge1(f, x, axis)
                assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), (fd, xd, shape)
                if n > 2:
                    x2 = x.copy(); x2[-1] += 0.5                                     # uneven: the general formula, numpy's own
                    assert np.array_equal(np.gradient(f, x2, axis=axis, edge_order=1), core._gradient_edge1(f, x2, axis))
    fi = np.arange(12).reshape(3, 4)
    assert np.array_equal(np.gradient(fi, np.arange(4.0), axis=1, edge_order=1), core._gradient_edge1(fi, np.arange(4.0), 1))


def test_gradient_wrt_area_in_the_library_is_the_numpy_formula_bit_for_bit():
    """cal_gradient_wrt_area (reference core.py:463-488) through xc_host_gradient_wrt_area -- a host-only entry point of the library,
    no device involved -- against the numpy statement np.gradient(var, k) / np.gradient(area, k): every float32 / float64 combination
    of the two arrays and their contour coordinates numpy computes without promoting (values AND dtype), rows of a larger array
    (cal_integral_within_contours_hist returns a slice of its (slab, channel, bin) result), one area profile for every row, two contours
    only, zero area steps (inf / NaN like numpy), and the cases the entry point leaves to numpy (uneven coordinate, float64 coordinate
    under float32 data, contour not last) through the facade."""
    rng = np.random.default_rng(3)
    lev = np.arange(4.0)
    cm = xa.Contour2D.__new__(xa.Contour2D)                                          # the method uses no state of the object

    def ref(v, k, a, ka, ax=-1, aax=-1):
        with np.errstate(all='ignore'):
            dv, da = np.gradient(v, k, axis=ax, edge_order=1), np.gradient(a, ka, axis=aax, edge_order=1)
            return dv / (da if da.ndim == dv.ndim or ax in (-1, dv.ndim - 1) else da[:, None])

    for n in (2, 3, 41):
        for vd in (np.float32, np.float64):
            for ad in (np.float32, np.float64):
                for kd in (np.float32, np.float64):
                    k = np.arange(n).astype(kd) * kd(0.5)
                    big = (rng.standard_normal((4, 3, n)) * 10.0 ** int(rng.integers(-6, 6))).astype(vd)
                    a2 = np.cumsum(rng.random((4, n)), axis=1).astype(ad)
                    a2[1, n // 2:] = a2[1, n // 2]                                   # zero steps of the area: 0 / 0 and x / 0
                    for v in (np.ascontiguousarray(big[:, 1, :]), big[:, 1, :]):
                        for a, adims in ((a2, ('lev', 'contour')), (a2[2], ('contour',))):
                            var = xa.DataArray(v, ('lev', 'contour'), {'lev': lev, 'contour': k}, 'q')
                            area = xa.DataArray(a, adims, {'lev': lev, 'contour': k} if len(adims) == 2 else {'contour': k}, 'A')
                            got = cm.cal_gradient_wrt_area(var, area)
                            want = ref(v, k, a, k)
                            assert got.values.dtype == want.dtype, (n, vd, ad, kd)
                            assert np.array_equal(got.values, want, equal_nan=True), (n, vd, ad, kd, adims)
                            assert got.dims == ('lev', 'contour') and got.name == 'dqdA' and np.array_equal(got.coords['contour'], k)
    # the library says "not mine" (return code 1) and writes nothing; the facade then takes numpy's general branch
    from xcontour_amd import _native as nat
    lib = nat.load()
    v = rng.standard_normal((3, 9)); a = np.cumsum(rng.random((3, 9)), axis=1); k = np.arange(9.0); ku = k.copy(); ku[4] += 0.25
    out = np.full((3, 9), 7.0)
    for kv, kdt, vv in ((ku, True, v), (k, True, v.astype(np.float32))):
        rc = lib.xc_host_gradient_wrt_area(vv.ctypes.data, vv.dtype.itemsize == 8, kv.ctypes.data, kdt, a.ctypes.data, True, k.ctypes.data, True,
                                           3, 9, 3, 9, 9, out.ctypes.data)
        assert rc == 1 and (out == 7.0).all()
    assert lib.xc_host_gradient_wrt_area(v.ctypes.data, True, k.ctypes.data, True, a.ctypes.data, True, k.ctypes.data, True, 3, 1, 3, 9, 9, out.ctypes.data) < 0
    var = xa.DataArray(v, ('lev', 'contour'), {'lev': np.arange(3.0), 'contour': ku}, 'q')
    area = xa.DataArray(a, ('lev', 'contour'), {'lev': np.arange(3.0), 'contour': ku}, 'A')
    assert np.array_equal(cm.cal_gradient_wrt_area(var, area).values, ref(v, ku, a, ku))
    varT = xa.DataArray(v.T.copy(), ('contour', 'lev'), {'lev': np.arange(3.0), 'contour': k}, 'q')          # contour first: numpy path
    areaT = xa.DataArray(a.T.copy(), ('contour', 'lev'), {'lev': np.arange(3.0), 'contour': k}, 'A')
    assert np.array_equal(cm.cal_gradient_wrt_area(varT, areaT).values, ref(v, k, a, k).T)


def test_lookup_coordinates_of_a_stack_equals_the_loop_over_slabs():
    """Table.lookup_coordinates on (level, contour) values with ONE table (round 6: one np.interp call for the stack) against the
    reference's per-slab np.interp (core.py:1136-1174, 1405-1434), increasing and decreasing tables"""
    rng = np.random.default_rng(1)
    lat = np.linspace(-90, 90, 181).astype(np.float32)
    for inc in (True, False):
        t = np.cumsum(rng.random(181)); t = t if inc else t[::-1].copy()
        tbl = xa.Table(xa.DataArray(t, ('lat',), {'lat': lat}, 'area'), 'lat')
        v = rng.random((5, 41)) * t.max() * 1.1 - 0.05
        vals = xa.DataArray(v, ('lev', 'contour'), {'lev': np.arange(5.0), 'contour': np.arange(41.0)}, 'area')
        got = tbl.lookup_coordinates(vals).values
        for s in range(5):
            ref = np.interp(v[s], t, lat) if inc else np.interp(v[s], t[::-1], lat[::-1])
            assert np.array_equal(got[s], ref.astype(got.dtype))
