"""-m gpu: the histogram pass and what hangs on it -- K1 min / max + levels (the reference notebook printout), K3 in its variants (float32 four cells per lane, E32,
chained min / max, per-slab weights, slab-major results, counts on request), the order-free deterministic sums, K4 |grad q|^2, K5 / K6 CDF + Keff epilogue.
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


@pytest.mark.parametrize('dt,cd', [(np.float64, np.float64), (np.float32, np.float32)])
def test_deterministic_pipeline_is_order_free(ctx, dt, cd):
    """xc_keff_desc.deterministic: two runs, and launch sets of 1 / 2 / all slabs (different block geometry, different
    partial layout), give the SAME bits in all nine vectors; levels and counts equal the default path's bits, sums agree
    with it and with the oracle to rounding"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N, S = 181, 360, 101, 6
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 1.0
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(dA.sum(1), True)
    kw = dict(dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True, nslots=4)
    plain = KeffPlan(ctx, S, ny, nx, N, dt, cd, **kw)
    plain.synth(lat, lon, 11, 0)
    plain.run(0)
    ref = plain.fetch(slot=0)
    q = plain.download_q()
    det = KeffPlan(ctx, S, ny, nx, N, dt, cd, deterministic=True, alloc_q=False, **kw)
    det.set_q_device(plain._q_ptr)
    outs = []
    for slot, group in enumerate((None, None, 1, 2)):
        det.run(slot, group, chain=(group == 2))                 # chained min / max (q_next) ride in the fixed-point pass
        outs.append(det.fetch(slot=slot))
    for o in outs[1:]:
        for k in NINE:
            assert np.array_equal(bits(o[k]), bits(outs[0][k])), k
        assert np.array_equal(o['counts'], outs[0]['counts'])
    d = outs[0]
    assert np.array_equal(d['ctr'], ref['ctr']) and np.array_equal(d['counts'], ref['counts'])
    assert rel(d['area'], ref['area']) < 1e-12 and rel(d['intgrdS'], ref['intgrdS']) < 1e-12
    for s in range(S):
        r = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=True, lt=True, dtype=cd)
        check_nine_det(d, s, r)
        # the oracle's own restatement of the fixed-point rule (deterministic_bin_sums): the SAME BITS, sums included
        rd = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=True, lt=True, dtype=cd, deterministic=True)
        assert np.array_equal(bits(d['area'][s]), bits(rd['area'])) and np.array_equal(bits(d['intgrdS'][s]), bits(rd['intgrdS']))
    det.free(); plain.free()


def test_deterministic_hist_channels_and_odd_inputs(ctx):
    """xc_hist_desc.deterministic through Context.hist: three channels (dA, a SIGNED integrand, |grad q|^2), NaN weights,
    NaN tracer cells, odd nx, float32 tracer; splitting the stack changes the launch geometry, not one bit"""
    rng = np.random.default_rng(5)
    for (S, ny, nx, dt) in ((5, 90, 131, np.float32), (4, 64, 256, np.float64), (3, 33, 2, np.float64)):
        q = rng.standard_normal((S, ny, nx)).astype(dt)
        q[0, 3, 1] = np.nan
        dA = rng.random((ny, nx)) + 0.1
        dA[5, 0] = np.nan                                                 # fillna(0), core.py:449
        g = rng.standard_normal((S, ny, nx)) * 10.0 ** rng.integers(-8, 8, (S, ny, nx))     # 16 decades, both signs
        edges = np.linspace(-2.5, 2.5, 41)
        rdx = rng.random(ny) + 0.5; rdy = rng.random(ny) + 0.5
        kw = dict(dA=dA, integrands=[g], grad=(rdx, rdy, True), last_closed=False, lt=True, want=('pdf', 'counts', 'cdf'))
        a = ctx.hist(q, edges, deterministic=True, **kw)
        b = ctx.hist(q, edges, deterministic=True, **kw)
        for k in ('pdf', 'cdf'):
            assert np.array_equal(bits(a[k]), bits(b[k]))
        parts = [ctx.hist(q[s:s + 1], edges, deterministic=True, **dict(kw, integrands=[g[s:s + 1]])) for s in range(S)]   # one slab per launch
        assert np.array_equal(bits(np.concatenate([p['pdf'] for p in parts])), bits(a['pdf']))
        plain = ctx.hist(q, edges, **kw)
        assert np.array_equal(a['counts'], plain['counts'])
        # against the default path on the scale of each channel's largest bin (a signed channel cancels inside a bin)
        for ch in range(3):
            scale = np.abs(plain['pdf'][:, ch]).max(axis=1, keepdims=True) + 1e-300
            assert (np.abs(a['pdf'][:, ch] - plain['pdf'][:, ch]) / scale).max() < 1e-10, ch
        # the area channel against numpy's histogram (no cell sits on the last edge, so the closed last bin is moot), and the
        # area + SIGNED integrand channels bit for bit against the oracle's restatement of the fixed-point rule
        w = np.where(np.isnan(dA), 0.0, dA)
        for s in range(S):
            assert not (q[s] == edges[-1]).any()
            ref, _ = np.histogram(q[s].astype(np.float64).ravel(), bins=edges, weights=w.ravel())
            assert rel(a['pdf'][s, 0], ref) < 1e-12
            od, _ = O.weighted_histogram(q[s].astype(np.float64), edges, w, 'numpy', deterministic=True)
            assert np.array_equal(bits(a['pdf'][s, 0]), bits(od))
            wg = g[s] * dA
            oi, _ = O.weighted_histogram(q[s].astype(np.float64), edges, np.where(np.isnan(wg), 0.0, wg), 'numpy', deterministic=True)
            assert np.array_equal(bits(a['pdf'][s, 1]), bits(oi))


def test_deterministic_infinite_weight_reports_nan(ctx):
    """an infinite weight makes ITS bin report NaN and leaves every other bin alone -- bit for bit the oracle's sums (round-5 advisor:
    the bound that positions the accumulator window came from extrema that kept +-inf, the window then sat at 2^1037 and every finite
    weight fell out of it: the other bins summed to 0.0, which `isfinite` did not notice).  The same for an infinite cell in a supplied
    integrand, for -inf, and for a weight array with no finite value at all."""
    rng = np.random.default_rng(3)
    q = np.linspace(0.05, 0.95, 64 * 128).reshape(1, 64, 128)
    edges = np.linspace(0, 1, 11)
    base = 0.5 + rng.random((64, 128))
    for sign in (1.0, -1.0):
        dA = base.copy(); dA[10, 7] = sign * np.inf
        a = ctx.hist(q, edges, dA=dA, deterministic=True, last_closed=False, want=('pdf', 'counts'))
        k = int(np.digitize(q[0, 10, 7], edges) - 1)
        od, _ = O.weighted_histogram(q[0], edges, dA, 'xhistogram', deterministic=True)
        assert np.isnan(a['pdf'][0, 0, k]) and np.isnan(od[k])
        keep = np.arange(10) != k
        assert (a['pdf'][0, 0][keep] > 0).all() and np.array_equal(bits(a['pdf'][0, 0][keep]), bits(od[keep]))
        assert a['counts'].sum() == 64 * 128
    # an infinite cell in a supplied integrand: its bin of THAT channel is NaN, the area channel and the other bins are the oracle's bits
    g = rng.standard_normal((1, 64, 128)); g[0, 33, 100] = np.inf
    a = ctx.hist(q, edges, dA=base, integrands=[g], deterministic=True, last_closed=False, want=('pdf',))
    k = int(np.digitize(q[0, 33, 100], edges) - 1)
    o0, _ = O.weighted_histogram(q[0], edges, base, 'xhistogram', deterministic=True)
    o1, _ = O.weighted_histogram(q[0], edges, g[0] * base, 'xhistogram', deterministic=True)
    assert np.array_equal(bits(a['pdf'][0, 0]), bits(o0))
    keep = np.arange(10) != k
    assert np.isnan(a['pdf'][0, 1, k]) and np.array_equal(bits(a['pdf'][0, 1][keep]), bits(o1[keep])) and (a['pdf'][0, 1][keep] != 0).all()
    # no finite weight at all: every bin that holds a cell reports NaN, nothing crashes
    a = ctx.hist(q, edges, dA=np.full((64, 128), np.inf), deterministic=True, last_closed=False, want=('pdf',))
    assert np.isnan(a['pdf'][0, 0][1:10]).all()


def test_cfg2_full_size_deterministic(ctx):
    """two runs at full cfg2 size give bit-identical area / intgrdS (and everything derived), equal to
    the oracle within the same bars as the default path; a facade object with deterministic=True does the same"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    ny, nx, N = 1801, 3600, 201
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.1
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
    plan = KeffPlan(ctx, 2, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, deterministic=True, nslots=2)
    plan.synth(lat, lon, 20241008, 0)
    plan.run(0)
    plan.run(1, group=1)
    a, b = plan.fetch(slot=0), plan.fetch(slot=1)
    for k in NINE:
        assert np.array_equal(bits(a[k]), bits(b[k])), k
    q = plan.download_q()
    r = O.keff_pipeline(q[1], dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float64)
    check_nine_det(a, 1, r)
    rd = O.keff_pipeline(q[1], dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float64, deterministic=True)
    assert np.array_equal(bits(a['area'][1]), bits(rd['area'])) and np.array_equal(bits(a['intgrdS'][1]), bits(rd['intgrdS']))   # 6.5 M cells, bit for bit
    plan.free()


def test_hist_all_nan_rows_and_land_mask(ctx):
    """rows that are entirely NaN (land in an ocean field) are skipped as a whole by K3; counts and sums
    are those of the oracle's histogram"""
    rng = np.random.default_rng(9)
    ny, nx = 120, 384
    q = rng.standard_normal((3, ny, nx))
    q[:, 10:40, :] = np.nan                                   # whole rows
    q[:, 60:90, 100:300] = np.nan                             # a continent
    q[1] = np.nan                                             # a slab with no ocean at all
    dA = rng.random((ny, nx)) + 0.5
    edges = np.linspace(-3, 3, 61)
    out = ctx.hist(q, edges, dA=dA, last_closed=True, want=('pdf', 'counts'))
    for s in range(3):
        ok = ~np.isnan(q[s])
        rc, _ = np.histogram(q[s][ok], bins=edges)
        rw, _ = np.histogram(q[s][ok], bins=edges, weights=dA[ok])
        assert np.array_equal(out['counts'][s].astype(np.int64), rc)
        assert rel(out['pdf'][s, 0], rw) < 1e-12


@pytest.mark.parametrize('nint', [0, 2])
def test_cdf_is_the_sequential_cumsum_of_the_pdf_bit_for_bit(ctx, nint):
    """k_finalize takes the cumulative sums systolically (lanes hold four elements each, 256 per chunk): for bin counts
    around the lane / chunk boundaries, for thousands of bins (work arrays in LDS and, with three channels, in global
    memory), for `lt` or not and both level orders, cdf is bit-identical to np.cumsum of the returned pdf (core.py:1320-1323)"""
    rng = np.random.default_rng(5)
    ny, nx = 40, 96
    q = rng.standard_normal((2, ny, nx))
    dA = rng.random((ny, nx)) + 0.1
    integ = [rng.standard_normal((2, ny, nx)) for _ in range(nint)]
    for nb in (1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 300, 513, 1023, 4000 if nint == 0 else 3500):
        edges = np.linspace(-3.0, 3.0, nb + 1)
        for lt in (True, False):
            for reverse in (False, True):
                out = ctx.hist(q, edges, dA, integ, lt=lt, reverse=reverse)
                pdf = out['pdf'][..., ::-1] if reverse else out['pdf']             # ascending-value order
                c = np.cumsum(pdf, axis=-1)
                if not lt:
                    c = c[..., -1:] - c
                if reverse:
                    c = c[..., ::-1]
                assert np.array_equal(bits(out['cdf']), bits(c)), (nb, lt, reverse)
                assert out['counts'].sum() <= 2 * ny * nx
    # and against the oracle's weighted histogram for one of them
    edges = np.linspace(-3.0, 3.0, 258)
    out = ctx.hist(q, edges, dA, integ, lt=True)
    for s in range(2):
        for ch, w in enumerate([dA] + [v[s] * dA for v in integ]):
            ref, cnt = O.weighted_histogram(q[s], edges, w, right_edge='numpy')
            assert np.array_equal(out['counts'][s].astype(np.int64), cnt)
            assert rel(out['pdf'][s].reshape(1 + nint, -1)[ch], ref) < TIGHT


@pytest.mark.parametrize('dt', [np.float64, np.float32])
def test_grad2_shapes_and_walls_bit_identical(ctx, dt):
    """K4 gives the bits of the normative order of operations (oracle.grad2_sphere) for widths around the 256-column
    workgroup boundary, one or a few rows, NaN cells, periodic or walled (one-sided differences at the walls)"""
    rng = np.random.default_rng(8)
    for ny, nx in ((1, 4), (2, 6), (17, 126), (33, 128), (40, 130), (5, 254), (35, 256), (16, 258), (31, 510), (32, 512), (3, 514),
                   (9, 1026), (7, 129), (19, 257)):
        q = rng.standard_normal((2, ny, nx)).astype(dt)
        q[1, ny // 2, nx // 3] = np.nan
        rdx = rng.random(ny) + 0.5
        rdy = rng.random(ny) + 0.5
        for periodic in (True, False):
            got = ctx.grad2(q, rdx, rdy, periodic)
            x = q.astype(np.float64)
            if periodic:
                gx = (np.roll(x, -1, axis=2) - np.roll(x, 1, axis=2)) * rdx[None, :, None]
            else:
                e = np.concatenate((x[:, :, 1:], x[:, :, -1:]), axis=2)
                w = np.concatenate((x[:, :, :1], x[:, :, :-1]), axis=2)
                f = np.ones(nx); f[0] = 2.0; f[-1] = 2.0
                gx = ((e - w) * rdx[None, :, None]) * f[None, None, :]
            jn = np.minimum(np.arange(ny) + 1, ny - 1); js = np.maximum(np.arange(ny) - 1, 0)
            gy = (x[:, jn, :] - x[:, js, :]) * rdy[None, :, None]
            want = gx * gx + gy * gy
            assert np.array_equal(got, want, equal_nan=True), (ny, nx, periodic)


@pytest.mark.parametrize('dt,cd,inc', [(np.float64, np.float64, True), (np.float32, np.float32, False)])
def test_slab_major_layout_is_the_dense_result_rearranged(ctx, dt, cd, inc):
    """KeffPlan(slab_major=True): the head of the slot is ONE [slab][9][N] block -- bit for bit the nine dense vectors, launch
    sets landing at their slab offset (the cfg4 sweep), counts / status / interp unchanged; and against the oracle"""
    from xcontour_amd.pipeline import KeffPlan, OUT_NAMES
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N, S = 181, 360, 41, 7
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 1.0
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(dA.sum(1), inc)                      # ylt = lt iff increase == coordinate increasing (core.py:180-188)
    pre = np.linspace(-80, 80, 33)
    kw = dict(dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=inc, lt=True, preY=pre, deterministic=True)
    dense = KeffPlan(ctx, S, ny, nx, N, dt, cd, **kw)
    dense.synth(lat, lon, 5, 0)
    dense.run(0)
    ref = dense.fetch()
    sm = KeffPlan(ctx, 3, ny, nx, N, dt, cd, alloc_q=False, out_slabs=S, slab_major=True, **kw)
    esz = ny * nx * np.dtype(dt).itemsize
    ctx._check(ctx.lib.xc_memset(ctx.handle, sm.out_ptr, 0, sm.slot_bytes))
    for c0 in range(0, S, 3):                                        # ragged launch sets 3 + 3 + 1 into one block
        m = min(3, S - c0)
        sm.set_q_device(dense._q_ptr + c0 * esz)
        sm._point(0, 0, m, out_s0=c0)
        sm.desc.q_next = None
        ctx._check(ctx.lib.xc_keff_dev(ctx.handle, __import__('ctypes').byref(sm.desc)))
    got = sm.fetch()
    assert sm.head_bytes == S * 9 * N * 8
    raw = np.empty(sm.head_bytes, dtype=np.uint8)
    ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, raw.ctypes.data, sm.out_ptr, sm.head_bytes))
    blk = raw.view(np.float64).reshape(S, 9, N)
    for i, k in enumerate(OUT_NAMES):
        assert np.array_equal(bits(got[k]), bits(ref[k])), k
        assert np.array_equal(bits(blk[:, i, :]), bits(ref[k])), k
    assert np.array_equal(got['counts'], ref['counts']) and np.array_equal(got['status'], ref['status'])
    for k in ref:
        if k.endswith('_eq'):
            assert np.array_equal(bits(got[k]), bits(ref[k])), k
    q = dense.download_q()
    r = O.keff_pipeline(q[4], dA, lat, N, lon=lon, increase=inc, lt=True, dtype=cd)
    assert np.array_equal(blk[4, 0], r['ctr'].astype(np.float64)) and np.array_equal(got['counts'][4].astype(np.int64), r['counts'])
    assert rel(blk[4, 1], r['area']) < TIGHT and rel(blk[4, 3], r['latEq']) < RTOL
    dense.free(); sm.free()


def test_out_stride_is_validated(ctx):
    from xcontour_amd import _native as nat
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N = 19, 36, 11
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 10.0
    dA = cell_area(lat, lon)
    p = KeffPlan(ctx, 1, ny, nx, N, dA=dA, lat=lat, lon=lon, tbl=table_from_rowsums(dA.sum(1), True), tbl_coord=lat)
    p.synth(lat, lon, 1, 0)
    p.desc.out_stride = N - 1
    with pytest.raises(nat.XContourHipError):
        p.run()
    p.free()


def test_e32_bin_search_ties_infinities_and_collapsed_levels(ctx):
    """the float32 variant of the histogram pass (raw float32 rows, float32 nearest-edge guess + ONE exact float32 compare):
    cells sitting exactly ON contour levels, on the dummy edge and on the bumped last edge, +-inf, NaN, denormal-scale fields
    whose float32 levels are not equally spaced (the variant must fall back to the exact search) -- counts bit for bit"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    rng = np.random.default_rng(44)
    ny, nx, N, S = 64, 1280, 201, 6
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * (360.0 / nx)
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
    q = (np.sin(np.deg2rad(lat))[None, :, None] + 0.05 * rng.standard_normal((S, ny, nx))).astype(np.float32)
    q[2] = (1e-38 * rng.random((ny, nx))).astype(np.float32)                       # denormal range
    q[3] = (300.0 + 0.05 * rng.standard_normal((ny, nx))).astype(np.float32)       # levels ~16 ulps apart: uneven, still "equally spaced to a quarter bin"
    q[4] = (300.0 + 1.2e-3 * rng.standard_normal((ny, nx))).astype(np.float32)     # levels 1-2 ulps apart: NOT equally spaced -> the exact search
    q[5] = np.float32(7.25)                                                        # constant field: all levels coincide (status 1)
    for s in (0, 1):
        ctr = O.cal_contours(q[s], N, True, np.float32)
        inner = (q[s] > q[s].min()) & (q[s] < q[s].max())
        idx = np.flatnonzero(inner.ravel())[:4000]
        q[s].ravel()[idx] = ctr[rng.integers(0, N, idx.size)]                      # exactly on levels (min / max cells untouched)
        assert np.array_equal(O.cal_contours(q[s], N, True, np.float32), ctr)
    q[1, 5, 7:11] = [np.inf, -np.inf, np.nan, np.inf]
    for inc in (True, False):
        plan = KeffPlan(ctx, S, ny, nx, N, np.float32, np.float32, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=inc, lt=True)
        plan.set_q(q)
        plan.run(0)
        b = plan.fetch(check=False)
        assert list(b['status']) == [0, 0, 0, 0, 0, 1] or b['status'][5] == 1
        for s in range(5):
            qs = q[s]
            if s == 1:
                continue                                                           # +-inf: the extremes are infinite, levels NaN -- compared below
            ctr = O.cal_contours(qs, N, inc, np.float32)
            assert np.array_equal(b['ctr'][s], ctr.astype(np.float64)), s
            if len(np.unique(ctr)) == N:
                _, cnt = O.cal_integral_within_contours_hist(qs, ctr, dA, None, True, return_counts=True)
                assert np.array_equal(b['counts'][s].astype(np.int64), cnt), (s, inc)
        plan.free()


def test_keff_without_counts_gives_the_same_vectors(ctx):
    """KeffPlan(counts=False) / xc_keff_desc.counts = NULL: the histogram pass skips the count adds (the reference's Keff
    sequence never looks at counts; Contour2D.keff runs this way) -- the nine vectors are the same bits with deterministic
    sums, float32 (E32) and float64, chained and not; xc_hist without 'counts' likewise"""
    from xcontour_amd.pipeline import KeffPlan, OUT_NAMES
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N, S = 91, 1280, 61, 4
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * (360.0 / nx)
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(dA.sum(1), True)
    for dt in (np.float64, np.float32):
        kw = dict(dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, deterministic=True, nslots=2)
        a = KeffPlan(ctx, S, ny, nx, N, dt, dt, **kw)
        a.synth(lat, lon, 3, 0)
        b = KeffPlan(ctx, S, ny, nx, N, dt, dt, alloc_q=False, counts=False, **kw)
        b.set_q_device(a._q_ptr)
        for slot, (group, chain) in enumerate(((None, False), (2, True))):
            a.run(slot, group, chain=chain); b.run(slot, group, chain=chain)
            ra, rb = a.fetch(slot=slot), b.fetch(slot=slot)
            for k in OUT_NAMES:
                assert np.array_equal(bits(ra[k]), bits(rb[k])), (k, dt, chain)
        q = a.download_q()
        r = O.keff_pipeline(q[1], dA, lat, N, lon=lon, increase=True, lt=True, dtype=dt)
        assert np.array_equal(ra['counts'][1].astype(np.int64), r['counts']) and rel(rb['area'][1], r['area']) < TIGHT
        a.free(); b.free()
    rng = np.random.default_rng(2)
    qh = rng.standard_normal((2, 40, 130))
    ed = np.linspace(-4, 4, 33)
    w = rng.random((40, 130))
    full = ctx.hist(qh, ed, dA=w, want=('pdf', 'cdf', 'counts'), deterministic=True)
    part = ctx.hist(qh, ed, dA=w, want=('pdf', 'cdf'), deterministic=True)
    assert np.array_equal(bits(full['pdf']), bits(part['pdf'])) and np.array_equal(bits(full['cdf']), bits(part['cdf']))
    plain = ctx.hist(qh, ed, dA=w, want=('cdf',))
    assert rel(plain['cdf'], full['cdf']) < 1e-13


def test_deterministic_sums_keep_49_bits_over_a_wide_dynamic_range(ctx):
    """round 5, the one-pass rule: a bin holds weights 2^120 apart (the pole row of a lat-lon grid does that to |grad q|^2) and BOTH ends
    keep their 49 bits -- the GPU's sums are the oracle's superaccumulator bit for bit, with the window the oracle is told (the
    bound the library derives: max |integrand| x max dA); a signed integrand, float32 and float64; and the sum of the
    cut weights is exact where float64 summation in cell order loses the small ones"""
    rng = np.random.default_rng(77)
    S, ny, nx, nb = 2, 96, 256, 37
    q = rng.random((S, ny, nx))
    edges = np.linspace(0.0, 1.0, nb + 1)
    mag = 10.0 ** rng.integers(-18, 18, (S, ny, nx))
    g = rng.standard_normal((S, ny, nx)) * mag
    g[0, 5, 7] = 3.0e30; g[0, 5, 8] = -3.0e30                               # cancel exactly: what is left are the small ones
    q[0, 5, 8] = q[0, 5, 7]
    dA = 0.5 + rng.random((ny, nx))
    dA[5, 8] = dA[5, 7]
    out = ctx.hist(q, edges, dA=dA, integrands=[g], last_closed=True, want=('pdf', 'counts'), deterministic=True)
    again = ctx.hist(q[:, :, ::1].copy(), edges, dA=dA, integrands=[g], last_closed=True, want=('pdf',), deterministic=True)
    assert np.array_equal(bits(out['pdf']), bits(again['pdf']))
    for s_ in range(S):
        w = g[s_] * dA
        top = O.det_window_top(max(abs(g[s_].min()), abs(g[s_].max())) * dA.max())
        od, cnt = O.weighted_histogram(q[s_], edges, w, 'numpy', deterministic=True, det_top=top)
        assert np.array_equal(out['counts'][s_].astype(np.int64), cnt)
        assert np.array_equal(bits(out['pdf'][s_, 1]), bits(od))
        oa, _ = O.weighted_histogram(q[s_], edges, np.broadcast_to(dA, q[s_].shape), 'numpy', deterministic=True, det_top=O.det_window_top(dA.max()))
        assert np.array_equal(bits(out['pdf'][s_, 0]), bits(oa))
    k = int(np.digitize(q[0, 5, 7], edges)) - 1
    plain = ctx.hist(q[:1], edges, dA=dA, integrands=[g[:1]], last_closed=True, want=('pdf',))['pdf'][0, 1, k]
    from fractions import Fraction
    sel = (np.digitize(q[0], edges) - 1) == k
    exact = float(sum(Fraction(float(v)) for v in (g[0] * dA)[sel]))
    assert abs(out['pdf'][0, 1, k] - exact) <= 1e-13 * np.abs((g[0] * dA)[sel & (np.abs(g[0]) < 1e30)]).sum()
    assert abs(plain - exact) > abs(out['pdf'][0, 1, k] - exact)             # float64 atomics lost what 3e30 - 3e30 swallowed


def test_contours_in_one_call_equals_minmax_then_levels(ctx, baro):
    """round 5: xc_contours (K1 + levels, one upload, one hand-over, one sync) gives the bits of xc_minmax followed by xc_levels -- float32
    and float64 tracers, both directions, both contour dtypes; cal_contours goes through it"""
    q0, lat, lon = baro
    rng = np.random.default_rng(1)
    for dt in (np.float32, np.float64):
        q = np.stack([q0.astype(dt) * (1 + 0.1 * s_) + dt(1e-5) * rng.standard_normal(q0.shape).astype(dt) for s_ in range(3)])
        q[1, 4, 5] = np.nan
        mm = ctx.minmax(q)
        for inc in (True, False):
            for cd in (np.float32, np.float64):
                ctr, mm2 = ctx.contours(q, 61, inc, cd, want_minmax=True)
                ref = ctx.levels(mm, q.dtype, 61, inc, cd)[0]
                assert np.array_equal(bits(ctr), bits(ref)) and np.array_equal(bits(mm2), bits(mm))
                for s_ in range(3):
                    assert np.array_equal(ctr[s_], O.cal_contours(q[s_], 61, inc, cd).astype(np.float64))


@pytest.mark.parametrize('single', [True, False])
@pytest.mark.parametrize('dt,counts', [(np.float64, True), (np.float32, True), (np.float64, False)])
def test_one_slab_alone_equals_the_same_slab_in_a_stack(ctx, dt, counts, single):
    """a launch of ONE slab against the same slab in a stack of 8 (per-block partials), on BOTH one-slab paths: `single` -- the
    single-read kernel (xc_keff1.hip: min/max, levels, histogram, epilogue in ONE launch, the slab held in registers), else the chain
    K1 -> K3 -> finalize, where ~256 blocks add their sums into the slab's accumulators (cleared by the K1 launch).  Levels and counts
    are identical, the sums agree to 1e-13, all nine vectors against the oracle; run twice (the accumulators must be cleared again)"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    ny, nx, N, S = 721, 1440, 201, 8
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.25
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
    kw = dict(dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True, counts=counts)
    stack = KeffPlan(ctx, S, ny, nx, N, dt, dt, **kw)
    stack.synth(lat, lon, 777, 0)
    stack.run()
    ref = stack.fetch()
    q = stack.download_q()
    one = KeffPlan(ctx, 1, ny, nx, N, dt, dt, single_read=single, **kw)
    for s in (0, 5):
        one.q_buf.upload(q[s])
        for _ in range(2):
            one.run()
            assert ctx.last_keff_path() == (1 if single else 0)
            got = one.fetch()
            assert one.replays == 0
            assert np.array_equal(got['ctr'][0], ref['ctr'][s])
            if counts:
                assert np.array_equal(got['counts'][0], ref['counts'][s])
            assert rel(got['area'][0], ref['area'][s]) < 1e-13 and rel(got['intgrdS'][0], ref['intgrdS'][s]) < 1e-13
        r = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=True, lt=True, dtype=dt)
        if not counts:
            got['counts'] = r['counts'][None]                  # (not produced: nothing to compare)
        check_nine(got, 0, r)
    one.free(); stack.free()


def test_small_inputs_are_recognised_by_content_never_by_address(ctx):
    """round 6: a host-form call keeps its small inputs (bin edges <= 64 KB) on the device and a later call with the SAME BYTES reads them
    there without a transfer.  The bytes decide, not the address: edges changed IN PLACE between two calls must give the new histogram,
    the old edges at a new address the old one; more distinct edge sets than the cache has entries (4) must all come out right, and a
    call that fails (non monotonic bins) must not leave anything behind that a later call could hit."""
    rng = np.random.default_rng(21)
    q = rng.standard_normal((3, 40, 64))
    w = rng.random((40, 64)) + 0.5

    def want(e):
        return np.stack([np.histogram(q[s].ravel(), bins=e, weights=w.ravel())[0] for s in range(3)])

    e = np.linspace(-3.0, 3.0, 25)
    r1 = ctx.hist(q, e, dA=w, want=('pdf',))['pdf'][:, 0, :]
    assert np.allclose(r1, want(e), rtol=1e-12, atol=0)
    keep = e.copy()
    e[:] = np.linspace(-2.0, 2.5, 25)                                   # same array, same address, new bytes
    r2 = ctx.hist(q, e, dA=w, want=('pdf',))['pdf'][:, 0, :]
    assert np.allclose(r2, want(e), rtol=1e-12, atol=0) and not np.allclose(r2, r1)
    r3 = ctx.hist(q, keep, dA=w, want=('pdf',))['pdf'][:, 0, :]         # the first bytes again, elsewhere: a hit, and the same result
    assert np.array_equal(r3, r1) or np.allclose(r3, r1, rtol=1e-13, atol=0)
    sets = [np.linspace(-3.0 + 0.1 * i, 3.0 - 0.07 * i, 25) for i in range(7)]
    for rep in range(2):
        for ee in sets:
            assert np.allclose(ctx.hist(q, ee, dA=w, want=('pdf',))['pdf'][:, 0, :], want(ee), rtol=1e-12, atol=0)
    bad = e.copy(); bad[5] = bad[4]
    with pytest.raises(Exception, match='non monotonic bins'):
        ctx.hist(q, bad, dA=w, want=('pdf',))
    assert np.allclose(ctx.hist(q, e, dA=w, want=('pdf',))['pdf'][:, 0, :], want(e), rtol=1e-12, atol=0)
    # the in-kernel gradient's metrics take the same road
    rdx = rng.random(40) + 0.1; rdy = rng.random(40) + 0.1
    g1 = ctx.hist(q, e, dA=w, grad=(rdx, rdy, True), want=('pdf',))['pdf']
    rdx2 = rdx.copy(); rdx[:] = rdx * 2.0
    g2 = ctx.hist(q, e, dA=w, grad=(rdx, rdy, True), want=('pdf',))['pdf']
    g3 = ctx.hist(q, e, dA=w, grad=(rdx2, rdy, True), want=('pdf',))['pdf']
    assert np.allclose(g1[:, 1], g3[:, 1], rtol=1e-12, atol=0) and not np.allclose(g1[:, 1], g2[:, 1], rtol=1e-3, atol=0)
    assert np.allclose(g1[:, 0], g2[:, 0], rtol=1e-12, atol=0)


def test_copy_kernel_switched_off_gives_the_same_facade_results(tmp_path):
    """XC_COPY_KERNEL=0 (one DMA copy per small array, the path of rounds 1-5) against the default (copy kernels, results written straight
    into the pinned buffer, levels reducing K1's partials): the reference's call sequence on a small stack, every result bit for bit"""
    code = '''
import sys, numpy as np
sys.path.insert(0, %r)
import xcontour_amd as xa
rng = np.random.default_rng(4)
lat = np.linspace(-88, 88, 45); lon = np.arange(64) * 5.625; lev = np.arange(3.0)
q = (np.sin(np.deg2rad(lat))[None, :, None] * (1 + 0.2 * lev[:, None, None]) + 0.1 * rng.standard_normal((3, 45, 64))).astype(np.float32)
c3 = {'lev': lev, 'lat': lat, 'lon': lon}; c2 = {'lat': lat, 'lon': lon}
tr = xa.DataArray(q, ('lev', 'lat', 'lon'), c3, 'pv')
dA = xa.DataArray(xa.cell_area(lat, lon), ('lat', 'lon'), c2, 'dA')
g2 = xa.DataArray(rng.random(q.shape).astype(np.float32), ('lev', 'lat', 'lon'), c3, 'grdS')
mask = xa.DataArray(np.ones((45, 64), np.float32), ('lat', 'lon'), c2, 'mask')
out = []
for resident in (False, True):
    cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, resident=resident, deterministic=True)
    for rep in range(2):
        table = cm.cal_area_eqCoord_table_hist(mask)
        ctr = cm.cal_contours(31)
        area = cm.cal_integral_within_contours_hist(ctr)
        intS = cm.cal_integral_within_contours_hist(ctr, integrand=g2)
        latEq = table.lookup_coordinates(area)
        d1 = cm.cal_gradient_wrt_area(ctr, area); d2 = cm.cal_gradient_wrt_area(intS, area)
        out += [table._table.values, ctr.values, area.values, intS.values, latEq.values, d1.values, d2.values]
np.save(sys.argv[1], np.concatenate([np.asarray(o, np.float64).ravel() for o in out]))
''' % ROOT
    res = []
    for ck in ('1', '0'):
        env = _clean_env(); env['XC_COPY_KERNEL'] = ck
        f = str(tmp_path / ('r%s.npy' % ck))
        r = subprocess.run([sys.executable, '-c', code, f], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(f))
    assert res[0].shape == res[1].shape and np.array_equal(res[0], res[1], equal_nan=True)
