#!/usr/bin/env python3
"""Differential fuzz: random shapes / dtypes / flags / NaN patterns, HIP path vs oracle, until the time budget is spent.
    python tools/gpu_fuzz.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import xcontour_oracle as O
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import table_from_rowsums, last_row_included

budget = float(sys.argv[1]) if len(sys.argv) > 1 and __name__ == '__main__' else 60.0
checked = {}


def tick(name):
    checked[name] = checked.get(name, 0) + 1

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 and __name__ == '__main__' else 12345)
ctx = nat.Context(0)
SC = int(os.environ.get('XC_FUZZ_SCALE', '1'))      # multiplies the maximum plane size (tiling edges of the big kernels)


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert np.array_equal(np.isnan(a), np.isnan(b)), 'NaN pattern'
    m = np.isfinite(b)
    assert np.array_equal(a[~m & ~np.isnan(b)], b[~m & ~np.isnan(b)]), 'inf pattern'
    return float(np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300))) if m.any() else 0.0


def field(S, ny, nx, dt):
    q = (np.linspace(-1, 1, ny)[None, :, None] * rng.uniform(0.2, 3) + rng.uniform(0.01, 1) * rng.standard_normal((S, ny, nx))).astype(dt)
    if rng.random() < 0.5:
        q[rng.random(q.shape) < rng.uniform(0, 0.1)] = np.nan
    if rng.random() < 0.2:
        q[0, rng.integers(ny), :] = q[0, 0, 0]
    return q


def case_hist():
    S, ny, nx = int(rng.integers(1, 5)), int(rng.integers(1, 90 * SC)), int(rng.integers(1, 300 * SC))
    dt = rng.choice([np.float32, np.float64])
    q = field(S, ny, nx, dt)
    nb = int(rng.integers(1, 80)) if rng.random() < 0.85 else int(rng.integers(80, 700))     # (past 256 bins: several chunks of the in-order scan)
    lo, hi = np.nanmin(q) if np.isfinite(q).any() else 0.0, np.nanmax(q) if np.isfinite(q).any() else 1.0
    ed = np.sort(rng.uniform(lo - 0.1, hi + 0.1, nb + 1)) if rng.random() < 0.5 else np.linspace(lo, hi + 1e-9, nb + 1)
    if len(np.unique(ed)) != len(ed):
        return
    dA = rng.random((ny, nx)) + 0.1
    if rng.random() < 0.3:
        dA[rng.integers(ny), rng.integers(nx)] = np.nan
    last = bool(rng.random() < 0.7)
    det = bool(rng.random() < 0.4)
    cap = ctx.max_batch_bytes
    if rng.random() < 0.3:
        ctx.max_batch_bytes = max(1, int(q[0].nbytes * rng.uniform(0.5, 2.5)))      # the stack goes through in batches of one or two slabs
    try:
        out = ctx.hist(q, ed, dA=dA, last_closed=last, lt=bool(rng.random() < 0.5), want=('counts', 'pdf'), deterministic=det)
        if det:                                                     # order-free sums: the same bits again, whatever the batching
            ctx.max_batch_bytes = cap if rng.random() < 0.5 else max(1, q[0].nbytes)
            again = ctx.hist(q, ed, dA=dA, last_closed=last, lt=True, want=('pdf',), deterministic=True)
            assert np.array_equal(again['pdf'].view(np.int64), out['pdf'].view(np.int64)), 'deterministic pdf bits'
            tick('hist_deterministic')
    finally:
        ctx.max_batch_bytes = cap
    for s in range(S):
        x = q[s].astype(np.float64).ravel(); w = np.nan_to_num(dA.ravel(), nan=0.0)
        if last:
            # np.histogram is the normative BINNING (counts); its weighted sums come from differences of a cumulative
            # sum (1e-16 of the total per bin: 5e-11 relative on a one-cell bin), so sums are checked against the oracle
            c, _ = np.histogram(x[~np.isnan(x)], bins=ed)
            assert np.array_equal(out['counts'][s].astype(np.int64), c), 'hist counts'
        p, oc = O.weighted_histogram(x, ed, w, 'numpy') if last else (None, None)
        if last:
            assert np.array_equal(oc, c), 'oracle counts'
            assert relerr(out['pdf'][s, 0], p) < 1e-12, 'hist pdf'
            if det:                                                 # the fixed-point rule restated in numpy: the same bits
                pd_, _ = O.weighted_histogram(x, ed, w, 'numpy', deterministic=True)
                assert np.array_equal(out['pdf'][s, 0].view(np.int64), pd_.view(np.int64)), 'deterministic pdf vs oracle bits'
                tick('hist_deterministic_oracle_bits')
            tick('hist')


def case_keff():
    S, ny, nx = int(rng.integers(1, 4)), int(rng.integers(8, 70 * SC)), int(rng.integers(8, 200 * SC))
    dt = rng.choice([np.float32, np.float64]); cdt = rng.choice([np.float32, np.float64])
    q = field(S, ny, nx, dt)
    q[~np.isfinite(q)] = 0.0
    # float32 coordinates half the time: under the xhistogram rule `lat[-1] + 1e-8 == lat[-1]` then, and the last
    # row drops out of the A(Yeq) table (the reference's own barotropic_vorticity.nc has float32 latitudes)
    lat = np.linspace(-80, 80, ny).astype(rng.choice([np.float32, np.float64])); lon = np.arange(nx) * (360.0 / nx)
    dA = O.cell_area(lat, lon)
    N = int(rng.integers(3, 60)) if rng.random() < 0.85 else int(rng.integers(60, 400)); inc = bool(rng.random() < 0.5); lt = bool(rng.random() < 0.5)
    re_ = str(rng.choice(['numpy', 'xhistogram']))
    ylt = lt if inc else (not lt)
    rows = ctx.rowsum(None, dA, ny, nx)
    tbl = table_from_rowsums(rows, ylt, last_row_included(lat, re_))          # the product's own table path (K2 + host rule)
    otbl, cs = O.cal_area_eqCoord_table_hist(np.ones((ny, nx)), dA, lat, inc, lt, re_)
    assert relerr(tbl, otbl) < 1e-12, 'keff table'
    det = bool(rng.random() < 0.4)
    plan = KeffPlan(ctx, S, ny, nx, N, dt, cdt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=cs, increase=inc, lt=lt,
                    right_edge=re_, deterministic=det, nslots=2)
    plan.set_q(q); plan.run(0)
    try:
        r = plan.fetch(slot=0)
    except Exception:
        plan.free(); return
    if det:                                                         # one slab per launch: other geometry, same bits
        plan.run(1, group=1)
        r2 = plan.fetch(check=False, slot=1)
        for k in ('area', 'intgrdS', 'latEq', 'nkeff'):
            assert np.array_equal(r[k].view(np.int64), r2[k].view(np.int64)), 'deterministic keff bits ' + k
        tick('keff_deterministic')
    plan.free()
    for s in range(S):
        o = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=inc, lt=lt, dtype=cdt, right_edge=re_)
        assert np.array_equal(r['ctr'][s], o['ctr'].astype(np.float64)), 'keff ctr'
        assert np.array_equal(r['counts'][s].astype(np.int64), o['counts']), 'keff counts'
        assert np.array_equal(np.isnan(r['latEq'][s]), np.isnan(o['latEq'])) and \
            np.nanmax(np.abs(r['latEq'][s] - o['latEq']), initial=0.0) < 1e-9 * 90, 'keff latEq'      # absolute: latEq passes through 0
        assert relerr(r['area'][s], o['area']) < 1e-11, 'keff area'
        assert relerr(r['intgrdS'][s], o['intgrdS']) < 1e-9, 'keff intgrdS'
        tick('keff')


def case_crossing():
    S, ny, nx = int(rng.integers(1, 4)), int(rng.integers(2, 80 * SC)), int(rng.integers(2, 300 * SC))
    dt = rng.choice([np.float32, np.float64])
    q = field(S, ny, nx, dt)
    stride = int(rng.choice([1, 1, 2, 3, 4, 6, 7, 9, 16, 33]))
    pad = stride + int(rng.integers(0, 3))
    mode = str(rng.choice(['edge', 'wrap', 'constant', 'reflect', 'symmetric']))
    if mode == 'reflect' and pad > nx - 1:
        return
    if mode == 'symmetric' and pad > nx:
        return
    cs = np.sort(rng.uniform(-2, 2, int(rng.integers(1, 40))))
    area = (rng.random((ny, nx)) + 0.2).astype(rng.choice([np.float32, np.float64]))
    full = bool(rng.random() < 0.5)
    lens, cnts = ctx.crossing(q, cs, area, stride=stride, pad_x=pad, pad_mode=mode, full_width=full)
    for s in range(S):
        ol, oc = O.contour_crossing(O.pad_x(q[s], pad, mode), cs, O.pad_x(area, pad, mode), stride, full)
        assert np.array_equal(cnts[s].astype(np.int64), oc), 'crossing counts %r' % ((S, ny, nx, stride, pad, mode, full),)
        assert relerr(lens[s], ol) < 1e-12, 'crossing lengths'
        tick('crossing')


def case_lwa():
    S, ny, nx = int(rng.integers(1, 3)), int(rng.integers(2, 90 * min(SC, 3))), int(rng.integers(1, 200 * SC))
    dt = rng.choice([np.float32, np.float64])
    q = field(S, ny, nx, dt)
    coord = np.linspace(-50, 50, ny) * (1 if rng.random() < 0.5 else -1)
    Q = np.sort(rng.standard_normal((S, ny)), axis=1) if rng.random() < 0.7 else rng.standard_normal((S, ny))
    dA = rng.random((ny, nx)) + 0.3
    inc = bool(rng.random() < 0.5); pc = int(rng.integers(0, 3)); var = int(rng.integers(0, 2))
    out, _ = ctx.lwa(q, Q, coord, dA, dA.max(), M=None, increase=inc, part=pc, variant=var)
    fn = O.cal_local_wave_activity2 if var else O.cal_local_wave_activity
    for s in range(S):
        with np.errstate(invalid='ignore'):
            ref = fn(q[s], Q[s], coord, dA, inc, ('all', 'upper', 'lower')[pc])
        if nx == 1:      # numpy sums a single contiguous column pairwise, not row by row: last-bit differences
            assert relerr(out[s], ref) < 1e-13, 'lwa nx=1'
        else:
            assert np.array_equal(out[s], ref, equal_nan=True), 'lwa %r' % ((S, ny, nx, inc, pc, var),)
        tick('lwa')


def case_lwa_interval():
    """planes of more than 512 rows: the O(ny log ny) interval kernel (monotone reference state) against the oracle's loop, <= 1e-11
    of the plane's largest value; a reference state that is not monotone must take the bit-exact band walk"""
    S, ny, nx = int(rng.integers(1, 3)), int(rng.integers(513, 640)), int(rng.integers(1, 40))
    dt = rng.choice([np.float32, np.float64])
    q = field(S, ny, nx, dt)
    up = bool(rng.random() < 0.5)
    coord = np.linspace(-50, 50, ny) * (1 if up else -1)
    inc = bool(rng.random() < 0.5)
    Q = np.sort(rng.standard_normal((S, ny)) * rng.uniform(0.3, 2), axis=1)
    if rng.random() < 0.3:
        Q[:, rng.integers(1, ny)] = Q[:, 0]                               # ties inside Q
        Q = np.sort(Q, axis=1)
    if not inc:
        Q = Q[:, ::-1].copy()
    idx = rng.integers(0, q.size, 50)
    q.reshape(-1)[idx] = Q[0, rng.integers(0, ny, 50)].astype(dt)        # cells exactly on reference levels
    broken = rng.random() < 0.2
    if broken:
        Q[0, [3, 4]] = Q[0, [4, 3]] + (0.1 if inc else -0.1) * np.array([1, -1])
    dA = rng.random((ny, nx)) + 0.3
    M = None if rng.random() < 0.4 else (rng.random(ny) + 0.5 if rng.random() < 0.5 else rng.random((ny, nx)) + 0.5)
    pc = int(rng.integers(0, 3))
    out, _ = ctx.lwa(q, Q, coord, dA, dA.max(), M=M, increase=inc, part=pc, variant=0)
    path = ctx.last_lwa_path()
    mono = all(np.all(np.diff(Q[s] if inc else -Q[s]) >= 0) for s in range(S))
    assert path == (1 if mono else 2), 'lwa interval path %d, monotone %r' % (path, mono)
    for s in range(S):
        with np.errstate(invalid='ignore'):
            ref = O.cal_local_wave_activity(q[s], Q[s], coord, dA, inc, ('all', 'upper', 'lower')[pc], metric=M)
        if path == 2 and nx > 1:
            assert np.array_equal(out[s], ref, equal_nan=True), 'lwa band walk behind a failed check'
        else:
            scale = max(float(np.abs(ref).max()), 1e-300)
            assert float(np.abs(out[s] - ref).max()) <= 1e-11 * scale, 'lwa interval %r: %g' % ((S, ny, nx, inc, up, pc), float(np.abs(out[s] - ref).max()) / scale)
        tick('lwa_interval')


def case_sort():
    S, ny, nx = int(rng.integers(1, 4)), int(rng.integers(1, 70 * SC)), int(rng.integers(1, 300 * SC))
    dt = rng.choice([np.float32, np.float64])
    q = field(S, ny, nx, dt)
    if rng.random() < 0.3:
        q = np.round(q, 1)                                  # many ties
    if rng.random() < 0.3:                                  # plateaus and clusters far below the range-key resolution
        q = np.where(rng.random(q.shape) < 0.5, q, (q[0, 0, 0] if np.isfinite(q[0, 0, 0]) else 0.5) +
                     rng.choice([1e-15, 1e-11, 1e-7]) * rng.integers(0, 50, q.shape)).astype(dt)
    dA = rng.random((ny, nx)) + 0.1
    neg = bool(rng.random() < 0.3)
    r = ctx.sort_profile(q, dA=dA, want_sorted=True, want_acum=True, negate=neg)
    tick('sort_path_%d' % ctx.last_sort_path())
    for s in range(S):
        x = q[s].astype(np.float64).ravel(); x = x[~np.isnan(x)]
        n = int(r['nvalid'][s])
        assert n == len(x), 'sort nvalid'
        _, oxs, oac = O.sorted_profile(-q[s] if neg else q[s], dA, [0.0])
        assert np.array_equal(r['q_sorted'][s][:n], oxs.astype(np.float64)), 'sort order'
        assert n == 0 or relerr(r['acum'][s][:n], oac) < 1e-12, 'sort payload order (stability)'
        got = r['q_sorted'][s][:n]
        want = np.sort(x) if np.array_equal(got, np.sort(got)) and not np.array_equal(got, -np.sort(-x)[::-1]) else None
        assert np.array_equal(np.sort(got), got), 'sortedness'
        assert np.array_equal(got, np.sort(x)) or np.array_equal(got, np.sort(-x)), 'sort values'
        tick('sort')


def case_facade():
    """the reference's call sequence through Contour2D / Table with random options, masks, NaNs, flipped coordinates"""
    import xcontour_amd as xa
    ny, nx = int(rng.integers(6, 60 * SC)), int(rng.integers(6, 150 * SC))
    dt = rng.choice([np.float32, np.float64]); cdt = rng.choice([np.float32, np.float64])
    inc, lt, flip = bool(rng.random() < 0.5), bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    lat = np.linspace(-85, 85, ny).astype(rng.choice([np.float32, np.float64])); lon = np.arange(nx) * (360.0 / nx)
    q = field(1, ny, nx, dt)[0]
    if flip:
        lat = lat[::-1].copy()
    dAv = O.cell_area(np.sort(lat), lon)[::-1].copy() if flip else O.cell_area(lat, lon)
    maskv = np.ones((ny, nx)); maskv[np.isnan(q)] = 0.0
    if rng.random() < 0.3:
        maskv[:, : max(1, nx // 7)] = 0.0
    c = {'lat': lat, 'lon': lon}
    tr = xa.DataArray(q, ('lat', 'lon'), c, 'pv'); dA = xa.DataArray(dAv, ('lat', 'lon'), c, 'dA')
    mask = xa.DataArray(maskv, ('lat', 'lon'), c, 'mask')
    g2v = rng.random((ny, nx)).astype(dt)
    g2 = xa.DataArray(g2v, ('lat', 'lon'), c, 'grdS')
    re_ = str(rng.choice(['numpy', 'xhistogram']))              # the two last-bin rules (oracle header)
    cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=inc, lt=lt, dtype=cdt, right_edge=re_)
    N = int(rng.integers(3, 70))
    try:
        ctr = cm.cal_contours(N)
        o_ctr = O.cal_contours(q, N, inc, cdt)
        assert np.array_equal(ctr.values, o_ctr, equal_nan=True), 'facade ctr'
        area = cm.cal_integral_within_contours_hist(ctr)
    except Exception as e:
        if 'non monotonic' in str(e) or 'bins' in str(e):
            return                                         # a constant / degenerate field: the reference raises too
        raise
    o_area = O.cal_integral_within_contours_hist(q, o_ctr, dAv, None, lt, re_)
    assert relerr(area.values, o_area) < 1e-11, 'facade area'
    S_ = cm.cal_integral_within_contours_hist(ctr, integrand=g2)
    assert relerr(S_.values, O.cal_integral_within_contours_hist(q, o_ctr, dAv, g2v, lt, re_)) < 1e-9, 'facade intS'
    a2 = cm.cal_integral_within_contours(ctr)
    assert relerr(a2.values, O.cal_integral_within_contours(q, o_ctr, dAv, None, lt)) < 1e-11, 'facade strict'
    for hist in (True, False):
        t = (cm.cal_area_eqCoord_table_hist if hist else cm.cal_area_eqCoord_table)(mask)
        ot, ocs = O.cal_area_eqCoord_table_hist(maskv, dAv, lat, inc, lt, re_) if hist else O.cal_area_eqCoord_table(maskv, dAv, lat, inc, lt)
        assert relerr(t._table.values, ot) < 1e-12, 'facade table'
        yeq = t.lookup_coordinates(area)
        # lookup on the SAME inputs: where a fully masked row makes the table flat, the interpolated coordinate jumps
        # with the last bit of the area (np.interp on duplicate knots), so end-to-end comparison is ill-posed there
        oy = O.lookup_coordinates(area.values, t._table.values, t._coord)
        assert np.array_equal(np.isnan(yeq.values), np.isnan(oy)) and np.nanmax(np.abs(yeq.values - oy), initial=0.0) < 1e-12 * 90, 'facade lookup'
    tick('facade')


cases = [case_lwa_interval, case_hist, case_keff, case_crossing, case_lwa, case_sort, case_facade]


def run(seconds, seed=None):
    """run random cases for `seconds`; returns (cases per kind, slab comparisons per kind); raises on a mismatch"""
    global rng
    if seed is not None:
        rng = np.random.default_rng(seed)
    t0 = time.time(); n = {c.__name__: 0 for c in cases}
    said = t0
    while time.time() - t0 < seconds:
        c = cases[int(rng.integers(len(cases)))]
        c()
        n[c.__name__] += 1
        if time.time() - said > 60.0:                     # a progress line a minute (a silent run looks hung to the GPU pool's watchdog)
            said = time.time()
            print('fuzz: %4.0f s, %d cases so far' % (said - t0, sum(n.values())), flush=True)
    return n, dict(checked)


if __name__ == '__main__':
    try:
        n, chk = run(budget)
    except AssertionError as e:
        print('FAIL', e)
        sys.exit(1)
    print('fuzz ok: cases', n, 'slab comparisons', chk)
