#!/usr/bin/env python3
"""Persistent single-read Keff kernel (xc_keffp.hip) against the two-pass path and the oracle on a spread of shapes /
flags, then a timing of both schedules on cfg2-sized slabs.
    python tools/gpu_persist_check.py [--time-only] [--slabs 16]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import xcontour_oracle as O
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included, cartesian_metrics

ctx = nat.Context(0)


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert np.array_equal(np.isnan(a), np.isnan(b)), 'NaN pattern'
    m = np.isfinite(b)
    return float(np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300))) if m.any() else 0.0


def one(ny, nx, N, S, dt, cd, inc, lt, rule, dar, periodic=True, nan=False, seed=0):
    rng = np.random.default_rng(seed)
    lat = np.linspace(-88, 88, ny); lon = np.arange(nx) * (360.0 / nx)
    dA2 = cell_area(lat, lon) * (1 + 0.1 * rng.random((ny, nx)))
    dA = {'none': None, 'row': dA2[:, 0].copy(), 'plane': dA2, 'slab': dA2[None] * (1 + 0.05 * rng.random((S, 1, 1)))}[dar]
    q = (np.sin(np.deg2rad(lat))[None, :, None] * (1 + 0.3 * rng.random((S, 1, 1))) + 0.05 * rng.standard_normal((S, ny, nx))).astype(dt)
    if nan:
        q[0, 5:9, 10:40] = np.nan
        q[-1, :, 3] = np.nan
    ylt = lt if inc else (not lt)
    rows = (np.ones((ny, nx)) if dA is None else (dA2 if dar != 'row' else np.repeat(dA[:, None], nx, 1))).sum(1)
    tbl = table_from_rowsums(rows, ylt, last_row_included(lat, rule))
    kw = dict(dA=dA, tbl=tbl, tbl_coord=lat, increase=inc, lt=lt, right_edge=rule, periodic_x=periodic)
    if periodic:
        kw.update(lat=lat, lon=lon)
    else:
        rdx, rdy = cartesian_metrics(lat, 2.0)
        kw.update(rdx=rdx, rdy=rdy)
    res = {}
    for mode in (nat.XC_KEFF_PERSISTENT, nat.XC_KEFF_TWO_PASS):
        ctx.set_keff_mode(mode)
        plan = KeffPlan(ctx, S, ny, nx, N, dt, cd, **kw)
        plan.set_q(q); plan.run()
        try:
            res[mode] = plan.fetch()
        except Exception as e:
            res[mode] = str(e)
        path = ctx.last_keff_path()
        plan.free()
        assert path == (1 if mode == nat.XC_KEFF_PERSISTENT else 0), ('path', mode, path)
    ctx.set_keff_mode(nat.XC_KEFF_AUTO)
    a, b = res[nat.XC_KEFF_PERSISTENT], res[nat.XC_KEFF_TWO_PASS]
    if isinstance(a, str) or isinstance(b, str):
        assert a == b, (a, b)
        return
    assert np.array_equal(a['ctr'], b['ctr']), 'ctr'
    assert np.array_equal(a['counts'], b['counts']), 'counts vs two-pass'
    assert rel(a['area'], b['area']) < 1e-12 and rel(a['intgrdS'], b['intgrdS']) < 1e-11, ('sums', rel(a['area'], b['area']), rel(a['intgrdS'], b['intgrdS']))
    for k in ('latEq', 'nkeff', 'Leq2'):
        assert rel(a[k], b[k]) < 1e-6, k
    # and the oracle on one slab (periodic sphere cases)
    if periodic and dar in ('plane', 'row', 'slab'):
        s = S - 1
        dAs = dA if dar == 'plane' else (np.repeat(dA[:, None], nx, 1) if dar == 'row' else dA[s])
        r = O.keff_pipeline(q[s], dAs, lat, N, lon=lon, increase=inc, lt=lt, dtype=cd, right_edge=rule)
        assert np.array_equal(a['counts'][s].astype(np.int64), r['counts']), 'counts vs oracle'
        assert rel(a['area'][s], r['area']) < 1e-11 and rel(a['intgrdS'][s], r['intgrdS']) < 1e-10, 'sums vs oracle'


def check():
    n = 0
    for (ny, nx) in ((361, 720), (300, 250), (721, 1440), (181, 2000), (1801, 3600)):
        for S in (1, 2, 9) if ny < 1000 else (3,):
            for dt, cd in ((np.float64, np.float64), (np.float32, np.float32), (np.float64, np.float32)):
                for dar in ('plane', 'row', 'slab', 'none'):
                    inc, lt = bool(n & 1), bool(n & 2)
                    rule = 'numpy' if n % 3 == 0 else 'xhistogram'
                    periodic = n % 5 != 0
                    N = (201, 121, 61, 33, 500)[n % 5]
                    if ny >= 1000 and (dar in ('slab', 'none') or dt == np.float32):
                        n += 1
                        continue
                    one(ny, nx, N, S, dt, cd, inc, lt, rule, dar, periodic, nan=(n % 4 == 1), seed=n)
                    n += 1
    print('persistent == two-pass on %d configurations' % n)
    # slabs the persistent kernel must hand to the two-pass path (status 3) or flag (status 1)
    ny, nx, N, S = 300, 400, 41, 4
    lat = np.linspace(-80, 80, ny); lon = np.arange(nx) * 0.9
    dA = cell_area(lat, lon); tbl = table_from_rowsums(dA.sum(1), True)
    rng = np.random.default_rng(3)
    q = np.sin(np.deg2rad(lat))[None, :, None] + 0.05 * rng.standard_normal((S, ny, nx))
    q[1] = 300.0 + 1e-5 * q[1]                 # float32 contours of 300 +- 1e-5: levels collapse -> 'non monotonic bins'
    q[2, 5, 5] = np.inf                        # infinite maximum: levels inf / nan
    q[3] = np.nan                              # nothing valid
    for cd in (np.float32, np.float64):
        res = []
        for mode in (nat.XC_KEFF_PERSISTENT, nat.XC_KEFF_TWO_PASS):
            ctx.set_keff_mode(mode)
            plan = KeffPlan(ctx, S, ny, nx, N, np.float64, cd, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True)
            plan.set_q(q); plan.run(); res.append(plan.fetch(check=False)); plan.free()
        ctx.set_keff_mode(nat.XC_KEFF_AUTO)
        a, b = res
        assert np.array_equal(a['status'], b['status']), (a['status'], b['status'])
        assert np.array_equal(a['ctr'], b['ctr'], equal_nan=True) and np.array_equal(a['counts'], b['counts'])
        ok = a['status'] == 0
        assert rel(a['area'][ok], b['area'][ok]) < 1e-12
        print('degenerate slabs, ctr %s: status %s, counts sums %s' % (np.dtype(cd).name, a['status'], a['counts'].sum(1)))


def timing(S, ny=1801, nx=3600):
    N = 201
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * (360.0 / nx)
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True)
    e0, e1 = ctx.event(), ctx.event()
    for rep_dA, name in ((False, 'plane dA'), (True, 'per-slab dA')):
        plan = KeffPlan(ctx, 2 * S, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                        increase=True, lt=True, replicate_dA=rep_dA, out_slabs=S)
        plan.synth(lat, lon, 20241008, 0)
        for mode, mname in ((nat.XC_KEFF_PERSISTENT, 'persistent'), (nat.XC_KEFF_TWO_PASS, 'two-pass')):
            ctx.set_keff_mode(mode)
            for k in range(3):
                plan.run_range(0, (k % 2) * S, S, out_s0=0)
            ctx.sync()
            ctx.record(e0)
            K = 10
            for k in range(K):
                plan.run_range(0, (k % 2) * S, S, out_s0=0)
            ctx.record(e1)
            ms = ctx.elapsed_ms(e0, e1) / K
            out = plan.fetch(check=False)
            print('%dx%d %-12s %-10s %d slabs/launch: %.3f ms per launch = %.2f us/slab  (counts ok: %s, path %d)'
                  % (ny, nx, name, mname, S, ms, ms * 1e3 / S, bool((out['counts'][:S].sum(1) == ny * nx).all()), ctx.last_keff_path()), flush=True)
        ctx.set_keff_mode(nat.XC_KEFF_AUTO)
        plan.free()


def stamps(S, ny=1801, nx=3600):
    N = 201
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.1
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True)
    plan = KeffPlan(ctx, S, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True)
    plan.synth(lat, lon, 20241008, 0)
    slots = int(os.environ.get('XC_PERSIST_SLOTS', '1'))
    nb = ctx.device_cus() * slots                  # workgroups of the launch (the kernel indexes the stamps by blockIdx)
    buf = ctx.alloc(S * nb * 8 * 8)
    ctx.set_keff_mode(nat.XC_KEFF_PERSISTENT)
    for k in range(3):
        plan.run()
    ctx._check(ctx.lib.xc_dbg_set_stamps(ctx.handle, buf.ptr))
    plan.run(); ctx.sync()
    ctx._check(ctx.lib.xc_dbg_set_stamps(ctx.handle, None))
    st = buf.download((S, nb, 8), np.uint64).astype(np.int64)
    live = st[:, :, 0] > 0                        # a workgroup stamps the slabs of its own group only
    t0 = st[:, :, 0][live].min()
    st = (st - t0) / 100.0                     # us
    names = ['top', 'minmax+barrier', 'publish issued', 'flushed prev', 'sync done', 'edges done', 'B done']
    for s in (0, 1, 2, S // 2, S - 1):
        m = live[s]
        print('slab %2d (%d workgroups):' % (s, m.sum()), '  '.join('%s %.1f/%.1f/%.1f' % (names[k], st[s, m, k].min(), np.median(st[s, m, k]), st[s, m, k].max()) for k in range(7)))
    d = np.diff(st, axis=2)
    sel = live.copy(); sel[:8] = False
    print('median phase lengths (us) over the later slabs: ', '  '.join('%s->%s %.2f' % (names[k], names[k + 1], np.median(d[:, :, k][sel])) for k in range(6)))
    per = []
    for w in range(nb):
        ss = np.nonzero(live[:, w])[0]
        if len(ss) > 3:
            per.append(np.median(np.diff(st[ss[2:], w, 0])))
    ng = int(np.median([np.median(np.diff(np.nonzero(live[:, w])[0])) for w in range(nb) if live[:, w].sum() > 1]))
    print('groups %d; period of a group (us): %.2f -> %.2f us per slab;  whole launch %.1f us for %d slabs = %.2f us per slab'
          % (ng, np.median(per), np.median(per) / ng, st[:, :, 6][live].max(), S, st[:, :, 6][live].max() / S))
    plan.free()


if __name__ == '__main__':
    if '--stamps' in sys.argv:
        if '--cfg4' in sys.argv:
            stamps(64, 721, 1440)
        else:
            stamps(16)
        sys.exit(0)
    S = 16
    if '--slabs' in sys.argv:
        S = int(sys.argv[sys.argv.index('--slabs') + 1])
    if '--time-only' not in sys.argv:
        check()
    if '--cfg4' in sys.argv:
        timing(S, 721, 1440)
    else:
        timing(S)
