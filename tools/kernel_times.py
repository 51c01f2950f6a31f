#!/usr/bin/env python3
"""Device-resident kernel timings on one MI355X (HIP events on the library's stream), one JSON line per measurement:

    python tools/kernel_times.py sort            K8: 6.48 M (f64, f64) / (f32, f64) pairs, spike + tie inputs, stacks of small planes
    python tools/kernel_times.py lwa             K7: cfg3 (barotropic 256x512, J = 256), stacks of it, one cfg2-sized slab
    python tools/kernel_times.py cross           K9: cfg2-sized slabs, strides, field variants
    python tools/kernel_times.py pipe            Keff pipeline: tracer / contour dtypes, supplied grdS, chained or not, deterministic
    python tools/kernel_times.py land            Keff pipeline on ocean-like slabs: ~30 % of the cells NaN (whole rows + a continent)
    python tools/kernel_times.py ncontours       Keff pipeline time per cfg2 slab against the number of contours
    python tools/kernel_times.py shapes          every kernel on awkward shapes (per-cell cost; catches pathological regimes)
    python tools/kernel_times.py single          every operator on ONE cfg2 slab next to its per-slab time in a 16-slab launch

The oracle is imported only by the sub-commands that spot-check a result (it is test infrastructure).
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat                                   # noqa: E402
from xcontour_amd.pipeline import KeffPlan                                # noqa: E402
from xcontour_amd.utils import cell_area, table_from_rowsums              # noqa: E402

if os.environ.get('XC_LIB'):
    nat.LIB_PATH = os.path.join(ROOT, 'xcontour_amd', os.environ['XC_LIB'])      # a diagnostic build (tools/build_variant.sh)
NY, NX, N = 1801, 3600, 201


def oracle():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import xcontour_oracle as O
    return O


class Timer(object):
    def __init__(self, ctx):
        self.ctx, self.e0, self.e1 = ctx, ctx.event(), ctx.event()

    def ms(self, fn, reps=10, warm=2):
        for _ in range(warm):
            fn()
        self.ctx.sync()
        self.ctx.record(self.e0)
        for _ in range(reps):
            fn()
        self.ctx.record(self.e1)
        return self.ctx.elapsed_ms(self.e0, self.e1) / reps


def emit(**kw):
    print(json.dumps(kw), flush=True)


def grid():
    lat = np.linspace(-90, 90, NY)
    lon = np.arange(NX) * 0.1
    return lat, lon, cell_area(lat, lon)


# ----------------------------------------------------------------------------------------------------------------- K8
def sort_call(ctx, dq, dt, S, ny, nx, dA=None, nv=None, mask=None):
    ctx._check(ctx.lib.xc_sort_profile_batch_dev(ctx.handle, dq.ptr, nat.dtype_code(dt), None if mask is None else mask.ptr, nat.XC_F64, 0,
                                                 None if dA is None else dA.ptr, nat.XC_DA_NONE if dA is None else nat.XC_DA_PLANE,
                                                 S, ny, nx, 0, None, 0, None, None, 0, None, None, None, nv.ptr, None))


def cmd_sort(ctx, T):
    rng = np.random.default_rng(0)
    n = NY * NX
    nv = ctx.alloc(4096)
    fields = {
        'normal noise': rng.standard_normal((NY, NX)),
        'PV-like (sin lat + noise)': np.sin(np.deg2rad(np.linspace(-90, 90, NY)))[:, None] + 0.02 * rng.standard_normal((NY, NX)),
        'ties: 1000 distinct values': rng.integers(0, 1000, (NY, NX)).astype(np.float64),
        'spike: 1e-12 noise around 1 and two outliers': 1.0 + 1e-12 * rng.standard_normal((NY, NX)),
    }
    fields['spike: 1e-12 noise around 1 and two outliers'][0, 0] = -5.0
    fields['spike: 1e-12 noise around 1 and two outliers'][1, 1] = 7.0
    dA = ctx.to_device(np.ones((NY, NX)))
    if os.environ.get('XC_SORT_ONLY'):
        fields = {k: v for k, v in list(fields.items())[:1]}
    for name, q in fields.items():
        for dt in (np.float64, np.float32):
            dq = ctx.to_device(q.astype(dt))
            ms = T.ms(lambda: sort_call(ctx, dq, dt, 1, NY, NX, dA, nv), reps=5)
            emit(kernel='K8 sort_profile', field=name, dtype=np.dtype(dt).name, pairs=n, ms=ms, gpairs_per_s=n / ms / 1e6, path=ctx.last_sort_path())
            dq.free()
    for (S, ny, nx) in (() if os.environ.get('XC_SORT_ONLY') else ((3, 100, 4480), (16, 256, 512), (64, 256, 512), (1, 100, 4480))):
        q = rng.standard_normal((S, ny, nx))
        dq = ctx.to_device(q)
        ms = T.ms(lambda: sort_call(ctx, dq, np.float64, S, ny, nx, None, nv), reps=10)
        emit(kernel='K8 sort_profile_batch', planes=S, shape=[ny, nx], dtype='float64', ms=ms, us_per_plane=ms / S * 1e3, path=ctx.last_sort_path())
        dq.free()


# ----------------------------------------------------------------------------------------------------------------- K7
def cmd_lwa(ctx, T):
    O = oracle()
    g = os.path.join(ROOT, 'tests', 'golden')
    q = np.load(g + '/baro_q.npy'); lat = np.load(g + '/baro_lat.npy'); lon = np.load(g + '/baro_lon.npy')
    L = np.load(g + '/baro_lwa_N121.npz')
    dA = O.cell_area(lat, lon)
    dc, dd, dM = ctx.to_device(lat.astype(np.float64)), ctx.to_device(dA), ctx.to_device(L['dy'])
    for S in (1, 8, 64):
        dq = ctx.to_device(np.repeat(q[None], S, 0)); dQ = ctx.to_device(np.repeat(L['Q'][None], S, 0))
        out = ctx.alloc(S * q.size * 8)
        fn = lambda: ctx._check(ctx.lib.xc_lwa_dev(ctx.handle, dq.ptr, nat.XC_F32, dQ.ptr, dc.ptr, dd.ptr, nat.XC_DA_PLANE, float(dA.max()),
                                                  dM.ptr, nat.XC_DA_ROW, S, 256, 512, 1, 0, 0, None, 0, out.ptr, None))
        ms = T.ms(fn, reps=20, warm=3)
        got = out.download((S, 256, 512), np.float64)
        ref = O.cal_local_wave_activity(q, L['Q'], lat, dA, True, 'all', metric=L['dy'])
        emit(kernel='K7 lwa', config='cfg3 barotropic 256x512 f32, J = 256', slabs=S, us_per_call=ms * 1e3, us_per_slab=ms / S * 1e3,
             cell_rows_per_s=S * 256 * 256 * 512 / ms * 1e3, bit_identical_to_oracle=bool(all(np.array_equal(got[s], ref) for s in range(S))))
        if S == 1:
            # the same plane through the interval kernel with its premises vouched for (mode 3: what `exact=False` does after looking
            # at Q on the host): one launch
            ctx._check(ctx.lib.xc_set_lwa_exact(ctx.handle, 3))
            ms3 = T.ms(fn, reps=20, warm=3)
            got3 = out.download((S, 256, 512), np.float64)
            ctx._check(ctx.lib.xc_set_lwa_exact(ctx.handle, 0))
            emit(kernel='K7 lwa', config='cfg3 barotropic 256x512 f32, J = 256, interval kernel (exact=False)', slabs=S, us_per_call=ms3 * 1e3,
                 max_err_over_max_value=float(np.abs(got3[0] - ref).max() / np.abs(ref).max()))
        for b in (dq, dQ, out):
            b.free()
    # one cfg2-sized slab, all J = 1801 target rows; six rows spot-checked bit for bit against the formula (core.py:752-789)
    lat, lon, dA = grid()
    qb = ctx.alloc(NY * NX * 8)
    lb_, lo_ = ctx.to_device(lat), ctx.to_device(lon)
    ctx._check(ctx.lib.xc_synth_dev(ctx.handle, qb.ptr, nat.XC_F64, 1, NY, NX, lb_.ptr, lo_.ptr, 20241008, int(os.environ.get('XC_VARIANT', '0'))))
    ctx.sync()
    q = qb.download((NY, NX), np.float64)
    Q = np.sort(q.mean(axis=1))
    dy = np.gradient(np.deg2rad(lat)) * 6371200.0
    dQ, dc, dd, dM = ctx.to_device(Q), ctx.to_device(lat), ctx.to_device(dA), ctx.to_device(dy)
    out = ctx.alloc(NY * NX * 8)
    dmax = float(dA.max())             # (outside the timed call: the max of a 52 MB host array is a millisecond of numpy)
    fn = lambda: ctx._check(ctx.lib.xc_lwa_dev(ctx.handle, qb.ptr, nat.XC_F64, dQ.ptr, dc.ptr, dd.ptr, nat.XC_DA_PLANE, dmax,
                                              dM.ptr, nat.XC_DA_ROW, 1, NY, NX, 1, 0, 0, None, 0, out.ptr, None))
    wei = dA / dA.max()
    rows = {}
    for j in (0, 300, 900, 901, 1500, 1800):
        qe = q - Q[j]
        m = (lat >= lat[j])[:, None]
        mask3 = np.where(np.logical_and(qe < 0, m), 1, np.where(m, 0, np.where(qe > 0, -1, 0))).astype(np.float64)
        rows[j] = -np.nansum(qe * mask3 * wei * dy[:, None], axis=0)
    for exact in (0, 1):                                           # the O(ny log ny) interval kernel (default for ny > 512), then the band walk
        ctx._check(ctx.lib.xc_set_lwa_exact(ctx.handle, exact))
        ms = T.ms(fn, reps=10, warm=2)
        got = out.download((NY, NX), np.float64)
        path = ctx.last_lwa_path()
        scale = max(float(np.abs(r).max()) for r in rows.values())
        err = max(float(np.abs(got[j] - r).max()) for j, r in rows.items()) / scale
        emit(kernel='K7 lwa', config='one cfg2-sized f64 slab, J = 1801', path={0: 'band walk (bit-exact)', 1: 'interval kernel', 2: 'check failed -> band walk'}[path],
             ms=ms, cell_rows_per_s=NY * NY * NX / ms * 1e3, six_rows_bit_identical=bool(all(np.array_equal(got[j], r) for j, r in rows.items())),
             six_rows_max_err_over_max_value=err)
    ctx._check(ctx.lib.xc_set_lwa_exact(ctx.handle, 0))


# ----------------------------------------------------------------------------------------------------------------- K9
def cmd_cross(ctx, T):
    O = oracle()
    S = int(os.environ.get('XC_SLABS', '16'))
    lat, lon, dA = grid()
    q = ctx.alloc(S * NY * NX * 8)
    lat_b, lon_b, dA_b = ctx.to_device(lat), ctx.to_device(lon), ctx.to_device(dA)
    out_l, out_c = ctx.alloc(S * N * 8), ctx.alloc(S * N * 8)
    for var in (0, 1, 2):                       # 0 PV-like + grid-scale noise, 1 white noise, 2 sin(lat) (smooth)
        ctx._check(ctx.lib.xc_synth_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY, NX, lat_b.ptr, lon_b.ptr, 20241008, var))
        ctx.sync()
        qh = q.download((S, NY, NX), np.float64)
        ctr, _, _ = ctx.levels(ctx.minmax(qh), np.float64, N, True, np.float64)
        ctr_b = ctx.to_device(ctr)
        for stride in [int(t) for t in os.environ.get('XC_STRIDES', '1,2,4,8,16,32').split(',')]:
            if var != 0 and stride > 1:
                continue
            for want_cnt in (1, 0):
                fn = lambda: ctx._check(ctx.lib.xc_crossing_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY, NX, stride, nat.XC_PAD_WRAP, ctr_b.ptr, N, 1,
                                                              dA_b.ptr, nat.XC_F64, 0, stride, 1, out_l.ptr, out_c.ptr if want_cnt else None))
                ms = T.ms(fn, reps=5)
                Jn, In = O.crossing_shape(NY, NX + stride, stride)
                cells = (Jn - 1) * stride * (In - 1) * stride
                by = cells * 8 + (Jn - 1) * (In - 1) * 8
                rec = dict(kernel='K9 crossing', variant=var, counts=bool(want_cnt), stride=stride, slabs=S, us_per_slab=ms / S * 1e3,
                           algorithmic_GBps=by * S / ms / 1e6, crossed_boxes_per_slab=float(out_c.download((S, N), np.uint64).sum() / S))
                if stride == 1 and want_cnt and os.environ.get('XC_CPU', '1') == '1':
                    t = time.perf_counter()
                    ol, oc = O.contour_crossing(O.pad_x(qh[0], 1, 'wrap'), ctr[0], O.pad_x(dA, 1, 'wrap'), 1, True)
                    rec['cpu_oracle_s_per_slab'] = time.perf_counter() - t
                    rec['counts_equal_oracle'] = bool(np.array_equal(out_c.download((S, N), np.uint64)[0].astype(np.int64), oc))
                    rec['len_rel_err'] = float(np.max(np.abs(out_l.download((S, N), np.float64)[0] - ol) / np.maximum(ol, 1)))
                emit(**rec)
        ctr_b.free()


# ----------------------------------------------------------------------------------------------------------------- Keff pipeline variants
def pipe_time(ctx, T, B, chain, **kw):
    plan = KeffPlan(ctx, 2 * B, NY, NX, kw.pop('N', N), kw.pop('dt', np.float64), kw.pop('cd', np.float64), out_slabs=B, **kw)
    lat, lon, _ = grid()
    plan.synth(lat, lon, 1, kw.get('variant', 0))
    if plan.grdS_buf is not None:
        ctx._check(ctx.lib.xc_memset(ctx.handle, plan.grdS_buf.ptr, 0, plan.grdS_buf.nbytes))
    k = [0]

    def step():
        s0 = (k[0] % 2) * B
        plan.run_range(0, s0, B, ((k[0] + 1) % 2) * B if chain else None, out_s0=0)
        k[0] += 1
    ms = T.ms(step, reps=10, warm=4)
    plan.free()
    return ms / B * 1e3


def cmd_pipe(ctx, T):
    lat, lon, dA = grid()
    tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True)
    base = dict(dA=dA, tbl=tbl, tbl_coord=lat, increase=True, lt=True)
    B = 32
    for dt, cd in ((np.float64, np.float64), (np.float32, np.float32), (np.float32, np.float64)):
        for supplied in (False, True):
            for det in (False, True):
                kw = dict(base, dt=dt, cd=cd, deterministic=det)
                if supplied:
                    kw['grdS_dtype'] = dt
                else:
                    kw.update(lat=lat, lon=lon)
                for chain in (False, True):
                    emit(kernel='Keff pipeline', tracer=np.dtype(dt).name, contours=np.dtype(cd).name, gradient='supplied grdS' if supplied else 'in-kernel',
                         deterministic=det, chained=chain, slabs_per_launch=B, us_per_slab=pipe_time(ctx, T, B, chain, **kw))


def cmd_land(ctx, T):
    """rows that are entirely NaN used to send 64 * VEC cells x 3 LDS adds onto the few addresses of the trash bin"""
    lat, lon, dA = grid()
    tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True)
    B = 8
    rng = np.random.default_rng(2)
    base = np.sin(np.deg2rad(lat))[:, None] + 0.25 * np.cos(3 * np.deg2rad(lon))[None, :] * np.cos(np.deg2rad(lat))[:, None] ** 2
    for frac_name, land in (('no land', None), ('30 % land (500 whole rows + a 600 x 1000 continent)', True)):
        q = np.stack([base + 0.02 * rng.standard_normal((NY, NX)) for _ in range(2 * B)])
        if land:
            q[:, 100:600, :] = np.nan
            q[:, 900:1500, 1000:2000] = np.nan
        plan = KeffPlan(ctx, 2 * B, NY, NX, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True, out_slabs=B)
        plan.set_q(q)
        for chain in (False, True):
            k = [0]

            def step():
                s0 = (k[0] % 2) * B
                plan.run_range(0, s0, B, ((k[0] + 1) % 2) * B if chain else None, out_s0=0)
                k[0] += 1
            emit(kernel='Keff pipeline', field=frac_name, nan_fraction=float(np.isnan(q).mean()), chained=chain, slabs_per_launch=B,
                 us_per_slab=T.ms(step, reps=10, warm=4) / B * 1e3)
        plan.free()


def cmd_ncontours(ctx, T):
    lat, lon, dA = grid()
    tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True)
    for n in (3, 11, 41, 201, 501, 1001, 3000, 6000):
        emit(kernel='Keff pipeline', contours=n, slabs_per_launch=16,
             us_per_slab=pipe_time(ctx, T, 16, True, N=n, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True))


def hist_desc(dq, S, ny, nx, de, dA, cdf):
    d = nat.HistDesc()
    d.q, d.q_dtype, d.nslab, d.ny, d.nx = dq.ptr, nat.XC_F64, S, ny, nx
    d.edges, d.nedge, d.edges_per_slab, d.last_closed = de.ptr, N + 1, 1, 1
    d.dA, d.dA_rank, d.lt, d.cdf = dA.ptr, nat.XC_DA_PLANE, 1, cdf.ptr
    return d


def cmd_shapes(ctx, T):
    rng = np.random.default_rng(0)
    for (S, ny, nx) in [(4, 1801, 3600), (4, 3600, 1801), (64, 256, 512), (1, 64800, 100), (1, 100, 64800), (512, 90, 180), (2, 6000, 6000), (1, 3, 2000000)]:
        cells = S * ny * nx
        q = np.sin(np.linspace(-1.5, 1.5, ny))[None, :, None] + 0.01 * rng.standard_normal((S, ny, nx))
        dq, dA = ctx.to_device(q), ctx.to_device(np.ones((ny, nx)))
        ctr, edges, _ = ctx.levels(ctx.minmax(q), np.float64, N, True, np.float64)
        de, cdf, mmb = ctx.to_device(edges), ctx.alloc(S * N * 8), ctx.alloc(S * 16)
        d = hist_desc(dq, S, ny, nx, de, dA, cdf)
        dc, ol, oc = ctx.to_device(ctr), ctx.alloc(S * N * 8), ctx.alloc(S * N * 8)
        rec = dict(shape=[S, ny, nx], cells=cells,
                   minmax_ns_per_kcell=T.ms(lambda: ctx._check(ctx.lib.xc_minmax_dev(ctx.handle, dq.ptr, nat.XC_F64, S, ny * nx, mmb.ptr)), 5) / cells * 1e9,
                   hist_ns_per_kcell=T.ms(lambda: ctx._check(ctx.lib.xc_hist_dev(ctx.handle, C.byref(d))), 5) / cells * 1e9,
                   crossing_ns_per_kcell=T.ms(lambda: ctx._check(ctx.lib.xc_crossing_dev(ctx.handle, dq.ptr, nat.XC_F64, S, ny, nx, 1, nat.XC_PAD_WRAP, dc.ptr, N, 1,
                                                                                         dA.ptr, nat.XC_F64, 0, 1, 1, ol.ptr, oc.ptr)), 5) / cells * 1e9)
        if ny <= 6000 and S * ny * ny * nx < 3e11:
            Q, co, out = ctx.to_device(np.sort(q.mean(axis=2), axis=1)), ctx.to_device(np.linspace(-80, 80, ny)), ctx.alloc(cells * 8)
            rec['lwa_ms'] = T.ms(lambda: ctx._check(ctx.lib.xc_lwa_dev(ctx.handle, dq.ptr, nat.XC_F64, Q.ptr, co.ptr, dA.ptr, nat.XC_DA_PLANE, 1.0,
                                                                         None, nat.XC_DA_NONE, S, ny, nx, 1, 0, 0, None, 0, out.ptr, None)), 2)
            for b in (Q, co, out):
                b.free()
        if ny * nx < 2 ** 31 and cells * 40 < 20e9:
            nv = ctx.alloc(S * 4)
            rec['sort_ns_per_kcell'] = T.ms(lambda: sort_call(ctx, dq, np.float64, S, ny, nx, None, nv), 3) / cells * 1e9
            nv.free()
        emit(**rec)
        for b in (dq, dA, de, cdf, mmb, dc, ol, oc):
            b.free()


def cmd_single(ctx, T):
    lat, lon, dA = grid()
    tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True)
    out = {}
    for S in (1, 16):
        plan = KeffPlan(ctx, S, NY, NX, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True)
        plan.synth(lat, lon, 1, 2)
        q = plan.q_buf
        mmb = ctx.alloc(S * 16)
        out['minmax', S] = T.ms(lambda: ctx._check(ctx.lib.xc_minmax_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY * NX, mmb.ptr)))
        out['keff pipeline', S] = T.ms(lambda: plan.run())
        ctr, edges, _ = ctx.levels(ctx.minmax(plan.download_q()), np.float64, N, True, np.float64)
        de, dAd, cdf = ctx.to_device(edges), ctx.to_device(dA), ctx.alloc(S * N * 8)
        d = hist_desc(q, S, NY, NX, de, dAd, cdf)
        out['hist (1 channel)', S] = T.ms(lambda: ctx._check(ctx.lib.xc_hist_dev(ctx.handle, C.byref(d))))
        dc, ol = ctx.to_device(ctr), ctx.alloc(S * N * 8)
        out['crossing stride 1', S] = T.ms(lambda: ctx._check(ctx.lib.xc_crossing_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY, NX, 1, nat.XC_PAD_WRAP, dc.ptr, N, 1,
                                                                                     dAd.ptr, nat.XC_F64, 0, 1, 1, ol.ptr, None)))
        g2, rdx, rdy = ctx.alloc(S * NY * NX * 8), ctx.to_device(np.ones(NY)), ctx.to_device(np.ones(NY))
        out['grad2', S] = T.ms(lambda: ctx._check(ctx.lib.xc_grad2_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY, NX, rdx.ptr, rdy.ptr, 1, g2.ptr)))
        nv = ctx.alloc(S * 4)
        out['sort', S] = T.ms(lambda: sort_call(ctx, q, np.float64, S, NY, NX, dAd, nv), 3)
        plan.free()
        for b in (mmb, de, dAd, cdf, dc, ol, g2, rdx, rdy, nv):
            b.free()
    for name in sorted(set(k[0] for k in out)):
        emit(operator=name, one_slab_us=out[name, 1] * 1e3, per_slab_of_16_us=out[name, 16] / 16 * 1e3)


CMDS = {'sort': cmd_sort, 'lwa': cmd_lwa, 'cross': cmd_cross, 'pipe': cmd_pipe, 'land': cmd_land, 'ncontours': cmd_ncontours, 'shapes': cmd_shapes, 'single': cmd_single}

if __name__ == '__main__':
    if len(sys.argv) < 2 or any(c not in CMDS for c in sys.argv[1:]):
        raise SystemExit(__doc__)
    ctx_ = nat.Context(0)
    for c in sys.argv[1:]:
        CMDS[c](ctx_, Timer(ctx_))
    ctx_.close()
