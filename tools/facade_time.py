#!/usr/bin/env python3
"""Wall time of the facade (numpy in / numpy out, host entry points, PCIe included) on one cfg2-sized
slab (and, with XC_FACADE_SMALL=1, on the 15 x 241 x 480 float32 stack of the reference's own demo size, cfg1): the reference's Keff call sequence (tests/test_Keff_atmos.py:75-92) call by call, and the fused
`Contour2D.keff`.  Prints a JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xcontour_amd as xa   # noqa: E402

NY, NX, N = 1801, 3600, 201
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
rng = np.random.default_rng(0)
q = np.sin(np.deg2rad(lat))[:, None] + 0.05 * rng.standard_normal((NY, NX))
c = {'lat': lat, 'lon': lon}
tr = xa.DataArray(q, ('lat', 'lon'), c, 'pv')
dA = xa.DataArray(xa.cell_area(lat, lon), ('lat', 'lon'), c, 'dA')
g2 = xa.DataArray(rng.random((NY, NX)), ('lat', 'lon'), c, 'grdS')
mask = xa.DataArray(np.ones((NY, NX)), ('lat', 'lon'), c, 'mask')
kw = dict(dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, dtype=np.float64)
kw.update(json.loads(os.environ.get('XC_FACADE_KW', '{}')))
cm = xa.Contour2D(tr, dA, **kw)
rec = {}


def timed(name, fn, reps=3):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    rec[name] = (time.perf_counter() - t) / reps * 1e3
    return out


table = timed('cal_area_eqCoord_table_hist', lambda: cm.cal_area_eqCoord_table_hist(mask))
ctr = timed('cal_contours', lambda: cm.cal_contours(N))
area = timed('cal_integral_within_contours_hist(area)', lambda: cm.cal_integral_within_contours_hist(ctr))
intS = timed('cal_integral_within_contours_hist(grdS)', lambda: cm.cal_integral_within_contours_hist(ctr, integrand=g2))
latEq = timed('lookup_coordinates', lambda: table.lookup_coordinates(area))
dq = timed('cal_gradient_wrt_area x2', lambda: (cm.cal_gradient_wrt_area(ctr, area), cm.cal_gradient_wrt_area(intS, area)))
timed('keff (fused, grdS supplied)', lambda: cm.keff(N, table, grdS=g2))
timed('keff (fused, in-kernel gradient)', lambda: cm.keff(N, table, lat=lat, lon=lon))
rec['sum of the call sequence'] = sum(v for k, v in rec.items() if not k.startswith('keff'))
print(json.dumps({'facade_ms_per_call_cfg2_slab': rec, 'kwargs': {k: str(v) for k, v in kw.items()}}))

# ---- a long host stack through the fused keff(): uploads in batches on the copy stream while the previous batch computes.
# Beside it: the bare hipMemcpy of the same bytes (pageable and pinned source), the floor of any host-side entry point.
S = int(os.environ.get('XC_FACADE_STACK', '0'))
if S:
    import ctypes as C
    from xcontour_amd import _native as nat
    stack = np.empty((S, NY, NX))
    for s in range(S):
        stack[s] = q * (1.0 + 0.01 * s)
    c3 = {'time': np.arange(S), 'lat': lat, 'lon': lon}
    tr3 = xa.DataArray(stack, ('time', 'lat', 'lon'), c3, 'pv')
    cm3 = xa.Contour2D(tr3, dA, **kw)
    cap = int(float(os.environ.get('XC_FACADE_CAP_GB', '4')) * (1 << 30))
    cm3.keff(N, table, lat=lat, lon=lon, max_batch_bytes=cap)                       # plan + buffers
    t = time.perf_counter()
    out = cm3.keff(N, table, lat=lat, lon=lon, max_batch_bytes=cap)
    t_keff = time.perf_counter() - t
    ctx = cm3.ctx
    dev = ctx.alloc(min(stack.nbytes, cap))
    hip = C.CDLL('libamdhip64.so')
    chunk = dev.nbytes // (NY * NX * 8)

    def bare(src):
        t0 = time.perf_counter()
        for s0 in range(0, S, chunk):
            m = min(chunk, S - s0)
            ctx._check(ctx.lib.xc_memcpy_h2d(ctx.handle, dev.ptr, src[s0:s0 + m].ctypes.data, m * NY * NX * 8))
        return time.perf_counter() - t0
    bare(stack)
    t_page = bare(stack)
    rc = hip.hipHostRegister(C.c_void_p(stack.ctypes.data), C.c_size_t(stack.nbytes), 0)
    t_pin = bare(stack) if rc == 0 else None
    if rc == 0:
        hip.hipHostUnregister(C.c_void_p(stack.ctypes.data))
    print(json.dumps({'facade_keff_stack': {'slabs': S, 'GB': stack.nbytes / 1e9, 'batch_cap_GB': cap / 2 ** 30, 'keff_s': t_keff,
                                            'keff_GBps': stack.nbytes / t_keff / 1e9, 'ms_per_slab': t_keff / S * 1e3,
                                            'bare_h2d_pageable_s': t_page, 'bare_h2d_pinned_s': t_pin,
                                            'keff_over_pinned_copy': None if not t_pin else t_keff / t_pin}}))

# ---- the reference's own problem size (cfg1: 15 levels of 241 x 480 float32): per-call latency, where fixed costs rule.
# `--breakdown`: every call's wall time split into (a) time inside the C library, call by call (ctypes wrappers around every xc_*
# function), itself split by the library's own stopwatch (xc_trace) into staging inputs / handing results over / waiting for the stream
# / everything else (argument checks, launches), and (b) the Python around it (labelled-array unwrap / wrap, numpy glue, ctypes marshalling).
class _LibTimer(object):
    """wraps every function of the ctypes library object in a stopwatch (per-name seconds and call counts)"""

    def __init__(self, lib):
        self.lib, self.t, self.n = lib, {}, {}

    def install(self, names):
        for name in names:
            fn = getattr(self.lib, name)

            def wrapped(*a, __fn=fn, __name=name):
                t0 = time.perf_counter()
                r = __fn(*a)
                self.t[__name] = self.t.get(__name, 0.0) + time.perf_counter() - t0
                self.n[__name] = self.n.get(__name, 0) + 1
                return r
            setattr(self.lib, name, wrapped)

    def reset(self):
        self.t, self.n = {}, {}


if os.environ.get('XC_FACADE_SMALL') or '--breakdown' in sys.argv:
    def small():
        rec = {}
        from xcontour_amd import _native as nat
        lt = None
        if '--breakdown' in sys.argv:
            lt = _LibTimer(nat.load())
            lt.install([n for n in nat.PROTOTYPES if n not in ('xc_last_error', 'xc_version', 'xc_trace')])

        def timed(name, fn, reps=20):
            fn(); fn()
            ctx = nat.default_context(0)
            if lt:
                lt.reset(); ctx.trace(True)
            t = time.perf_counter()
            for _ in range(reps):
                out = fn()
            wall = (time.perf_counter() - t) / reps * 1e6
            rec[name] = wall
            if lt:
                tr = ctx.trace(True)
                inlib = sum(lt.t.values()) / reps * 1e6
                rec[name] = {'us': wall, 'python_us': wall - inlib, 'library_us': inlib,
                             'library_calls': {k: round(lt.t[k] / reps * 1e6, 1) for k in sorted(lt.t, key=lambda k: -lt.t[k])},
                             'library_split_us': {'stage_inputs': tr['stage_in_s'] / reps * 1e6, 'hand_over_results': tr['hand_over_s'] / reps * 1e6,
                                                  'wait_for_stream': tr['sync_wait_s'] / reps * 1e6,
                                                  'checks_and_launches': inlib - (tr['stage_in_s'] + tr['hand_over_s'] + tr['sync_wait_s']) / reps * 1e6}}
            return out
        NL1, NY1, NX1, N1 = 15, 241, 480, 201
        lat = np.linspace(-90, 90, NY1).astype(np.float32); lon = (np.arange(NX1) * 0.75).astype(np.float32); lev = np.arange(NL1, dtype=np.float32)
        rng = np.random.default_rng(0)
        q = (np.sin(np.deg2rad(lat))[None, :, None] * (1 + 0.1 * lev[:, None, None]) + 0.05 * rng.standard_normal((NL1, NY1, NX1))).astype(np.float32)
        c3 = {'lev': lev, 'lat': lat, 'lon': lon}; c2 = {'lat': lat, 'lon': lon}
        tr = xa.DataArray(q, ('lev', 'lat', 'lon'), c3, 'pv')
        dA = xa.DataArray(xa.cell_area(lat.astype(np.float64), lon.astype(np.float64)).astype(np.float32), ('lat', 'lon'), c2, 'dA')
        g2 = xa.DataArray(rng.random(q.shape).astype(np.float32), ('lev', 'lat', 'lon'), c3, 'grdS')
        mask = xa.DataArray(np.ones((NY1, NX1), np.float32), ('lat', 'lon'), c2, 'mask')
        cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, resident=bool(kw.get('resident', False)))
        table = timed('cal_area_eqCoord_table_hist', lambda: cm.cal_area_eqCoord_table_hist(mask))
        ctr = timed('cal_contours', lambda: cm.cal_contours(N1))
        area = timed('integral_hist(area)', lambda: cm.cal_integral_within_contours_hist(ctr))
        intS = timed('integral_hist(grdS)', lambda: cm.cal_integral_within_contours_hist(ctr, integrand=g2))
        latEq = timed('lookup_coordinates', lambda: table.lookup_coordinates(area))
        dq = timed('gradient_wrt_area x2', lambda: (cm.cal_gradient_wrt_area(ctr, area), cm.cal_gradient_wrt_area(intS, area)))
        timed('keff fused (grdS supplied)', lambda: cm.keff(N1, table, grdS=g2))
        # ... and the sequence as an analysis runs it: the seven calls one after the other on a tracer that CHANGES from one pass to the
        # next (six fields round-robin, each with its own levels: the library's cache of small inputs -- four entries -- cannot serve the
        # first binning call of a pass, only the second, as in real use; the per-call rows above repeat one call on one field)
        if kw.get('resident', False):
            objs = []
            for i in range(6):
                qi = (q * np.float32(1.0 + 0.03 * i) + np.float32(0.01 * i)).astype(np.float32)
                objs.append(xa.Contour2D(xa.DataArray(qi, ('lev', 'lat', 'lon'), c3, 'pv'), dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'},
                                         increase=True, lt=True, resident=True))

            def sequence(c):
                tb = c.cal_area_eqCoord_table_hist(mask)
                ct = c.cal_contours(N1)
                ar = c.cal_integral_within_contours_hist(ct)
                iS = c.cal_integral_within_contours_hist(ct, integrand=g2)
                le = tb.lookup_coordinates(ar)
                return le, c.cal_gradient_wrt_area(ct, ar), c.cal_gradient_wrt_area(iS, ar)
            for c in objs:
                sequence(c); sequence(c)
            t = time.perf_counter()
            for _ in range(10):
                for c in objs:
                    sequence(c)
            rec['whole sequence, changing tracer'] = (time.perf_counter() - t) / 60 * 1e6
        print(json.dumps({'facade_us_per_call_cfg1_stack_15x241x480_f32': rec, 'resident': bool(kw.get('resident', False))}))
    small()
