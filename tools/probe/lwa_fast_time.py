#!/usr/bin/env python3
"""K7F alone: one cfg2-sized f64 slab (J = 1801) through the interval kernel, premises vouched for (mode 3: ONE launch), and
through the default path (check + interval kernel + gated band walk).  Prints ms per call and the error of six rows against
the formula of core.py:752-789 (numpy).   python tools/probe/lwa_fast_time.py [slabs]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from xcontour_amd import _native as nat
from kernel_times import Timer, grid, NY, NX
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = nat.Context(0); T = Timer(ctx)
lat, lon, dA = grid()
qb = ctx.alloc(S * NY * NX * 8)
lb_, lo_ = ctx.to_device(lat), ctx.to_device(lon)
ctx._check(ctx.lib.xc_synth_dev(ctx.handle, qb.ptr, nat.XC_F64, S, NY, NX, lb_.ptr, lo_.ptr, 20241008, int(os.environ.get('XC_VARIANT', '0'))))
ctx.sync()
q = qb.download((S, NY, NX), np.float64)
Q = np.sort(q.mean(axis=2), axis=1)
dy = np.gradient(np.deg2rad(lat)) * 6371200.0
dQ, dc, dd, dM = ctx.to_device(Q), ctx.to_device(lat), ctx.to_device(dA), ctx.to_device(dy)
out = ctx.alloc(S * NY * NX * 8)
dmax = float(dA.max())
fn = lambda: ctx._check(ctx.lib.xc_lwa_dev(ctx.handle, qb.ptr, nat.XC_F64, dQ.ptr, dc.ptr, dd.ptr, nat.XC_DA_PLANE, dmax,
                                          dM.ptr, nat.XC_DA_ROW, S, NY, NX, 1, 0, 0, None, 0, out.ptr, None))
wei = dA / dA.max()
rows = {}
for j in (0, 300, 900, 901, 1500, 1800):
    qe = q[S - 1] - Q[S - 1][j]
    m = (lat >= lat[j])[:, None]
    mask3 = np.where(np.logical_and(qe < 0, m), 1, np.where(m, 0, np.where(qe > 0, -1, 0))).astype(np.float64)
    rows[j] = -np.nansum(qe * mask3 * wei * dy[:, None], axis=0)
scale = max(float(np.abs(r).max()) for r in rows.values())
for mode in (3, 0):
    ctx._check(ctx.lib.xc_set_lwa_exact(ctx.handle, mode))
    ms = T.ms(fn, reps=20, warm=3)
    got = out.download((S, NY, NX), np.float64)[S - 1]
    err = max(float(np.abs(got[j] - r).max()) for j, r in rows.items()) / scale
    print(json.dumps({'mode': mode, 'slabs': S, 'ms_per_call': ms, 'us_per_slab': ms / S * 1e3, 'six_rows_max_err_over_max_value': err,
                      'GBps_of_24B_per_cell': S * NY * NX * 24 / ms / 1e6}), flush=True)
