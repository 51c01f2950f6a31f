#!/usr/bin/env python3
"""K7 at cfg2 size: LWA of one 1801 x 3600 float64 PV-like slab for all J = 1801 target rows, device
resident; six target rows are checked bit for bit against the oracle's formula (core.py:752-789)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from xcontour_amd import _native as nat
from xcontour_amd.utils import cell_area
NY, NX = 1801, 3600
ctx = nat.Context(0)
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
dA = cell_area(lat, lon)
qb = ctx.alloc(NY * NX * 8)
lb_, lo_ = ctx.to_device(lat), ctx.to_device(lon)
ctx._check(ctx.lib.xc_synth_dev(ctx.handle, qb.ptr, nat.XC_F64, 1, NY, NX, lb_.ptr, lo_.ptr, 20241008, int(os.environ.get('XC_VARIANT', '0'))))
ctx.sync()
q = qb.download((NY, NX), np.float64)
Q = np.sort(q.mean(axis=1))                      # a monotone reference profile (the zonal mean, sorted)
dy = np.gradient(np.deg2rad(lat)) * 6371200.0
dQ, dc, dd, dM = ctx.to_device(Q), ctx.to_device(lat), ctx.to_device(dA), ctx.to_device(dy)
out = ctx.alloc(NY * NX * 8)
e0, e1 = ctx.event(), ctx.event()
def run():
    ctx._check(ctx.lib.xc_lwa_dev(ctx.handle, qb.ptr, nat.XC_F64, dQ.ptr, dc.ptr, dd.ptr, nat.XC_DA_PLANE, float(dA.max()),
                                  dM.ptr, nat.XC_DA_ROW, 1, NY, NX, 1, 0, 0, None, 0, out.ptr, None))
run(); ctx.sync()
ctx.record(e0)
for _ in range(3): run()
ctx.record(e1)
ms = ctx.elapsed_ms(e0, e1) / 3
got = out.download((NY, NX), np.float64)
wei = dA / dA.max()
ok = True
t = time.perf_counter()
for j in (0, 300, 900, 901, 1500, 1800):
    qe = q - Q[j]
    m = (lat >= lat[j])[:, None]
    mask3 = np.where(np.logical_and(qe < 0, m), 1, np.where(m, 0, np.where(qe > 0, -1, 0))).astype(np.float64)
    ref = -np.nansum(qe * mask3 * wei * dy[:, None], axis=0)
    ok &= bool(np.array_equal(got[j], ref))
tc = (time.perf_counter() - t) / 6
rmn, rmx = q.min(1), q.max(1)
need = np.array([np.where(lat >= lat[j], rmn < Q[j], rmx > Q[j]).mean() for j in range(0, NY, 25)]).mean()
print(json.dumps({'kernel': 'k_lwa<double,false,4>', 'shape': [NY, NX], 'J': NY, 'gpu_ms': ms,
                  'nominal_cell_rows_per_s': NY * NY * NX / ms * 1e3, 'contributing_row_fraction': float(need),
                  'six_rows_bit_identical_to_oracle': ok, 'cpu_oracle_s_per_target_row': tc,
                  'cpu_oracle_s_all_rows_estimate': tc * NY}))
