"""Scratch: cfg2 pipeline timing + parity on a real GPU."""
import os, sys, time, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
pkg = types.ModuleType('xcontour_amd'); pkg.__path__ = [os.path.join(ROOT, 'xcontour_amd')]; sys.modules['xcontour_amd'] = pkg
import xcontour_amd._native as nat
import xcontour_amd.pipeline as pl
import xcontour_amd.utils as U
import xcontour_oracle as O

ctx = nat.Context(0)
ny, nx, N = 1801, 3600, 201
lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.1
dA = U.cell_area(lat, lon)
rows = dA.sum(1)
tbl = U.table_from_rowsums(rows, True)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
plan = pl.KeffPlan(ctx, B, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                   preY=lat, increase=True, lt=True)
plan.synth(lat, lon, 20241008, variant)
ctx.set_kernel_timing(True)
for _ in range(3):
    plan.run()
ctx.sync()
e0, e1 = ctx.event(), ctx.event()
K = 20
hist_ms = []
ctx.record(e0)
for _ in range(K):
    plan.run()
ctx.record(e1)
tot = ctx.elapsed_ms(e0, e1) / K
for _ in range(10):
    plan.run(); hist_ms.append(ctx.last_hist_ms())
cells = ny * nx
print('variant', variant, 'B', B, 'pipeline ms/step', tot, 'per slab us', tot / B * 1e3, 'hist kernel ms', np.median(hist_ms), min(hist_ms))
print('pipeline GB/s (16B/cell)', cells * B * 16 / tot / 1e6, 'hist GB/s', cells * B * 16 / np.median(hist_ms) / 1e6)
out = plan.fetch()
q = plan.download_q()
t = time.time()
r = O.keff_pipeline(q[0], dA, lat, N, lon=lon, dtype=np.float64, preLats=lat)
print('oracle s', time.time() - t)
print('counts equal', np.array_equal(out['counts'][0].astype(np.int64), r['counts']), r['counts'].sum())
for k in ('ctr', 'area', 'intgrdS', 'latEq', 'Lmin', 'dqdA', 'dintSdA', 'Leq2', 'nkeff'):
    a, b = out[k][0], r[k]
    m = np.isfinite(b)
    print(k, 'nan-pattern', np.array_equal(np.isnan(a), np.isnan(b)), 'maxrel', np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300)) if m.any() else None)
    a, b = out[k + '_eq'][0], r[k + '_eq']
    m = np.isfinite(b)
    print('   _eq nan-pattern', np.array_equal(np.isnan(a), np.isnan(b)), 'maxrel', np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), 1e-300)) if m.any() else None)
