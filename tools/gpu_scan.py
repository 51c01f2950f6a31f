"""Scratch: hist kernel time vs ny (fixed overhead vs slope)."""
import os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = types.ModuleType('xcontour_amd'); pkg.__path__ = [os.path.join(ROOT, 'xcontour_amd')]; sys.modules['xcontour_amd'] = pkg
import xcontour_amd._native as nat
import xcontour_amd.pipeline as pl
import xcontour_amd.utils as U
ctx = nat.Context(0)
nx, N = 3600, 201
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for ny in [int(a) for a in sys.argv[2:]] or [33, 257, 901, 1801, 3601, 7201]:
    lat = np.linspace(-89.9, 89.9, ny); lon = np.arange(nx) * 0.1
    dA = U.cell_area(lat, lon)
    tbl = U.table_from_rowsums(dA.sum(1), True)
    plan = pl.KeffPlan(ctx, 1, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True)
    plan.synth(lat, lon, 1, variant)
    ctx.set_kernel_timing(True)
    ms = []
    for _ in range(12):
        plan.run(); ms.append(ctx.last_hist_ms())
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(20): plan.run()
    ctx.record(e1)
    tot = ctx.elapsed_ms(e0, e1) / 20
    print('ny', ny, 'hist us', round(np.median(ms) * 1e3, 2), 'min', round(min(ms) * 1e3, 2), 'pipeline us', round(tot * 1e3, 2),
          'hist GB/s', round(ny * nx * 16 / np.median(ms) / 1e6, 1))
    plan.free()
