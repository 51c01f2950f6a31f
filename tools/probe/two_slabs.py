"""diagnostic: a call of TWO cfg2 slabs through the single-read kernel and through the chain (us per call, warm)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
ctx = nat.Context(0)
ny, nx, N = 1801, 3600, 201
lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.1
dA = cell_area(lat, lon)
tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
for nslab in (1, 2):
    for single in (True, False):
        p = KeffPlan(ctx, nslab, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, single_read=single)
        p.synth(lat, lon, 5, 0)
        e0, e1 = ctx.event(), ctx.event()
        ts = []
        for r in range(25):
            ctx.sync(); ctx.record(e0); p.run(); ctx.record(e1); ms = ctx.elapsed_ms(e0, e1)
            if r >= 5:
                ts.append(ms * 1e3)
        print(nslab, 'single' if single else 'chain', 'path', ctx.last_keff_path(), 'us per call', round(float(np.median(ts)), 2), 'min', round(float(np.min(ts)), 2))
        p.free()
