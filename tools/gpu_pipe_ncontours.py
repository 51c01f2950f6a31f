#!/usr/bin/env python3
"""Keff pipeline time per cfg2 slab as a function of the number of contours (LDS copies shrink as N grows)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums
ctx = nat.Context(0)
NY, NX, B = 1801, 3600, 16
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
dA = cell_area(lat, lon)
tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True)
e0, e1 = ctx.event(), ctx.event()
for N in (3, 11, 41, 201, 501, 1001, 3000, 6000):
    for variant in (0, 2):
        plan = KeffPlan(ctx, 2 * B, NY, NX, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True, out_slabs=B)
        plan.synth(lat, lon, 1, variant)
        def step(k):
            s0 = (k % 2) * B; nxt = ((k + 1) % 2) * B
            plan.run_range(0, s0, B, nxt, out_s0=0)
        for k in range(3): step(k)
        ctx.sync(); ctx.record(e0)
        for k in range(10): step(k)
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 10
        print('N = %5d  variant %d  %.1f us/slab' % (N, variant, ms / B * 1e3), flush=True)
        plan.free()
