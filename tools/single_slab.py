#!/usr/bin/env python3
"""One cfg2 slab (1801 x 3600 float64, 201 contours) through the Keff pipeline, N times with a sync in between (a user with one
field): run it under `rocprofv3 --kernel-trace --stats` to see every launch of the sequence, or alone for the event time.
    python tools/single_slab.py [reps] [f32]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat                                   # noqa: E402
from xcontour_amd.pipeline import KeffPlan                                # noqa: E402
from xcontour_amd.utils import cell_area, table_from_rowsums              # noqa: E402

NY, NX, N = 1801, 3600, 201
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = np.float32 if 'f32' in sys.argv[2:] else np.float64
ctx = nat.Context(0)
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
dA = cell_area(lat, lon)
tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True)
plan = KeffPlan(ctx, 1, NY, NX, N, dt, dt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True)
plan.synth(lat, lon, 1, 0)
big = ctx.alloc(600 << 20)                                                # evict the slab from the Infinity Cache between runs
e0, e1 = ctx.event(), ctx.event()
ts = []
for r in range(reps + 3):
    ctx._check(ctx.lib.xc_memset(ctx.handle, big.ptr, r & 255, big.nbytes))
    ctx.sync()
    ctx.record(e0)
    plan.run()
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)
    if r >= 3:
        ts.append(ms * 1e3)
ts = np.array(ts)
print(json.dumps({'keff_one_slab_us_cold': {'mean': float(ts.mean()), 'min': float(ts.min()), 'median': float(np.median(ts))}, 'dtype': np.dtype(dt).name}))
ts = []
for r in range(reps + 3):
    ctx.sync()
    ctx.record(e0)
    plan.run()
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)
    if r >= 3:
        ts.append(ms * 1e3)
ts = np.array(ts)
print(json.dumps({'keff_one_slab_us_warm': {'mean': float(ts.mean()), 'min': float(ts.min()), 'median': float(np.median(ts))}, 'dtype': np.dtype(dt).name}))
ctx.close()
