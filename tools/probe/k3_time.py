#!/usr/bin/env python3
"""K3 launch time on the headline workload (two launch sets of 64 cfg2 slabs chained round-robin, as bench.py schedules them), no result checks:
for timing VARIANT builds (XC_LIB_PATH) whose sums are deliberately wrong.   python tools/probe/k3_time.py [f32|f64] [chain|nochain] [variant]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums
NY, NX, N, B = 1801, 3600, 201, 64
dt = np.float32 if 'f32' in sys.argv[1:] else np.float64
chain = 'nochain' not in sys.argv[1:]
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = nat.Context(0)
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
dA = cell_area(lat, lon)
tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True)
plan = KeffPlan(ctx, 2 * B, NY, NX, N, dt, dt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True, counts=False)
plan.synth(lat, lon, 1, variant)
ctx._check(ctx.lib.xc_set_kernel_timing(ctx.handle, 1))
ts = []
for it in range(12):
    plan.run(0, B, chain=chain)                    # two launch sets per call, each with the other one as its NEXT set
    ctx.sync()
    if it >= 2:
        ts.append(ctx.last_hist_ms())
print(json.dumps({'lib': os.environ.get('XC_LIB_PATH', 'default'), 'dtype': np.dtype(dt).name, 'chain': chain, 'variant': variant,
                  'k3_ms_per_64_slabs': float(np.median(ts)), 'us_per_slab': float(np.median(ts)) / B * 1e3}))
