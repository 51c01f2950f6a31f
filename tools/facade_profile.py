#!/usr/bin/env python3
"""cProfile of the facade's reference call sequence at the reference's demo size (15 x 241 x 480 float32, resident inputs): where the
Python around the library calls spends its time.  GPU box only.   python3 tools/facade_profile.py [name-filter]"""
import cProfile
import io
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xcontour_amd as xa   # noqa: E402

NL1, NY1, NX1, N1 = 15, 241, 480, 201
lat = np.linspace(-90, 90, NY1).astype(np.float32); lon = (np.arange(NX1) * 0.75).astype(np.float32); lev = np.arange(NL1, dtype=np.float32)
rng = np.random.default_rng(0)
q = (np.sin(np.deg2rad(lat))[None, :, None] * (1 + 0.1 * lev[:, None, None]) + 0.05 * rng.standard_normal((NL1, NY1, NX1))).astype(np.float32)
c3 = {'lev': lev, 'lat': lat, 'lon': lon}; c2 = {'lat': lat, 'lon': lon}
tr = xa.DataArray(q, ('lev', 'lat', 'lon'), c3, 'pv')
dA = xa.DataArray(xa.cell_area(lat.astype(np.float64), lon.astype(np.float64)).astype(np.float32), ('lat', 'lon'), c2, 'dA')
g2 = xa.DataArray(rng.random(q.shape).astype(np.float32), ('lev', 'lat', 'lon'), c3, 'grdS')
mask = xa.DataArray(np.ones((NY1, NX1), np.float32), ('lat', 'lon'), c2, 'mask')
cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, resident=True)
table = cm.cal_area_eqCoord_table_hist(mask)
ctr = cm.cal_contours(N1)
area = cm.cal_integral_within_contours_hist(ctr)
intS = cm.cal_integral_within_contours_hist(ctr, integrand=g2)
calls = {
    'table': lambda: cm.cal_area_eqCoord_table_hist(mask),
    'contours': lambda: cm.cal_contours(N1),
    'integral_area': lambda: cm.cal_integral_within_contours_hist(ctr),
    'integral_grdS': lambda: cm.cal_integral_within_contours_hist(ctr, integrand=g2),
    'lookup': lambda: table.lookup_coordinates(area),
    'gradient': lambda: cm.cal_gradient_wrt_area(intS, area),
    'keff': lambda: cm.keff(N1, table, grdS=g2),
}
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for name, fn in calls.items():
    if flt and flt not in name:
        continue
    for _ in range(5):
        fn()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        fn()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
    print('=====', name)
    print('\n'.join(l for l in s.getvalue().splitlines() if l.strip() and 'Ordered by' not in l and 'function calls' not in l)[:2600])
