#!/usr/bin/env python3
"""cProfile of the facade's fused keff() and of the histogram twin on the reference's demo size (15 x 241 x 480 float32), resident inputs:
where the Python microseconds of a call go.   python tools/probe/keff_pyprof.py [keff|hist|contours|lookup|grad]"""
import cProfile, os, pstats, sys, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import xcontour_amd as xa
what = sys.argv[1] if len(sys.argv) > 1 else 'keff'
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
fn = {'keff': lambda: cm.keff(N1, table, grdS=g2), 'hist': lambda: cm.cal_integral_within_contours_hist(ctr, integrand=g2),
      'contours': lambda: cm.cal_contours(N1), 'lookup': lambda: table.lookup_coordinates(area),
      'grad': lambda: cm.cal_gradient_wrt_area(ctr, area), 'table': lambda: cm.cal_area_eqCoord_table_hist(mask)}[what]
for _ in range(5):
    fn()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    fn()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue())
