#!/usr/bin/env python3
"""50 x each GPU call of the reference's sequence at its demo size (resident), for a kernel trace"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import xcontour_amd as xa
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
only = os.environ.get('XC_LOOP', '')
for _ in range(50 if not only else 1):
    table = cm.cal_area_eqCoord_table_hist(mask)
for _ in range(50 if not only else 1):
    ctr = cm.cal_contours(N1)
for _ in range(50):
    area = cm.cal_integral_within_contours_hist(ctr)
for _ in range(50):
    intS = cm.cal_integral_within_contours_hist(ctr, integrand=g2)
