#!/usr/bin/env python3
"""60 x cal_local_wave_activity on one 241 x 480 float32 plane (resident weights), for a kernel / copy trace"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import xcontour_amd as xa
NY1, NX1 = 241, 480
lat = np.linspace(-90, 90, NY1).astype(np.float32); lon = (np.arange(NX1) * 0.75).astype(np.float32)
rng = np.random.default_rng(0)
q = (np.sin(np.deg2rad(lat))[:, None] + 0.05 * rng.standard_normal((NY1, NX1))).astype(np.float32)
c2 = {'lat': lat, 'lon': lon}
tr = xa.DataArray(q, ('lat', 'lon'), c2, 'pv')
dA = xa.DataArray(xa.cell_area(lat.astype(np.float64), lon.astype(np.float64)).astype(np.float32), ('lat', 'lon'), c2, 'dA')
Q = xa.DataArray(np.sort(q.mean(axis=1)).astype(np.float32), ('lat',), {'lat': lat}, 'Q')
cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, resident=True)
for _ in range(60):
    out = cm.cal_local_wave_activity(tr, Q)
