#!/usr/bin/env python3
"""Wall time of the facade (numpy in / numpy out, host entry points, PCIe included) on one cfg2-sized
slab: the reference's Keff call sequence (tests/test_Keff_atmos.py:75-92) call by call, and the fused
`Contour2D.keff`.  Prints a JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xcontour_amd as xa   # noqa: E402

NY, NX, N = 1801, 3600, 201
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
rng = np.random.default_rng(0)
q = np.sin(np.deg2rad(lat))[:, None] + 0.05 * rng.standard_normal((NY, NX))
c = {'lat': lat, 'lon': lon}
tr = xa.DataArray(q, ('lat', 'lon'), c, 'pv')
dA = xa.DataArray(xa.cell_area(lat, lon), ('lat', 'lon'), c, 'dA')
g2 = xa.DataArray(rng.random((NY, NX)), ('lat', 'lon'), c, 'grdS')
mask = xa.DataArray(np.ones((NY, NX)), ('lat', 'lon'), c, 'mask')
kw = dict(dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, dtype=np.float64)
kw.update(json.loads(os.environ.get('XC_FACADE_KW', '{}')))
cm = xa.Contour2D(tr, dA, **kw)
rec = {}


def timed(name, fn, reps=3):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    rec[name] = (time.perf_counter() - t) / reps * 1e3
    return out


table = timed('cal_area_eqCoord_table_hist', lambda: cm.cal_area_eqCoord_table_hist(mask))
ctr = timed('cal_contours', lambda: cm.cal_contours(N))
area = timed('cal_integral_within_contours_hist(area)', lambda: cm.cal_integral_within_contours_hist(ctr))
intS = timed('cal_integral_within_contours_hist(grdS)', lambda: cm.cal_integral_within_contours_hist(ctr, integrand=g2))
latEq = timed('lookup_coordinates', lambda: table.lookup_coordinates(area))
dq = timed('cal_gradient_wrt_area x2', lambda: (cm.cal_gradient_wrt_area(ctr, area), cm.cal_gradient_wrt_area(intS, area)))
timed('keff (fused, grdS supplied)', lambda: cm.keff(N, table, grdS=g2))
timed('keff (fused, in-kernel gradient)', lambda: cm.keff(N, table, lat=lat, lon=lon))
rec['sum of the call sequence'] = sum(v for k, v in rec.items() if not k.startswith('keff'))
print(json.dumps({'facade_ms_per_call_cfg2_slab': rec, 'kwargs': {k: str(v) for k, v in kw.items()}}))
