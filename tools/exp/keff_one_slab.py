#!/usr/bin/env python3
"""What a caller of Contour2D.keff sees for ONE cfg2 slab per call (resident inputs): wall time, time inside the library call by call."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import xcontour_amd as xa
from xcontour_amd import _native as nat
NY, NX, N = 1801, 3600, 201
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
rng = np.random.default_rng(0)
q = np.sin(np.deg2rad(lat))[:, None] + 0.05 * rng.standard_normal((NY, NX))
c = {'lat': lat, 'lon': lon}
tr = xa.DataArray(q, ('lat', 'lon'), c, 'pv')
dA = xa.DataArray(xa.cell_area(lat, lon), ('lat', 'lon'), c, 'dA')
mask = xa.DataArray(np.ones((NY, NX)), ('lat', 'lon'), c, 'mask')
cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, dtype=np.float64, resident=True)
table = cm.cal_area_eqCoord_table_hist(mask)
fn = lambda: cm.keff(N, table, lat=lat, lon=lon)
for _ in range(5):
    fn()
lib = nat.load(); T = {}; Nn = {}
for name in nat.PROTOTYPES:
    if name in ('xc_last_error', 'xc_version', 'xc_trace'):
        continue
    f = getattr(lib, name)
    def w(*a, __f=f, __n=name):
        t0 = time.perf_counter(); r = __f(*a); T[__n] = T.get(__n, 0.0) + time.perf_counter() - t0; Nn[__n] = Nn.get(__n, 0) + 1; return r
    setattr(lib, name, w)
reps = 200
t = time.perf_counter()
for _ in range(reps):
    fn()
wall = (time.perf_counter() - t) / reps * 1e6
inlib = sum(T.values()) / reps * 1e6
print(json.dumps({'keff_one_cfg2_slab_us': round(wall, 1), 'python_us': round(wall - inlib, 1), 'library_us': round(inlib, 1),
                  'calls': {k: [round(T[k] / reps * 1e6, 1), Nn[k] // reps] for k in sorted(T, key=lambda k: -T[k])}, 'path': cm.ctx.last_keff_path()}))
