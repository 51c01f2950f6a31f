#!/usr/bin/env python3
"""cal_local_wave_activity through the facade at the reference's demo size: one (lat, lon) plane per call (the loop of the reference's
tests/LWA.py) and the 15-level stack in one call; wall time, time inside the library call by call."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import xcontour_amd as xa
from xcontour_amd import _native as nat
NL1, NY1, NX1 = 15, 241, 480
lat = np.linspace(-90, 90, NY1).astype(np.float32); lon = (np.arange(NX1) * 0.75).astype(np.float32); lev = np.arange(NL1, dtype=np.float32)
rng = np.random.default_rng(0)
q = (np.sin(np.deg2rad(lat))[None, :, None] * (1 + 0.1 * lev[:, None, None]) + 0.05 * rng.standard_normal((NL1, NY1, NX1))).astype(np.float32)
c3 = {'lev': lev, 'lat': lat, 'lon': lon}; c2 = {'lat': lat, 'lon': lon}
tr = xa.DataArray(q, ('lev', 'lat', 'lon'), c3, 'pv')
dA = xa.DataArray(xa.cell_area(lat.astype(np.float64), lon.astype(np.float64)).astype(np.float32), ('lat', 'lon'), c2, 'dA')
Qv = np.sort(q.mean(axis=2), axis=1).astype(np.float32)
Q = xa.DataArray(Qv, ('lev', 'lat'), {'lev': lev, 'lat': lat}, 'Q')
res = {}
for resident in (False, True):
    cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, resident=resident)
    q0 = xa.DataArray(q[3], ('lat', 'lon'), c2, 'pv'); Q0 = xa.DataArray(Qv[3], ('lat',), {'lat': lat}, 'Q')
    calls = {'one plane': lambda: cm.cal_local_wave_activity(q0, Q0), 'stack of 15': lambda: cm.cal_local_wave_activity(tr, Q)}
    lib = nat.load()
    for name, fn in calls.items():
        for _ in range(5):
            fn()
        T = {}
        saved = {}
        for n_ in nat.PROTOTYPES:
            if n_ in ('xc_last_error', 'xc_version', 'xc_trace'):
                continue
            f = getattr(lib, n_); saved[n_] = f
            def w(*a, __f=f, __n=n_):
                t0 = time.perf_counter(); r = __f(*a); T[__n] = T.get(__n, 0.0) + time.perf_counter() - t0; return r
            setattr(lib, n_, w)
        reps = 100
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        wall = (time.perf_counter() - t) / reps * 1e6
        for n_, f in saved.items():
            setattr(lib, n_, f)
        inlib = sum(T.values()) / reps * 1e6
        res['%s, resident=%s' % (name, resident)] = {'us': round(wall, 1), 'python_us': round(wall - inlib, 1),
                                                     'library': {k: round(T[k] / reps * 1e6, 1) for k in sorted(T, key=lambda k: -T[k])}}
    cm.close()
print(json.dumps(res, indent=1))
