#!/usr/bin/env python3
"""Wall time of every facade method on a cfg1-like stack (15 levels x 241 x 480 float32) and on 40 time steps of the
barotropic-sized plane: looks for Python-side overheads (per-slab loops, repeated uploads)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xcontour_amd as xa

rng = np.random.default_rng(0)


def timed(fn, reps=3):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t) / reps * 1e3, r


for (S, ny, nx, dt) in [(15, 241, 480, np.float32), (40, 256, 512, np.float32), (4, 721, 1440, np.float64)]:
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * (360.0 / nx)
    q = (np.sin(np.deg2rad(lat))[None, :, None] * (1 + 0.1 * np.arange(S))[:, None, None] + 0.05 * rng.standard_normal((S, ny, nx))).astype(dt)
    c = {'lev': np.arange(S, dtype=float), 'lat': lat, 'lon': lon}
    tr = xa.DataArray(q, ('lev', 'lat', 'lon'), c, 'pv')
    dA = xa.DataArray(xa.cell_area(lat, lon), ('lat', 'lon'), {'lat': lat, 'lon': lon}, 'dA')
    g2 = xa.DataArray(rng.random((S, ny, nx)).astype(dt), tr.dims, c, 'grdS')
    mask = xa.DataArray(np.ones((ny, nx)), ('lat', 'lon'), {'lat': lat, 'lon': lon}, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True)
    rec = {}
    rec['table'], table = timed(lambda: cm.cal_area_eqCoord_table_hist(mask))
    rec['cal_contours'], ctr = timed(lambda: cm.cal_contours(121))
    rec['integral_hist(area)'], area = timed(lambda: cm.cal_integral_within_contours_hist(ctr))
    rec['integral_hist(grdS)'], intS = timed(lambda: cm.cal_integral_within_contours_hist(ctr, integrand=g2))
    rec['integral(strict)'], _ = timed(lambda: cm.cal_integral_within_contours(ctr))
    rec['lookup'], latEq = timed(lambda: table.lookup_coordinates(area))
    rec['contour_mean_hist'], _ = timed(lambda: cm.cal_contour_mean_hist(ctr, g2, g2))
    rec['keff(fused)'], ds = timed(lambda: cm.keff(121, table, preY=lat))
    Q = ds['ctr_eq'].rename({'new': 'lat'})
    rec['lwa'], _ = timed(lambda: cm.cal_local_wave_activity(tr, Q), reps=2)
    rec['sorted_profile'], _ = timed(lambda: cm.cal_sorted_profile(table))
    rec['crossing[1,2,4,8]'], _ = timed(lambda: cm.cal_contour_crossing(ctr, stride=[1, 2, 4, 8], mode='wrap'))
    rec['contours_at_hist'], _ = timed(lambda: cm.cal_contours_at_hist(lat, table), reps=2)
    print(json.dumps({'stack': [S, ny, nx, np.dtype(dt).name], 'mbytes': q.nbytes / 1e6, 'ms': {k: round(v, 2) for k, v in rec.items()}}), flush=True)
    cm.close()
