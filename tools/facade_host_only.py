#!/usr/bin/env python3
"""The Python around the library calls, alone: the reference's Keff call sequence at its demo size (15 x 241 x 480 float32, resident
inputs) against a stand-in library whose entry points return at once (results: fixed monotone vectors).  No GPU needed -- this is the
harness the facade's host overhead is tuned with (`--profile`: cProfile per call).  What it prints is the `python_us` column of
tools/facade_time.py --breakdown, measured where there is no device."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xcontour_amd as xa                 # noqa: E402
from xcontour_amd import _native as nat   # noqa: E402

NL1, NY1, NX1, N1 = 15, 241, 480, 201
rng = np.random.default_rng(0)


class _Lib(object):
    """every xc_* entry point: returns XC_OK; the three the sequence reads results from fill them with monotone numbers"""

    def __getattr__(self, name):
        if name == 'xc_hist':
            return self._hist
        if name == 'xc_contours':
            return self._contours
        if name == 'xc_rowsum':
            return self._rowsum
        if name == 'xc_malloc':
            return self._malloc
        if name == 'xc_memcpy_d2h':
            return lambda h, dst, src, n: (C.memset(dst, 0, n), 0)[1]
        if name == 'xc_resident_lookup':
            return self._malloc2
        return lambda *a: 0

    _next = 1 << 40

    def _malloc(self, h, n, pref):
        pref._obj.value = _Lib._next
        _Lib._next += (int(n) + 4095) & ~4095
        return 0

    def _malloc2(self, h, p, n, pref):
        return self._malloc(h, n, pref)

    def _hist(self, h, dref):
        d = dref._obj
        n = d.nslab * (1 + d.nint + d.grad) * (d.nedge - 1)
        if d.cdf:
            C.memmove(d.cdf, self._mono(n).ctypes.data, n * 8)
        return 0

    def _contours(self, h, q, qd, nslab, ncell, N, inc, cd, re, mm, ctr, e, s):
        C.memmove(ctr, self._mono(nslab * N).ctypes.data, nslab * N * 8)
        return 0

    def _rowsum(self, h, m, md, dA, rank, ny, nx, mul, out):
        C.memmove(out, np.full(ny, 1.5).ctypes.data, ny * 8)
        return 0

    _cache = {}

    def _mono(self, n):
        if n not in self._cache:
            self._cache[n] = (np.arange(n, dtype=np.float64) % N1 + 1.0) * 0.37
        return self._cache[n]


ctx = nat.Context.__new__(nat.Context)
ctx.lib, ctx.handle, ctx.device = _Lib(), None, 0
ctx._buffers, ctx._resident, ctx._staged, ctx._ev_pool = [], {}, [], []
ctx.max_batch_bytes = 8 << 30
nat.default_context = lambda device=0: ctx

lat = np.linspace(-90, 90, NY1).astype(np.float32); lon = (np.arange(NX1) * 0.75).astype(np.float32); lev = np.arange(NL1, dtype=np.float32)
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
calls = [
    ('table', lambda: cm.cal_area_eqCoord_table_hist(mask)),
    ('contours', lambda: cm.cal_contours(N1)),
    ('integral_area', lambda: cm.cal_integral_within_contours_hist(ctr)),
    ('integral_grdS', lambda: cm.cal_integral_within_contours_hist(ctr, integrand=g2)),
    ('lookup', lambda: table.lookup_coordinates(area)),
    ('gradient x2', lambda: (cm.cal_gradient_wrt_area(ctr, area), cm.cal_gradient_wrt_area(intS, area))),
    ('keff (fused)', lambda: cm.keff(N1, table, grdS=g2)),
]
flt = [a for a in sys.argv[1:] if not a.startswith('--')]
tot = 0.0
for name, fn in calls:
    if flt and not any(f in name for f in flt):
        continue
    for _ in range(20):
        fn()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(400):
            fn()
        best = min(best, (time.perf_counter() - t) / 400 * 1e6)
    tot += best
    print('%-16s %7.1f us' % (name, best))
    if '--profile' in sys.argv:
        import cProfile, io, pstats
        pr = cProfile.Profile(); pr.enable()
        for _ in range(400):
            fn()
        pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
        print('\n'.join(l[:150] for l in s.getvalue().splitlines() if l.strip() and 'Ordered by' not in l and 'function calls' not in l))
print('%-16s %7.1f us' % ('sum', tot))
