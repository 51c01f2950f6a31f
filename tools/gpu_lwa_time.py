"""Scratch: device-resident timing of K7 (cfg3: barotropic LWA) and K8."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from xcontour_amd import _native as nat
import xcontour_oracle as O
ctx = nat.Context(0)
g = os.path.join(ROOT, 'tests', 'golden')
q = np.load(g + '/baro_q.npy'); lat = np.load(g + '/baro_lat.npy'); lon = np.load(g + '/baro_lon.npy')
L = np.load(g + '/baro_lwa_N121.npz')
dA = O.cell_area(lat, lon)
dq, dQ, dc, dd, dM = ctx.to_device(q), ctx.to_device(L['Q']), ctx.to_device(lat.astype(np.float64)), ctx.to_device(dA), ctx.to_device(L['dy'])
out = ctx.alloc(q.size * 8)
e0, e1 = ctx.event(), ctx.event()
def run():
    ctx._check(ctx.lib.xc_lwa_dev(ctx.handle, dq.ptr, nat.XC_F32, dQ.ptr, dc.ptr, dd.ptr, nat.XC_DA_PLANE, float(dA.max()),
                                  dM.ptr, nat.XC_DA_ROW, 1, 256, 512, 1, 0, 0, None, 0, out.ptr, None))
for _ in range(3): run()
ctx.record(e0)
for _ in range(20): run()
ctx.record(e1)
ms = ctx.elapsed_ms(e0, e1) / 20
print('K7 cfg3 (256x512 f32, J=256): %.1f us per call -> %.2f G cell-rows/s' % (ms * 1e3, 256 * 256 * 512 / ms / 1e6))
t = __import__('time').time(); ref = O.cal_local_wave_activity(q, L['Q'], lat, dA, True, 'all', metric=L['dy']); print('oracle LWA s', __import__('time').time() - t)
