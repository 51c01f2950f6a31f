"""Scratch: per-phase timing inside k_hist from the diagnostic stamps build."""
import os, sys, types, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = types.ModuleType('xcontour_amd'); pkg.__path__ = [os.path.join(ROOT, 'xcontour_amd')]; sys.modules['xcontour_amd'] = pkg
import xcontour_amd._native as nat
nat.LIB_PATH = os.path.join(ROOT, 'xcontour_amd', 'libxc_stamps.so')
import xcontour_amd.pipeline as pl
import xcontour_amd.utils as U
ctx = nat.Context(0)
nx, N = 3600, 201
for ny in [int(a) for a in sys.argv[1:]] or [33, 1801]:
    lat = np.linspace(-89.9, 89.9, ny); lon = np.arange(nx) * 0.1
    dA = U.cell_area(lat, lon)
    tbl = U.table_from_rowsums(dA.sum(1), True)
    plan = pl.KeffPlan(ctx, 1, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat)
    plan.synth(lat, lon, 1, 0)
    nb = 256
    st = ctx.alloc(nb * 8 * 8)
    ctx.lib.xc_dbg_set_hist_stamps.argtypes = [C.c_void_p]
    for _ in range(3): plan.run()
    ctx.sync()
    assert ctx.lib.xc_dbg_set_hist_stamps(st.ptr) == 0
    plan.run(); ctx.sync()
    s = st.download((nb, 8), np.uint64).astype(np.int64)
    t0 = s[:, 0].min()
    rel = (s[:, :6] - t0) * 10  # ns
    print('ny', ny, 'phase end times (ns, median over blocks):', np.median(rel, axis=0), 'max', rel.max(axis=0))
    print('   start spread', rel[:, 0].max(), ' durations median', np.median(np.diff(rel, axis=1), axis=0))
    ctx.lib.xc_dbg_set_hist_stamps(None)
    plan.free()
