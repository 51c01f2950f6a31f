"""Scratch: per-phase timing inside k_hist for the BENCH configuration (cfg2, 64 slabs per launch, chained) from the
diagnostic stamps build (tools/build_variant.sh stamps "-DXC_STAMPS").  Prints phase durations and CU utilisation."""
import os, sys, types, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = types.ModuleType('xcontour_amd'); pkg.__path__ = [os.path.join(ROOT, 'xcontour_amd')]; sys.modules['xcontour_amd'] = pkg
import xcontour_amd._native as nat
nat.LIB_PATH = os.path.join(ROOT, 'xcontour_amd', 'libxc_stamps.so')
import xcontour_amd.pipeline as pl
import xcontour_amd.utils as U
ctx = nat.Context(0)
ny, nx, N, B = 1801, 3600, 201, 64
chain = '--no-chain' not in sys.argv
lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.1
dA = U.cell_area(lat, lon)
tbl = U.table_from_rowsums(dA.sum(1), True)
plan = pl.KeffPlan(ctx, 2 * B, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, out_slabs=B)
plan.synth(lat, lon, 1, 0)
nb = 8 * 8 * B                      # upper bound of the grid (bps <= 64)
st = ctx.alloc(nb * 8 * 8)
ctx.lib.xc_dbg_set_hist_stamps.argtypes = [C.c_void_p]
for k in range(4):
    plan.run_range(0, (k % 2) * B, B, ((k + 1) % 2) * B if chain else None, out_s0=0)
ctx.sync()
st.upload(np.zeros(nb * 8, np.uint64))
assert ctx.lib.xc_dbg_set_hist_stamps(st.ptr) == 0
plan.run_range(0, 0, B, B if chain else None, out_s0=0); ctx.sync()
s = st.download((nb, 8), np.uint64).astype(np.int64)
s = s[(s[:, 0] > 0) & (s[:, 5] > 0)]
t0 = s[:, 0].min()
rel = (s[:, :6] - t0) / 100.0       # us (100 MHz clock)
d = np.diff(rel, axis=1)
span = rel[:, 5].max()
print('blocks', len(s), 'kernel span %.1f us' % span)
print('phase durations us (zero LDS | edges | main loop | flush+minmax | partials): median', np.round(np.median(d, axis=0), 2),
      'p90', np.round(np.percentile(d, 90, axis=0), 2), 'max', np.round(d.max(axis=0), 2))
life = rel[:, 5] - rel[:, 0]
print('block lifetime us: median %.1f p10 %.1f p90 %.1f;  CU utilisation (sum of lifetimes / 256 / span) %.3f;  main-loop share %.3f'
      % (np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.sum() / 256 / span, d[:, 2].sum() / 256 / span))
order = np.argsort(rel[:, 0])
starts = rel[order, 0]
print('start times us of blocks #0,255,256,511,512,...:', np.round(starts[[0, 255, 256, 511, 512, 767, 768, min(1023, len(starts) - 1), min(1024, len(starts) - 1), len(starts) - 1]], 1))
