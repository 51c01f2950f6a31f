"""Phase stamps of the single-read Keff kernel (xc_keff1.hip) on one cfg2 slab: wall-clock (100 MHz) stamps of thread 0 of every
workgroup at the phase boundaries -> medians over workgroups of every phase, and the spread of arrival.  GPU box only.
  python3 tools/single_stamps.py [--dtype f64|f32] [--ny 1801 --nx 3600] [--reps 20] [--cold]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included

NAMES = ['start', 'tile landed + min/max', 'publish', 'wait for the grid', 'edges', 'bin', 'flush (adds issued)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f64')
    ap.add_argument('--ny', type=int, default=1801)
    ap.add_argument('--nx', type=int, default=3600)
    ap.add_argument('--N', type=int, default=201)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--cold', action='store_true')
    a = ap.parse_args()
    dt = np.float64 if a.dtype == 'f64' else np.float32
    ctx = nat.Context(0)
    lat = np.linspace(-90, 90, a.ny); lon = np.arange(a.nx) * (360.0 / a.nx)
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, a.ny, a.nx), True, last_row_included(lat))
    p = KeffPlan(ctx, 1, a.ny, a.nx, a.N, dt, dt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, counts=False)
    p.synth(lat, lon, 20241015, 0)
    big = ctx.alloc(600 << 20) if a.cold else None
    ptr, slots = ctx.single_stamps(True)
    cus = ctx.device_cus()
    n = 2 * cus * slots
    acc = []
    fst = []
    e0, e1 = ctx.event(), ctx.event()
    tms = []
    for r in range(a.reps + 3):
        if big is not None:
            ctx._check(ctx.lib.xc_memset(ctx.handle, big.ptr, r & 255, big.nbytes))
        ctx.sync()
        ctx.record(e0); p.run(); ctx.record(e1)
        tms.append(ctx.elapsed_ms(e0, e1) * 1e3)
        assert ctx.last_keff_path() == 1
        st = np.empty(n, np.uint64)
        ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, st.ctypes.data, ptr, n * 8))
        st = st.reshape(2, cus, slots)[0].astype(np.float64)
        f8 = st[0, 8:13].copy()                   # the finalize kernel's stamps (slots 8.. of workgroup 0)
        st = st[st[:, 0] > 0]                      # workgroups of the grid
        if r >= 3:
            acc.append(st); fst.append(f8)
    out = p.fetch()
    assert not out['status'].any()
    rows = []
    G = acc[0].shape[0]
    t0 = np.array([s[:, 0].min() for s in acc])
    for k in range(1, 7):
        d = np.array([np.median(s[:, k] - s[:, k - 1]) for s in acc]) / 100.0
        end = np.array([np.median(s[:, k]) - t for s, t in zip(acc, t0)]) / 100.0
        last = np.array([s[:, k].max() - t for s, t in zip(acc, t0)]) / 100.0
        rows.append({'phase': NAMES[k], 'median_us': round(float(np.median(d)), 2), 'ends_at_us_median_wg': round(float(np.median(end)), 2),
                     'ends_at_us_last_wg': round(float(np.median(last)), 2)})
    raw = [s_ for s_ in acc]
    sk = float(np.median([np.median(s_[:, 7] - s_[:, 5]) for s_ in acc])) / 100.0
    res = {'flush_barrier_wait_us': round(sk, 2), 'workgroups': int(G), 'start_spread_us': round(float(np.median([s[:, 0].max() - s[:, 0].min() for s in acc])) / 100.0, 2),
           'phases': rows,
           'finalize_kernel_us_after_the_last_flush': dict(zip(['first instruction', 'sums + levels + table in LDS', 'cumsum', 'look-ups | gradients', 'nkeff (end)'],
                                                               [round(float(np.median([(f[k] - s_[:, 6].max()) / 100.0 for f, s_ in zip(fst, raw)])), 2) for k in range(5)])),
           'event_us_median': round(float(np.median(tms[3:])), 2), 'event_us_min': round(float(np.min(tms[3:])), 2),
           'dtype': a.dtype, 'shape': [a.ny, a.nx], 'N': a.N, 'cold': bool(a.cold)}
    print(json.dumps(res))
    ctx.single_stamps(False)
    p.free()
    ctx.close()


if __name__ == '__main__':
    main()
