#!/usr/bin/env python3
"""Device-resident time of each operator on ONE cfg2 slab (what a facade call launches) next to its per-slab time
in a 16-slab launch: catches tails that only show with few slabs."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums
ctx = nat.Context(0)
NY, NX, N = 1801, 3600, 201
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
dA = cell_area(lat, lon)
tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True)
e0, e1 = ctx.event(), ctx.event()


def dev_time(fn, reps=10):
    fn(); fn(); ctx.sync()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1) / reps * 1e3


out = {}
for S in (1, 16):
    plan = KeffPlan(ctx, S, NY, NX, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True)
    plan.synth(lat, lon, 1, 2)
    q = plan.q_buf
    mmb = ctx.alloc(S * 16)
    out['minmax', S] = dev_time(lambda: ctx._check(ctx.lib.xc_minmax_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY * NX, mmb.ptr))) / S
    out['keff pipeline', S] = dev_time(lambda: plan.run()) / S
    mm = ctx.minmax(plan.download_q())
    ctr, edges, _ = ctx.levels(mm, np.float64, N, True, np.float64)
    de, dAd, cdf = ctx.to_device(edges), ctx.to_device(dA), ctx.alloc(S * N * 8)
    d = nat.HistDesc()
    d.q, d.q_dtype, d.nslab, d.ny, d.nx = q.ptr, nat.XC_F64, S, NY, NX
    d.edges, d.nedge, d.edges_per_slab, d.last_closed = de.ptr, N + 1, 1, 1
    d.dA, d.dA_rank, d.lt, d.cdf = dAd.ptr, nat.XC_DA_PLANE, 1, cdf.ptr
    out['hist (1 channel)', S] = dev_time(lambda: ctx._check(ctx.lib.xc_hist_dev(ctx.handle, C.byref(d)))) / S
    dc, ol = ctx.to_device(ctr), ctx.alloc(S * N * 8)
    out['crossing stride 1', S] = dev_time(lambda: ctx._check(ctx.lib.xc_crossing_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY, NX, 1, nat.XC_PAD_WRAP, dc.ptr, N, 1,
                                                                                  dAd.ptr, nat.XC_F64, 0, 1, 1, ol.ptr, None))) / S
    g2 = ctx.alloc(S * NY * NX * 8)
    rdx = ctx.to_device(np.ones(NY)); rdy = ctx.to_device(np.ones(NY))
    out['grad2', S] = dev_time(lambda: ctx._check(ctx.lib.xc_grad2_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY, NX, rdx.ptr, rdy.ptr, 1, g2.ptr))) / S
    if S == 1:
        rows = ctx.alloc(NY * 8)
        out['rowsum', 1] = dev_time(lambda: ctx._check(ctx.lib.xc_rowsum_dev(ctx.handle, None, nat.XC_F64, dAd.ptr, nat.XC_DA_PLANE, NY, NX, 0, rows.ptr)))
    nv = ctx.alloc(S * 4)
    out['sort', S] = dev_time(lambda: ctx._check(ctx.lib.xc_sort_profile_batch_dev(ctx.handle, q.ptr, nat.XC_F64, None, nat.XC_F64, 0, dAd.ptr, nat.XC_DA_PLANE,
                                                                                S, NY, NX, 0, None, 0, None, None, 0, None, None, None, nv.ptr, None)), reps=3) / S
    plan.free()
    for b in (mmb, de, dAd, cdf, dc, ol, g2, rdx, rdy, nv):
        b.free()
for name in sorted(set(k[0] for k in out)):
    print('%-22s one slab %8.1f us   16 slabs %8.1f us / slab' % (name, out.get((name, 1), float('nan')), out.get((name, 16), float('nan'))))
