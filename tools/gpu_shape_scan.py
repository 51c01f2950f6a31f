#!/usr/bin/env python3
"""Timing scan over awkward shapes (per-cell cost relative to the cfg2-like shape) to catch pathological regimes."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat
ctx = nat.Context(0)
rng = np.random.default_rng(0)
e0, e1 = ctx.event(), ctx.event()


def dev_time(fn, reps=5):
    fn(); ctx.sync()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1) / reps


for (S, ny, nx) in [(4, 1801, 3600), (4, 3600, 1801), (64, 256, 512), (1, 64800, 100), (1, 100, 64800), (512, 90, 180), (2, 6000, 6000), (1, 3, 2000000)]:
    cells = S * ny * nx
    q = np.sin(np.linspace(-1.5, 1.5, ny))[None, :, None] + 0.01 * rng.standard_normal((S, ny, nx))
    dq = ctx.to_device(q)
    dA = ctx.to_device(np.ones((ny, nx)))
    N = 201
    mm = ctx.minmax(q)
    ctr, edges, _ = ctx.levels(mm, np.float64, N, True, np.float64)
    de = ctx.to_device(edges)
    cdf = ctx.alloc(S * N * 8)
    d = nat.HistDesc()
    d.q, d.q_dtype, d.nslab, d.ny, d.nx = dq.ptr, nat.XC_F64, S, ny, nx
    d.edges, d.nedge, d.edges_per_slab, d.last_closed = de.ptr, N + 1, 1, 1
    d.dA, d.dA_rank, d.lt, d.cdf = dA.ptr, nat.XC_DA_PLANE, 1, cdf.ptr
    import ctypes as C
    t_hist = dev_time(lambda: ctx._check(ctx.lib.xc_hist_dev(ctx.handle, C.byref(d))))
    mmb = ctx.alloc(S * 16)
    t_mm = dev_time(lambda: ctx._check(ctx.lib.xc_minmax_dev(ctx.handle, dq.ptr, nat.XC_F64, S, ny * nx, mmb.ptr)))
    dc = ctx.to_device(ctr)
    ol, oc = ctx.alloc(S * N * 8), ctx.alloc(S * N * 8)
    t_cr = dev_time(lambda: ctx._check(ctx.lib.xc_crossing_dev(ctx.handle, dq.ptr, nat.XC_F64, S, ny, nx, 1, nat.XC_PAD_WRAP, dc.ptr, N, 1,
                                                             dA.ptr, nat.XC_F64, 0, 1, 1, ol.ptr, oc.ptr)))
    rec = {'shape': [S, ny, nx], 'cells': cells, 'minmax_ns_per_kcell': t_mm / cells * 1e9, 'hist_ns_per_kcell': t_hist / cells * 1e9,
           'crossing_ns_per_kcell': t_cr / cells * 1e9}
    if ny <= 6000 and S * ny * ny * nx < 3e11:
        Q = ctx.to_device(np.sort(q.mean(axis=2), axis=1))
        co = ctx.to_device(np.linspace(-80, 80, ny))
        out = ctx.alloc(cells * 8)
        t_lwa = dev_time(lambda: ctx._check(ctx.lib.xc_lwa_dev(ctx.handle, dq.ptr, nat.XC_F64, Q.ptr, co.ptr, dA.ptr, nat.XC_DA_PLANE, 1.0,
                                                             None, nat.XC_DA_NONE, S, ny, nx, 1, 0, 0, None, 0, out.ptr, None)), reps=2)
        rec['lwa_ms'] = t_lwa
        for b in (Q, co, out):
            b.free()
    if ny * nx < 2 ** 31 and cells * 40 < 20e9:
        nv = ctx.alloc(S * 4)
        t_sort = dev_time(lambda: ctx._check(ctx.lib.xc_sort_profile_batch_dev(ctx.handle, dq.ptr, nat.XC_F64, None, nat.XC_F64, 0, None, nat.XC_DA_NONE,
                                                                            S, ny, nx, 0, None, 0, None, None, 0, None, None, None, nv.ptr, None)), reps=3)
        rec['sort_ns_per_kcell'] = t_sort / cells * 1e9
        nv.free()
    print(json.dumps(rec), flush=True)
    for b in (dq, dA, de, cdf, mmb, dc, ol, oc):
        b.free()
