#!/usr/bin/env python3
"""Streaming read bandwidth of K1 (min/max) by working-set size: does a set that fits the 256 MiB Infinity Cache stream
faster than one that does not?  (decides whether re-reading a slab shortly after its first read is cheaper than HBM)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat
ctx = nat.Context(0)
ny, nx = 1801, 3600
lat = ctx.to_device(np.linspace(-90, 90, ny)); lon = ctx.to_device(np.arange(nx) * 0.1)
e0, e1 = ctx.event(), ctx.event()
out = ctx.alloc(64 * 16)
for S in (1, 2, 3, 4, 6, 8, 16, 64):
    buf = ctx.alloc(S * ny * nx * 8)
    ctx._check(ctx.lib.xc_synth_dev(ctx.handle, buf.ptr, nat.XC_F64, S, ny, nx, lat.ptr, lon.ptr, 1, 0))
    for _ in range(3):
        ctx._check(ctx.lib.xc_minmax_dev(ctx.handle, buf.ptr, nat.XC_F64, S, ny * nx, out.ptr))
    ctx.record(e0)
    K = max(4, 64 // S)
    for _ in range(K):
        ctx._check(ctx.lib.xc_minmax_dev(ctx.handle, buf.ptr, nat.XC_F64, S, ny * nx, out.ptr))
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1) / K
    print('working set %6.1f MB: %.3f ms per pass = %.2f TB/s (incl. the small final kernel)' % (S * ny * nx * 8 / 1e6, ms, S * ny * nx * 8 / ms / 1e9), flush=True)
    buf.free()
