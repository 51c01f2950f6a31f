#!/usr/bin/env python3
"""Device-resident timing of K9 (box-counting contour crossing) on cfg2-sized slabs:
S slabs of 1801 x 3600 float64 (xc_synth_dev, PV-like), 201 per-slab levels, 2-D f64 area.
Prints JSON lines; the oracle (all contours, numpy) is timed on one slab beside it."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from xcontour_amd import _native as nat      # noqa: E402
from xcontour_amd.utils import cell_area     # noqa: E402
import xcontour_oracle as O                  # noqa: E402

if os.environ.get('XC_LIB'):
    nat.LIB_PATH = os.path.join(ROOT, 'xcontour_amd', os.environ['XC_LIB'])      # a diagnostic build (tools/build_variant.sh)
S, NY, NX, N = int(os.environ.get('XC_SLABS', '16')), 1801, 3600, 201
ctx = nat.Context(0)
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
dA = cell_area(lat, lon)
q = ctx.alloc(S * NY * NX * 8)
lat_b, lon_b, dA_b = ctx.to_device(lat), ctx.to_device(lon), ctx.to_device(dA)
VAR = int(os.environ.get('XC_VARIANT', '0'))      # 0 PV-like + grid-scale noise, 1 white noise, 2 sin(lat) (smooth)
ctx._check(ctx.lib.xc_synth_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY, NX, lat_b.ptr, lon_b.ptr, 20241008, VAR))
ctx.sync()
q0 = q.download((S, NY, NX), np.float64)[:1]
mm = ctx.minmax(q.download((S, NY, NX), np.float64))
ctr, _, _ = ctx.levels(mm, np.float64, N, True, np.float64)
ctr_b = ctx.to_device(ctr)
out_l, out_c = ctx.alloc(S * N * 8), ctx.alloc(S * N * 8)
e0, e1 = ctx.event(), ctx.event()
STRIDES = [int(t) for t in os.environ.get('XC_STRIDES', '1,2,4').split(',')]
for stride, pad in [(t, t) for t in STRIDES]:
    for full, want_cnt in ((0, 1), (1, 1), (1, 0)):
        def run():
            ctx._check(ctx.lib.xc_crossing_dev(ctx.handle, q.ptr, nat.XC_F64, S, NY, NX, pad, nat.XC_PAD_WRAP, ctr_b.ptr, N, 1,
                                               dA_b.ptr, nat.XC_F64, 0, stride, full, out_l.ptr, out_c.ptr if want_cnt else None))
        for _ in range(2):
            run()
        ctx.record(e0)
        for _ in range(5):
            run()
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 5
        Jn, In = O.crossing_shape(NY, NX + pad, stride)
        nbi = (In - 1) if full else (min(Jn, In) - 1)
        cells = (Jn - 1) * stride * nbi * stride                 # fine cells under the scanned boxes
        by = cells * 8 + (Jn - 1) * nbi * 8                      # tracer once + one area value per box
        cnt = out_c.download((S, N), np.uint64)
        rec = {'kernel': 'k_crossing', 'variant': VAR, 'counts': bool(want_cnt), 'stride': stride, 'pad_x': pad, 'full_width': bool(full), 'slabs': S,
               'us_per_slab': ms / S * 1e3, 'scanned_cells_per_slab': cells, 'algorithmic_GBps': by * S / ms / 1e6,
               'cells_contours_per_s': cells * N * S / ms * 1e3, 'crossed_boxes_per_slab': float(cnt.sum() / S)}
        if stride == 1 and pad == 1 and want_cnt and os.environ.get('XC_CPU', '1') == '1':
            t = time.perf_counter()
            ol, oc = O.contour_crossing(O.pad_x(q0[0], pad, 'wrap'), ctr[0], O.pad_x(dA, pad, 'wrap'), stride, bool(full))
            rec['cpu_oracle_s_per_slab'] = time.perf_counter() - t
            rec['counts_equal_oracle'] = bool(np.array_equal(cnt[0].astype(np.int64), oc))
            rec['len_rel_err'] = float(np.max(np.abs(out_l.download((S, N), np.float64)[0] - ol) / np.maximum(ol, 1)))
        print(json.dumps(rec), flush=True)
