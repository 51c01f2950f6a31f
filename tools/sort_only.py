#!/usr/bin/env python3
"""A batched sort 200 (or XC_REPS) times, for kernel traces of the K8 chain alone: the cfg5 stand-in's 3 x 100 x 4480 planes by default,
`python tools/sort_only.py S NY NX` for another stack (1 1801 3600: the 6.48 M-pair sort of a cfg2 slab)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xcontour_amd import _native as nat
ctx = nat.Context(0)
rng = np.random.default_rng(0)
S, nz, nx = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 100, 4480)
q = ctx.to_device(rng.standard_normal((S, nz, nx)))
nv = ctx.alloc(4096)
for _ in range(int(os.environ.get('XC_REPS', '200'))):
    ctx._check(ctx.lib.xc_sort_profile_batch_dev(ctx.handle, q.ptr, nat.XC_F64, None, nat.XC_F64, 0, None, nat.XC_DA_NONE, S, nz, nx, 0,
                                                 None, 0, None, None, 0, None, None, None, nv.ptr, None))
ctx.sync()
