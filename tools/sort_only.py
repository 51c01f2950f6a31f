#!/usr/bin/env python3
"""The cfg5 stand-in's batched sort 200 times (for kernel traces of the K8 chain alone)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xcontour_amd import _native as nat
ctx = nat.Context(0)
rng = np.random.default_rng(0)
S, nz, nx = 3, 100, 4480
q = ctx.to_device(rng.standard_normal((S, nz, nx)))
nv = ctx.alloc(4096)
for _ in range(200):
    ctx._check(ctx.lib.xc_sort_profile_batch_dev(ctx.handle, q.ptr, nat.XC_F64, None, nat.XC_F64, 0, None, nat.XC_DA_NONE, S, nz, nx, 0,
                                                 None, 0, None, None, 0, None, None, None, nv.ptr, None))
ctx.sync()
