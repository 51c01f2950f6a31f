import sys, numpy as np
sys.path.insert(0, ".")
from xcontour_amd import _native as nat
import os
if os.environ.get("XC_LIB"):
    nat.LIB_PATH = os.path.join("xcontour_amd", os.environ["XC_LIB"])      # a diagnostic build (tools/build_variant.sh)
ctx = nat.Context(0)
n = 1801*3600
q = np.random.default_rng(0).standard_normal((1801, 3600))
dq = ctx.to_device(q)
e0, e1 = ctx.event(), ctx.event()
nv = ctx.alloc(64)
for rep in range(4):
    ctx.record(e0)
    ctx._check(ctx.lib.xc_sort_profile_dev(ctx.handle, dq.ptr, 1, None, 1, None, 0, 1801, 3600, 0, None, 0, None, None, 0, None, None, None, nv.ptr, None))
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)
    print("sort 6.48M (key f64, payload f64): %.3f ms -> %.2f Gpairs/s; pass traffic %.0f GB/s" % (ms, n/ms/1e6, 8*(8+32)*n/ms/1e6))
q32 = q.astype(np.float32)
dq32 = ctx.to_device(q32)
for rep in range(3):
    ctx.record(e0)
    ctx._check(ctx.lib.xc_sort_profile_dev(ctx.handle, dq32.ptr, 0, None, 1, None, 0, 1801, 3600, 0, None, 0, None, None, 0, None, None, None, nv.ptr, None))
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1)
    print("sort 6.48M (key f32 -> 32-bit keys, payload f64, 4 passes): %.3f ms -> %.2f Gpairs/s; pass traffic %.0f GB/s" % (ms, n/ms/1e6, 4*(4+24)*n/ms/1e6))
