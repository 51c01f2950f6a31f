"""Scratch: device-resident timing of the fused Keff pipeline with a SUPPLIED squared-gradient field (the reference's own
workflow: grdS computed outside and passed as the integrand) next to the in-kernel gradient, cfg2-sized stacks."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums
ctx = nat.Context(0)
NY, NX, N, B = 1801, 3600, 201, 16
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
dA = cell_area(lat, lon)
tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True)
e0, e1 = ctx.event(), ctx.event()
for dt, gdt in ((np.float64, np.float64), (np.float32, np.float32)):
    for supplied in (False, True):
        kw = dict(dA=dA, tbl=tbl, tbl_coord=lat, increase=True, lt=True, out_slabs=B)
        if supplied:
            kw['grdS_dtype'] = gdt
        else:
            kw.update(lat=lat, lon=lon)
        try:
            plan = KeffPlan(ctx, 2 * B, NY, NX, N, dt, dt, **kw)
        except TypeError as e:
            print('KeffPlan signature:', e); break
        plan.synth(lat, lon, 1, 0)
        if supplied:
            plan.grdS_buf.upload(np.ones((2 * B, NY, NX), dtype=gdt))
        for chain in (False, True):
            def step(k):
                s0 = (k % 2) * B; nxt = ((k + 1) % 2) * B
                plan.run_range(0, s0, B, nxt if chain else None, out_s0=0)
            for k in range(4): step(k)
            ctx.sync(); ctx.record(e0)
            for k in range(10): step(k)
            ctx.record(e1)
            ms = ctx.elapsed_ms(e0, e1) / 10
            print(np.dtype(dt).name, 'grdS supplied' if supplied else 'in-kernel gradient', 'chain' if chain else 'plain', '%.1f us/slab' % (ms / B * 1e3), flush=True)
        plan.free()
