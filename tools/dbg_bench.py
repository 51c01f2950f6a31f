import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums
NY, NX, N = 1801, 3600, 201
ctx = nat.Context(0)
lat = np.linspace(-90, 90, NY); lon = np.arange(NX) * 0.1
dA = cell_area(lat, lon); tbl = table_from_rowsums(dA.sum(1), True)
B, K = 8, 3
slot = KeffPlan.out_bytes(B, N)
res = torch.zeros(slot * K // 8, dtype=torch.float64, device='cuda')
plan = KeffPlan(ctx, B, NY, NX, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, nslots=K, out_ptr=res.data_ptr())
print(slot, plan.slot_bytes)
plan.synth(lat, lon, 1, 0)
for k in range(K):
    plan.run(k, None)
ctx.sync()
for k in range(K):
    out = plan.fetch(slot=k, check=False)
    print(k, out['counts'].sum(axis=1), out['status'], out['area'][:, -1])
plan2 = KeffPlan(ctx, B, NY, NX, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat)
plan2.set_q_device(plan._q_ptr)
plan2.run(); out = plan2.fetch(check=False)
print('own buf', out['counts'].sum(axis=1), out['status'])
