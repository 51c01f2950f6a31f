"""diagnostic: does a single-read Keff launch leave anything behind that breaks ncclCommInitRank in the same process?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
ctx = nat.Context(0)
ny, nx, N = 721, 1440, 201
lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.25
dA = cell_area(lat, lon)
tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
mode = sys.argv[1] if len(sys.argv) > 1 else 'single'
p = KeffPlan(ctx, 1, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, single_read=(mode == 'single'))
p.synth(lat, lon, 1, 0)
if mode != 'none':
    p.run(); r = p.fetch(); print('path', ctx.last_keff_path(), 'status', r['status'])
uid = ctx.comm_unique_id()
try:
    c = ctx.comm_create(1, 0, uid)
    ctx.comm_attach(c, 1, 0)
    print(mode, 'comm ok', ctx.comm_info())
except Exception as e:
    print(mode, 'comm FAILED', e)
