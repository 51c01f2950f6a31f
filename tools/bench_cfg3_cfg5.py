#!/usr/bin/env python3
"""Secondary configs of BASELINE.json, device-resident timings next to the numpy oracle:
  cfg3: barotropic_vorticity local wave activity (256x512 f32, 121 contours for the sorted
        state, J = 256 target latitudes), K7;
  cfg5 stand-in: exact adiabatic sort + background-state integral of an X-Z section
        (internalwave.nc is missing: synthetic nz x 4480 buoyancy with topography), K8;
  plus the K8 sort of one cfg2-sized slab (6.48 M pairs).
Prints one JSON object per config.  Run on a GPU box: python tools/bench_cfg3_cfg5.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from xcontour_amd import _native as nat   # noqa: E402
import xcontour_oracle as O               # noqa: E402

ctx = nat.Context(0)
e0, e1 = ctx.event(), ctx.event()


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    return ctx.elapsed_ms(e0, e1) / reps


# ---------------------------------------------------------------- cfg3
g = os.path.join(ROOT, 'tests', 'golden')
q = np.load(g + '/baro_q.npy'); lat = np.load(g + '/baro_lat.npy'); lon = np.load(g + '/baro_lon.npy')
L = np.load(g + '/baro_lwa_N121.npz')
dA = O.cell_area(lat, lon)
dq, dQ, dc = ctx.to_device(q), ctx.to_device(L['Q']), ctx.to_device(lat.astype(np.float64))
dd, dM = ctx.to_device(dA), ctx.to_device(L['dy'])
out = ctx.alloc(q.size * 8)
ms = timed(lambda: ctx._check(ctx.lib.xc_lwa_dev(ctx.handle, dq.ptr, nat.XC_F32, dQ.ptr, dc.ptr, dd.ptr, nat.XC_DA_PLANE,
                                                float(dA.max()), dM.ptr, nat.XC_DA_ROW, 1, 256, 512, 1, 0, 0, None, 0,
                                                out.ptr, None)))
t = time.perf_counter(); ref = O.cal_local_wave_activity(q, L['Q'], lat, dA, True, 'all', metric=L['dy']); tc = time.perf_counter() - t
got = out.download((256, 512), np.float64)
work = 256 * 256 * 512
print(json.dumps({'config': 'cfg3: barotropic LWA, 256x512 f32, J=256', 'kernel': 'k_lwa', 'gpu_ms': ms,
                  'gpu_cell_rows_per_s': work / ms * 1e3, 'cpu_oracle_s': tc, 'cpu_cell_rows_per_s': work / tc,
                  'bit_identical_to_oracle': bool(np.array_equal(got, ref)), 'lwa_max': float(got.max())}))

# ---------------------------------------------------------------- cfg5 stand-in
nz, nxx = 100, 4480
Z = -(np.arange(nz) + 0.5) * 2.0
X = (np.arange(nxx) + 0.5) * 2.0
xx, zz = np.meshgrid(X, Z)
T = 20 + 5 * np.tanh((zz + 60 + 15 * np.sin(2 * np.pi * xx / 3000.0)) / 20.0)
depth = 200 - 80 * np.exp(-((X - 4480.0) / 1500.0) ** 2)
maskC = (zz > -depth[None, :]).astype(np.float64)
b = 2e-4 * (np.where(maskC == 1, T, np.nan) - 20) * 9.81
yA = np.full((nz, nxx), 4.0)
tbl, cs = O.cal_area_eqCoord_table_hist(maskC, yA, Z, False, False)
db, dmk, dya = ctx.to_device(b), ctx.to_device(maskC), ctx.to_device(yA)
dt_, dcs = ctx.to_device(tbl), ctx.to_device(cs)
dQx = ctx.alloc(nz * 8); nv = ctx.alloc(64); dbpe = ctx.alloc(8)
ms = timed(lambda: ctx._check(ctx.lib.xc_sort_profile_dev(ctx.handle, db.ptr, nat.XC_F64, dmk.ptr, nat.XC_F64, dya.ptr,
                                                         nat.XC_DA_PLANE, nz, nxx, 0, dt_.ptr, nz, dt_.ptr, dcs.ptr, nz,
                                                         dQx.ptr, None, None, nv.ptr, dbpe.ptr)))
t = time.perf_counter(); bo = O.bpe_integral(b, yA, tbl, cs, maskC); tc = time.perf_counter() - t
bg = float(dbpe.download((1,), np.float64)[0])
print(json.dumps({'config': 'cfg5 stand-in: X-Z section %dx%d f64, topography, exact sort + Q(z*) + BPE integral' % (nz, nxx),
                  'kernel': 'K8 radix sort (8 passes) + scan + profile + integral', 'gpu_ms': ms,
                  'cells_per_s': nz * nxx / ms * 1e3, 'cpu_oracle_s': tc, 'bpe_rel_err': abs(bg / bo - 1)}))

# ---------------------------------------------------------------- cfg5 as a stack: 3 sections in one set of launches
b3 = np.stack([b, b * 1.01, b[:, ::-1]])
db3 = ctx.to_device(b3)
dQ3 = ctx.alloc(3 * nz * 8); nv3 = ctx.alloc(64); dbpe3 = ctx.alloc(3 * 8)
ms3 = timed(lambda: ctx._check(ctx.lib.xc_sort_profile_batch_dev(ctx.handle, db3.ptr, nat.XC_F64, dmk.ptr, nat.XC_F64, 0, dya.ptr,
                                                                nat.XC_DA_PLANE, 3, nz, nxx, 0, dt_.ptr, nz, dt_.ptr, dcs.ptr, nz,
                                                                dQ3.ptr, None, None, nv3.ptr, dbpe3.ptr)))
print(json.dumps({'config': 'cfg5 stand-in x 3 time steps, one batched (segmented) sort', 'gpu_ms': ms3,
                  'ms_per_section': ms3 / 3, 'one_section_alone_ms': ms}))

# ---------------------------------------------------------------- 16 barotropic-sized f32 planes: loop vs batch
qb16 = np.stack([q * (1 + 0.01 * k) for k in range(16)]).astype(np.float32)
dqb = ctx.to_device(qb16)
dAb = ctx.to_device(dA)
tgt = ctx.to_device(np.cumsum(dA.sum(1)))
Qo = ctx.alloc(16 * 256 * 8); nvb = ctx.alloc(64)
msb = timed(lambda: ctx._check(ctx.lib.xc_sort_profile_batch_dev(ctx.handle, dqb.ptr, nat.XC_F32, None, nat.XC_F64, 0, dAb.ptr, nat.XC_DA_PLANE,
                                                                16, 256, 512, 0, tgt.ptr, 256, None, None, 0, Qo.ptr, None, None, nvb.ptr, None)))
ms1 = timed(lambda: ctx._check(ctx.lib.xc_sort_profile_dev(ctx.handle, dqb.ptr, nat.XC_F32, None, nat.XC_F64, dAb.ptr, nat.XC_DA_PLANE,
                                                          256, 512, 0, tgt.ptr, 256, None, None, 0, Qo.ptr, None, None, nvb.ptr, None)))
print(json.dumps({'config': '16 planes of 256x512 f32 (32-bit keys, 4 passes), exact sorted profile at 256 targets', 'batched_gpu_ms': msb,
                  'ms_per_plane_batched': msb / 16, 'ms_per_plane_alone': ms1}))

# ---------------------------------------------------------------- K8 on a cfg2-sized slab
n = 1801 * 3600
qq = np.random.default_rng(0).standard_normal((1801, 3600))
dqq = ctx.to_device(qq)
ms = timed(lambda: ctx._check(ctx.lib.xc_sort_profile_dev(ctx.handle, dqq.ptr, nat.XC_F64, None, nat.XC_F64, None, 0, 1801,
                                                         3600, 0, None, 0, None, None, 0, None, None, None, nv.ptr, None)), reps=5)
t = time.perf_counter(); np.sort(qq.ravel(), kind='stable'); tc = time.perf_counter() - t
print(json.dumps({'config': 'K8: 6 483 600 (f64 key, f64 payload) pairs', 'gpu_ms': ms, 'pairs_per_s': n / ms * 1e3,
                  'pass_traffic_GBps': 8 * 40 * n / ms / 1e6, 'numpy_stable_sort_s': tc}))
