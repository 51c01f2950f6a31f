"""Scratch: first GPU sanity run of the native layer against the oracle."""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
spec = importlib.util.spec_from_file_location('xcnat', os.path.join(ROOT, 'xcontour_amd', '_native.py'))
nat = importlib.util.module_from_spec(spec); spec.loader.exec_module(nat)
import xcontour_oracle as O

ctx = nat.Context(0)
print(ctx.device_name(), ctx.device_cus())
g = os.path.join(ROOT, 'tests', 'golden')
q = np.load(g + '/baro_q.npy'); lat = np.load(g + '/baro_lat.npy'); lon = np.load(g + '/baro_lon.npy')
dA = O.cell_area(lat, lon)

mm = ctx.minmax(q[None])
print('minmax', mm, q.min(), q.max(), mm[0, 0] == q.min(), mm[0, 1] == q.max())
for N in (121, 201):
    for inc in (True, False):
        for cd in (np.float32, np.float64):
            ctr, edges, st = ctx.levels(mm, q.dtype, N, inc, cd)
            ref = O.cal_contours(q, N, inc, cd)
            e_ref, _ = O.hist_edges(ref)
            print('levels', N, inc, cd.__name__, np.array_equal(ctr[0], ref.astype(np.float64)),
                  np.array_equal(edges[0], e_ref.astype(np.float64)), st)
ctr = O.cal_contours(q, 121, True, np.float32)
edges, _ = O.hist_edges(ctr)
rdx, rdy = O.grad_metrics(lat, lon)
g2 = O.grad2_sphere(q, lat, lon)
t = time.time()
out = ctx.hist(q[None], edges.astype(np.float64), dA=dA, grad=(rdx, rdy, True), lt=True)
print('hist time (incl staging)', time.time() - t)
pdf0, cnt = O.weighted_histogram(q, edges, dA)
pdf1, _ = O.weighted_histogram(q, edges, np.where(np.isnan(g2 * dA), 0, g2 * dA))
print('counts equal', np.array_equal(out['counts'][0].astype(np.int64), cnt), cnt.sum(), out['counts'].sum())
print('pdf dA rel', np.max(np.abs(out['pdf'][0, 0] - pdf0) / np.maximum(np.abs(pdf0), 1e-300)))
print('pdf g2 rel', np.max(np.abs(out['pdf'][0, 1] - pdf1) / np.maximum(np.abs(pdf1), 1e-300)))
print('cdf rel', np.max(np.abs(out['cdf'][0, 0] - np.cumsum(pdf0)) / np.maximum(np.cumsum(pdf0), 1e-300)))
# supplied integrand path
out2 = ctx.hist(q[None], edges.astype(np.float64), dA=dA, integrands=[g2[None]], lt=True)
print('pdf integrand rel', np.max(np.abs(out2['pdf'][0, 1] - pdf1) / np.maximum(np.abs(pdf1), 1e-300)))
# random f64 data, odd nx, NaNs
rng = np.random.default_rng(0)
x = rng.standard_normal((3, 37, 131)); x[0, 3, 5] = np.nan; x[1, :, 7] = np.nan
w = rng.random((37, 131))
ed = np.linspace(-2, 2, 52)
o3 = ctx.hist(x, ed, dA=w)
for s in range(3):
    p, c = O.weighted_histogram(x[s], ed, w)
    print('rand slab', s, np.array_equal(o3['counts'][s].astype(np.int64), c), np.max(np.abs(o3['pdf'][s, 0] - p)))
# rowsum, grad2, lwa
rs = ctx.rowsum(np.ones_like(q), dA, 256, 512)
print('rowsum rel', np.max(np.abs(rs - dA.sum(1)) / dA.sum(1)))
gg = ctx.grad2(q[None], rdx, rdy, True)
print('grad2 equal', np.array_equal(gg[0], g2), np.nanmax(np.abs(gg[0] - g2) / np.maximum(g2, 1e-300)))
r = O.keff_pipeline(q, dA, lat, 121, lon=lon, preLats=lat)
Q = r['ctr_eq']
dy = np.gradient(np.deg2rad(lat.astype(np.float64))) * O.Rearth
t = time.time()
lw, mo = ctx.lwa(q[None], Q[None], lat, dA, dA.max(), M=dy, increase=True, part=0, mask_idx=[37, 125, 170, 213])
print('lwa time', time.time() - t)
lref, cs, ms = O.cal_local_wave_activity(q, Q, lat, dA, True, 'all', [37, 125, 170, 213], metric=dy)
print('lwa max', lw.max(), 'equal', np.array_equal(lw[0], lref), np.max(np.abs(lw[0] - lref)) / lref.max())
print('masks equal', all(np.array_equal(mo[0, i], ms[i]) for i in range(4)))
for part, pn in ((1, 'upper'), (2, 'lower')):
    lw2, _ = ctx.lwa(q[None], Q[None], lat, dA, dA.max(), M=None, increase=True, part=part)
    lr2 = O.cal_local_wave_activity(q, Q, lat, dA, True, pn)
    print('lwa', pn, np.max(np.abs(lw2[0] - lr2)) / np.abs(lr2).max())
print('DONE')
