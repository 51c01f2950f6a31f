import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import xcontour_oracle as O
from xcontour_amd import _native as nat
ctx = nat.Context(0)
rng = np.random.default_rng(0)
# many tiny slabs
S, ny, nx = 70000, 8, 16
q = rng.standard_normal((S, ny, nx)).astype(np.float32)
try:
    mm = ctx.minmax(q); print('minmax ok', np.array_equal(mm[:,0], q.reshape(S,-1).min(1)))
except Exception as e: print('minmax ERR', e)
try:
    ed = np.linspace(-3,3,12)
    out = ctx.hist(q, ed, dA=np.ones((ny,nx)), want=('counts',))
    ref = np.stack([np.histogram(q[s], bins=ed)[0] for s in (0, 1, S-1)])
    print('hist ok', np.array_equal(out['counts'][[0,1,S-1]].astype(np.int64), ref))
except Exception as e: print('hist ERR', e)
# one huge-ish row count / narrow
for (ny, nx) in [(1, 5000), (5000, 1), (2, 2), (70000, 3), (3, 70000)]:
    q = rng.standard_normal((2, ny, nx))
    try:
        ed = np.linspace(-3,3,7)
        out = ctx.hist(q, ed, dA=np.ones((ny,nx)), want=('counts','cdf'))
        ref = np.stack([np.histogram(q[s], bins=ed)[0] for s in range(2)])
        print((ny,nx), 'hist ok', np.array_equal(out['counts'].astype(np.int64), ref))
    except Exception as e: print((ny,nx), 'hist ERR', e)
    try:
        if ny >= 2:
            g = ctx.grad2(q, np.ones(ny), np.ones(ny), True); print((ny,nx),'grad2 ok', g.shape)
    except Exception as e: print((ny,nx), 'grad2 ERR', e)
    try:
        l, c = ctx.crossing(q, np.linspace(-2,2,5), np.ones((ny,nx)), 1, 1, 'wrap', True)
        ol, oc = O.contour_crossing(O.pad_x(q[0],1,'wrap'), np.linspace(-2,2,5), O.pad_x(np.ones((ny,nx)),1,'wrap'), 1, True)
        print((ny,nx),'crossing ok', np.array_equal(c[0].astype(np.int64), oc))
    except Exception as e: print((ny,nx), 'crossing ERR', e)
# many contours
q = rng.standard_normal((1, 300, 400))
for N in (2, 3000, 6000):
    try:
        ed = np.linspace(-4,4,N+1)
        out = ctx.hist(q, ed, dA=np.ones((300,400)), want=('counts',))
        print(N, 'bins ok', np.array_equal(out['counts'][0].astype(np.int64), np.histogram(q[0], bins=ed)[0]))
    except Exception as e: print(N, 'bins ERR', e)
    try:
        cs = np.linspace(-3,3,N)
        l, c = ctx.crossing(q, cs, np.ones((300,400)), 1, 1, 'edge', True)
        ol, oc = O.contour_crossing(O.pad_x(q[0],1,'edge'), cs, np.ones((300,401)), 1, True)
        print(N, 'crossing ok', np.array_equal(c[0].astype(np.int64), oc))
    except Exception as e: print(N, 'crossing ERR', e)
# sort extremes
for n in [(1,1),(1,63),(3,4097),(1,5)]:
    q = rng.standard_normal(n)
    try:
        r = ctx.sort_profile(q, dA=None, targets=np.array([0.5, 2.0]), want_sorted=True)
        print(n, 'sort ok', np.array_equal(r['q_sorted'][:q.size], np.sort(q.ravel())))
    except Exception as e: print(n, 'sort ERR', repr(e)[:200])
