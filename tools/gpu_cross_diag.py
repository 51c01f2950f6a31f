"""Scratch: per-level error of K9 against the oracle on a small noisy slab."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from xcontour_amd import _native as nat
import xcontour_oracle as O
ctx = nat.Context(0)
rng = np.random.default_rng(3)
ny, nx, N = 200, 300, 21
lat = np.linspace(-80, 80, ny)
q = (np.sin(np.deg2rad(lat))[:, None] + 0.1 * rng.standard_normal((ny, nx)))[None]
area = (np.cos(np.deg2rad(lat))[:, None] * np.ones((ny, nx))) * 1e8 + 1.0
cs = np.linspace(q.min(), q.max(), N)
lens, cnts = ctx.crossing(q, cs, area, stride=1, pad_x=1, pad_mode='wrap', full_width=True)
ol, oc = O.contour_crossing(O.pad_x(q[0], 1, 'wrap'), cs, O.pad_x(area, 1, 'wrap'), 1, True)
print('counts equal', np.array_equal(cnts[0].astype(np.int64), oc))
print('rel err per level', np.array2string((lens[0] - ol) / np.maximum(ol, 1), precision=2))
print('abs err / typical w', np.array2string((lens[0] - ol) / np.sqrt(area.mean()), precision=3))
