#!/usr/bin/env python3
"""Writes baro_crossing_N41.npz: box-counting contour crossing (oracle.contour_crossing) of the
barotropic fixture, 41 float32 levels (increase=True), X padded by 4 columns with mode='wrap',
strides 1 / 2 / 4.  Inputs: baro_*.npy of this directory (see extract_barotropic.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
import xcontour_oracle as O   # noqa: E402

q = np.load(os.path.join(HERE, 'baro_q.npy'))
lat = np.load(os.path.join(HERE, 'baro_lat.npy'))
lon = np.load(os.path.join(HERE, 'baro_lon.npy'))
dA = O.cell_area(lat, lon)
ctr = O.cal_contours(q, 41, True, np.float32)
out = {'ctr': ctr}
dp, ap = O.pad_x(q, 4, 'wrap'), O.pad_x(dA, 4, 'wrap')
for s in (1, 2, 4):
    L, C = O.contour_crossing(dp, ctr, ap, s)
    out['len_s%d' % s] = L.astype(np.float32)
    out['cnt_s%d' % s] = C
out['cnt_sorted_s1'] = out['cnt_s1']            # levels ascend: sorted order == level order
np.savez_compressed(os.path.join(HERE, 'baro_crossing_N41.npz'), **out)

# the reference's own call (tests/test_fractal.py:30-75): N = 121, strides 1..32, mode='edge'
ctr121 = O.cal_contours(q, 121, True, np.float32)
strides = [1, 2, 4, 8, 16, 32]
res = O.cal_contour_crossing(q, ctr121, dA, strides, 'edge')
np.savez_compressed(os.path.join(HERE, 'baro_fractal_N121.npz'), ctr=ctr121, strides=np.array(strides),
                    **{'bclens%d' % s: r for s, r in zip(strides, res)})
print('wrote baro_crossing_N41.npz', out['cnt_s1'][:6], out['len_s1'][:4])
