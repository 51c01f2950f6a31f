#!/usr/bin/env python3
"""Generate the committed golden vectors from the oracle on the barotropic fixture.

    python tests/golden/make_golden.py        (numpy only; run in the build container)

Inputs : tests/golden/baro_{q,lat,lon}.npy   (extract_barotropic.py)
Outputs: tests/golden/baro_keff_N{121,201}.npz  Keff call sequence, increase, lt,
                                                 float32 contours, preLats = lat, last-bin rule
                                                 'xhistogram' (the default); ..._numpy.npz: rule 'numpy'
         tests/golden/baro_contours_at.npz       cal_contours_at(_hist) (core.py:269-360)
         tests/golden/baro_lwa_N121.npz          sorted state Q(lat) + LWA (legacy dy metric and
                                                 snapshot dA metric), masks at rows 37/125/170/213
The oracle is pinned against SURVEY.md 8(c)'s known answers in tests/test_oracle_golden.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
import xcontour_oracle as O  # noqa: E402

q = np.load(os.path.join(HERE, 'baro_q.npy'))
lat = np.load(os.path.join(HERE, 'baro_lat.npy'))
lon = np.load(os.path.join(HERE, 'baro_lon.npy'))
dA = O.cell_area(lat, lon)

# both last-bin rules: 'xhistogram' (default = what the reference runs; file names without suffix) and
# 'numpy' (closed last bin; the rule SURVEY.md 8(c)'s known answers were derived with)
for rule, sfx in (('xhistogram', ''), ('numpy', '_numpy')):
    for N in (121, 201):
        r = O.keff_pipeline(q, dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float32, preLats=lat,
                            right_edge=rule)
        np.savez_compressed(os.path.join(HERE, 'baro_keff_N%d%s.npz' % (N, sfx)), **r)

r = O.keff_pipeline(q, dA, lat, 121, lon=lon, increase=True, lt=True, dtype=np.float32, preLats=lat)
Q = r['ctr_eq']
dy = np.gradient(np.deg2rad(lat.astype(np.float64))) * O.Rearth
idx = [37, 125, 170, 213]
lwa_dy, _, masks = O.cal_local_wave_activity(q, Q, lat, dA, True, 'all', idx, metric=dy)
lwa_dA = O.cal_local_wave_activity(q, Q, lat, dA, True, 'all')
lwa_up = O.cal_local_wave_activity(q, Q, lat, dA, True, 'upper', metric=dy)
lwa_lo = O.cal_local_wave_activity(q, Q, lat, dA, True, 'lower', metric=dy)
np.savez_compressed(os.path.join(HERE, 'baro_lwa_N121.npz'), Q=Q, dy=dy, lwa_dy=lwa_dy, lwa_dA=lwa_dA,
                    lwa_upper=lwa_up, lwa_lower=lwa_lo, mask_idx=np.array(idx),
                    masks=np.stack(masks).astype(np.int8))

# f3: contours at prescribed equivalent latitudes (core.py:269-360), both twins x both rules x (increase, lt)
pre = np.linspace(-85, 85, 69).astype(np.float32)
at = {'predef': pre}
for rule in ('xhistogram', 'numpy'):
    for inc in (True, False):
        for lt in (True, False):
            tbl, cs = O.cal_area_eqCoord_table_hist(np.ones_like(q), dA, lat, inc, lt, rule)
            for hist in (True, False):
                qi, _ = O.cal_contours_at(q, pre, tbl, cs, dA, inc, lt, np.float32, hist, rule)
                at['%s_inc%d_lt%d_%s' % (rule, inc, lt, 'hist' if hist else 'cond')] = qi
np.savez_compressed(os.path.join(HERE, 'baro_contours_at.npz'), **at)
print('wrote golden npz files')
