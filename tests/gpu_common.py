"""helpers shared by the -m gpu test files (no tests here)"""
import os

import numpy as np

from test_gpu_parity import rel, RTOL, TIGHT, LMIN_FLOOR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

NINE = ('ctr', 'area', 'intgrdS', 'latEq', 'dqdA', 'dintSdA', 'Leq2', 'Lmin', 'nkeff')

def check_nine(out, s, r, with_eq=False):
    """all nine result vectors of slab `s` against the oracle's dict `r`"""
    assert np.array_equal(out['counts'][s].astype(np.int64), r['counts'])
    assert np.array_equal(out['ctr'][s], r['ctr'].astype(np.float64))
    assert rel(out['area'][s], r['area']) < TIGHT and rel(out['intgrdS'][s], r['intgrdS']) < TIGHT
    for k in ('latEq', 'dqdA', 'dintSdA', 'Leq2'):
        assert rel(out[k][s], r[k]) < RTOL, k
    assert rel(out['Lmin'][s], r['Lmin'], LMIN_FLOOR) < RTOL
    ok = r['Lmin'] > LMIN_FLOOR
    assert rel(out['nkeff'][s][ok], r['nkeff'][ok]) < RTOL
    if with_eq:
        for k in ('ctr', 'area', 'intgrdS', 'latEq'):
            assert rel(out[k + '_eq'][s], r[k + '_eq']) < RTOL, k

def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)

def check_nine_det(out, s, r):
    """check_nine for the fixed-point sums.  Lmin = 2 pi R cos(latEq) of a contour that encloses all but ~1e-13 of the sphere
    is a 1e-4 m quantity on a 4e7 m scale whose value IS the rounding of the area sum (cos near 90 degrees): such contours
    (Lmin below one metre) are compared through latEq only."""
    assert np.array_equal(out['counts'][s].astype(np.int64), r['counts'])
    assert np.array_equal(out['ctr'][s], r['ctr'].astype(np.float64))
    assert rel(out['area'][s], r['area']) < TIGHT and rel(out['intgrdS'][s], r['intgrdS']) < TIGHT
    for k in ('latEq', 'dqdA', 'dintSdA', 'Leq2'):
        assert rel(out[k][s], r[k]) < RTOL, k
    ok = r['Lmin'] > 1.0
    assert rel(out['Lmin'][s][ok], r['Lmin'][ok]) < RTOL
    assert rel(out['nkeff'][s][ok], r['nkeff'][ok]) < RTOL

def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'XC_DIST_TOKEN')}
