"""Property tests (hypothesis) of the oracle and of the host-side logic -- CPU only.
The same invariants are asserted for the HIP path at full size in tests/test_gpu_parity.py."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
from hypothesis.extra import numpy as hnp

import xcontour_oracle as O
from xcontour_amd import core
from xcontour_amd.utils import table_from_rowsums

finite = st.floats(min_value=-1e6, max_value=1e6, allow_nan=False, allow_infinity=False, width=32)


@settings(max_examples=60, deadline=None)
@given(hnp.arrays(np.float32, hnp.array_shapes(min_dims=2, max_dims=2, min_side=2, max_side=24), elements=finite),
       st.integers(2, 40), st.booleans(), st.booleans(), st.sampled_from([np.float32, np.float64]))
def test_cdf_invariants(q, N, increase, lt, dtype):
    if q.min() == q.max():
        return                                             # the reference raises 'non monotonic bins'
    rng = np.random.default_rng(q.size)
    dA = rng.random(q.shape) + 0.1
    ctr = O.cal_contours(q, N, increase, dtype)
    if not np.diff(ctr).all():
        return
    assert (ctr[0] == (q.min() if increase else q.max()))
    cdf, cnt = O.cal_integral_within_contours_hist(q, ctr, dA, None, lt, return_counts=True)
    d = np.diff(cdf if increase == lt else cdf[::-1])
    assert (d >= -1e-9 * dA.sum()).all()                   # monotone in the direction the flags imply
    assert cnt.sum() <= q.size and q.size - cnt.sum() <= (q == (q.max() if increase else q.min())).sum() + \
        (q == (q.min() if increase else q.max())).sum()    # only extreme cells can fall off the rounded end levels
    assert 0 <= cdf.min() and cdf.max() <= dA.sum() * (1 + 1e-12)
    # hist API == strict conditional API wherever no cell sits exactly on a level (tests/test_hist.py overlay)
    strict = O.cal_integral_within_contours(q, ctr, dA, None, lt)
    on_edge = np.isin(ctr.astype(np.float64), q.astype(np.float64))
    k = np.arange(N)
    inner = ~on_edge & (k > 0) & (k < N - 1)
    if cnt.sum() != q.size:
        return                                             # an extreme cell fell off the rounded end level
    assert np.allclose(cdf[inner], strict[inner], rtol=1e-10, atol=1e-9 * dA.sum())


@settings(max_examples=60, deadline=None)
@given(hnp.arrays(np.float64, st.integers(3, 60), elements=st.floats(-1e3, 1e3, allow_nan=False), unique=True),
       st.booleans(), st.sampled_from([np.float32, np.float64]))
def test_edges_from_levels_property(b, descending, dtype):
    b = np.sort(b).astype(dtype)
    if not np.diff(b).all():
        return
    if descending:
        b = b[::-1].copy()
    e, binc, closed = core._edges_from_levels(b[None, :], 'numpy')
    eo, bo = O.hist_edges(b)
    assert binc == bo == (not descending) and closed
    assert np.array_equal(e[0], eo.astype(np.float64)) and (np.diff(e[0]) > 0).all()


@settings(max_examples=60, deadline=None)
@given(hnp.arrays(np.float64, st.integers(2, 50), elements=st.floats(0.01, 1e6)), st.booleans())
def test_table_from_rowsums_property(r, ylt):
    t = table_from_rowsums(r, ylt)
    tot = r.sum()
    if ylt:
        assert t[0] == 0 and abs(t[-1] - tot) <= 1e-12 * tot and (np.diff(t) >= 0).all()
    else:
        assert t[-1] == 0 and abs(t[0] - tot) <= 1e-12 * tot and (np.diff(t) <= 1e-9 * tot).all()


@settings(max_examples=40, deadline=None)
@given(hnp.arrays(np.float64, hnp.array_shapes(min_dims=2, max_dims=2, min_side=2, max_side=16),
                  elements=st.floats(-100, 100, allow_nan=False)))
def test_sorted_profile_property(q):
    rng = np.random.default_rng(q.size)
    dA = rng.random(q.shape) + 0.1
    targets = np.linspace(0, dA.sum(), 9)
    Q, xs, acum = O.sorted_profile(q, dA, targets)
    assert (np.diff(xs) >= 0).all() and (np.diff(Q) >= 0).all()
    assert abs(acum[-1] - dA.sum()) <= 1e-12 * dA.sum() and Q[0] == xs[0] and Q[-1] == xs[-1]


def test_sorted_profile_of_a_plane_without_valid_cells():
    """an all-NaN (or fully masked) plane has no sorted state: NaN for every target, empty sorted arrays"""
    q = np.full((3, 4), np.nan)
    Q, xs, acum = O.sorted_profile(q, np.ones((3, 4)), [0.0, 1.0])
    assert np.isnan(Q).all() and Q.shape == (2,) and len(xs) == 0 and len(acum) == 0
    Q, xs, acum = O.sorted_profile(np.arange(12.0).reshape(3, 4), np.ones((3, 4)), [0.0], mask=np.zeros((3, 4)))
    assert np.isnan(Q).all() and len(xs) == 0


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 20000), st.integers(1, 16))
def test_shard_slabs_property(S, G):
    from xcontour_amd.pipeline import shard_slabs
    parts = [shard_slabs(S, r, G) for r in range(G)]
    assert parts[0][0] == 0 and parts[-1][1] == S
    assert all(parts[i][1] == parts[i + 1][0] for i in range(G - 1))
    assert all(0 <= hi - lo <= -(-S // G) for lo, hi in parts)
