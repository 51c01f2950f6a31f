"""-m gpu: K8 exact adiabatic sort -- the three-pass range-key path with its repair, its fallback, planes without valid cells.
(Regrouped in round 5 from the per-round files of rounds 2-4; nothing dropped.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import xcontour_oracle as O
from test_gpu_parity import rel, RTOL, TIGHT, LMIN_FLOOR, _baro_da
from gpu_common import GOLD, NINE, ROOT, bits, check_nine, check_nine_det, _clean_env

pytestmark = pytest.mark.gpu


def _sort_equals_oracle(ctx, q, dA=None, mask=None, negate=False):
    r = ctx.sort_profile(q, dA=dA, mask=mask, want_sorted=True, want_acum=True, negate=negate)
    _, xs, acum = O.sorted_profile(-q if negate else q, np.ones(q.shape) if dA is None else dA, [0.0], mask)
    n = r['nvalid']
    assert n == len(xs)
    assert np.array_equal(r['q_sorted'][:n], xs)                            # exact order, ties included
    assert rel(r['acum'][:n], acum) < 1e-12                                 # the payload travelled with its key (stable)
    return ctx.last_sort_path()


def test_sort_range_path_fields_ties_masks(ctx):
    """float64 tracers: sorted by three passes over the 24-bit range key + repair of the short runs (path 1), the same
    stable order as the oracle's argsort -- noise, heavy ties (stability decides the payload order), a land mask that
    drops a third of the cells, NaNs, negated input, +-inf, a constant field, a two-cell plane"""
    rng = np.random.default_rng(12)
    ny, nx = 301, 700
    dA = rng.random((ny, nx)) + 0.5
    q = rng.standard_normal((ny, nx))
    assert _sort_equals_oracle(ctx, q, dA) == 1
    ties = rng.integers(0, 40, (ny, nx)).astype(np.float64)
    assert _sort_equals_oracle(ctx, ties, dA) == 1                           # runs of ~5000 equal keys: already in order
    mask = (rng.random((ny, nx)) > 0.33).astype(np.float64)
    qn = q.copy(); qn[::7, ::5] = np.nan
    assert _sort_equals_oracle(ctx, qn, dA, mask) == 1                       # dropped cells gather behind the maximum
    assert _sort_equals_oracle(ctx, q, dA, negate=True) == 1
    qi = q.copy(); qi[3, 4] = np.inf; qi[5, 6] = -np.inf
    assert _sort_equals_oracle(ctx, qi, dA) in (1, 2)                        # an infinite range collapses the range key
    assert _sort_equals_oracle(ctx, np.full((ny, nx), 2.5), dA) == 1
    assert _sort_equals_oracle(ctx, np.array([[3.0, -1.0]])) == 1
    # ties AND inversions inside one range-key run: few distinct values 1e-13 apart, payloads must follow the stable order
    tq = 0.25 + 1e-13 * rng.integers(0, 4, (ny, nx))
    tq[::3] = rng.standard_normal((len(range(0, ny, 3)), nx))
    assert _sort_equals_oracle(ctx, tq, dA) in (1, 2)
    tq2 = np.linspace(0, 1, ny * nx).reshape(ny, nx)
    for off in range(500, ny * nx - 64, 9973):
        tq2.ravel()[off:off + 40] = tq2.ravel()[off] + 1e-14 * rng.integers(0, 3, 40)
    assert _sort_equals_oracle(ctx, tq2, dA) == 1
    # float32 tracers keep the four key passes
    r = ctx.sort_profile(q.astype(np.float32), dA=dA, want_sorted=True)
    assert ctx.last_sort_path() == 0 and np.array_equal(r['q_sorted'], np.sort(q.astype(np.float32).ravel()).astype(np.float64))


def test_sort_range_path_spike_falls_back(ctx):
    """distinct values packed into less than 2^-24 of the (robust) range: the runs of equal range key are thousands of cells long
    and out of order -- the check fails, the eight key passes sort the stack (path 2).  Round 4: a FEW stray cells no longer do
    that (the equalised range runs between the 9th smallest / largest K1 block extrema, strays go to the outer zones: path 1);
    outliers in more blocks than the trim covers still do."""
    rng = np.random.default_rng(13)
    q = 1.0 + 1e-12 * rng.standard_normal((200, 512))
    q[0, 0], q[1, 1] = -5.0, 7.0
    dA = rng.random(q.shape) + 0.5
    assert _sort_equals_oracle(ctx, q, dA) == 1                          # two strays: trimmed, three passes suffice
    q[::7, 3] = -5.0; q[::9, 5] = 7.0                                     # strays in every K1 block: the robust range is [-5, 7] again
    assert _sort_equals_oracle(ctx, q, dA) == 2
    # the same spike in a stack next to a harmless plane: the batch falls back as a whole, every plane is right
    st = np.stack([rng.standard_normal(q.shape), q])
    r = ctx.sort_profile(st, dA=dA, want_sorted=True)
    assert ctx.last_sort_path() == 2
    for s in range(2):
        assert np.array_equal(r['q_sorted'][s], np.sort(st[s].ravel()))
    # an unmasked fill value next to ordinary data (the realistic stray): the field keeps its three passes
    f = rng.standard_normal((300, 700)) * 10 + 280
    f[17, 33] = 1e20; f[250, 600] = -9999.0
    assert _sort_equals_oracle(ctx, f, rng.random(f.shape) + 0.5) == 1
    # runs under / over the repair limit: 50 distinct values inside one range-key bucket are repaired in LDS, 400 are not
    base = np.linspace(0.0, 1.0, 4096 * 8).reshape(64, 512)
    for off in (1000, 2047, 2048 + 17, 4096 - 25):                       # also runs that straddle two repair blocks
        b2 = base.copy()
        b2.ravel()[off:off + 50] = base.ravel()[off] + 1e-13 * rng.permutation(50)
        assert _sort_equals_oracle(ctx, b2) == 1
    b3 = base.copy()
    b3.ravel()[1000:1400] = base.ravel()[1000] + 1e-13 * rng.permutation(400)
    assert _sort_equals_oracle(ctx, b3) == 2


def test_bpe_through_both_paths_and_twice_on_the_fallback(ctx):
    """round 5: the last-arriving k_bpe block sums the partials (no k_sum_parts launch) behind an arrival ticket that the chain's first
    kernel clears -- and that the last arriver clears again, because a stack that fails the range-key check runs the tail TWICE in one
    call.  BPE of a plane that takes path 1, of one that falls back (path 2), and again: the oracle's value every time"""
    rng = np.random.default_rng(3)
    nz, nxx = 64, 512
    Z = -(np.arange(nz) + 0.5) * 2.0
    yA = np.full((nz, nxx), 40.0)
    tbl, cs = O.cal_area_eqCoord_table_hist(np.ones((nz, nxx)), yA, Z, False, False)
    ok = rng.standard_normal((nz, nxx))
    spike = 1.0 + 1e-12 * rng.standard_normal((nz, nxx)); spike[::7, 3] = -5.0; spike[::9, 5] = 7.0
    for _ in range(2):
        for q, want in ((ok, 1), (spike, 2), (ok, 1)):
            r = ctx.sort_profile(q, dA=yA, tbl=tbl, coord=cs)
            assert ctx.last_sort_path() == want
            ref = O.bpe_integral(q, yA, tbl, cs)
            assert abs(r['bpe'] / ref - 1) < 1e-10 and r['nvalid'] == q.size
    st = np.stack([ok, spike, ok * 2])
    r = ctx.sort_profile(st, dA=yA, tbl=tbl, coord=cs)                      # the batch falls back as a whole
    assert ctx.last_sort_path() == 2
    for s_ in range(3):
        assert abs(r['bpe'][s_] / O.bpe_integral(st[s_], yA, tbl, cs) - 1) < 1e-10


def test_sort_profile_of_planes_without_valid_cells(ctx):
    """an all-NaN plane inside a stack: nvalid 0 and NaN for every target on that plane (the oracle's rule), the other planes
    unaffected; both sort paths"""
    rng = np.random.default_rng(3)
    ny, nx = 37, 130
    dA = rng.random((ny, nx)) + 0.1
    tg = np.linspace(0.0, dA.sum(), 7)
    for dt in (np.float64, np.float32):
        q = rng.standard_normal((3, ny, nx)).astype(dt)
        q[1] = np.nan
        r = ctx.sort_profile(q, dA=dA, targets=tg, want_sorted=True, want_acum=True)
        assert list(r['nvalid']) == [ny * nx, 0, ny * nx]
        for s in range(3):
            Q, xs, acum = O.sorted_profile(q[s], dA, tg)
            assert np.array_equal(r['Q'][s], Q.astype(np.float64), equal_nan=True)
            n = int(r['nvalid'][s])
            assert np.array_equal(r['q_sorted'][s][:n], xs.astype(np.float64))


@pytest.mark.parametrize('dt', [np.float64, np.float32])
def test_sort_tile_choice_both_sides_of_the_switch(ctx, dt):
    """xc_sort.hip small_tiles(): stacks with at most 400 full-size tiles run the radix passes on half tiles (2048 pairs per
    block), larger ones on 4096-pair tiles.  One plane shape with a ragged last tile in BOTH tilings, 44 planes (396 tiles:
    half tiles) and 45 planes (405: full tiles): every plane's order, payloads and count against the oracle's stable argsort."""
    rng = np.random.default_rng(77)
    ny, nx = 193, 170                                           # 32810 cells = 8 full tiles + 42 cells; 16 half tiles + 42
    for S in (44, 45):
        q = rng.standard_normal((S, ny, nx)).astype(dt)
        q[rng.random(q.shape) < 0.01] = np.nan
        q[2, 5:9] = 0.25                                         # ties: the payload order is the stable one
        dA = rng.random((ny, nx)) + 0.5
        r = ctx.sort_profile(q, dA=dA, want_sorted=True, want_acum=True)
        assert ctx.last_sort_path() == (1 if dt == np.float64 else 0)
        for s in range(S):
            _, xs, ac = O.sorted_profile(q[s], dA, [0.0], None)
            n = len(xs)
            assert int(r['nvalid'][s]) == n
            assert np.array_equal(r['q_sorted'][s][:n], xs.astype(np.float64)), (S, s)
            assert rel(r['acum'][s][:n], ac) < 1e-12, (S, s)
