"""-m gpu: the single-read Keff kernel (xcontour_amd/csrc/xc_keff1.hip) -- xc_keff_dev calls of ONE or TWO slabs, the reference's own
call pattern (one (time, level) plane per call: tests/LWA.py:40-43; core.py:224-225 min / max, then 1307 the histogram, per object):
min/max, levels, histogram, CDF and the Keff epilogue in one launch with the slab held in registers between them.  Checked against the
oracle and, option by option, against the min/max + histogram + finalize chain it replaces (levels and counts bit for bit, sums to
1e-13: both add float64 atomically in arrival order)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import xcontour_oracle as O
from test_gpu_parity import rel, RTOL, TIGHT, LMIN_FLOOR
from gpu_common import NINE, ROOT, check_nine, _clean_env

pytestmark = pytest.mark.gpu


def _plan_pair(ctx, nslab, ny, nx, N, dt, cdt=None, **kw):
    from xcontour_amd.pipeline import KeffPlan
    cdt = dt if cdt is None else cdt
    return (KeffPlan(ctx, nslab, ny, nx, N, dt, cdt, single_read='force' if nslab > 1 else True, **kw),
            KeffPlan(ctx, nslab, ny, nx, N, dt, cdt, single_read=False, **kw))


def _close(x, y, tol=1e-13):
    """sums of the two paths: both add float64 atomically in arrival order, so they agree to rounding -- measured against the LARGEST
    value of the vector (with lt=False the results are cdf[-1] - cdf: in the tail the two big numbers cancel and an element-wise
    relative bar would measure the cancellation, not the kernels)"""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    if not np.array_equal(np.isnan(x), np.isnan(y)):
        return False
    m = np.isfinite(y)
    if not np.array_equal(x[~m & ~np.isnan(y)], y[~m & ~np.isnan(y)]):
        return False
    if not m.any():
        return True
    scale = np.abs(y[m]).max()
    return bool(np.abs(x[m] - y[m]).max() <= tol * max(scale, 1e-300))


def _same(a, b, counts=True, tol=1e-13):
    assert np.array_equal(a['status'], b['status'])
    assert np.array_equal(a['ctr'], b['ctr'], equal_nan=True)
    if counts:
        assert np.array_equal(a['counts'], b['counts'])
    for k in ('area', 'intgrdS'):
        assert _close(a[k], b[k], tol), k
    for k in NINE[3:]:
        x, y = a[k], b[k]
        assert np.array_equal(np.isnan(x), np.isnan(y)), k


@pytest.mark.parametrize('dt', [np.float64, np.float32])
def test_cfg2_slab_single_read_against_the_oracle(ctx, dt):
    """north_star's literal unit: ONE 3600 x 1801 slab, 201 contours, 2-D float64 dA, through the one launch -- all nine vectors and the
    counts against the oracle; then the same call twice more (the kernel works in alternating sets of records that its successor clears)"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    ny, nx, N = 1801, 3600, 201
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.1
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
    p = KeffPlan(ctx, 1, ny, nx, N, dt, dt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True, preY=lat[::50])
    p.synth(lat, lon, 4242, 0)
    q = p.download_q()[0]
    r = O.keff_pipeline(q, dA, lat, N, lon=lon, increase=True, lt=True, dtype=dt, preLats=lat[::50])
    first = None
    for _ in range(3):
        p.run()
        assert ctx.last_keff_path() == 1
        got = p.fetch()
        assert p.replays == 0 and not got['status'].any()
        check_nine(got, 0, r, with_eq=True)
        assert int(got['counts'][0].sum()) == ny * nx - (dt == np.float32)     # (float32 levels: max + 1e-8 == max, the max cell falls on the open last edge)
        if first is None:
            first = got
        else:
            assert np.array_equal(got['counts'], first['counts']) and np.array_equal(got['ctr'], first['ctr'])
            assert _close(got['area'], first['area'])
    p.free()


CASES = [
    # ny, nx, N, dt, cdt, kwargs of the plan, what is done to the tracer
    (721, 1440, 201, np.float64, None, {}, None),
    (721, 1440, 121, np.float32, None, {}, None),
    (700, 1000, 64, np.float64, None, dict(increase=False), None),                 # decreasing levels; nx not a multiple of 124
    (513, 770, 33, np.float64, None, dict(lt=False), 'nan'),                       # NaN cells are dropped
    (400, 512, 50, np.float32, np.float64, dict(right_edge='numpy'), None),        # closed last bin: not the FAST instantiation
    (400, 512, 50, np.float64, None, dict(periodic_x=False), None),                # walls
    (1024, 64, 21, np.float64, None, dict(periodic_x=False, increase=False, lt=False), 'nan'),   # one narrow strip
    (300, 256, 700, np.float64, None, {}, None),                                   # many contours: fewer LDS copies
    (256, 512, 2, np.float64, None, {}, None),                                     # the smallest N
    (721, 1440, 201, np.float64, None, dict(counts=False), None),
    (361, 720, 101, np.float64, None, dict(dA_mode='row'), None),                  # one weight per row
    (361, 720, 101, np.float64, None, dict(dA_mode='none'), None),                 # no weights at all
    (361, 720, 101, np.float64, None, dict(dA_mode='nan'), None),                  # dA with NaN cells: fillna(0), core.py:449
    (361, 720, 101, np.float32, None, dict(dA_mode='slab'), None),                 # per-slab weights
]


@pytest.mark.parametrize('nslab', [1, 2])
@pytest.mark.parametrize('case', range(len(CASES)))
def test_single_read_equals_the_chain(ctx, case, nslab):
    """every option of the Keff call on both paths: levels, counts and status bit for bit, sums to 1e-13, the same NaN pattern in the
    derived vectors; one and two slabs per call; slab 0 also against the oracle"""
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    ny, nx, N, dt, cdt, kw, mod = CASES[case]
    kw = dict(kw)
    mode = kw.pop('dA_mode', 'plane')
    lat = np.linspace(-89.5, 89.5, ny); lon = np.arange(nx) * (360.0 / nx)
    dA2 = cell_area(lat, lon)
    rng = np.random.default_rng(1000 + case)
    if mode == 'row':
        dA = np.ascontiguousarray(dA2[:, 0])
    elif mode == 'none':
        dA = None
    elif mode == 'nan':
        dA = dA2.copy(); dA[rng.integers(0, ny, 50), rng.integers(0, nx, 50)] = np.nan
    elif mode == 'slab':
        dA = np.stack([dA2 * (1.0 + 0.1 * s) for s in range(nslab)])
    else:
        dA = dA2
    dA_tbl = dA2 if dA is None or mode in ('nan', 'slab') else (np.repeat(dA[:, None], nx, 1) if mode == 'row' else dA)
    inc = kw.get('increase', True)
    tbl = table_from_rowsums(ctx.rowsum(None, np.ascontiguousarray(dA_tbl), ny, nx), kw.get('lt', True) == inc, last_row_included(lat))
    counts = kw.get('counts', True)
    a, b = _plan_pair(ctx, nslab, ny, nx, N, dt, cdt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, **kw)
    a.synth(lat, lon, 99 + case, 0)
    q = a.download_q()
    if mod == 'nan':
        q[:, rng.integers(0, ny, 300), rng.integers(0, nx, 300)] = np.nan
        q[:, 5, :] = np.nan
    a.set_q(q); b.set_q(q)
    a.run(); pa = ctx.last_keff_path(); ra = a.fetch()
    b.run(); pb = ctx.last_keff_path(); rb = b.fetch()
    assert (pa, pb) == (1, 0) and a.replays == 0
    _same(ra, rb, counts=counts)
    if mode in ('plane', 'row') and kw.get('periodic_x', True):
        r = O.keff_pipeline(q[0], dA_tbl, lat, N, lon=lon, increase=inc, lt=kw.get('lt', True), dtype=np.dtype(cdt or dt).type,
                            right_edge=kw.get('right_edge', 'xhistogram'))
        if counts:
            assert np.array_equal(ra['counts'][0].astype(np.int64), r['counts'])
        assert np.array_equal(ra['ctr'][0], r['ctr'].astype(np.float64))
        assert _close(ra['area'][0], r['area'], TIGHT) and _close(ra['intgrdS'][0], r['intgrdS'], TIGHT)
    a.free(); b.free()


def test_levels_that_are_not_equally_spaced_take_the_general_search(ctx):
    """float32 contours of a range of a few hundred ulps (potential temperature near 300 K on one level) are NOT equally spaced to a
    quarter of a bin: the kernel then bins with the general bracket search of the two-pass kernel (rows re-read from the caches) --
    the same counts, bit for bit; also an all-NaN slab and a slab with an infinite cell"""
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    ny, nx, N = 361, 720, 201
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.5
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
    rng = np.random.default_rng(5)
    base = (300.0 + 0.0104 * rng.random((ny, nx))).astype(np.float32)   # 201 levels over ~340 ulps: steps of 1 or 2 ulps, all distinct
    e, _ = O.hist_edges(O.cal_contours(base, N, True, np.float32))
    e = e.astype(np.float64); h = (e[-1] - e[0]) / N
    assert (np.abs(e - (e[0] + np.arange(N + 1) * h)) > 0.25 * h).any()         # NOT equally spaced to a quarter of a bin
    for name, q in (('tiny', base), ('allnan', np.full((ny, nx), np.nan, np.float32)),
                    ('inf', np.where(rng.random((ny, nx)) < 1e-5, np.float32(np.inf), base))):
        a, b = _plan_pair(ctx, 1, ny, nx, N, np.float32, None, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat)
        a.set_q(q[None]); b.set_q(q[None])
        a.run(); assert ctx.last_keff_path() == 1
        b.run(); assert ctx.last_keff_path() == 0
        ra, rb = a.fetch(check=False), b.fetch(check=False)
        assert np.array_equal(ra['status'], rb['status']), name
        assert np.array_equal(ra['ctr'], rb['ctr'], equal_nan=True), name
        assert np.array_equal(ra['counts'], rb['counts']), name
        for k in ('area', 'intgrdS'):
            x, y = ra[k], rb[k]
            assert _close(x, y), (name, k)
        if name == 'tiny':
            r = O.keff_pipeline(q, dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float32)
            assert np.array_equal(ra['counts'][0].astype(np.int64), r['counts'])
        a.free(); b.free()


def test_shapes_the_kernel_does_not_take_fall_to_the_chain(ctx):
    """an odd nx (no 16-byte pairs), three slabs, two slabs unless forced, a tiny plane, deterministic sums, a supplied gradient:
    the chain runs, as before"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    for nslab, ny, nx, kw in ((1, 400, 511, {}), (3, 400, 512, {}), (2, 400, 512, {}), (1, 90, 180, {}), (1, 400, 512, dict(deterministic=True)),
                              (1, 400, 512, dict(grdS_dtype=np.float64)), (3, 400, 512, dict(single_read='force'))):
        lat = np.linspace(-89, 89, ny); lon = np.arange(nx) * (360.0 / nx)
        dA = cell_area(lat, lon)
        tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
        p = KeffPlan(ctx, nslab, ny, nx, 41, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, **kw)
        p.synth(lat, lon, 3, 0)
        if 'grdS_dtype' in kw:
            p.set_grdS(np.stack([O.grad2_sphere(x, lat, lon) for x in p.download_q()]))
        p.run()
        assert ctx.last_keff_path() == 0
        r = p.fetch()
        assert int(r['counts'].sum()) == nslab * ny * nx
        p.free()


def test_timeout_gives_status_2_and_fetch_repeats_the_call_on_the_chain():
    """every wait of the kernel on another workgroup is bounded by the wall clock (XC_KEFF_SINGLE_TIMEOUT_US; here 1 us, which no grid
    can meet): the call comes back -- it never hangs -- with status 2 for every slab and the result vectors untouched; KeffPlan.fetch
    sees the status and repeats the launch set on the min/max + histogram + finalize chain.  In a child process: the knob is read once,
    when the context is created."""
    code = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'oracle'))
os.environ['XC_KEFF_SINGLE_TIMEOUT_US'] = '1'
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
import xcontour_oracle as O
ctx = nat.Context(0)
ny, nx, N = 721, 1440, 101
lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.25
dA = cell_area(lat, lon)
tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
p = KeffPlan(ctx, 2, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, single_read='force')
p.synth(lat, lon, 11, 0)
q = p.download_q()
ctx._check(ctx.lib.xc_memset(ctx.handle, p.out_ptr, 0xff, p.slot_bytes))      # poison: a skipped slab must stay untouched
p.run()
assert ctx.last_keff_path() == 1
raw = np.empty(p.slot_bytes, np.uint8)
ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, raw.ctypes.data, p.out_ptr, p.slot_bytes))
out = p.unpack(raw)
assert (out['status'] == 2).all(), out['status']
assert (raw[:p.head_bytes] == 0xff).all()                                     # nothing was written to the nine vectors
got = p.fetch()                                                               # ... sees status 2 and repeats the call on the chain
assert p.replays == 1 and ctx.last_keff_path() == 0 and not got['status'].any()
for s in range(2):
    r = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float64)
    assert np.array_equal(got['counts'][s].astype(np.int64), r['counts'])
    assert np.array_equal(got['ctr'][s], r['ctr'])
    assert np.allclose(got['area'][s], r['area'], rtol=1e-11, atol=0)
# and the NEXT launch of the kernel works in records its aborted predecessor left behind: with a sane bound it must succeed
p.free(); ctx.close()
os.environ['XC_KEFF_SINGLE_TIMEOUT_US'] = '200000'
ctx = nat.Context(0)
p = KeffPlan(ctx, 2, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, single_read='force')
p.set_q(q)
for _ in range(3):
    p.run(); assert ctx.last_keff_path() == 1
    got = p.fetch()
    assert p.replays == 0 and not got['status'].any()
    assert np.array_equal(got['counts'][1].astype(np.int64), r['counts'])
print('OK')
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, '-c', code], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True, timeout=600)
    assert r.returncode == 0 and 'OK' in r.stdout, r.stdout[-3000:]


def test_facade_keff_takes_the_single_read_kernel(ctx, baro):
    """Contour2D.keff on one plane (the reference's call pattern) goes through the one launch and equals the oracle"""
    import xcontour_amd as xa
    ny, nx, N = 361, 720, 121
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.5
    rng = np.random.default_rng(2)
    q = (np.sin(np.deg2rad(lat))[:, None] + 0.05 * rng.standard_normal((ny, nx))).astype(np.float32)
    c = {'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(q, ('latitude', 'longitude'), c, 'q')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), c, 'rA')
    mask = xa.DataArray(np.ones_like(q), ('latitude', 'longitude'), c, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    table = cm.cal_area_eqCoord_table_hist(mask)
    ds = cm.keff(N, table, lat=lat, lon=lon)
    assert cm.ctx.last_keff_path() == 1
    r = O.keff_pipeline(q, dA.values, lat, N, lon=lon, increase=True, lt=True, dtype=np.float32)
    assert np.array_equal(ds['ctr'].values, r['ctr'].astype(np.float64))
    for k in ('area', 'intgrdS', 'latEq'):
        assert np.allclose(ds[k].values, r[k], rtol=1e-6, atol=0, equal_nan=True), k


def test_single_read_fuzz_against_the_chain(ctx):
    """a seeded differential fuzz: 120 random (shape, N, dtype, direction, last-bin rule, periodicity, weights, NaN pattern) cases, one slab
    each, through the single-read kernel and through the chain -- status, levels and counts bit for bit, the sums to 1e-13, the same NaN
    pattern in every derived vector.  Shapes include strips that end one or two columns into the last strip, chunks of 4 rows (tall
    narrow planes), planes just above the 65 536-cell threshold and ragged last chunks."""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    rng = np.random.default_rng(20261004)
    taken = 0
    for case in range(120):
        nx = int(rng.choice([126, 128, 250, 372, 374, 496, 500, 620, 746, 994, 1240, 1488, 2 * int(rng.integers(65, 900))]))
        ny = int(rng.integers(max(2, 65536 // nx + 1), max(65536 // nx + 40, 900)))
        if ny * nx < 65536:
            ny = 65536 // nx + 1
        N = int(rng.choice([2, 3, 17, 64, 121, 201, 333]))
        dt = [np.float64, np.float32][int(rng.integers(2))]
        cdt = dt if rng.random() < 0.7 else np.float64
        kw = dict(increase=bool(rng.integers(2)), lt=bool(rng.integers(2)), periodic_x=bool(rng.random() < 0.7),
                  right_edge='xhistogram' if rng.random() < 0.7 else 'numpy', counts=bool(rng.random() < 0.8))
        lat = np.linspace(-88, 88, ny); lon = np.arange(nx) * (360.0 / nx)
        dA = cell_area(lat, lon)
        mode = int(rng.integers(4))
        if mode == 1:
            dA = np.ascontiguousarray(dA[:, 0])
        elif mode == 2:
            dA = dA.copy(); dA[rng.integers(0, ny, 20), rng.integers(0, nx, 20)] = np.nan
        elif mode == 3:
            dA = None
        tbl = table_from_rowsums(ctx.rowsum(None, cell_area(lat, lon), ny, nx), True, last_row_included(lat))
        a = KeffPlan(ctx, 1, ny, nx, N, dt, cdt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, single_read=True, **kw)
        b = KeffPlan(ctx, 1, ny, nx, N, dt, cdt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, single_read=False, **kw)
        a.synth(lat, lon, 1000 + case, int(rng.integers(3)))
        q = a.download_q()
        if rng.random() < 0.5:
            q[0, rng.integers(0, ny, 50), rng.integers(0, nx, 50)] = np.nan
        if rng.random() < 0.2:
            q[0, int(rng.integers(ny))] = np.nan                      # a whole row of land
        if rng.random() < 0.1:
            q[0] = np.round(q[0] * 4) / 4                             # many cells exactly on levels and ties in the extrema
        a.set_q(q); b.set_q(q)
        a.run(); pa = ctx.last_keff_path(); ra = a.fetch(check=False)
        b.run(); pb = ctx.last_keff_path(); rb = b.fetch(check=False)
        assert pb == 0 and a.replays == 0
        taken += pa
        tag = (case, ny, nx, N, np.dtype(dt).name, kw, mode)
        assert np.array_equal(ra['status'], rb['status']), tag
        assert np.array_equal(ra['ctr'], rb['ctr'], equal_nan=True), tag
        if kw['counts']:
            assert np.array_equal(ra['counts'], rb['counts']), tag
        for k in ('area', 'intgrdS'):
            x, y = ra[k], rb[k]
            assert _close(x, y), (tag, k)
        for k in NINE[3:]:
            assert np.array_equal(np.isnan(ra[k]), np.isnan(rb[k])), (tag, k)
        a.free(); b.free()
    assert taken >= 110                                               # (nearly every case fits the register tiles)


def test_two_processes_on_one_gpu_never_hang_and_never_return_garbage():
    """the failure the bounded waits exist for, for real: TWO processes launch the single-read kernel on the SAME GPU at the same time --
    neither grid can be co-resident while the other holds compute units.  Whatever the hardware scheduler does with them (serialise
    them, or interleave them until a wait expires), every call of both processes must come back (status 0, or status 2 and the replay on
    the chain inside KeffPlan.fetch) with the chain's own levels and counts; the processes report how many launch sets they repeated."""
    code = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
os.environ['XC_KEFF_SINGLE_TIMEOUT_US'] = '20000'
from xcontour_amd import _native as nat
from xcontour_amd.pipeline import KeffPlan
from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
ctx = nat.Context(0)
ny, nx, N = 1801, 3600, 201
lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.1
dA = cell_area(lat, lon)
tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
kw = dict(dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat)
ref = KeffPlan(ctx, 1, ny, nx, N, np.float64, np.float64, single_read=False, **kw)
ref.synth(lat, lon, 77 + int(sys.argv[1]), 0)
ref.run(); want = ref.fetch()
p = KeffPlan(ctx, 1, ny, nx, N, np.float64, np.float64, alloc_q=False, **kw)
p.set_q_device(ref._q_ptr)
while time.time() < float(sys.argv[2]):          # both processes start their loops at the same wall-clock instant
    pass
took = 0
for it in range(150):
    p.run(); took += ctx.last_keff_path()
    got = p.fetch()
    assert not got['status'].any(), (it, got['status'])
    assert np.array_equal(got['ctr'], want['ctr']) and np.array_equal(got['counts'], want['counts']), it
    assert np.allclose(got['area'], want['area'], rtol=1e-12, atol=0), it
print('OK single-read launches %%d replays %%d' %% (took, p.replays))
''' % ROOT
    import time
    t0 = time.time() + 20.0
    procs = [subprocess.Popen([sys.executable, '-c', code, str(r), repr(t0)], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              universal_newlines=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and 'OK single-read launches 150' in o, o[-3000:]
    print(' | '.join(o.strip().splitlines()[-1] for o in outs))
