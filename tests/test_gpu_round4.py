"""-m gpu, round 4: bench.py launching its own ranks (torch-free), the slab-major result layout, the library's RCCL gather with
two ranks (when two devices are visible)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import xcontour_oracle as O
from test_gpu_parity import rel, RTOL, TIGHT

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'XC_DIST_TOKEN')}


# ---------------------------------------------------------------- bench.py --gpus 2 on the one GPU
def test_bench_launches_two_ranks_by_itself_on_one_gpu():
    """the driver's own command form, `python3 bench.py --gpus 2 ...` with no launcher around it: bench.py starts two rank
    processes (both on this box's one GPU, the one gather staged through the host), the line says n_gpus 2, the gathered blocks
    arrived in rank order (asserted inside) and the cfg4 block's checks hold -- rank 1's first slab recomputed on rank 0, two
    slabs against the oracle; no torch in any of the three processes"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1',
                        '--batch', '4', '--cfg4-slabs', '64', '--cfg4-reps', '1', '--cpu-slabs', '1', '--cpu-workers', '2'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=_clean_env(), timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
    assert d['config']['launcher'].startswith('bench.py itself') and 'no torch' in d['config']['host_code']
    assert d['config']['slabs_per_step_per_gpu'] == 4
    c4 = d['cfg4_strong']
    assert c4['n_gpus'] == 2 and c4['slabs'] == 64 and c4['slabs_per_gpu'] == 32 and c4['scaling'] == 'strong'
    assert c4['checks']['rank0_block_bit_identical'] and c4['checks']['first_slab_of_each_rank_recomputed'] == [0, 32]
    assert c4['checks']['oracle_checked_slabs'] == 2 and c4['checks']['finite_nkeff_fraction'] > 0.5
    b = c4['budget']
    assert len(b['sweep_ms_by_rank']) == 2 and len(b['gather_ms_by_rank']) == 2 and b['pack_ms'] == 0.0
    assert all(x > 0 for x in b['sweep_ms_by_rank'])


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    """`--gpus 8` inside a 1-rank environment used to run one rank and print n_gpus 1 (round-3 review): now an error"""
    env = _clean_env()
    env.update({'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0', '--no-cpu'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr and r.stdout.strip() == ''


# ---------------------------------------------------------------- slab-major result layout (xc_keff_desc.out_stride)
@pytest.mark.parametrize('dt,cd,inc', [(np.float64, np.float64, True), (np.float32, np.float32, False)])
def test_slab_major_layout_is_the_dense_result_rearranged(ctx, dt, cd, inc):
    """KeffPlan(slab_major=True): the head of the slot is ONE [slab][9][N] block -- bit for bit the nine dense vectors, launch
    sets landing at their slab offset (the cfg4 sweep), counts / status / interp unchanged; and against the oracle"""
    from xcontour_amd.pipeline import KeffPlan, OUT_NAMES
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N, S = 181, 360, 41, 7
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 1.0
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(dA.sum(1), inc)                      # ylt = lt iff increase == coordinate increasing (core.py:180-188)
    pre = np.linspace(-80, 80, 33)
    kw = dict(dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=inc, lt=True, preY=pre, deterministic=True)
    dense = KeffPlan(ctx, S, ny, nx, N, dt, cd, **kw)
    dense.synth(lat, lon, 5, 0)
    dense.run(0)
    ref = dense.fetch()
    sm = KeffPlan(ctx, 3, ny, nx, N, dt, cd, alloc_q=False, out_slabs=S, slab_major=True, **kw)
    esz = ny * nx * np.dtype(dt).itemsize
    ctx._check(ctx.lib.xc_memset(ctx.handle, sm.out_ptr, 0, sm.slot_bytes))
    for c0 in range(0, S, 3):                                        # ragged launch sets 3 + 3 + 1 into one block
        m = min(3, S - c0)
        sm.set_q_device(dense._q_ptr + c0 * esz)
        sm._point(0, 0, m, out_s0=c0)
        sm.desc.q_next = None
        ctx._check(ctx.lib.xc_keff_dev(ctx.handle, __import__('ctypes').byref(sm.desc)))
    got = sm.fetch()
    assert sm.head_bytes == S * 9 * N * 8
    raw = np.empty(sm.head_bytes, dtype=np.uint8)
    ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, raw.ctypes.data, sm.out_ptr, sm.head_bytes))
    blk = raw.view(np.float64).reshape(S, 9, N)
    for i, k in enumerate(OUT_NAMES):
        assert np.array_equal(bits(got[k]), bits(ref[k])), k
        assert np.array_equal(bits(blk[:, i, :]), bits(ref[k])), k
    assert np.array_equal(got['counts'], ref['counts']) and np.array_equal(got['status'], ref['status'])
    for k in ref:
        if k.endswith('_eq'):
            assert np.array_equal(bits(got[k]), bits(ref[k])), k
    q = dense.download_q()
    r = O.keff_pipeline(q[4], dA, lat, N, lon=lon, increase=inc, lt=True, dtype=cd)
    assert np.array_equal(blk[4, 0], r['ctr'].astype(np.float64)) and np.array_equal(got['counts'][4].astype(np.int64), r['counts'])
    assert rel(blk[4, 1], r['area']) < TIGHT and rel(blk[4, 3], r['latEq']) < RTOL
    dense.free(); sm.free()


def test_out_stride_is_validated(ctx):
    from xcontour_amd import _native as nat
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N = 19, 36, 11
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 10.0
    dA = cell_area(lat, lon)
    p = KeffPlan(ctx, 1, ny, nx, N, dA=dA, lat=lat, lon=lon, tbl=table_from_rowsums(dA.sum(1), True), tbl_coord=lat)
    p.synth(lat, lon, 1, 0)
    p.desc.out_stride = N - 1
    with pytest.raises(nat.XContourHipError):
        p.run()
    p.free()


# ---------------------------------------------------------------- the library's own RCCL gather with two ranks
_RCCL_RANK = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from xcontour_amd import _native as nat
from xcontour_amd.distributed import SocketGroup
assert 'torch' not in sys.modules
g = SocketGroup()
ctx = nat.Context(g.rank)                          # one GPU per rank
g.init_device(ctx)
n = 1 << 20
send = ctx.to_device(np.full(n, g.rank + 1, dtype=np.float64))
recv = ctx.alloc(g.world * n * 8)
for _ in range(2):
    g.allgather_device(send.ptr, recv.ptr, n * 8)
ctx.sync()
out = recv.download((g.world, n), np.float64)
assert all((out[r] == r + 1).all() for r in range(g.world)), out[:, :4]
g.barrier(); ctx.comm_finalize(); ctx.close(); g.close()
print('rank %%d ok' %% g.rank)
'''


def test_native_rccl_allgather_two_ranks_two_gpus(tmp_path):
    """xc_comm_* with world 2: ncclGetUniqueId on rank 0, the id through the sockets, ncclCommInitRank, ncclAllGather on each
    context's stream.  Needs two visible devices (RCCL refuses two ranks on one GPU): skipped on the one-GPU test box."""
    import ctypes as C
    import importlib.util
    from xcontour_amd import _native as nat
    n = C.c_int(0)
    nat.load().xc_device_count(C.byref(n))
    if n.value < 2:
        pytest.skip('one visible device: RCCL needs one GPU per rank')
    prog = tmp_path / 'rccl_rank.py'
    prog.write_text(_RCCL_RANK % {'root': ROOT})
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    saved = dict(os.environ)
    try:
        os.environ.clear(); os.environ.update(_clean_env())
        assert m.launch_ranks(2, [], program=str(prog)) == 0
    finally:
        os.environ.clear(); os.environ.update(saved)


def test_rccl_unavailable_lands_on_the_ipc_carrier_loudly():
    """two ranks on ONE GPU with the default backend: ncclCommInitRank fails on every rank ('duplicate GPU'), every rank
    learns it (SocketGroup.init_device), the job moves on to the HIP IPC carrier -- a device carrier, not the TCP one of round 4 -- and
    the line says so, with the trials it made"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2',
                        '--no-cpu', '--no-cfg4'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=_clean_env(), timeout=600, cwd=ROOT)
    import ctypes as C
    from xcontour_amd import _native as nat
    n = C.c_int(0)
    nat.load().xc_device_count(C.byref(n))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 2 and '[preflight] rank 1 -> device' in r.stderr and '[preflight] carrier:' in r.stderr
    tr = d['config']['carrier_trials']
    if n.value < 2:
        assert 'rccl unavailable' in d['config']['collective_note'] and 'carrier ipc' in d['config']['parallelism']
        assert 'error' in tr['rccl'] and tr['ipc']['ms_1MB'] > 0 and tr['ipc']['ms_32MB'] > 0 and 'host' not in tr
    else:
        assert d['config']['collective_note'] is None and 'carrier rccl' in d['config']['parallelism'] and tr['rccl']['ms_32MB'] > 0


@pytest.mark.parametrize('world', [2, 4])
def test_bench_ipc_ranks_on_one_gpu_equal_the_one_rank_job(world, tmp_path):
    """`python3 bench.py --gpus N --backend ipc` (the driver's command form) with N = 2 and 4 ranks sharing this box's GPU: the HIP IPC
    carrier gathers every rank's block to rank 0 -- a piece per launch set on the comm stream -- and the gathered (S, 9, N) cfg4
    result equals the ONE-rank job's bit for bit (deterministic sums: a rank count must not change a bit); the gather's exposed
    time stays in the milliseconds (round 4, through TCP: 400-510 ms); N > 1 lines carry roofline and cpu_baseline"""
    outs = {}
    for w in (1, world):
        f = str(tmp_path / ('cfg4_%d.npy' % w))
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(w), '--steps', '2', '--warmup', '1', '--batch', '2', '--deterministic',
               '--cfg4-slabs', '100', '--cfg4-chunk', '8', '--cfg4-reps', '1', '--dump-cfg4', f, '--no-extras', '--cpu-slabs', '1', '--cpu-workers', '2']
        if w > 1:
            cmd += ['--backend', 'ipc']
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=_clean_env(), timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[w] = (json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0]), np.load(f))
    d, full = outs[world]
    assert d['n_gpus'] == world and 'carrier ipc' in d['config']['parallelism'] and d['config']['collective_note'] is None
    assert d['roofline']['frac'] > 0 and d['cpu_baseline']['value'] > 0 and d['cpu_baseline']['parity_checked_slabs'] >= 1
    c4 = d['cfg4_strong']
    per = -(-100 // world)
    assert c4['n_gpus'] == world and c4['slabs_per_gpu'] == per and c4['pieces_per_job'] == -(-per // 8) and 'HIP IPC' in c4['gather']
    assert c4['checks']['first_slab_of_each_rank_recomputed'] == [r * per for r in range(world) if r * per < 100]
    assert c4['budget']['gather_ms_max'] < 5.0, c4['budget']
    assert full.shape == outs[1][1].shape == (100, 9, 201)
    assert np.array_equal(bits(full), bits(outs[1][1]))                   # 100 = 4 x 25: ragged launch sets, and at N = 4 pieces of 8, 8, 8, 1


# ---------------------------------------------------------------- K3 E32: float32 tracer + float32 levels, float32 bin search
def test_e32_bin_search_ties_infinities_and_collapsed_levels(ctx):
    """the float32 variant of the histogram pass (raw float32 rows, float32 nearest-edge guess + ONE exact float32 compare):
    cells sitting exactly ON contour levels, on the dummy edge and on the bumped last edge, +-inf, NaN, denormal-scale fields
    whose float32 levels are not equally spaced (the variant must fall back to the exact search) -- counts bit for bit"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    rng = np.random.default_rng(44)
    ny, nx, N, S = 64, 1280, 201, 6
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * (360.0 / nx)
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
    q = (np.sin(np.deg2rad(lat))[None, :, None] + 0.05 * rng.standard_normal((S, ny, nx))).astype(np.float32)
    q[2] = (1e-38 * rng.random((ny, nx))).astype(np.float32)                       # denormal range
    q[3] = (300.0 + 0.05 * rng.standard_normal((ny, nx))).astype(np.float32)       # levels ~16 ulps apart: uneven, still "equally spaced to a quarter bin"
    q[4] = (300.0 + 1.2e-3 * rng.standard_normal((ny, nx))).astype(np.float32)     # levels 1-2 ulps apart: NOT equally spaced -> the exact search
    q[5] = np.float32(7.25)                                                        # constant field: all levels coincide (status 1)
    for s in (0, 1):
        ctr = O.cal_contours(q[s], N, True, np.float32)
        inner = (q[s] > q[s].min()) & (q[s] < q[s].max())
        idx = np.flatnonzero(inner.ravel())[:4000]
        q[s].ravel()[idx] = ctr[rng.integers(0, N, idx.size)]                      # exactly on levels (min / max cells untouched)
        assert np.array_equal(O.cal_contours(q[s], N, True, np.float32), ctr)
    q[1, 5, 7:11] = [np.inf, -np.inf, np.nan, np.inf]
    for inc in (True, False):
        plan = KeffPlan(ctx, S, ny, nx, N, np.float32, np.float32, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=inc, lt=True)
        plan.set_q(q)
        plan.run(0)
        b = plan.fetch(check=False)
        assert list(b['status']) == [0, 0, 0, 0, 0, 1] or b['status'][5] == 1
        for s in range(5):
            qs = q[s]
            if s == 1:
                continue                                                           # +-inf: the extremes are infinite, levels NaN -- compared below
            ctr = O.cal_contours(qs, N, inc, np.float32)
            assert np.array_equal(b['ctr'][s], ctr.astype(np.float64)), s
            if len(np.unique(ctr)) == N:
                _, cnt = O.cal_integral_within_contours_hist(qs, ctr, dA, None, True, return_counts=True)
                assert np.array_equal(b['counts'][s].astype(np.int64), cnt), (s, inc)
        plan.free()


# ---------------------------------------------------------------- resident mirrors keyed on the source's identity
def test_resident_memo_follows_reassignment_and_is_private(ctx, baro):
    """Contour2D(resident=True): `c.tracer = other` / `c.dA = other` must not be served the OLD mirror (round-3 advisor), and
    what is registered is a private copy: a second, non-resident object that hands the SAME ndarray to the library after an
    in-place change gets the new values, not the first object's mirror"""
    import xcontour_amd as xa
    q0, lat, lon = baro
    c2 = {'latitude': lat, 'longitude': lon}
    q = np.ascontiguousarray(q0.astype(np.float64))
    tr = xa.DataArray(q, ('latitude', 'longitude'), c2, 'absolute_vorticity')
    dAv = O.cell_area(lat, lon)
    dA = xa.DataArray(dAv, ('latitude', 'longitude'), c2, 'rA')
    kw = dict(dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True, deterministic=True)

    def area(cm):
        return cm.cal_integral_within_contours_hist(cm.cal_contours(41)).values

    res = xa.Contour2D(tr, dA, resident=True, **kw)
    a0 = area(res)
    # 1. reassign the tracer and the weights
    q2 = np.ascontiguousarray(np.roll(q, 17, axis=0) * 1.5)
    res.tracer = xa.DataArray(q2, ('latitude', 'longitude'), c2, 'absolute_vorticity')
    a1 = area(res)
    ref1 = area(xa.Contour2D(res.tracer, dA, **kw))
    assert np.array_equal(bits(a1), bits(ref1)) and not np.array_equal(bits(a1), bits(a0))
    res.dA = xa.DataArray(dAv * 2.0, ('latitude', 'longitude'), c2, 'rA')
    a2 = area(res)
    assert np.array_equal(bits(a2), bits(area(xa.Contour2D(res.tracer, res.dA, **kw))))
    assert np.array_equal(bits(a2), bits(2.0 * a1))
    # 2. the registered host memory is not the caller's array
    assert all(not np.shares_memory(arr, q2) for arr in res.ctx._resident.values())
    q2[:] = np.roll(q2, 5, axis=0)                                   # in place, no touch(): res keeps its (documented) old mirror ...
    other = xa.Contour2D(res.tracer, res.dA, **kw)                  # ... but a non-resident object must see the new values
    b = area(other)
    res.touch()
    assert np.array_equal(bits(b), bits(area(res))) and not np.array_equal(bits(b), bits(a2))
    res.close(); other.close()


# ---------------------------------------------------------------- xarray in, xarray out (a test double of xarray)
def test_facade_call_sequences_through_the_xarray_branch():
    """the Keff (tests/test_hist.py), contour-mean and LWA (tests/test_LWA.py) call sequences of the reference with xarray
    objects in: every result is an xarray object with the reference's dims and names ('AeqCTbl', 'd...dA', 'Leq2', 'nkeff',
    'LWA', 'LAPE', 'cm...'), values bit-identical to the same calls on the in-house DataArray, the fused pipeline against
    the oracle (subprocess: the double must be importable BEFORE the package is)"""
    env = dict(_clean_env(), PYTHONPATH=os.path.join(ROOT, 'tests', 'fake_xarray'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'xarray_branch_script.py'), 'gpu'], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok gpu'), r.stderr[-3000:]


# ---------------------------------------------------------------- lazy stacks stay lazy
class _Budget(object):
    """a lazy (time, level, lat, lon) source that refuses to hand out more than `limit` bytes at once"""

    def __init__(self, a, limit):
        self.a, self.shape, self.dtype, self.limit, self.peak, self.reads = a, a.shape, a.dtype, limit, 0, 0

    def __getitem__(self, k):
        r = self.a[k]
        self.peak = max(self.peak, r.nbytes)
        self.reads += 1
        if r.nbytes > self.limit:
            raise MemoryError('asked for %d bytes at once, budget %d' % (r.nbytes, self.limit))
        return r


def test_lazy_stack_goes_through_in_batches(baro, tmp_path):
    """the reference's histogram API is lazy (dask='allowed', core.py:242, 258): a lazy tracer stack -- here a source that
    RAISES when more than two slabs are requested at once, and a multi-record .nc opened with lazy=True -- is pulled through
    cal_contours / the histogram integrals / contour means / keff / LWA / crossing batch by batch under max_batch_bytes; results equal the eager run bit for bit"""
    import xcontour_amd as xa
    from xcontour_amd import ncio
    q0, lat, lon = baro
    T, Z = 3, 2
    rng = np.random.default_rng(8)
    q = np.stack([q0 * (1 + 0.05 * k) for k in range(T * Z)]).reshape(T, Z, *q0.shape).astype(np.float32)
    g = rng.random(q.shape).astype(np.float32)
    c4 = {'time': np.arange(T), 'level': np.arange(Z), 'latitude': lat, 'longitude': lon}
    c2 = {'latitude': lat, 'longitude': lon}
    d4 = ('time', 'level', 'latitude', 'longitude')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), c2, 'rA')
    mask = xa.DataArray(np.ones_like(q0), ('latitude', 'longitude'), c2, 'mask')
    kw = dict(dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True, deterministic=True)
    L = np.load(os.path.join(ROOT, 'tests', 'golden', 'baro_lwa_N121.npz'))

    def run(tr, grd):
        cm = xa.Contour2D(tr, dA, **kw)
        table = cm.cal_area_eqCoord_table_hist(mask)
        ctr = cm.cal_contours(41)
        area = cm.cal_integral_within_contours_hist(ctr)
        intS = cm.cal_integral_within_contours_hist(ctr, integrand=grd)
        strict = cm.cal_integral_within_contours(ctr)
        mean = cm.cal_contour_mean_hist(ctr, grd, grd)
        ds = cm.keff(41, table, lat=lat, lon=lon, max_batch_bytes=2 * q0.nbytes + 64)
        Q = xa.DataArray(L['Q'], ('latitude',), {'latitude': lat}, 'Q')
        lwa = cm.cal_local_wave_activity(tr, Q, metric=L['dy'])
        cross = cm.cal_contour_crossing(ctr, stride=2)
        cm.close()
        return [ctr.values, area.values, intS.values, strict.values, mean.values, ds['nkeff'].values, ds['area'].values, lwa.values, cross.values]

    eager = run(xa.DataArray(q, d4, c4, 'pv'), xa.DataArray(g, d4, c4, 'grdS'))
    from xcontour_amd import _native as nat
    ctx = nat.default_context(0)                                     # the facade's own context
    old = ctx.max_batch_bytes
    try:
        ctx.max_batch_bytes = 2 * q0.nbytes + 64                     # at most two slabs per batch, one when an integrand rides along
        src, gsrc = _Budget(q, 2 * q0.nbytes), _Budget(g, 2 * q0.nbytes)
        lazy = run(xa.DataArray(src, d4, c4, 'pv'), xa.DataArray(gsrc, d4, c4, 'grdS'))
        assert 0 < src.peak <= 2 * q0.nbytes and src.reads >= 3 * 6 and gsrc.peak <= 2 * q0.nbytes
        for a, b in zip(lazy, eager):
            assert np.array_equal(bits(a), bits(b))
        # the same from a file: a classic NetCDF stack opened lazily
        from scipy.io import netcdf_file
        path = str(tmp_path / 'stack.nc')
        with netcdf_file(path, 'w', version=2) as f:
            f.createDimension('time', None); f.createDimension('level', Z); f.createDimension('latitude', len(lat)); f.createDimension('longitude', len(lon))
            for n_, v_ in (('latitude', lat), ('longitude', lon)):
                w = f.createVariable(n_, 'f4', (n_,)); w[:] = v_
            w = f.createVariable('pv', 'f4', d4); w[:] = q
            w = f.createVariable('grdS', 'f4', d4); w[:] = g
        ds = ncio.open_dataset(path, lazy=True)
        assert isinstance(ds.pv.data, ncio.LazyVariable) and ds.pv.dims == d4
        filed = run(ds.pv, ds.grdS)
        for a, b in zip(filed, eager):
            assert np.array_equal(bits(a), bits(b))
        assert ds.pv.data.rows_read >= T and isinstance(ds.pv.data, ncio.LazyVariable)      # still lazy afterwards
    finally:
        ctx.max_batch_bytes = old


# ---------------------------------------------------------------- K7F: the O(ny log ny) interval kernel of large planes
def _lwa_case(rng, ny, nx, dt, increase, coord_up, with_nan):
    lat = np.linspace(-80, 80, ny) if coord_up else np.linspace(80, -80, ny)
    prof = np.sin(np.deg2rad(np.linspace(-80, 80, ny)))
    if not increase:
        prof = -prof
    q = (prof[:, None] + 0.3 * np.sin(np.linspace(0, 12, nx))[None, :] * np.cos(np.deg2rad(lat))[:, None]
         + 0.05 * rng.standard_normal((ny, nx))).astype(dt)
    Q = np.sort(q.astype(np.float64).mean(axis=1))
    if not increase:
        Q = Q[::-1].copy()
    # ties: some cells exactly ON reference levels
    idx = rng.integers(0, ny * nx, 500)
    q.ravel()[idx] = Q[rng.integers(0, ny, 500)].astype(dt)
    if with_nan:
        q[rng.integers(0, ny, 40), rng.integers(0, nx, 40)] = np.nan
        q[5:9, 10:30] = np.nan
    dA = np.abs(np.cos(np.deg2rad(lat)))[:, None] * np.ones((1, nx)) * 1e9 + 1e7 * rng.random((ny, nx))
    return lat, q, Q, dA


@pytest.mark.parametrize('dt,increase,coord_up,part,mkind', [
    (np.float64, True, True, 'all', 'row'), (np.float32, True, False, 'upper', 'plane'), (np.float64, False, True, 'lower', None),
    (np.float32, False, False, 'all', 'row'), (np.float64, True, True, 'upper', None), (np.float64, False, False, 'upper', 'plane')])
def test_lwa_interval_kernel_matches_the_oracle(ctx, dt, increase, coord_up, part, mkind):
    """planes of more than 512 rows: one binary search in the (monotone) reference state per cell + difference arrays + prefix
    sums instead of the band walk -- against the oracle's literal python loop (core.py:752-791) for both directions of the
    tracer and of the coordinate, every `part`, the three metric forms, float32 / float64 tracers, NaN cells, ties with
    levels: <= 1e-11 of the plane's largest value (summation order; the bit-exact band walk is `exact=True`)"""
    rng = np.random.default_rng(int(increase) * 4 + int(coord_up) * 2 + (mkind is not None))
    ny, nx = 600, 334                                       # 334: a ragged last column group
    lat, q, Q, dA = _lwa_case(rng, ny, nx, dt, increase, coord_up, True)
    M = None if mkind is None else (np.abs(np.gradient(np.deg2rad(lat))) * 6.371e6 if mkind == 'row' else 1.0 + rng.random((ny, nx)))
    pcode = {'all': 0, 'upper': 1, 'lower': 2}[part]
    got, _ = ctx.lwa(q[None], Q[None], lat, dA, float(dA.max()), M=M, increase=increase, part=pcode)
    assert ctx.last_lwa_path() == 1
    ref = O.cal_local_wave_activity(q, Q, lat, dA, increase, part, metric=M)
    scale = np.abs(ref).max()
    assert scale > 0 and np.abs(got[0] - ref).max() <= 1e-11 * scale
    ex, _ = ctx.lwa(q[None], Q[None], lat, dA, float(dA.max()), M=M, increase=increase, part=pcode, exact=True)
    assert ctx.last_lwa_path() == 0 and np.array_equal(ex[0], ref)              # the band walk: numpy's own summation order


def test_lwa_interval_kernel_premises_and_stacks(ctx):
    """a reference state that is NOT monotone (or holds a NaN), or a coordinate with a repeated value, sends the call to the
    bit-exact band walk (path 2); a stack of slabs with per-slab Q; masks for mask_idx stay exact on the fast path; planes of
    up to 512 rows never take it"""
    rng = np.random.default_rng(9)
    ny, nx = 520, 128
    lat, q, Q, dA = _lwa_case(rng, ny, nx, np.float64, True, True, False)
    Qbad = Q.copy(); Qbad[100], Qbad[101] = Q[101], Q[100]
    for Qx, c in ((Qbad, lat), (np.where(np.arange(ny) == 7, np.nan, Q), lat), (Q, np.where(np.arange(ny) == 300, lat[299], lat))):
        got, _ = ctx.lwa(q[None], Qx[None], c, dA, float(dA.max()))
        assert ctx.last_lwa_path() == 2
        assert np.array_equal(got[0], O.cal_local_wave_activity(q, Qx, c, dA, True, 'all'), equal_nan=True)
    S = 3
    qs = np.stack([q * (1 + 0.1 * s) for s in range(S)])
    Qs = np.stack([Q * (1 + 0.1 * s) for s in range(S)])
    got, masks = ctx.lwa(qs, Qs, lat, dA, float(dA.max()), mask_idx=[3, 400])
    assert ctx.last_lwa_path() == 1
    for s in range(S):
        ref, _, mref = O.cal_local_wave_activity(qs[s], Qs[s], lat, dA, True, 'all', mask_idx=[3, 400])
        assert np.abs(got[s] - ref).max() <= 1e-11 * np.abs(ref).max()
        assert np.array_equal(masks[s, 0], mref[0]) and np.array_equal(masks[s, 1], mref[1])
    small = _lwa_case(rng, 256, 128, np.float64, True, True, False)
    sref = O.cal_local_wave_activity(small[1], small[2], small[0], small[3], True, 'all')
    g2, _ = ctx.lwa(small[1][None], small[2][None], small[0], small[3], float(small[3].max()))
    assert ctx.last_lwa_path() == 0 and np.array_equal(g2[0], sref)
    # exact=False: the premises are checked on the host and vouched for -- ONE launch of the interval kernel, any plane size
    g3, _ = ctx.lwa(small[1][None], small[2][None], small[0], small[3], float(small[3].max()), exact=False)
    assert ctx.last_lwa_path() == 1 and np.abs(g3[0] - sref).max() <= 1e-11 * np.abs(sref).max() and not np.array_equal(g3[0], sref)
    g4, _ = ctx.lwa(q[None], Qbad[None], lat, dA, float(dA.max()), exact=False)            # not monotone: the host check sends it to the band walk
    assert ctx.last_lwa_path() == 0 and np.array_equal(g4[0], O.cal_local_wave_activity(q, Qbad, lat, dA, True, 'all'))


def test_keff_without_counts_gives_the_same_vectors(ctx):
    """KeffPlan(counts=False) / xc_keff_desc.counts = NULL: the histogram pass skips the count adds (the reference's Keff
    sequence never looks at counts; Contour2D.keff runs this way) -- the nine vectors are the same bits with deterministic
    sums, float32 (E32) and float64, chained and not; xc_hist without 'counts' likewise"""
    from xcontour_amd.pipeline import KeffPlan, OUT_NAMES
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N, S = 91, 1280, 61, 4
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * (360.0 / nx)
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(dA.sum(1), True)
    for dt in (np.float64, np.float32):
        kw = dict(dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, deterministic=True, nslots=2)
        a = KeffPlan(ctx, S, ny, nx, N, dt, dt, **kw)
        a.synth(lat, lon, 3, 0)
        b = KeffPlan(ctx, S, ny, nx, N, dt, dt, alloc_q=False, counts=False, **kw)
        b.set_q_device(a._q_ptr)
        for slot, (group, chain) in enumerate(((None, False), (2, True))):
            a.run(slot, group, chain=chain); b.run(slot, group, chain=chain)
            ra, rb = a.fetch(slot=slot), b.fetch(slot=slot)
            for k in OUT_NAMES:
                assert np.array_equal(bits(ra[k]), bits(rb[k])), (k, dt, chain)
        q = a.download_q()
        r = O.keff_pipeline(q[1], dA, lat, N, lon=lon, increase=True, lt=True, dtype=dt)
        assert np.array_equal(ra['counts'][1].astype(np.int64), r['counts']) and rel(rb['area'][1], r['area']) < TIGHT
        a.free(); b.free()
    rng = np.random.default_rng(2)
    qh = rng.standard_normal((2, 40, 130))
    ed = np.linspace(-4, 4, 33)
    w = rng.random((40, 130))
    full = ctx.hist(qh, ed, dA=w, want=('pdf', 'cdf', 'counts'), deterministic=True)
    part = ctx.hist(qh, ed, dA=w, want=('pdf', 'cdf'), deterministic=True)
    assert np.array_equal(bits(full['pdf']), bits(part['pdf'])) and np.array_equal(bits(full['cdf']), bits(part['cdf']))
    plain = ctx.hist(qh, ed, dA=w, want=('cdf',))
    assert rel(plain['cdf'], full['cdf']) < 1e-13
