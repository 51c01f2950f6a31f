"""-m gpu, round 3: deterministic (order-free fixed-point) sums, and the N > 1 path with REAL pipeline output -- two
processes sharing the one GPU of the test box, each running a KeffPlan over its block of a 1440 x 721 stack, one
gloo all-gather, against the 1-rank run."""
import os
import socket
import sys

import numpy as np
import pytest

import xcontour_oracle as O
from test_gpu_parity import rel, RTOL, TIGHT
from test_gpu_round2 import check_nine, NINE

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NY4, NX4, N4, SEED4 = 721, 1440, 201, 20241008


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


# ---------------------------------------------------------------- deterministic sums
@pytest.mark.parametrize('dt,cd', [(np.float64, np.float64), (np.float32, np.float32)])
def test_deterministic_pipeline_is_order_free(ctx, dt, cd):
    """xc_keff_desc.deterministic: two runs, and launch sets of 1 / 2 / all slabs (different block geometry, different
    partial layout), give the SAME bits in all nine vectors; levels and counts equal the default path's bits, sums agree
    with it and with the oracle to rounding"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums
    ny, nx, N, S = 181, 360, 101, 6
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 1.0
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(dA.sum(1), True)
    kw = dict(dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True, nslots=4)
    plain = KeffPlan(ctx, S, ny, nx, N, dt, cd, **kw)
    plain.synth(lat, lon, 11, 0)
    plain.run(0)
    ref = plain.fetch(slot=0)
    q = plain.download_q()
    det = KeffPlan(ctx, S, ny, nx, N, dt, cd, deterministic=True, alloc_q=False, **kw)
    det.set_q_device(plain._q_ptr)
    outs = []
    for slot, group in enumerate((None, None, 1, 2)):
        det.run(slot, group, chain=(group == 2))                 # chained min / max (q_next) ride in the fixed-point pass
        outs.append(det.fetch(slot=slot))
    for o in outs[1:]:
        for k in NINE:
            assert np.array_equal(bits(o[k]), bits(outs[0][k])), k
        assert np.array_equal(o['counts'], outs[0]['counts'])
    d = outs[0]
    assert np.array_equal(d['ctr'], ref['ctr']) and np.array_equal(d['counts'], ref['counts'])
    assert rel(d['area'], ref['area']) < 1e-12 and rel(d['intgrdS'], ref['intgrdS']) < 1e-12
    for s in range(S):
        r = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=True, lt=True, dtype=cd)
        check_nine_det(d, s, r)
        # the oracle's own restatement of the fixed-point rule (deterministic_bin_sums): the SAME BITS, sums included
        rd = O.keff_pipeline(q[s], dA, lat, N, lon=lon, increase=True, lt=True, dtype=cd, deterministic=True)
        assert np.array_equal(bits(d['area'][s]), bits(rd['area'])) and np.array_equal(bits(d['intgrdS'][s]), bits(rd['intgrdS']))
    det.free(); plain.free()


def test_deterministic_hist_channels_and_odd_inputs(ctx):
    """xc_hist_desc.deterministic through Context.hist: three channels (dA, a SIGNED integrand, |grad q|^2), NaN weights,
    NaN tracer cells, odd nx, float32 tracer; splitting the stack changes the launch geometry, not one bit"""
    rng = np.random.default_rng(5)
    for (S, ny, nx, dt) in ((5, 90, 131, np.float32), (4, 64, 256, np.float64), (3, 33, 2, np.float64)):
        q = rng.standard_normal((S, ny, nx)).astype(dt)
        q[0, 3, 1] = np.nan
        dA = rng.random((ny, nx)) + 0.1
        dA[5, 0] = np.nan                                                 # fillna(0), core.py:449
        g = rng.standard_normal((S, ny, nx)) * 10.0 ** rng.integers(-8, 8, (S, ny, nx))     # 16 decades, both signs
        edges = np.linspace(-2.5, 2.5, 41)
        rdx = rng.random(ny) + 0.5; rdy = rng.random(ny) + 0.5
        kw = dict(dA=dA, integrands=[g], grad=(rdx, rdy, True), last_closed=False, lt=True, want=('pdf', 'counts', 'cdf'))
        a = ctx.hist(q, edges, deterministic=True, **kw)
        b = ctx.hist(q, edges, deterministic=True, **kw)
        for k in ('pdf', 'cdf'):
            assert np.array_equal(bits(a[k]), bits(b[k]))
        parts = [ctx.hist(q[s:s + 1], edges, deterministic=True, **dict(kw, integrands=[g[s:s + 1]])) for s in range(S)]   # one slab per launch
        assert np.array_equal(bits(np.concatenate([p['pdf'] for p in parts])), bits(a['pdf']))
        plain = ctx.hist(q, edges, **kw)
        assert np.array_equal(a['counts'], plain['counts'])
        # against the default path on the scale of each channel's largest bin (a signed channel cancels inside a bin)
        for ch in range(3):
            scale = np.abs(plain['pdf'][:, ch]).max(axis=1, keepdims=True) + 1e-300
            assert (np.abs(a['pdf'][:, ch] - plain['pdf'][:, ch]) / scale).max() < 1e-10, ch
        # the area channel against numpy's histogram (no cell sits on the last edge, so the closed last bin is moot), and the
        # area + SIGNED integrand channels bit for bit against the oracle's restatement of the fixed-point rule
        w = np.where(np.isnan(dA), 0.0, dA)
        for s in range(S):
            assert not (q[s] == edges[-1]).any()
            ref, _ = np.histogram(q[s].astype(np.float64).ravel(), bins=edges, weights=w.ravel())
            assert rel(a['pdf'][s, 0], ref) < 1e-12
            od, _ = O.weighted_histogram(q[s].astype(np.float64), edges, w, 'numpy', deterministic=True)
            assert np.array_equal(bits(a['pdf'][s, 0]), bits(od))
            wg = g[s] * dA
            oi, _ = O.weighted_histogram(q[s].astype(np.float64), edges, np.where(np.isnan(wg), 0.0, wg), 'numpy', deterministic=True)
            assert np.array_equal(bits(a['pdf'][s, 1]), bits(oi))


def test_deterministic_infinite_weight_reports_nan(ctx):
    q = np.linspace(0.05, 0.95, 64 * 128).reshape(1, 64, 128)
    dA = np.ones((64, 128)); dA[10, 7] = np.inf
    edges = np.linspace(0, 1, 11)
    a = ctx.hist(q, edges, dA=dA, deterministic=True, last_closed=False, want=('pdf', 'counts'))
    k = int(np.digitize(q[0, 10, 7], edges) - 1)
    assert np.isnan(a['pdf'][0, 0, k]) and np.isfinite(np.delete(a['pdf'][0, 0], k)).all()
    assert a['counts'].sum() == 64 * 128


def test_cfg2_full_size_deterministic(ctx):
    """two runs at full cfg2 size give bit-identical area / intgrdS (and everything derived), equal to
    the oracle within the same bars as the default path; a facade object with deterministic=True does the same"""
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    ny, nx, N = 1801, 3600, 201
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 0.1
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, ny, nx), True, last_row_included(lat))
    plan = KeffPlan(ctx, 2, ny, nx, N, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, deterministic=True, nslots=2)
    plan.synth(lat, lon, 20241008, 0)
    plan.run(0)
    plan.run(1, group=1)
    a, b = plan.fetch(slot=0), plan.fetch(slot=1)
    for k in NINE:
        assert np.array_equal(bits(a[k]), bits(b[k])), k
    q = plan.download_q()
    r = O.keff_pipeline(q[1], dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float64)
    check_nine_det(a, 1, r)
    rd = O.keff_pipeline(q[1], dA, lat, N, lon=lon, increase=True, lt=True, dtype=np.float64, deterministic=True)
    assert np.array_equal(bits(a['area'][1]), bits(rd['area'])) and np.array_equal(bits(a['intgrdS'][1]), bits(rd['intgrdS']))   # 6.5 M cells, bit for bit
    plan.free()


# ---------------------------------------------------------------- N > 1 with real pipeline output
def _cfg4_block(ctx, lo, hi, chunk, det):
    """this rank's (hi - lo, 9, N) block of the cfg4-shaped stack: chained launch sets of `chunk` slabs (ragged last set),
    result slots -> slab-major block (the flow of bench.py's cfg4_strong)"""
    import ctypes as C
    from xcontour_amd import _native as nat
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.distributed import chunks_to_slabs
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    lat = np.linspace(-90, 90, NY4); lon = np.arange(NX4) * 0.25
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, NY4, NX4), True, last_row_included(lat, 'xhistogram'))
    n = hi - lo
    if n == 0:
        return np.empty((0, 9, N4))
    Cn = min(chunk, n)
    nchunk = -(-n // Cn)
    sb = NY4 * NX4 * 8
    qbuf = ctx.alloc(n * sb)
    lat_b, lon_b = ctx.to_device(lat), ctx.to_device(lon)
    for c0 in range(0, n, Cn):
        m = min(Cn, n - c0)
        ctx._check(ctx.lib.xc_synth_dev(ctx.handle, qbuf.ptr + c0 * sb, nat.XC_F64, m, NY4, NX4, lat_b.ptr, lon_b.ptr, SEED4 + lo + c0, 0))
    ctx.sync()
    plan = KeffPlan(ctx, Cn, NY4, NX4, N4, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, nslots=nchunk, alloc_q=False, deterministic=det)
    for ci in range(nchunk):
        c0 = ci * Cn
        m = min(Cn, n - c0)
        plan.set_q_device(qbuf.ptr + c0 * sb)
        nxt = ((ci + 1) % nchunk) * Cn
        plan._point(ci, 0, m)
        plan.desc.q_next = (qbuf.ptr + nxt * sb) if min(Cn, n - nxt) == m else None
        ctx._check(ctx.lib.xc_keff_dev(ctx.handle, C.byref(plan.desc)))
    ctx.sync()
    res = plan.out_buf.download((nchunk * plan.slot_bytes // 8,), np.float64)
    mine = chunks_to_slabs(res, plan.slot_bytes // 8, Cn, n, N4)
    plan.free(); qbuf.free(); lat_b.free(); lon_b.free()
    return mine


def _gpu_rank(rank, world, port, S, chunk, det, carrier, outq):
    """one rank = one fresh process with its own context on the (shared) GPU; the one gather travels over torch.distributed
    gloo or over the package's own torch-free SocketGroup"""
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        from xcontour_amd import _native as nat
        from xcontour_amd.pipeline import shard_slabs
        from xcontour_amd.distributed import all_gather_slabs, SocketGroup
        ctx = nat.Context(0)
        lo, hi = shard_slabs(S, rank, world)
        mine = np.ascontiguousarray(_cfg4_block(ctx, lo, hi, chunk, det))
        if carrier == 'socket':
            g = SocketGroup(rank, world, '127.0.0.1', port)
            full = all_gather_slabs(mine, S, rank, world, group=g)
            g.barrier()
            outq.put((rank, np.asarray(full).copy(), None))
            g.close()
            ctx.close()
            return
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
        full = all_gather_slabs(torch.from_numpy(mine), S, rank, world)
        dist.barrier()
        outq.put((rank, full.numpy().copy(), None))
        ctx.close()
        dist.destroy_process_group()
    except Exception as e:                      # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        outq.put((rank, None, traceback.format_exc() + repr(e)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('det,carrier,world,S', [(True, 'gloo', 2, 11), (False, 'gloo', 2, 11), (True, 'socket', 2, 11),
                                                 (True, 'gloo', 3, 11), (True, 'socket', 4, 9)])
def test_two_ranks_share_one_gpu_real_pipeline(ctx, det, carrier, world, S):
    """SURVEY 8(e) end to end on the hardware at hand: 2 - 4 spawned processes (never a re-exec of a process that touched the
    GPU), each a real KeffPlan over its shard_slabs block of a small 1440 x 721 stack (ragged blocks: 6 + 5, 4 + 4 + 3, and
    3 + 3 + 3 + 0 -- a rank WITHOUT slabs; ragged launch sets of 4), chunks_to_slabs -> all_gather_slabs over gloo or the
    package's own sockets.  Every rank must hold the 1-rank result: all nine vectors bit for bit with deterministic sums;
    levels bit for bit and sums to 1e-12 with the default float64 atomics."""
    import torch.multiprocessing as mp
    chunk = 4
    ref = _cfg4_block(ctx, 0, S, 5, det)                               # 1 rank, other launch-set size on purpose
    mpc = mp.get_context('spawn')
    outq = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_gpu_rank, args=(r, world, port, S, chunk, det, carrier, outq)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, arr, err = outq.get(timeout=600)
        assert err is None, 'rank %d failed:\n%s' % (r, err)
        got[r] = arr
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ref.shape == (S, 9, N4)
    ictr = NINE.index('ctr')
    for r in range(world):
        assert got[r].shape == ref.shape
        assert np.array_equal(bits(got[r][:, ictr]), bits(ref[:, ictr]))
        if det:
            assert np.array_equal(bits(got[r]), bits(ref))             # all nine vectors, every slab, every rank
        else:
            for k in ('area', 'intgrdS', 'latEq'):
                i = NINE.index(k)
                assert rel(got[r][:, i], ref[:, i]) < 1e-12, k
    assert all(np.array_equal(bits(got[0]), bits(got[r])) for r in range(1, world))   # the gather hands every rank the same bytes
    # and the stack really is per-slab data in slab order: slab 7 against the oracle
    from xcontour_amd import _native as nat
    from xcontour_amd.utils import cell_area
    lat = np.linspace(-90, 90, NY4); lon = np.arange(NX4) * 0.25
    buf = ctx.alloc(NY4 * NX4 * 8)
    lb_, lo_ = ctx.to_device(lat), ctx.to_device(lon)
    ctx._check(ctx.lib.xc_synth_dev(ctx.handle, buf.ptr, nat.XC_F64, 1, NY4, NX4, lb_.ptr, lo_.ptr, SEED4 + 7, 0))
    q7 = buf.download((NY4, NX4), np.float64)
    r7 = O.keff_pipeline(q7, cell_area(lat, lon), lat, N4, lon=lon, increase=True, lt=True, dtype=np.float64)
    assert np.array_equal(ref[7, ictr], r7['ctr'])
    assert rel(ref[7, NINE.index('area')], r7['area']) < TIGHT and rel(ref[7, NINE.index('intgrdS')], r7['intgrdS']) < TIGHT
    assert rel(ref[7, NINE.index('latEq')], r7['latEq']) < RTOL
    buf.free(); lb_.free(); lo_.free()


# ---------------------------------------------------------------- K8: three range-key passes + short-run repair
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


# ---------------------------------------------------------------- byte-bounded staging of the facade (missing #4)
def test_facade_methods_stream_large_stacks_in_batches(ctx, baro):
    """a stack larger than a deliberately small staging cap goes through the device in batches of whole slabs -- histogram
    integrals, crossing, LWA, sorted profile, the fused keff() with its double-buffered uploads -- with the results of one
    big launch (counts / levels / exact kernels bit for bit, float sums to rounding)"""
    import xcontour_amd as xa
    q0, lat, lon = baro
    S = 7
    rng = np.random.default_rng(4)
    q = np.stack([q0 * (1 + 0.1 * s) + 1e-6 * rng.standard_normal(q0.shape).astype(np.float32) for s in range(S)])
    c = {'time': np.arange(S), 'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(q, ('time', 'latitude', 'longitude'), c, 'absolute_vorticity')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'rA')
    mask = xa.DataArray(np.ones_like(q0), ('latitude', 'longitude'), {'latitude': lat, 'longitude': lon}, 'mask')
    cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
    table = cm.cal_area_eqCoord_table_hist(mask)
    ctr = cm.cal_contours(41)
    dy = np.gradient(np.deg2rad(lat.astype(np.float64))) * O.Rearth
    Qeq = xa.DataArray(np.sort(q.mean(axis=2), axis=1), ('time', 'latitude'), {'time': np.arange(S), 'latitude': lat}, 'absolute_vorticity')

    def everything(cap_keff):
        return dict(area=cm.cal_integral_within_contours_hist(ctr).values,
                    cross=cm.cal_contour_crossing(ctr, stride=[1, 2]),
                    lwa=cm.cal_local_wave_activity(tr, Qeq, metric=dy).values,
                    prof=cm.cal_sorted_profile(table).values,
                    keff=cm.keff(41, table, lat=lat, lon=lon, max_batch_bytes=cap_keff))

    cap = cm.ctx.max_batch_bytes
    big = everything(8 << 30)
    try:
        cm.ctx.max_batch_bytes = 3 * q0.nbytes + 1000                 # room for one or two slabs' worth of staged bytes
        assert len(cm.ctx._batches(S, q0.nbytes)) >= 3
        small = everything(5 * q0.nbytes)                              # keff: two device halves of two slabs each, four batches
    finally:
        cm.ctx.max_batch_bytes = cap
    assert rel(small['area'], big['area']) < 1e-12
    for a, b in zip(small['cross'], big['cross']):
        assert rel(a.values, b.values) < 1e-12
    assert np.array_equal(small['lwa'], big['lwa'])                    # sequential sums: bit-identical
    assert np.array_equal(small['prof'], big['prof'])
    for k in ('ctr', 'area', 'intgrdS', 'latEq', 'nkeff'):
        a, b = small['keff'][k].values, big['keff'][k].values
        assert a.shape == (S, 41)
        assert np.array_equal(a, b) if k == 'ctr' else rel(a, b) < 1e-9, k
    cm.close()


def test_crossing_uncrossed_interior_levels_are_exact_zeros(ctx):
    """two regions of the plane separated by NaN columns hold values in [0, 0.3] and [0.7, 1]; no box has corners in
    both, so the levels in between are crossed by nothing and must come out as the exact 0 of the reference's per-contour
    loop (the difference-array accumulation used to leave ~1e-16-of-the-mass residues there), and nothing is negative"""
    rng = np.random.default_rng(8)
    ny, nx = 96, 260
    q = np.empty((2, ny, nx))
    q[:, :, :128] = 0.3 * rng.random((2, ny, 128))
    q[:, :, 128:132] = np.nan
    q[:, :, 132:] = 0.7 + 0.3 * rng.random((2, ny, nx - 132))
    area = 1e9 * (1 + rng.random((ny, nx)))
    ctr = np.linspace(0.0, 1.0, 101)
    lens, cnts = ctx.crossing(q, ctr, area, stride=1, full_width=True)
    mid = (ctr > 0.305) & (ctr < 0.695)
    assert (cnts[:, mid] == 0).all() and (cnts[:, (ctr > 0.05) & (ctr < 0.25)] > 0).all() and (cnts[:, (ctr > 0.75) & (ctr < 0.95)] > 0).all()
    assert (lens[:, mid] == 0.0).all()
    assert (lens >= 0).all()
    for s in range(2):
        ol, oc = O.contour_crossing(q[s], ctr, area, 1, True)
        assert np.array_equal(cnts[s].astype(np.int64), oc) and rel(lens[s], ol) < 1e-13


def test_hist_all_nan_rows_and_land_mask(ctx):
    """rows that are entirely NaN (land in an ocean field) are skipped as a whole by K3; counts and sums
    are those of the oracle's histogram"""
    rng = np.random.default_rng(9)
    ny, nx = 120, 384
    q = rng.standard_normal((3, ny, nx))
    q[:, 10:40, :] = np.nan                                   # whole rows
    q[:, 60:90, 100:300] = np.nan                             # a continent
    q[1] = np.nan                                             # a slab with no ocean at all
    dA = rng.random((ny, nx)) + 0.5
    edges = np.linspace(-3, 3, 61)
    out = ctx.hist(q, edges, dA=dA, last_closed=True, want=('pdf', 'counts'))
    for s in range(3):
        ok = ~np.isnan(q[s])
        rc, _ = np.histogram(q[s][ok], bins=edges)
        rw, _ = np.histogram(q[s][ok], bins=edges, weights=dA[ok])
        assert np.array_equal(out['counts'][s].astype(np.int64), rc)
        assert rel(out['pdf'][s, 0], rw) < 1e-12


def test_keff_double_buffered_batches_with_supplied_grdS_and_per_slab_dA(ctx, baro):
    """the multi-batch path of Contour2D.keff (two device halves, uploads on the copy stream) with everything that travels per
    slab -- the tracer, a supplied squared gradient, time-varying weights -- against the single-batch result and the oracle"""
    import xcontour_amd as xa
    q0, lat, lon = baro
    S = 5
    rng = np.random.default_rng(14)
    q = np.stack([q0 * (1 + 0.05 * s) for s in range(S)])
    g = (rng.random(q.shape) * 1e-16).astype(np.float32)
    dA0 = O.cell_area(lat, lon)
    dA3 = np.stack([dA0 * (1 + 0.01 * s) for s in range(S)])
    c3 = {'time': np.arange(S), 'latitude': lat, 'longitude': lon}
    c2 = {'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(q, ('time', 'latitude', 'longitude'), c3, 'absolute_vorticity')
    gs = xa.DataArray(g, ('time', 'latitude', 'longitude'), c3, 'grdS')
    mask = xa.DataArray(np.ones_like(q0), ('latitude', 'longitude'), c2, 'mask')
    for dA in (xa.DataArray(dA0, ('latitude', 'longitude'), c2, 'rA'), xa.DataArray(dA3, ('time', 'latitude', 'longitude'), c3, 'rA')):
        cm = xa.Contour2D(tr, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True)
        table = xa.Contour2D(tr, xa.DataArray(dA0, ('latitude', 'longitude'), c2, 'rA'), dims={'X': 'longitude', 'Y': 'latitude'},
                             dimEq={'Y': 'latitude'}, increase=True, lt=True).cal_area_eqCoord_table_hist(mask)
        one = cm.keff(31, table, grdS=gs)
        per = q0.nbytes + g[0].nbytes + (dA0.nbytes if dA.values.ndim == 3 else 0)
        many = cm.keff(31, table, grdS=gs, max_batch_bytes=2 * per * 2 + 100)          # two slabs per half: batches 2 + 2 + 1
        for k in ('ctr', 'area', 'intgrdS', 'latEq', 'nkeff'):
            a, b = many[k].values, one[k].values
            assert np.array_equal(a, b) if k == 'ctr' else rel(a, b) < 1e-9, k
        for s in (0, S - 1):
            w = dA.values if dA.values.ndim == 2 else dA.values[s]
            r = O.keff_pipeline(q[s], w, lat, 31, grdS=g[s], increase=True, lt=True, dtype=np.float32)
            assert np.array_equal(many['ctr'].values[s], r['ctr'].astype(np.float64))
            assert rel(many['area'].values[s], r['area']) < TIGHT and rel(many['intgrdS'].values[s], r['intgrdS']) < 1e-6
        cm.close()


# ---------------------------------------------------------------- K5: the cumulative sums keep np.cumsum's order
@pytest.mark.parametrize('nint', [0, 2])
def test_cdf_is_the_sequential_cumsum_of_the_pdf_bit_for_bit(ctx, nint):
    """k_finalize takes the cumulative sums systolically (lanes hold four elements each, 256 per chunk): for bin counts
    around the lane / chunk boundaries, for thousands of bins (work arrays in LDS and, with three channels, in global
    memory), for `lt` or not and both level orders, cdf is bit-identical to np.cumsum of the returned pdf (core.py:1320-1323)"""
    rng = np.random.default_rng(5)
    ny, nx = 40, 96
    q = rng.standard_normal((2, ny, nx))
    dA = rng.random((ny, nx)) + 0.1
    integ = [rng.standard_normal((2, ny, nx)) for _ in range(nint)]
    for nb in (1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 300, 513, 1023, 4000 if nint == 0 else 3500):
        edges = np.linspace(-3.0, 3.0, nb + 1)
        for lt in (True, False):
            for reverse in (False, True):
                out = ctx.hist(q, edges, dA, integ, lt=lt, reverse=reverse)
                pdf = out['pdf'][..., ::-1] if reverse else out['pdf']             # ascending-value order
                c = np.cumsum(pdf, axis=-1)
                if not lt:
                    c = c[..., -1:] - c
                if reverse:
                    c = c[..., ::-1]
                assert np.array_equal(bits(out['cdf']), bits(c)), (nb, lt, reverse)
                assert out['counts'].sum() <= 2 * ny * nx
    # and against the oracle's weighted histogram for one of them
    edges = np.linspace(-3.0, 3.0, 258)
    out = ctx.hist(q, edges, dA, integ, lt=True)
    for s in range(2):
        for ch, w in enumerate([dA] + [v[s] * dA for v in integ]):
            ref, cnt = O.weighted_histogram(q[s], edges, w, right_edge='numpy')
            assert np.array_equal(out['counts'][s].astype(np.int64), cnt)
            assert rel(out['pdf'][s].reshape(1 + nint, -1)[ch], ref) < TIGHT


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


# ---------------------------------------------------------------- K4 on odd shapes
@pytest.mark.parametrize('dt', [np.float64, np.float32])
def test_grad2_shapes_and_walls_bit_identical(ctx, dt):
    """K4 gives the bits of the normative order of operations (oracle.grad2_sphere) for widths around the 256-column
    workgroup boundary, one or a few rows, NaN cells, periodic or walled (one-sided differences at the walls)"""
    rng = np.random.default_rng(8)
    for ny, nx in ((1, 4), (2, 6), (17, 126), (33, 128), (40, 130), (5, 254), (35, 256), (16, 258), (31, 510), (32, 512), (3, 514),
                   (9, 1026), (7, 129), (19, 257)):
        q = rng.standard_normal((2, ny, nx)).astype(dt)
        q[1, ny // 2, nx // 3] = np.nan
        rdx = rng.random(ny) + 0.5
        rdy = rng.random(ny) + 0.5
        for periodic in (True, False):
            got = ctx.grad2(q, rdx, rdy, periodic)
            x = q.astype(np.float64)
            if periodic:
                gx = (np.roll(x, -1, axis=2) - np.roll(x, 1, axis=2)) * rdx[None, :, None]
            else:
                e = np.concatenate((x[:, :, 1:], x[:, :, -1:]), axis=2)
                w = np.concatenate((x[:, :, :1], x[:, :, :-1]), axis=2)
                f = np.ones(nx); f[0] = 2.0; f[-1] = 2.0
                gx = ((e - w) * rdx[None, :, None]) * f[None, None, :]
            jn = np.minimum(np.arange(ny) + 1, ny - 1); js = np.maximum(np.arange(ny) - 1, 0)
            gy = (x[:, jn, :] - x[:, js, :]) * rdy[None, :, None]
            want = gx * gx + gy * gy
            assert np.array_equal(got, want, equal_nan=True), (ny, nx, periodic)


# ---------------------------------------------------------------- resident inputs (xc_keep_resident)
def test_resident_inputs_give_the_same_results(ctx, baro):
    """Contour2D(resident=True): tracer and weights are uploaded once and the host-form calls copy from the device mirror --
    same bits as without, for the reference's Keff call sequence on a stack, also when the stack goes through in batches of
    whole slabs (slices of the registered array) and after touch() following an in-place change"""
    import xcontour_amd as xa
    q0, lat, lon = baro
    S = 4
    q = np.stack([q0 * (1 + 0.1 * s) for s in range(S)])
    c3 = {'time': np.arange(S), 'latitude': lat, 'longitude': lon}
    c2 = {'latitude': lat, 'longitude': lon}
    tr = xa.DataArray(q, ('time', 'latitude', 'longitude'), c3, 'absolute_vorticity')
    dA = xa.DataArray(O.cell_area(lat, lon), ('latitude', 'longitude'), c2, 'rA')
    g = xa.DataArray(np.random.default_rng(2).random(q.shape).astype(np.float32), ('time', 'latitude', 'longitude'), c3, 'grdS')
    mask = xa.DataArray(np.ones_like(q0), ('latitude', 'longitude'), c2, 'mask')
    kw = dict(dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'}, increase=True, lt=True, deterministic=True)

    def sequence(cm):
        table = cm.cal_area_eqCoord_table_hist(mask)
        ctr = cm.cal_contours(61)
        area = cm.cal_integral_within_contours_hist(ctr)
        intS = cm.cal_integral_within_contours_hist(ctr, integrand=g)
        ds = cm.keff(61, table, lat=lat, lon=lon)                    # the fused pipeline uploads through xc_memcpy_h2d_async: mirror-aware too
        return [table.lookup_coordinates(area).values, ctr.values, area.values, intS.values, ds['area'].values, ds['nkeff'].values]

    plain = xa.Contour2D(tr, dA, **kw)
    ref = sequence(plain)
    res = xa.Contour2D(tr, dA, resident=True, **kw)
    n0 = len(res.ctx._resident)
    got = sequence(res)
    assert len(res.ctx._resident) == n0 + 4                         # the tracer stack, the float64 weights, the (time-invariant) mask and (round 5) the last integrand, once each
    for a, b in zip(got, ref):
        assert np.array_equal(bits(a), bits(b))
    old = res.ctx.max_batch_bytes
    try:
        res.ctx.max_batch_bytes = 2 * q0.nbytes + 100                # two slabs per batch: slices of the registered stack
        for a, b in zip(sequence(res), ref):
            assert np.array_equal(bits(a), bits(b))
    finally:
        res.ctx.max_batch_bytes = old
    q[1] = np.roll(q[1], 9, axis=0)                                 # in place (rows meet other weights): the mirror is stale until touch()
    res.touch()
    again = sequence(res)
    fresh = sequence(xa.Contour2D(tr, dA, **kw))
    for a, b in zip(again, fresh):
        assert np.array_equal(bits(a), bits(b))
    assert not np.array_equal(bits(again[2]), bits(ref[2]))
    res.close(); plain.close()
    assert len(res.ctx._resident) == n0


# ---------------------------------------------------------------- the stub of INTEGRATION.md, executed as printed
def test_integration_md_stub_runs_as_printed(baro):
    """the two python blocks INTEGRATION.md shows a maintainer of the reference (ctypes binding of xc_hist and xc_crossing) are
    executed verbatim against the built library: histogram_cdf == the oracle's _histogram restatement (counts-exact levels,
    sums to rounding), contour_crossing == the oracle's box counting"""
    import re
    from xcontour_amd import _native as nat
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    stub = next(b for b in blocks if 'class _HistDesc' in b)
    cross = next(b for b in blocks if 'def contour_crossing' in b)
    ns = {}
    exec(stub.replace("C.CDLL('libxcontour_hip.so')", 'C.CDLL(%r)' % nat.LIB_PATH), ns)
    exec(cross, ns)
    q, lat, lon = baro
    dA = O.cell_area(lat, lon)
    for inc in (True, False):
        ctr = O.cal_contours(q, 61, inc, np.float32)
        for lt in (True, False):
            got = ns['histogram_cdf'](q[None], ctr, dA, lt)
            want = O.histogram_cdf(q, ctr, dA, lt)                 # ascending-value order like _histogram's return
            want = want[0] if isinstance(want, tuple) else want
            assert rel(got[0], np.asarray(want, dtype=np.float64)) < TIGHT, (inc, lt)
    levels = np.sort(np.linspace(float(np.nanmin(q)), float(np.nanmax(q)), 9)[1:-1]).astype(np.float64)
    got = ns['contour_crossing'](q[None].astype(np.float64), levels[None], dA, 2, 2, 'edge')      # stride 2, padded by max_stride = 2 columns
    want, _ = O.contour_crossing(O.pad_x(q.astype(np.float64), 2, 'edge'), levels, O.pad_x(dA, 2, 'edge'), 2)
    assert rel(got[0], np.asarray(want)) < 1e-12


def test_facade_cycles_do_not_leak_device_memory(ctx):
    """create / use / close Contour2D objects over and over (resident or not, deterministic or not, every operator): the
    free device memory settles -- plans, resident mirrors and work buffers are returned"""
    import ctypes as C
    import xcontour_amd as xa
    hip = C.CDLL('libamdhip64.so')

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    ny, nx, S = 91, 180, 4
    lat = np.linspace(-90, 90, ny); lon = np.arange(nx) * 2.0
    rng = np.random.default_rng(0)
    c3 = {'t': np.arange(S), 'lat': lat, 'lon': lon}; c2 = {'lat': lat, 'lon': lon}
    dA = xa.DataArray(O.cell_area(lat, lon), ('lat', 'lon'), c2, 'dA')
    mask = xa.DataArray(np.ones((ny, nx)), ('lat', 'lon'), c2, 'mask')
    base = None
    for it in range(45):
        q = np.sin(np.deg2rad(lat))[None, :, None] + 0.05 * rng.standard_normal((S, ny, nx))
        tr = xa.DataArray(q, ('t', 'lat', 'lon'), c3, 'pv')
        cm = xa.Contour2D(tr, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True,
                          resident=(it % 2 == 0), deterministic=(it % 3 == 0))
        table = cm.cal_area_eqCoord_table_hist(mask)
        ctr = cm.cal_contours(31)
        cm.cal_integral_within_contours_hist(ctr)
        ds = cm.keff(31, table, preY=lat, lat=lat, lon=lon)
        if it % 5 == 0:
            cm.cal_local_wave_activity(tr, ds['ctr_eq'].rename({'new': 'lat'}))
            cm.cal_sorted_profile(table)
            cm.cal_contour_crossing(ctr, stride=[1, 2])
        cm.close()
        del cm
        if it == 14:
            base = free_bytes()
    assert base - free_bytes() < (1 << 20), 'device memory keeps shrinking: %d bytes since iteration 14' % (base - free_bytes())
