"""-m gpu: the N > 1 path on the hardware at hand -- ranks sharing one GPU with REAL pipeline output, bench.py launching its own ranks, the carrier
ladder (RCCL -> HIP IPC -> host), the gather-to-root against the one-rank job bit for bit, contexts on every visible device.
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
NY4, NX4, N4, SEED4 = 721, 1440, 201, 20241008          # the cfg4 slab shape


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
                    increase=True, lt=True, nslots=nchunk, alloc_q=False, deterministic=det, single_read=False)
    # (single_read=False: this helper calls xc_keff_dev directly and never looks at status 2 -- ranks that share ONE GPU cannot all hold it
    # for the single-read kernel at once; KeffPlan.fetch would repeat such a launch set on the chain, tests/test_gpu_single.py)
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
    # the weak-scaling line itself says what every rank did and what RCCL reported (here: the host carrier, so `rccl` is null)
    pr = d['per_rank']
    assert len(pr['sweep_ms']) == 2 and all(x > 0 for x in pr['sweep_ms']) and len(pr['wall_ms']) == 2 and pr['device'] == [0, 0]
    assert d['config']['rccl'] is None and 'carrier host' in d['config']['parallelism']


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    """`--gpus 8` inside a 1-rank environment used to run one rank and print n_gpus 1 (round-3 review): now an error"""
    env = _clean_env()
    env.update({'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0', '--no-cpu'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr and r.stdout.strip() == ''


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
    assert len(d['per_rank']['sweep_ms']) == 2 and all(x > 0 for x in d['per_rank']['sweep_ms'])
    if n.value < 2:
        assert 'rccl unavailable' in d['config']['collective_note'] and 'carrier ipc' in d['config']['parallelism']
        assert 'error' in tr['rccl'] and tr['ipc']['ms_1MB'] > 0 and tr['ipc']['ms_32MB'] > 0 and 'host' not in tr
        assert d['config']['rccl'] is None and d['per_rank']['device'] == [0, 0]          # RCCL carried nothing: nothing of it is reported
    else:
        assert d['config']['collective_note'] is None and 'carrier rccl' in d['config']['parallelism'] and tr['rccl']['ms_32MB'] > 0
        rc = d['config']['rccl']                                                           # read back from the library, not from the launcher
        assert rc['consistent'] and rc['comm_count'] == [2, 2] and rc['comm_rank'] == [0, 1] and rc['rank_devices'] == [0, 1]
        assert rc['version'] > 20000 and 'rccl' in rc['librccl']


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
    assert d['config']['rccl'] is None and len(d['per_rank']['sweep_ms']) == world and d['per_rank']['device'] == [0] * world
    assert outs[1][0]['config']['rccl'] is None and len(outs[1][0]['per_rank']['sweep_ms']) == 1      # the fields are there at N = 1 too
    assert d['roofline']['frac'] > 0 and d['cpu_baseline']['value'] > 0 and d['cpu_baseline']['parity_checked_slabs'] >= 1
    c4 = d['cfg4_strong']
    per = -(-100 // world)
    assert c4['n_gpus'] == world and c4['slabs_per_gpu'] == per and c4['pieces_per_job'] == -(-per // 8) and 'HIP IPC' in c4['gather']
    assert c4['checks']['first_slab_of_each_rank_recomputed'] == [r * per for r in range(world) if r * per < 100]
    assert c4['budget']['gather_ms_max'] < 5.0, c4['budget']
    assert full.shape == outs[1][1].shape == (100, 9, 201)
    assert np.array_equal(bits(full), bits(outs[1][1]))                   # 100 = 4 x 25: ragged launch sets, and at N = 4 pieces of 8, 8, 8, 1


def test_contexts_on_every_visible_device():
    """INTEGRATION.md threading contract: one context per GPU, driven from threads of ONE process.  The >64 KB LDS
    opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-device property of a kernel: every device's FIRST
    launch of the big-LDS kernels must work.  Skips on a single-GPU box."""
    import threading
    from xcontour_amd import _native as nat
    ctxs = []
    for d in range(16):
        try:
            ctxs.append(nat.Context(d))
        except nat.XContourHipError:
            break
    try:
        if len(ctxs) < 2:
            pytest.skip('needs at least two visible GPUs (this box has %d)' % len(ctxs))
        rng = np.random.default_rng(9)
        q = rng.standard_normal((3, 120, 200))
        ed = np.linspace(-4, 4, 202)
        dA = rng.random((120, 200)) + 0.5
        _, cnt = zip(*[O.weighted_histogram(q[s], ed, dA, 'numpy') for s in range(3)])
        out, errs = [None] * len(ctxs), []

        def work(i):
            try:
                out[i] = ctxs[i].hist(q, ed, dA=dA, want=('counts', 'cdf'))     # ~100 KB of LDS histogram copies
            except Exception as e:            # pragma: no cover
                errs.append((i, e))
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for o in out:
            for s in range(3):
                assert np.array_equal(o['counts'][s].astype(np.int64), cnt[s])
    finally:
        for c in ctxs:
            c.close()


def test_comm_stream_fences_and_root_gather_with_one_rank(ctx):
    """the comm stream on its own (all one GPU and one rank can show of it): a device-to-device push waits for the compute stream's
    producer (xc_comm_wait_compute), the compute stream's consumer waits for the push (xc_compute_wait_comm), xc_streams_idle polls
    without blocking, sync_within gives up or returns True; and xc_comm_gather_dev of a one-rank communicator is its own block copied
    into place on that stream (the root's share of every real gather)"""
    n = 1 << 20
    x = np.arange(n, dtype=np.float64)
    src, mid, dst = ctx.to_device(x), ctx.alloc(n * 8), ctx.alloc(3 * n * 8)
    ctx._check(ctx.lib.xc_memset(ctx.handle, dst.ptr, 0, 3 * n * 8))
    ctx.comm_memcpy_d2d(mid.ptr, src.ptr, n * 8)                          # (producer stand-in on the comm stream itself)
    ctx.comm_wait_compute()
    ctx.comm_memcpy_d2d(dst.ptr + n * 8, mid.ptr, n * 8)
    ctx.compute_wait_comm()
    assert ctx.sync_within(30.0) and ctx.streams_idle()
    got = dst.download((3, n), np.float64)
    assert np.array_equal(got[1], x) and not got[0].any() and not got[2].any()
    assert ctx.comm_info()['comm_count'] == 0                             # no communicator yet
    uid = ctx.comm_unique_id()
    comm = ctx.comm_create(1, 0, uid)                                     # (the two-step form: created off-context, attached by the owner)
    ctx.comm_attach(comm, 1, 0)
    info = ctx.comm_info()                                                # what RCCL ITSELF says: ncclCommCount / UserRank / CuDevice / version
    assert info['comm_count'] == 1 and info['comm_rank'] == 0 and info['comm_device'] == info['ctx_device'] == ctx.device
    assert info['rccl_version'] > 20000 and 'rccl' in info['rccl_path']
    ctx.comm_wait_compute()
    ctx.comm_gather(src.ptr, n * 8, dst.ptr + 2 * n * 8, n * 8, 0)       # rank 0's block lands at recv + 0 * stride
    ctx.compute_wait_comm(); ctx.sync()
    assert np.array_equal(dst.download((3, n), np.float64)[2], x)
    ctx.comm_finalize()
    for b in (src, mid, dst):
        b.free()


@pytest.mark.parametrize('world', [2, 3])
def test_ipc_pushes_between_rank_processes_on_one_gpu(world):
    """xc_ipc_export / xc_ipc_open / xc_comm_memcpy_d2d below bench.py: rank processes that share this box's GPU push patterned blocks into a
    buffer the root exported, the root compares every word (tools/probe/ipc_probe.py, the probe that answered "does HIP IPC work here?")"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'probe', 'ipc_probe.py'), str(world), '8'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, env=_clean_env(), timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['ipc_push_ok'] is True and d['world'] == world


def test_a_rank_that_never_joins_rccl_does_not_hang_the_job():
    """the failure the deadline on communicator creation exists for, with the REAL RCCL: rank 1 never calls ncclCommInitRank (fault injection), so
    rank 0 sits in RCCL's bootstrap waiting for it.  After XC_COMM_TIMEOUT_S its helper thread is given up, every rank learns the verdict, the
    ladder moves on to HIP IPC, the job completes with its JSON line, and rank 0 -- which still owns a thread stuck inside librccl -- leaves
    through os._exit instead of waiting for it"""
    import time
    env = _clean_env()
    env.update({'XC_TEST_SKIP_COMM_INIT_RANK': '1', 'XC_COMM_TIMEOUT_S': '5'})
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2', '--no-cpu',
                        '--cfg4-slabs', '32', '--cfg4-reps', '1', '--deadline-s', '240'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=400, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert time.time() - t0 < 200
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    tr = d['config']['carrier_trials']
    assert 'rank 0: ncclCommInitRank did not return within 5 s' in tr['rccl']['error'] and 'rank 1: comm_init skipped' in tr['rccl']['error']
    assert 'carrier ipc' in d['config']['parallelism'] and d['n_gpus'] == 2 and d['cfg4_strong']['checks']['rank0_block_bit_identical']
