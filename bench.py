#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the contour-coordinate hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic 3600 x 1801 float64 PV-like slabs, 2-D
float64 cell areas, 201 contours, the FULL Keff pipeline per slab (min/max -> levels ->
one histogram pass with in-kernel |grad q|^2 -> CDF -> A(Yeq) lookup -> d/dA -> Leq2 ->
Lmin -> nkeff).  A "step" is one pass of that pipeline over one batch of `--batch`
distinct slabs resident in HBM; two such batches alternate (each larger than the 256 MiB
Infinity Cache, so every step really reads its tracer from HBM, and the batch whose min/max is
folded into a histogram pass is different data).  Metric: lat-lon cells x contours per second,
whole job.  N > 1: every rank owns its own batch of independent slabs (weak scaling), no
data-path collective during compute, ONE RCCL all-gather of all per-slab result vectors at
the end of the timed region.

Steady-state schedule (default, `--chain`): the stack is processed as a software pipeline -- the
histogram pass of step k also streams the batch of step k+1 and leaves its min/max partials
(`xc_keff_desc.q_next`), which step k+1 turns into its levels.  Every step therefore does one
batch of min/max AND one batch of histogram + epilogue (nothing is skipped or reused; results are
bit-identical to the unchained order, tests/test_gpu_parity.py::test_chained_minmax_is_bit_identical);
the stand-alone min/max launch merely disappears.  `--no-chain` runs K1 then K3 per step.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel = the histogram pass, timed
with HIP events on its own stream around every launch of the timed region) and, at N=1,
`cpu_baseline` (the numpy oracle = a port of the reference's xarray/xhistogram call
sequence, timed on this host's cores on a bounded sample of the same slabs).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
NY, NX, NCONT = 1801, 3600, 201
SEED = 20241008
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_CELL = 16              # algorithmic: tracer f64 once + dA f64 once; 8 with --row-dA


# ----------------------------------------------------------------------------- CPU baseline worker
CHECK_NAMES = ('ctr', 'area', 'intgrdS', 'latEq', 'dqdA', 'dintSdA', 'Leq2', 'Lmin', 'nkeff')


def _cpu_keff_worker(args):
    """One slab through the oracle's Keff call sequence (runs in a spawned process)."""
    path, idx, want = args
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import xcontour_oracle as O
    q = np.load(path, mmap_mode='r')[idx]
    lat = np.linspace(-90, 90, NY)
    lon = np.arange(NX) * 0.1
    dA = O.cell_area(lat, lon)
    t = time.perf_counter()
    r = O.keff_pipeline(np.asarray(q), dA, lat, NCONT, lon=lon, increase=True, lt=True, dtype=np.float64)
    dt = time.perf_counter() - t
    if not want:
        return dt, None
    return dt, {k: np.asarray(r[k], dtype=np.float64) for k in CHECK_NAMES + ('counts',)}


def _mem_available_bytes():
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 16 << 30


def _compare_with_oracle(gpu, ref, s):
    """GPU result vectors of slab `s` against the oracle's (outside every timed region); raises on a mismatch.
    Bars: counts and levels bit-exact, float64 sums 1e-11, derived quantities 1e-6 (north_star)."""
    def rel(a, b, floor=1e-300):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        if not np.array_equal(np.isnan(a), np.isnan(b)):
            return np.inf
        m = np.isfinite(b)
        return float(np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), floor))) if m.any() else 0.0
    bad = []
    if not np.array_equal(gpu['counts'][s].astype(np.int64), ref['counts'].astype(np.int64)):
        bad.append('counts')
    if not np.array_equal(gpu['ctr'][s], ref['ctr']):
        bad.append('ctr')
    ok = ref['Lmin'] > 2 * np.pi * 6371200.0 * 1e-6                      # nkeff = Leq2 / Lmin^2 is ill-conditioned AT the pole
    for k, tol in (('area', 1e-11), ('intgrdS', 1e-11), ('latEq', 1e-6), ('dqdA', 1e-6), ('dintSdA', 1e-6), ('Leq2', 1e-6)):
        if not rel(gpu[k][s], ref[k]) < tol:
            bad.append(k)
    if not rel(gpu['Lmin'][s], ref['Lmin'], 2 * np.pi * 6371200.0 * 1e-6) < 1e-6:
        bad.append('Lmin')
    if not rel(gpu['nkeff'][s][ok], ref['nkeff'][ok]) < 1e-6:
        bad.append('nkeff')
    if bad:
        raise RuntimeError('bench parity check against the oracle FAILED for slab %d: %s' % (s, ', '.join(bad)))


def cpu_baseline(q_host, gpu_out, ncheck):
    """Oracle on a bounded sample: single-thread time per slab, then every logical core of the host in parallel
    (bounded by memory: ~1.2 GB of numpy temporaries per worker).  The first `ncheck` slabs' vectors are compared
    with the GPU's (`gpu_out`, same slabs) -- the CPU leg is the checker of the timed GPU result, not only a clock."""
    import multiprocessing as mp
    import shutil
    import tempfile
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, int(_mem_available_bytes() * 0.5 // (1.2 * (1 << 30)))))
    nd = q_host.shape[0]
    # the host's real best: all logical cores, and (this path is memory-bound numpy: more processes than memory channels
    # slow it down) 1/2, 1/4, 1/8, 1/16 of them; the best rate is `value`, every tried count goes into `by_workers`
    tries = sorted({max(1, workers // 16), max(1, workers // 8), max(1, workers // 4), max(1, workers // 2), workers})
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
    tmp = tempfile.mkdtemp(prefix='xc_bench_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    rates = {}
    try:
        path = os.path.join(tmp, 'q.npy')
        np.save(path, q_host)
        t1, r0 = _cpu_keff_worker((path, 0, True))                   # single thread, in-process
        _compare_with_oracle(gpu_out, r0, 0)
        ctx = mp.get_context('spawn')
        with ctx.Pool(workers) as pool:
            pool.map(_cpu_keff_worker, [(path, 0, False)] * workers, chunksize=1)     # warm the workers (imports, page cache)
            for w in tries:                                          # w tasks in flight on w idle workers
                t = time.perf_counter()
                res = pool.map(_cpu_keff_worker, [(path, i % nd, i < ncheck) for i in range(w)], chunksize=1)
                rates[w] = (w, time.perf_counter() - t)
                for i in range(min(ncheck, w)):                      # ten 201-vectors per checked slab: not a timing factor
                    _compare_with_oracle(gpu_out, res[i][1], i % nd)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    best = max(rates, key=lambda w: rates[w][0] / rates[w][1])
    n, wall = rates[best]
    nchecked = min(ncheck, tries[-1], nd)
    work = NY * NX * NCONT
    model, phys = 'unknown CPU', set()
    try:
        pid = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name') and model == 'unknown CPU':
                model = line.split(':', 1)[1].strip()
            elif line.startswith('physical id'):
                pid = line.split(':', 1)[1].strip()
            elif line.startswith('core id'):
                phys.add((pid, line.split(':', 1)[1].strip()))
    except OSError:
        pass
    return {
        'value': n * work / wall, 'unit': 'cells*contours/s', 'cores': best, 'kind': 'port',
        'single_thread_value': work / t1, 'parity_checked_slabs': max(1, nchecked),
        'by_workers': {str(w): rates[w][0] * work / rates[w][1] for w in tries},
        'sample': '%d slabs (%d distinct) of %dx%d f64, %d contours, numpy oracle (port of the reference xarray/'
                  'xhistogram Keff call sequence) in %d concurrent processes (best of %s): %.2f s wall; single thread %.2f s/slab '
                  '= %.3e cells*contours/s; host: %s, %d logical / %d physical cores; %d slabs compared with the GPU vectors '
                  '(counts + levels bit-exact, sums 1e-11, derived 1e-6)'
                  % (n, nd, NX, NY, NCONT, best, tries, wall, t1, work / t1, model, cores, len(phys) or cores, max(1, nchecked)),
    }


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='slabs per step per GPU')
    ap.add_argument('--group', type=int, default=0, help='slabs per launch set (0: whole batch)')
    ap.add_argument('--variant', type=int, default=0, help='0 PV-like, 1 noise, 2 sin(lat)')
    ap.add_argument('--chain', dest='chain', action='store_true', default=True,
                    help="(default) software-pipelined stack processing: this step's histogram pass also streams the "
                         "NEXT step's batch and leaves its min/max partials (xc_keff_desc.q_next), so the stand-alone "
                         'min/max pass disappears: every step still computes one batch of min/max and one batch of '
                         'histogram + epilogue, bit-identical results; the fused kernel streams 24 B/cell against '
                         'the 16 B/cell roofline numerator')
    ap.add_argument('--no-chain', dest='chain', action='store_false',
                    help='stand-alone min/max pass (K1) before every histogram pass (K3)')
    ap.add_argument('--row-dA', action='store_true',
                    help='let the plan detect that the lat-lon dA plane has constant rows and read it as a '
                         'per-row vector (8 B/cell algorithmic instead of 16); off by default: the headline '
                         'keeps the generic 2-D dA read')
    ap.add_argument('--slab-dA', action='store_true',
                    help='per-slab (time-varying) weights: every slab reads ITS OWN 2-D f64 dA plane from HBM (XC_DA_SLAB; '
                         'the reference allows weights with a time dim, core.py:1271-1274).  This is the configuration in '
                         'which the 16 B/cell roofline numerator is exactly the unique HBM traffic')
    ap.add_argument('--native-rccl', action='store_true',
                    help="do the one end-of-job gather with the library's own RCCL communicator (xc_comm_*) on its "
                         'own stream instead of torch.distributed (the id travels through the torch store)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="process-group backend; 'nccl' IS RCCL on ROCm (default).  'gloo' stages the one gather "
                         'through the host: only for exercising the multi-rank path on a box with fewer GPUs than ranks')
    ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg3', 'cfg4', 'cfg5'],
                    help="BASELINE.json configuration: cfg2 (default, the headline metric's), or one of the secondary ones as "
                         'a bench line of the same contract (tools/bench_configs.py; single GPU; --steps / --warmup apply)')
    ap.add_argument('--persistent', action='store_true',
                    help='run the Keff pipeline through the persistent single-read kernel (xc_keffp.hip, XC_KEFF_PERSISTENT): '
                         'the tracer crosses the fabric once; slower than the chained streaming schedule on MI355X (DESIGN.md)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--cpu-slabs', type=int, default=0, help='distinct slabs in the CPU sample, all parity-checked (0: 8)')
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError('bench.py needs an MI355X (no CPU fallback)')
    local = local if local < ndev else 0          # a launcher may expose one device per rank
    torch.cuda.set_device(local)
    if world > 1:
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')
    if a.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (a.gpus, world), file=sys.stderr)

    sys.path.insert(0, ROOT)
    from xcontour_amd import _native as nat
    if not os.path.exists(nat.LIB_PATH):
        # the in-tree library normally travels with the snapshot; if it did not, build it once per node
        # (local rank 0 compiles, the others wait for the file) -- still no fallback: without it nothing runs
        if int(os.environ.get('LOCAL_RANK', '0')) == 0:
            import __graft_entry__
            __graft_entry__.build()
        for _ in range(600):
            if os.path.exists(nat.LIB_PATH):
                break
            time.sleep(0.5)
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included

    ctx = nat.Context(local)
    if a.config != 'cfg2':
        if world > 1:
            raise SystemExit('--config %s is a single-GPU line (the multi-GPU cfg4 driver is tools/bench_cfg4.py)' % a.config)
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import bench_configs
        print(json.dumps(bench_configs.run(a.config, ctx, a.steps, a.warmup)), flush=True)
        ctx.close()
        return
    if a.persistent:
        ctx.set_keff_mode(nat.XC_KEFF_PERSISTENT)
    B, K, W = a.batch, a.steps, a.warmup
    lat = np.linspace(-90, 90, NY)
    lon = np.arange(NX) * 0.1
    dA = cell_area(lat, lon)
    rows = ctx.rowsum(None, dA, NY, NX)                           # K2: A(Yeq) table, once per mask
    tbl = table_from_rowsums(rows, True, last_row_included(lat, 'xhistogram'))   # f64 latitudes: the last row stays in

    # Two resident batches (A, B) of `B` distinct slabs each; steps alternate between them like a
    # time loop over a long record, so the batch whose min/max rides along in a histogram pass
    # (q_next) is genuinely different data.  All K steps keep their per-slab result vectors on
    # the device; one gather at the end.
    NB = 2
    slot = KeffPlan.out_bytes(B, NCONT)
    res = torch.empty(slot * K // 8, dtype=torch.float64, device='cuda')
    wres = torch.empty(slot // 8, dtype=torch.float64, device='cuda')        # warm-up slot
    plan = KeffPlan(ctx, NB * B, NY, NX, NCONT, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl,
                    tbl_coord=lat, increase=True, lt=True, nslots=K, out_ptr=res.data_ptr(), detect_row_dA=a.row_dA,
                    out_slabs=B, replicate_dA=a.slab_dA, right_edge='xhistogram')
    plan.synth(lat, lon, SEED + rank * NB * B, a.variant)         # slab s of rank r: seed + r*2B + s
    group = a.group or B
    chain = bool(a.chain) and not a.persistent          # the persistent kernel finds its min/max itself

    def step(k, slot_idx):
        s0 = (k % NB) * B                                         # this step's batch
        nxt = ((k + 1) % NB) * B                                  # the batch of the next step
        for g0 in range(s0, s0 + B, group):
            n = min(group, s0 + B - g0)
            g1 = g0 + group if g0 + group < s0 + B else nxt       # what runs after this launch set
            plan.run_range(slot_idx, g0, n, g1 if (chain and min(group, B) == n) else None, out_s0=g0 - s0)

    plan.out_ptr = wres.data_ptr()
    for k in range(-W, 0):                                        # ends on batch B; its pass carries batch A's min/max
        step(k, 0)
    plan.out_ptr = res.data_ptr()
    ctx.sync()
    torch.cuda.synchronize()
    ev = [(ctx.event(), ctx.event()) for _ in range(K)]
    if world > 1 and a.native_rccl:
        uid = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world, rank, uid[0])
    gdev = 'cuda' if a.backend == 'nccl' else 'cpu'
    gathered = torch.empty(res.numel() * world, dtype=torch.float64, device=gdev) if world > 1 else None
    if world > 1:
        # warm-up of the collective, like the W warm-up steps of the compute: the first all-gather of a communicator sets up
        # its channels / proxy connections (tens of ms), which is start-up cost and not part of a steady-state job
        if a.native_rccl:
            ctx.comm_allgather(res.data_ptr(), gathered.data_ptr(), res.numel() * 8)
            ctx.sync()
        else:
            dist.all_gather_into_tensor(gathered, res if a.backend == 'nccl' else res.cpu())
        torch.cuda.synchronize()
        res.zero_(); gathered.zero_()                               # the timed region fills them again

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        if group == B:
            ctx.set_hist_events(ev[k][0], ev[k][1])               # events around the K3 launch only
        step(k, k)
    if world > 1 and a.native_rccl:
        ctx.comm_allgather(res.data_ptr(), gathered.data_ptr(), res.numel() * 8)         # the one collective, same stream
    ctx.sync()                                                    # the library's own HIP stream
    if world > 1 and not a.native_rccl:
        dist.all_gather_into_tensor(gathered, res if a.backend == 'nccl' else res.cpu())   # the one collective (RCCL)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    el = t1 - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=gdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        if rank == 0:
            # every rank's block arrived, in rank order, and rank r's slabs differ from rank 0's (seed + r*2B + s)
            g = gathered.view(world, -1).view(torch.int64).cpu()      # bit patterns: results hold NaNs
            mine = res.view(torch.int64).cpu()
            assert torch.equal(g[0], mine) and all(not torch.equal(g[r], g[0]) for r in range(1, world))

    if rank == 0:
        work_step = world * B * NY * NX * NCONT
        line = {
            'metric': 'lat-lon cells*contours/s, full Keff pipeline', 'value': work_step * K / el,
            'unit': 'cells*contours/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': el / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'cfg2: synthetic %dx%d float64 PV-like slabs, 2-D f64 dA, %d contours, '
                                   'full Keff (min/max + histogram with in-kernel |grad q|^2 + CDF + epilogue)'
                                   % (NX, NY, NCONT),
                       'slabs_per_step_per_gpu': B, 'slabs_per_launch': group, 'resident_batches': NB, 'variant': a.variant, 'dA': 'per-row vector (detected constant rows)' if a.row_dA else ('2-D f64 plane PER SLAB (time-varying weights)' if a.slab_dA else '2-D f64 plane shared by the slabs'),
                       'minmax': ('inside the persistent single-read kernel' if a.persistent else
                                  ('folded into the previous histogram pass (q_next)' if chain else 'stand-alone K1 pass')),
                       'parallelism': 'independent slabs per GPU, one RCCL all-gather at the end' if world > 1 else 'single GPU',
                       'device': ctx.device_name()},
        }
        cells = B * NY * NX
        alg = cells * (8 if a.row_dA else BYTES_PER_CELL)              # tracer once + dA once per slab
        # bytes that MUST cross HBM once per launch: this batch's tracer + the weights that are not shared
        # (a dA plane shared by the B slabs of a launch is fetched once; per-slab dA planes B times; a per-row vector ~0)
        uniq = cells * 8 + (cells * 8 if a.slab_dA else (NY * 8 if a.row_dA else NY * NX * 8))
        if group == B:
            ms = np.array([ctx.elapsed_ms(e0, e1) for e0, e1 in ev])
            ach = alg / (ms.mean() * 1e-3) / 1e9
            traffic, tcommit = None, None
            tf = os.path.join(ROOT, 'profiles', 'hist_traffic.json')
            if os.path.exists(tf) and not a.row_dA and a.variant == 0:
                try:
                    tj = json.load(open(tf))
                    tj = tj.get(('slab_' if a.slab_dA else '') + ('persistent' if a.persistent else ('chain' if chain else 'nochain')), {})
                    if tj.get('slabs_per_launch') == B:                # PMC passes were taken at the default batch
                        traffic = tj.get('hbm_bytes_per_launch')
                        tcommit = tj.get('commit')
                except Exception:
                    traffic = None
            line['roofline'] = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                'frac': ach / HBM_PEAK_GBS, 'traffic': traffic,
                                'traffic_source': None if traffic is None else
                                'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_bench_traffic.sh), measured at commit %s; '
                                'not re-measured in this run' % tcommit,
                                'kernel': ('k_keff_persist<double,%s>' if a.persistent else 'k_hist<double,%s,%s>')
                                          % (('DA_SLAB' if a.slab_dA else ('DA_ROW' if a.row_dA else 'DA_PLANE'),) if a.persistent else
                                             ('DA_SLAB' if a.slab_dA else ('DA_ROW' if a.row_dA else 'DA_PLANE'), 'NEXT' if chain else 'plain')),
                                'launch_ms': float(ms.mean()),
                                'algorithmic_bytes_per_launch': alg,
                                'hbm_unique_bytes_per_launch': uniq,
                                'hbm_unique_frac': uniq / (ms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                'streamed_bytes_per_launch': alg + (cells * 8 if chain else 0),
                                'pipeline_frac': (alg * K / el / 1e9) / HBM_PEAK_GBS,
                                'pipeline_hbm_unique_frac': (uniq * K / el / 1e9) / HBM_PEAK_GBS,
                                'note': 'frac = 16 B/cell (SURVEY 8d) / launch time / 8 TB/s; hbm_unique_frac counts only bytes that '
                                        'must come from HBM (a dA plane shared by the %d slabs of a launch counts once); '
                                        'with --slab-dA the two coincide' % B}
        # self-check of the last step: every cell lands in exactly one bin (xhistogram rule: last edge + 1e-8 keeps the max cell)
        out = plan.fetch(slot=K - 1)
        if not (out['counts'].sum(axis=1).astype(np.int64) == NY * NX).all() or out['status'].any():
            raise RuntimeError('bench self-check failed: counts %r status %r' % (out['counts'].sum(axis=1), out['status']))
        if world == 1 and chain and group == B and not a.persistent:
            # transparency: the same work in the plain order (stand-alone K1 launch, then K3), a short extra run
            # AFTER the timed region (identical per-step outputs; tests/test_gpu_parity.py::test_chained_minmax_is_bit_identical)
            chain = False
            K2 = max(5, min(20, K))
            plan.out_ptr = wres.data_ptr()
            for k in range(-3, 0):
                step(k, 0)
            ctx.sync()
            t2 = time.perf_counter()
            for k in range(K2):
                step(k, 0)
            ctx.sync()
            el2 = time.perf_counter() - t2
            chain = True
            plan.out_ptr = res.data_ptr()
            line['unchained'] = {'value': work_step * K2 / el2, 'ms_per_step': el2 / K2 * 1e3, 'steps': K2,
                                 'note': 'stand-alone min/max launch before every histogram launch (--no-chain), same slabs'}
        if world == 1 and not a.no_cpu:
            # the oracle on slabs of the LAST timed step's batch; their vectors are compared with that step's GPU result
            nd = max(1, min(a.cpu_slabs or 8, B))
            s0 = ((K - 1) % NB) * B
            esz = NY * NX * 8
            qh = np.empty((nd, NY, NX), dtype=np.float64)
            ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, qh.ctypes.data, plan._q_ptr + s0 * esz, nd * esz))
            line['cpu_baseline'] = cpu_baseline(qh, out, nd)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
