#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the contour-coordinate hot path on MI355X.  No PyTorch anywhere in this file: device
memory, streams, events and the one collective come from libxcontour_hip.so (ctypes), the rendezvous from
xcontour_amd.distributed.SocketGroup (TCP on MASTER_ADDR / MASTER_PORT + 1).

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N ...            # N > 1 without WORLD_SIZE in the environment: this process only LAUNCHES N fresh
                                            # rank processes (one per GPU, before anything touches the GPU) and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W       # any launcher that exports RANK / WORLD_SIZE / MASTER_* works

Workload (BASELINE.json configs[1]): synthetic 3600 x 1801 float64 PV-like slabs, 2-D
float64 cell areas, 201 contours, the FULL Keff pipeline per slab (min/max -> levels ->
one histogram pass with in-kernel |grad q|^2 -> CDF -> A(Yeq) lookup -> d/dA -> Leq2 ->
Lmin -> nkeff).  A "step" is one pass of that pipeline over one batch of `--batch`
distinct slabs resident in HBM; two such batches alternate (each larger than the 256 MiB
Infinity Cache, so every step really reads its tracer from HBM, and the batch whose min/max is
folded into a histogram pass is different data).  Metric: lat-lon cells x contours per second,
whole job.  N > 1: every rank owns its own batch of independent slabs (weak scaling), no
data-path collective during compute, ONE gather of all per-slab result vectors to rank 0 at the end of the timed region
(SURVEY 8e).  Carriers of that gather, in the order they are tried (class Gather): RCCL over xGMI (grouped ncclSend / ncclRecv
through the library's own communicator, xc_comm_*), HIP IPC pushes into the root's receive buffer (no RCCL; works with several
ranks on ONE GPU too: the rehearsal this file's tests run), the rendezvous sockets (host; last resort).  Every first contact
-- ncclCommInitRank, the first transfers -- runs under a deadline; the launcher has one for the whole job (`--deadline-s`)
and reports every rank's last stage when it expires.  A < 10 s preflight goes to stderr before any timed work.

Steady-state schedule (default, `--chain`): the stack is processed as a software pipeline -- the
histogram pass of step k also streams the batch of step k+1 and leaves its min/max partials
(`xc_keff_desc.q_next`), which step k+1 turns into its levels.  Every step therefore does one
batch of min/max AND one batch of histogram + epilogue (nothing is skipped or reused; results are
bit-identical to the unchained order, tests/test_gpu_parity.py::test_chained_minmax_is_bit_identical);
the stand-alone min/max launch merely disappears.  `--no-chain` runs K1 then K3 per step.

After the timed region rank 0's line also carries (all outside the timed region, all parity-checked):
  * `variants.slab_dA` -- the same chained schedule with per-slab (time-varying) weights: every byte of the 16 B/cell
    numerator is then unique HBM traffic, so `frac == hbm_unique_frac` is a genuine HBM fraction;
  * `variants.f32` -- the same schedule on float32 tracers and float32 contours (the dtype of every file the reference ships
    and its default `dtype`, core.py:21), two of its slabs checked against the oracle;
  * `long_run` -- >= 0.5 s of the same steps with HIP events around every histogram launch (mean / spread);
  * `cfg4_strong` -- BASELINE.json configs[3]: 18 944 slabs of 1440 x 721 float64 partitioned contiguously over the
    ranks (strong scaling: total work fixed), Keff per slab, results written slab-major so that the rank's block IS the
    send buffer, ONE gather of all nine result vectors inside its own timed region (barrier + sync on both sides, max over
    ranks) and a `budget` of where each rank's job time goes.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel = the histogram pass, timed
with HIP events on its own stream around every launch of the timed region) and
`cpu_baseline` (the numpy oracle = a port of the reference's xarray/xhistogram call
sequence, timed on this host's cores by rank 0 on a bounded sample of the same slabs).
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
    path, idx, want = args[:3]
    cdt = np.dtype(args[3]) if len(args) > 3 else np.float64
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import xcontour_oracle as O
    q = np.load(path, mmap_mode='r')[idx]
    lat = np.linspace(-90, 90, NY)
    lon = np.arange(NX) * 0.1
    dA = O.cell_area(lat, lon)
    t = time.perf_counter()
    r = O.keff_pipeline(np.asarray(q), dA, lat, NCONT, lon=lon, increase=True, lt=True, dtype=cdt)
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


def cpu_baseline(q_host, gpu_out, ncheck, cdt='float64', max_workers=0):
    """Oracle on a bounded sample: single-thread time per slab, then every logical core of the host in parallel
    (bounded by memory: ~1.2 GB of numpy temporaries per worker).  The first `ncheck` slabs' vectors are compared
    with the GPU's (`gpu_out`, same slabs) -- the CPU leg is the checker of the timed GPU result, not only a clock."""
    import multiprocessing as mp
    import shutil
    import tempfile
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, int(_mem_available_bytes() * 0.5 // (1.2 * (1 << 30)))))
    if max_workers > 0:
        workers = min(workers, int(max_workers))                     # (--cpu-workers: tests bound the leg's time)
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
        t1, r0 = _cpu_keff_worker((path, 0, True, cdt))              # single thread, in-process
        _compare_with_oracle(gpu_out, r0, 0)
        ctx = mp.get_context('spawn')
        with ctx.Pool(workers) as pool:
            pool.map(_cpu_keff_worker, [(path, 0, False, cdt)] * workers, chunksize=1)     # warm the workers (imports, page cache)
            for w in tries:                                          # w tasks in flight on w idle workers
                t = time.perf_counter()
                res = pool.map(_cpu_keff_worker, [(path, i % nd, i < ncheck, cdt) for i in range(w)], chunksize=1)
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
        'sample': '%d slabs (%d distinct) of %dx%d %s, %d contours, numpy oracle (port of the reference xarray/'
                  'xhistogram Keff call sequence) in %d concurrent processes (best of %s): %.2f s wall; single thread %.2f s/slab '
                  '= %.3e cells*contours/s; host: %s, %d logical / %d physical cores; %d slabs compared with the GPU vectors '
                  '(counts + levels bit-exact, sums 1e-11, derived 1e-6)'
                  % (n, nd, NX, NY, q_host.dtype.name, NCONT, best, tries, wall, t1, work / t1, model, cores, len(phys) or cores, max(1, nchecked)),
    }


# ----------------------------------------------------------------------------- helpers
def hist_source_sha():
    """sha256 over the sources of the dominant kernel: a stored PMC traffic figure is quoted only while they are unchanged"""
    import hashlib
    h = hashlib.sha256()
    for f in ('xc_hist.hip', 'xc_hist_kernel.h', 'xc_binning.h'):
        h.update(open(os.path.join(ROOT, 'xcontour_amd', 'csrc', f), 'rb').read())
    return h.hexdigest()


def stored_traffic(key, B):
    """fabric bytes per launch of the dominant kernel from profiles/hist_traffic.json (rocprofv3 --pmc, separate passes:
    tools/pmc_bench_traffic.sh) -- or None when the file is missing, was taken at another batch size, or was taken
    with OTHER kernel sources than the ones in this tree (sha256 of xc_hist.hip + xc_hist_kernel.h + xc_binning.h)."""
    tf = os.path.join(ROOT, 'profiles', 'hist_traffic.json')
    try:
        tj = json.load(open(tf))
        if tj.get('source_sha256') != hist_source_sha():
            return None, 'profiles/hist_traffic.json was measured on other kernel sources (sha256 mismatch): not quoted'
        e = tj.get(key, {})
        if e.get('slabs_per_launch') != B or e.get('hbm_bytes_per_launch') is None:
            return None, 'no PMC entry for this schedule / batch size'
        when, age = tj.get('measured_at_utc'), None
        if when:
            import datetime
            try:
                age = (datetime.datetime.utcnow() - datetime.datetime.strptime(when, '%Y-%m-%dT%H:%M:%SZ')).total_seconds() / 86400.0
            except ValueError:
                age = None
        return e['hbm_bytes_per_launch'], ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_bench_traffic.sh) at commit %s, '
                                           'same kernel sources as this tree (sha256 %s); the record: taken %s (%s days before this run) on %s; '
                                           'a PMC pass cannot run inside a timed bench (the counters serialise the kernels): not re-measured in this run'
                                           % (e.get('commit', tj.get('commit')), tj['source_sha256'][:16], when or 'at an unrecorded time',
                                              ('%.2f' % age) if age is not None else '?', tj.get('device', 'an unrecorded box')))
    except Exception as ex:            # noqa: BLE001 -- a missing / malformed file only nulls the optional figure
        return None, 'profiles/hist_traffic.json unreadable: %s' % ex


def stream_ceilings():
    """what pure streams reach on this part (tools/probe/bw_probe.hip, stored in profiles/): context for `frac`, not its denominator"""
    try:
        rows = json.load(open(os.path.join(ROOT, 'profiles', 'r03_bw_probe.json')))['rows']
        best = lambda pred: max((r['TBps'] for r in rows if pred(r['kernel'])), default=None)
        return {'read_plain_TBps': best(lambda k: k in ('stride_u4', 'stride_u8', 'chunk_u8')), 'read_nontemporal_TBps': best(lambda k: k == 'stride_u8_nt'),
                'copy_TBps': best(lambda k: k.startswith('copy')), 'source': 'profiles/r03_bw_probe.json (tools/probe/bw_probe.hip; best of each family, not re-measured in this run)'}
    except (OSError, ValueError, KeyError):
        return None


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return np.inf
    m = np.isfinite(b) & (b != 0)
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))) if m.any() else 0.0


# ----------------------------------------------------------------------------- launcher + the ranks' process group
# HSA_ENABLE_IPC_MODE_LEGACY=0: on this driver stack (ROCm 7.2 user space over a host kernel driver that only implements dmabuf
# IPC) the ROCr default, the legacy KFD IPC ioctls, is refused: hipIpcGetMemHandle fails with "invalid argument", and with it
# everything that shares device memory between processes -- RCCL's intra-node P2P transport and this file's own HIP IPC carrier.
# The image exports the variable already; the launcher and every rank set it if it is missing (never overriding a value the
# environment chose), because a rank started by some other launcher with a scrubbed environment would otherwise lose both device
# carriers and fall to the host carrier.  It has no effect on single-process work.
IPC_ENV = ('HSA_ENABLE_IPC_MODE_LEGACY', '0')
STAGE_ENV = 'XC_BENCH_STAGE_DIR'


def stage(word):
    """one word saying where this rank is, into $XC_BENCH_STAGE_DIR/rank<r> (set by the launcher): what the launcher reports for
    every rank when its deadline expires.  Nothing happens without the variable."""
    d = os.environ.get(STAGE_ENV)
    if not d:
        return
    try:
        with open(os.path.join(d, 'rank%s' % os.environ.get('RANK', '0')), 'w') as f:
            f.write('%s %.1f\n' % (word, time.time()))
    except OSError:
        pass


def launch_ranks(n, argv, deadline_s=480.0, program=None):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N fresh rank processes (one per GPU), each a
    new interpreter running this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, and return the
    worst exit code.  This process imports nothing that can touch the GPU and never execs: the ranks are children.
    Rank 0's JSON line goes to this process's stdout (inherited); a rank that dies takes the others down with it.
    `deadline_s`: the whole job's time limit.  When it expires the children still alive are terminated BY PID, the launcher says
    on stderr which ranks were alive and the last stage each rank reported (`stage()`), and returns 124 -- a job that hangs in a
    first contact with a collective ends with a diagnosis inside the driver's own limit instead of with its kill."""
    import secrets
    import shutil
    import socket
    import subprocess
    import tempfile
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:      # a free port pair: P for a launcher's store (unused here), P + 1 for SocketGroup
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    port = port - 1 if port > 1024 else port
    sdir = tempfile.mkdtemp(prefix='xc_bench_stage_')
    env = dict(os.environ)
    env.update({'WORLD_SIZE': str(n), 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'XC_DIST_TOKEN': secrets.token_hex(16),
                IPC_ENV[0]: env.get(IPC_ENV[0], IPC_ENV[1]), 'XC_BENCH_LAUNCHED': '1', STAGE_ENV: sdir})
    procs = []
    for r in range(n):
        e = dict(env)
        e.update({'RANK': str(r), 'LOCAL_RANK': str(r)})
        procs.append(subprocess.Popen([sys.executable, program or os.path.abspath(__file__)] + list(argv), env=e))
    t_end = time.time() + float(deadline_s)
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            c = p.poll()
            if c is None:
                continue
            alive.remove(p)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                for q in alive:                                        # exactly the processes started above, by PID
                    q.terminate()
        if alive and time.time() > t_end:
            stages = {}
            for r in range(n):
                try:
                    w = open(os.path.join(sdir, 'rank%d' % r)).read().split()
                    stages[r] = '%s (%.0f s ago)' % (w[0], time.time() - float(w[1]))
                except (OSError, IndexError, ValueError):
                    stages[r] = 'no stage reported'
            live = [procs.index(p) for p in alive]
            print('bench.py launcher: deadline of %.0f s expired; ranks still alive: %s; last stage per rank: %s -- terminating them'
                  % (deadline_s, live, ', '.join('rank %d: %s' % (r, stages[r]) for r in range(n))), file=sys.stderr, flush=True)
            for q in alive:
                q.terminate()
            rc = 124
            break
    for p in procs:
        try:
            p.wait(timeout=15)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    shutil.rmtree(sdir, ignore_errors=True)
    return rc


class RecvBlock(object):
    """where the root collects the ranks' blocks: `stride` bytes per rank.  Device carriers: a device buffer on the root (`buf`),
    `dst` = its address as THIS rank sees it (the root: its own pointer; HIP IPC: the mapping opened from the root's handle; RCCL:
    nothing, the library addresses the root by rank).  Host carrier: `host` on the root after finish()."""

    def __init__(self, stride):
        self.stride, self.buf, self.dst, self.opened, self.host, self.pending = int(stride), None, None, None, None, []


class Gather(object):
    """The one collective of a job: every rank's result block to rank 0 (north_star: "RCCL gather"; SURVEY 8e: "all-gather or
    gather-to-root").  A rank PUSHES pieces of its block as they become final -- a launch set's results leave on the context's
    comm stream, behind an event, while the next launch set computes -- and `finish()` ends the job: only the last piece is exposed.
    Carriers, tried in this order (`--backend nccl`, the default) until one passes its trial on EVERY rank:
      'rccl'  grouped ncclSend / ncclRecv over xGMI through the library's own communicator (xc_comm_gather_dev);
      'ipc'   HIP IPC: the root's receive buffer mapped into every rank, device-to-device pushes (peer writes over xGMI between
              GPUs; plain copies when ranks share one GPU in a rehearsal) -- needs no RCCL;
      'host'  device -> host, the rendezvous sockets to rank 0 (TCP; rehearsal only: ~0.5 s per 274 MB).
    A trial = communicator / mapping set-up under a deadline + a 1 MB and a 32 MB gather whose bytes the root checks, each with a
    deadline on the stream wait (xc_streams_idle polling: never a blocking sync on a collective that has not proved itself).
    `self.trials` records what was tried, how long it took and why a carrier was passed over; it goes into the JSON line."""

    ORDER = {'nccl': ('rccl', 'ipc', 'host'), 'rccl': ('rccl', 'ipc', 'host'), 'ipc': ('ipc', 'host'), 'gloo': ('host',), 'host': ('host',),
             'auto': ('rccl', 'ipc', 'host')}

    def __init__(self, ctx, group, backend, timeout=None):
        self.ctx, self.group, self.note, self.trials = ctx, group, None, {}
        self.timeout = float(os.environ.get('XC_COMM_TIMEOUT_S', 60)) if timeout is None else float(timeout)
        self.carrier = 'none'
        if group.world == 1:
            return
        passed, notes = [], []
        for c in self.ORDER[backend]:
            if c == 'host' and passed:
                break                                                   # ('auto': the host carrier only when no device carrier works)
            ok, why = self._try(c)
            if ok:
                passed.append(c)
                if backend != 'auto':
                    break
            else:
                notes.append('%s unavailable (%s)' % (c, why))
        if not passed:
            raise RuntimeError('bench.py: no carrier for the gather works: ' + '; '.join(notes))
        if backend == 'auto' and len(passed) > 1:                     # the fastest 32 MB trial wins (rank 0's clock, same answer everywhere)
            best = min(passed, key=lambda c: self.trials[c].get('ms_32MB', 1e9))
            passed = [group.broadcast_bytes(best.encode() if group.rank == 0 else b'').decode()]
        self.carrier = passed[0]
        if self.carrier != 'rccl' and not group.stuck:
            try:
                ctx.comm_finalize()                                     # (a no-op without a communicator)
            except Exception:                                           # noqa: BLE001
                pass
        if notes:
            self.note = '; '.join(notes) + ' -> gathered over ' + self.carrier
            if group.rank == 0:
                print('warning: ' + self.note, file=sys.stderr)

    # -- set-up + trial of one carrier; the verdict is the same on every rank
    def _try(self, c):
        g, ctx, t = self.group, self.ctx, {}
        self.trials[c] = t
        t0 = time.perf_counter()
        try:
            if c == 'rccl':
                g.init_device(ctx, timeout=self.timeout)               # consensus inside: raises on every rank or on none
            t['setup_ms'] = (time.perf_counter() - t0) * 1e3
        except Exception as e:                                          # noqa: BLE001
            t['error'] = str(e)
            return False, str(e)
        self.carrier = c
        why = ''
        try:
            for mb in (1, 32):
                ms, ok = self._trial(int(mb) << 20)
                if not ok:
                    why = 'the %d MB trial gather %s' % (mb, 'did not finish within %.0f s' % self.timeout if ms is None else 'delivered wrong bytes')
                    break
                t['ms_%dMB' % mb] = ms
        except Exception as e:                                          # noqa: BLE001 -- this rank's failure; the others learn it below
            why = str(e)
        bad = [p.decode('utf-8', 'replace') for p in g.allgather_bytes(why.encode())]
        self.carrier = 'none'
        if any(bad):
            why = '; '.join('rank %d: %s' % (r, w) for r, w in enumerate(bad) if w)
            t['error'] = why
            if c == 'rccl':
                try:
                    ctx.comm_abort()                                    # never wait for a collective that did not finish
                except Exception:                                       # noqa: BLE001
                    pass
            return False, why
        return True, ''

    def _trial(self, nbytes):
        """one gather of `nbytes` per rank with a known pattern; the root compares every byte.  Returns (ms or None on a timeout, ok)."""
        g, ctx = self.group, self.ctx
        pat = (np.arange(nbytes // 8, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(g.rank + 1)).view(np.uint8)
        send = ctx.to_device(pat)
        rb = self.recv_block(nbytes)
        if g.rank == 0 and rb.buf is not None:
            ctx._check(ctx.lib.xc_memset(ctx.handle, rb.buf.ptr, 0xA5, g.world * nbytes))     # stale lines in the root's caches on purpose
        ctx.sync()
        g.barrier()
        t0 = time.perf_counter()
        self.push(rb, send.ptr, nbytes)
        done = self.finish(rb, deadline=self.timeout)
        ms = (time.perf_counter() - t0) * 1e3
        ok = True
        if done and g.rank == 0:
            got = self.fetch(rb)
            for r in range(g.world):
                e = (np.arange(nbytes // 8, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(r + 1)).view(np.uint8)
                ok = ok and bool(np.array_equal(got[r], e))
        if done:
            self.free_block(rb)
            send.free()
        return (ms if done else None), (ok and done)

    def describe(self):
        return {'none': 'single rank (no collective)',
                'rccl': 'gather to rank 0 over xGMI: grouped ncclSend / ncclRecv (library communicator xc_comm_*, RCCL) on a comm stream, one piece per launch set',
                'ipc': 'gather to rank 0 by HIP IPC: the root\'s receive buffer mapped into every rank, device-to-device pushes on a comm stream, one piece per launch set',
                'host': 'device -> host, TCP gather to rank 0 (xcontour_amd.distributed.SocketGroup), host result'}[self.carrier]

    # -- the job's interface
    def recv_block(self, stride):
        """collective: the root's landing area for `stride` bytes per rank"""
        rb = RecvBlock(stride)
        g, ctx = self.group, self.ctx
        if self.carrier in ('rccl', 'ipc') and g.rank == 0:
            rb.buf = ctx.alloc(g.world * rb.stride)
            rb.dst = rb.buf.ptr
        if self.carrier == 'ipc':
            h = g.broadcast_bytes(ctx.ipc_export(rb.buf.ptr) if g.rank == 0 else b'')
            if g.rank != 0:
                rb.opened = ctx.ipc_open(h)
                rb.dst = rb.opened
        return rb

    def push(self, rb, src_ptr, nbytes, offset=0):
        """enqueue: bytes [src_ptr, + nbytes) of this rank -> the root's block at rank * stride + offset, behind everything enqueued
        on the compute stream so far; returns at once.  Every rank pushes the same sizes in the same order."""
        ctx, r = self.ctx, self.group.rank
        if self.carrier == 'none' or nbytes <= 0:
            return
        if self.carrier == 'host':
            rb.pending.append((int(src_ptr), int(nbytes), int(offset)))
            return
        ctx.comm_wait_compute()
        if self.carrier == 'rccl':
            ctx.comm_gather(src_ptr, nbytes, (rb.dst + offset) if r == 0 else None, rb.stride, 0)
        else:
            ctx.comm_memcpy_d2d(rb.dst + r * rb.stride + offset, src_ptr, nbytes)

    def finish(self, rb, deadline=None):
        """the end of the job on this rank: its pushes have landed (and, on the root, everybody's).  `deadline`: give up after that many
        seconds instead of blocking (trials); returns False then."""
        ctx, g = self.ctx, self.group
        if self.carrier == 'none':
            ctx.sync()
            return True
        if self.carrier == 'host':
            ctx.sync()
            mine = np.zeros(rb.stride, dtype=np.uint8)
            for src, n, off in rb.pending:
                ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, mine[off:].ctypes.data, src, n))
            del rb.pending[:]
            parts = g.gather_bytes(mine.tobytes())
            if g.rank == 0:
                rb.host = np.stack([np.frombuffer(p, dtype=np.uint8) for p in parts])
            return True
        ctx.compute_wait_comm()
        if deadline is None:
            ctx.sync()
            if self.carrier == 'ipc':
                g.barrier()                                            # the root learns that every rank's pushes have landed
            return True
        # a trial: this rank's verdict goes to EVERY rank in ONE exchange that every rank reaches whatever its own outcome (round-5
        # advisor: a rank that timed out used to skip the barrier below and the collective free_block, and its next frame -- the
        # 'why' string -- met the others' barrier frames: garbage verdicts or a hang until the socket timeout).  The exchange is also
        # the IPC carrier's "every push has landed" barrier.  The same answer on all ranks: all go on, or all take the abort path.
        mine = ctx.sync_within(deadline)
        return all(p == b'1' for p in g.allgather_bytes(b'1' if mine else b'0'))

    def fetch(self, rb):
        """root: the gathered bytes on the host, (world, stride) uint8 (outside every timed region)"""
        if rb.host is not None:
            return rb.host
        return rb.buf.download((self.group.world, rb.stride), np.uint8)

    def free_block(self, rb):
        """collective"""
        if rb.opened is not None:
            self.ctx.ipc_close(rb.opened)
            rb.opened = None
        if self.carrier == 'ipc':
            self.group.barrier()                                        # nobody still maps the buffer the root is about to free
        if rb.buf is not None:
            rb.buf.free()
            rb.buf = None


def preflight(ctx, nat, group, dev, nd):
    """< 10 s, before any timed work, to stderr (rank 0): what this node offers the N > 1 path -- devices, librccl, the peer-access
    matrix row of every rank's device.  (The carriers' trial gathers, with their times, follow in Gather.)"""
    import ctypes as C
    info = {'rank': group.rank, 'devices_visible': nd, 'device': dev, 'name': ctx.device_name()}
    try:
        C.CDLL('librccl.so.1')
        info['librccl'] = True
    except OSError as e:
        info['librccl'] = str(e)
    row = []
    for j in range(nd):
        v = C.c_int(0)
        nat.load().xc_device_can_access_peer(dev, j, C.byref(v))
        row.append(int(v.value))
    info['peer_access'] = row
    info[IPC_ENV[0]] = os.environ.get(IPC_ENV[0])
    allinfo = [json.loads(p.decode()) for p in group.allgather_bytes(json.dumps(info).encode())]
    if group.rank == 0:
        print('[preflight] world %d, %d device(s) visible to rank 0%s' % (group.world, nd, '' if nd >= group.world else
              ' -- FEWER than ranks: ranks share GPUs (a rehearsal; RCCL refuses that, the gather will use HIP IPC)'), file=sys.stderr)
        for i in allinfo:
            print('[preflight] rank %d -> device %d (%s), librccl %s, peer access %s, %s=%s'
                  % (i['rank'], i['device'], i['name'], 'loadable' if i['librccl'] is True else 'NOT loadable: %s' % i['librccl'],
                     i['peer_access'], IPC_ENV[0], i[IPC_ENV[0]]), file=sys.stderr)
        sys.stderr.flush()
    return allinfo


# ----------------------------------------------------------------------------- cfg4: strong scaling over the ranks
def cfg4_launch_set(n, cus=256, blocks_per_slab=3):
    """Slabs per launch set for a rank's block of `n` cfg4 slabs.  A set that TILES the block keeps every set chained (a ragged
    last set and the set after it run a stand-alone min/max pass: 2 368 slabs per rank, the N = 8 block, 7.67 ms with sets of 256
    + 64 against 7.32 ms with 4 x 592); the histogram grid is three workgroups per slab on 256 CUs, so a set should fill whole
    rounds (296 slabs = 3.47 rounds: 8.12 ms); larger sets amortise the per-launch epilogue (N = 1: 512 -> 56.1 ms, 592 -> 56.5,
    256 -> 57.5).  Among the divisors of n up to 640: best round efficiency minus a small bonus for size; no good divisor: 256, ragged."""
    best, score = None, 0.0
    for d in range(128, 641):
        if n % d:
            continue
        rounds = blocks_per_slab * d / float(cus)
        sc = rounds / np.ceil(rounds) - 0.02 * 256.0 / d
        if sc > score:
            best, score = d, sc
    return best if best is not None and score >= 0.96 else 256


def cfg4_strong(ctx, nat, a, group, gather):
    """BASELINE.json configs[3]: `--cfg4-slabs` (18 944 = 512 x 37) slabs of 1440 x 721 float64, Keff per slab with per-slab
    levels; the flattened (time, level) index is cut into contiguous blocks of ceil(S/G) slabs (pipeline.shard_slabs), every
    rank sweeps its block in chained launch sets of `--cfg4-chunk` slabs whose results land slab-major -- [slab][9][N],
    xc_keff_desc.out_stride -- straight in the rank's send block, and ONE all-gather of the nine per-slab result vectors ends
    the job.  A job = sweep + gather; `--cfg4-reps` jobs are timed between barriers, max over ranks."""
    from xcontour_amd.pipeline import KeffPlan, shard_slabs, OUT_NAMES
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included
    import ctypes as C
    world, rank = group.world, group.rank
    NY4, NX4 = 721, 1440
    S, R = int(a.cfg4_slabs), max(1, int(a.cfg4_reps))
    lat = np.linspace(-90, 90, NY4)
    lon = np.arange(NX4) * 0.25
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, NY4, NX4), True, last_row_included(lat, 'xhistogram'))
    per = -(-S // world)
    lo, hi = shard_slabs(S, rank, world)
    n = hi - lo
    # launch sets are cut from the PADDED block (per slabs on every rank, a short last block ends in zero rows): every rank pushes
    # the same piece sizes in the same order, whatever it owns
    Cn = min(int(a.cfg4_chunk) if a.cfg4_chunk > 0 else cfg4_launch_set(per), max(per, 1))
    nchunk = -(-per // Cn)
    slab_bytes = NY4 * NX4 * 8
    qbuf, err = None, ''
    try:
        qbuf = ctx.alloc(max(n, 1) * slab_bytes)                 # this rank's block of the stack, resident in HBM
    except nat.XContourHipError as e:
        err = str(e)
    if group.allreduce_min(0 if qbuf is None else 1) < 1:        # every rank skips if one could not hold its block
        if qbuf is not None:
            qbuf.free()
        return {'skipped': 'a rank could not allocate its %d-slab block (%.1f GB): %s' % (n, n * slab_bytes / 1e9, err)}
    lat_b, lon_b = ctx.to_device(lat), ctx.to_device(lon)
    for c0 in range(0, n, Cn):                                   # slab s of the stack: seed + s, whatever rank owns it
        m = min(Cn, n - c0)
        ctx._check(ctx.lib.xc_synth_dev(ctx.handle, qbuf.ptr + c0 * slab_bytes, nat.XC_F64, m, NY4, NX4,
                                        lat_b.ptr, lon_b.ptr, SEED + lo + c0, 0))
    ctx.sync()
    # ONE result slot of `per` slabs, slab-major: its head is the rank's (per, 9, N) block (short blocks: zero padding)
    plan = KeffPlan(ctx, Cn, NY4, NX4, NCONT, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, nslots=1, out_slabs=per, alloc_q=False, right_edge='xhistogram', slab_major=True,
                    deterministic=a.deterministic, single_read=False)   # (a stack job: a ragged last launch set of ONE slab stays on the chain -- nothing
                                                                      # here looks at status 2, and ranks that share a GPU in a rehearsal could not all hold it)
    block_bytes = plan.head_bytes                                # per * 9 * N * 8
    ctx._check(ctx.lib.xc_memset(ctx.handle, plan.out_ptr, 0, plan.slot_bytes))
    rb = gather.recv_block(block_bytes)
    sb = 9 * NCONT * 8                                           # result bytes per slab
    sets = [(c0, min(Cn, n - c0), min(Cn, per - c0)) for c0 in range(0, per, Cn)]     # (first slab, slabs computed here, slabs pushed)

    def sweep(push=True):
        for ci, (c0, m, mp) in enumerate(sets):
            if m > 0:
                plan.set_q_device(qbuf.ptr + c0 * slab_bytes)
                nc0, nm, _ = sets[(ci + 1) % nchunk]
                plan._point(0, 0, m, out_s0=c0)                    # results of slab c0 + i -> block[c0 + i][:][:]
                plan.desc.q_next = (qbuf.ptr + nc0 * slab_bytes) if nm == m else None   # equal-shape launch sets chain their min/max
                ctx._check(ctx.lib.xc_keff_dev(ctx.handle, C.byref(plan.desc)))
            if push:                                               # this set's rows leave on the comm stream while the next set computes
                gather.push(rb, plan.out_ptr + c0 * sb, mp * sb, c0 * sb)

    def job():
        sweep()
        gather.finish(rb)                                          # only the last set's piece is exposed

    stage('cfg4_warmup')
    job()                                                        # warm-up: kernels, scratch growth, the carrier's channels
    group.barrier()
    ctx.sync()
    stage('cfg4_timed')
    t0 = time.perf_counter()
    for _ in range(R):
        job()
    group.barrier()
    el = group.allreduce_max(time.perf_counter() - t0)
    # where a job's time goes: one extra job per rank, device events around the sweep and behind the gather's last piece (no sync
    # between them), host clock around the whole; every rank reports, rank 0 prints the per-rank lists and the worst of each
    stage('cfg4_budget')
    e0, e1, e2 = ctx.event(), ctx.event(), ctx.event()
    group.barrier()
    ts = time.perf_counter()
    ctx.record(e0); sweep(); ctx.record(e1)
    tg = time.perf_counter()
    if gather.carrier in ('rccl', 'ipc'):
        ctx.compute_wait_comm()                                  # (what finish() does first: e2 then marks the last piece's arrival / departure)
    ctx.record(e2)
    gather.finish(rb)
    t_end = time.perf_counter()
    sweep_ms = ctx.elapsed_ms(e0, e1)
    gather_ms = (t_end - tg) * 1e3 if gather.carrier == 'host' else ctx.elapsed_ms(e1, e2)
    st = np.array([sweep_ms, gather_ms, (t_end - ts) * 1e3, (t_end - tg) * 1e3])
    stages = group.allgather(st)                                 # (world, 4)
    block = None
    if rank == 0:
        if world == 1:
            full = np.empty(plan.head_bytes, dtype=np.uint8)
            ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, full.ctypes.data, plan.out_ptr, plan.head_bytes))
            full = full[None]
        else:
            full = gather.fetch(rb)
        full = full.view(np.float64).reshape(world * per, 9, NCONT)[:S]
        mine = plan.fetch(check=False)
        mine_blk = np.stack([mine[k] for k in OUT_NAMES], axis=1)[:n]
        if a.dump_cfg4:
            np.save(a.dump_cfg4, full)                               # (tests: an N-rank job against the 1-rank job, bit for bit with --deterministic)
        fb = full.view(np.int64)                                   # bit patterns: the vectors hold NaNs
        assert np.array_equal(fb[lo:hi], mine_blk.view(np.int64)), 'rank 0 block is not where it belongs'
        # every other rank's block sits at its place with that rank's data: recompute the FIRST slab of each block here
        checked = []
        one = KeffPlan(ctx, 1, NY4, NX4, NCONT, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                       increase=True, lt=True, right_edge='xhistogram', deterministic=a.deterministic)
        for r in range(world):
            rlo, rhi = shard_slabs(S, r, world)
            if rhi <= rlo:
                continue
            one.synth(lat, lon, SEED + rlo, 0)
            one.run()
            o = one.fetch()
            if not np.array_equal(o['ctr'][0], full[rlo, OUT_NAMES.index('ctr')]):
                raise RuntimeError('cfg4_strong: slab %d (first of rank %d) does not carry that slab\'s levels' % (rlo, r))
            for k in ('area', 'intgrdS', 'latEq'):
                e = rel_err(full[rlo, OUT_NAMES.index(k)], o[k][0])
                if not e < 1e-11:
                    raise RuntimeError('cfg4_strong: slab %d (first of rank %d): %s differs by %g' % (rlo, r, k, e))
            checked.append(rlo)
        oracle_checked = 0
        if not a.no_cpu:
            sys.path.insert(0, os.path.join(ROOT, 'oracle'))
            import xcontour_oracle as O
            qh = np.empty((2, NY4, NX4))
            ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, qh.ctypes.data, qbuf.ptr, min(2, n) * slab_bytes))
            for s_ in range(min(2, n)):
                rr = O.keff_pipeline(qh[s_], dA, lat, NCONT, lon=lon, increase=True, lt=True, dtype=np.float64)
                gpu = {k: full[lo:lo + 2, OUT_NAMES.index(k)] for k in OUT_NAMES}
                gpu['counts'] = mine['counts'][:2]
                _compare_with_oracle(gpu, rr, s_)
                oracle_checked += 1
        one.free()
        nk = full[:, OUT_NAMES.index('nkeff'), :]
        work = S * NY4 * NX4 * NCONT
        block = {
            'metric': 'lat-lon cells*contours/s, full Keff pipeline, cfg4 stack (sweep + one gather)', 'value': work * R / el,
            'unit': 'cells*contours/s', 'n_gpus': world, 'scaling': 'strong', 'jobs_timed': R, 'ms_per_job': el / R * 1e3,
            'slabs': S, 'slab_shape': [NY4, NX4], 'slabs_per_gpu': per, 'slabs_per_launch': Cn,
            'us_per_slab_per_gpu': el / R / max(1, per) * 1e6,
            'budget': {'sweep_ms_by_rank': [float(x) for x in stages[:, 0]], 'gather_ms_by_rank': [float(x) for x in stages[:, 1]],
                       'job_ms_by_rank': [float(x) for x in stages[:, 2]], 'finish_host_ms_by_rank': [float(x) for x in stages[:, 3]], 'pack_ms': 0.0,
                       'sweep_ms_max': float(stages[:, 0].max()), 'gather_ms_max': float(stages[:, 1].max()),
                       'note': 'one extra job after the timed ones.  sweep_ms: HIP events around the launch sets.  gather_ms: what the gather ADDS behind the '
                               'sweep\'s last kernel -- every launch set\'s rows leave on the comm stream while the next set computes, so this is the last '
                               'piece (device carriers: event behind the comm stream\'s last piece minus the sweep\'s end; host carrier: host clock around '
                               'the whole staged gather).  finish_host_ms: host clock from the last enqueue to the end of finish() (stream wait + the one '
                               'rendezvous barrier of the HIP IPC carrier).  No pack stage: the launch sets write [slab][9][N] straight into the send block'},
            'gathered_bytes': int(world * block_bytes), 'gather_to': 'rank 0', 'pieces_per_job': len(sets),
            'gather': gather.describe(), 'gather_note': gather.note,
            'algorithmic_bytes': int(S * NY4 * NX4 * BYTES_PER_CELL), 'pipeline_frac': (S * NY4 * NX4 * BYTES_PER_CELL * R / el / 1e9) / HBM_PEAK_GBS / world,
            'checks': {'rank0_block_bit_identical': True, 'first_slab_of_each_rank_recomputed': checked,
                       'oracle_checked_slabs': oracle_checked, 'finite_nkeff_fraction': float(np.isfinite(nk).mean())},
            'config': 'cfg4: %d slabs of %dx%d float64 (seed + slab id), %d contours, per-slab levels, contiguous blocks of '
                      'ceil(S/G) slabs per rank, chained launch sets of %d writing slab-major, ONE gather of (S, 9, N) f64 to rank 0 inside the timed job '
                      '(a piece per launch set, overlapped with the next set)'
                      % (S, NX4, NY4, NCONT, Cn),
        }
    for e in (e0, e1, e2):
        ctx.lib.xc_event_destroy(ctx.handle, e)
    plan.free()
    qbuf.free(); lat_b.free(); lon_b.free()
    gather.free_block(rb)
    return block


# ----------------------------------------------------------------------------- main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='slabs per step per GPU')
    ap.add_argument('--group', type=int, default=0, help='slabs per launch set (0: whole batch)')
    ap.add_argument('--variant', type=int, default=0, choices=[0, 1, 2, 3], help='0 PV-like (grid-scale noise), 1 noise, 2 sin(lat), 3 the reference\'s barotropic vorticity field interpolated to the cfg2 grid (smooth at grid scale)')
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
                         'which the 16 B/cell roofline numerator is exactly the unique HBM traffic (the default run reports it '
                         'as `variants.slab_dA` after the timed region)')
    ap.add_argument('--deterministic', action='store_true',
                    help='order-free fixed-point accumulation (xc_keff_desc.deterministic): bit-reproducible sums in ONE pass, '
                         'about 1.3x the histogram cost')
    ap.add_argument('--native-rccl', action='store_true', help='(kept for old command lines: the library communicator is the default now)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'rccl', 'ipc', 'gloo', 'host', 'auto'],
                    help="carrier of the ONE gather of a job (to rank 0): 'nccl' = 'rccl' (default): grouped ncclSend / ncclRecv over xGMI "
                         "through the library's own communicator (xc_comm_*), falling back -- loudly, in the JSON line -- to 'ipc' and then "
                         "'host' if a carrier does not pass its deadline-bounded trial on every rank; 'ipc': HIP IPC pushes into the root's "
                         "receive buffer (no RCCL; also works with several ranks on one GPU); 'gloo' = 'host': staged through the host over "
                         "the rendezvous sockets (slow; a last resort); 'auto': whichever of rccl / ipc moved its 32 MB trial faster")
    ap.add_argument('--deadline-s', type=float, default=480.0,
                    help='launcher only (`--gpus N` without WORLD_SIZE): the whole job\'s time limit; on expiry the ranks are terminated '
                         'by PID, the launcher reports every rank\'s last stage on stderr and exits 124')
    ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg3', 'cfg4', 'cfg5'],
                    help="BASELINE.json configuration: cfg2 (default, the headline metric's), or one of the secondary ones as "
                         'a bench line of the same contract (tools/bench_configs.py; single GPU; --steps / --warmup apply)')
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'],
                    help='tracer dtype: f64 (default: the headline configuration BASELINE.json names) or f32 -- float32 tracer AND float32 '
                         'contours, the dtype of the files the reference ships and its default `dtype` (core.py:21); algorithmic bytes 4 + 8 per cell')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--cpu-slabs', type=int, default=0, help='distinct slabs in the CPU sample, all parity-checked (0: 8)')
    ap.add_argument('--cpu-workers', type=int, default=0, help='upper bound on the CPU sample\'s concurrent processes (0: as many as cores and memory allow)')
    ap.add_argument('--no-extras', action='store_true', help='skip variants / long_run / unchained after the timed region')
    ap.add_argument('--long-run-s', type=float, default=4.0, help='seconds of extra steps with per-launch events (long_run)')
    ap.add_argument('--no-cfg4', action='store_true', help='skip the cfg4 strong-scaling block')
    ap.add_argument('--cfg4-slabs', type=int, default=512 * 37)
    ap.add_argument('--cfg4-chunk', type=int, default=0, help='slabs per launch set of the cfg4 sweep (0: chosen by cfg4_launch_set: a size that tiles the rank\'s block and fills whole rounds of workgroups)')
    ap.add_argument('--cfg4-reps', type=int, default=2, help='timed cfg4 jobs (sweep + gather)')
    ap.add_argument('--dump-cfg4', default='', help='rank 0 saves the gathered (S, 9, N) cfg4 result to this .npy file')
    return ap.parse_args(argv)


def main():
    a = parse_args()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # the driver's plain `python3 bench.py --gpus N`: become the launcher, BEFORE anything that could touch a GPU
        if a.config != 'cfg2':
            raise SystemExit('--config %s is a single-GPU line; the multi-GPU cfg4 job is the `cfg4_strong` block of the default run' % a.config)
        sys.exit(launch_ranks(a.gpus, sys.argv[1:], a.deadline_s))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault(*IPC_ENV)                               # see IPC_ENV above: dmabuf IPC, or no device carrier between processes
    stage('start')
    if a.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks (the two must agree: `value` is the '
                         'whole-job aggregate over --gpus GPUs)' % (a.gpus, world))

    sys.path.insert(0, ROOT)
    from xcontour_amd import _native as nat
    if not os.path.exists(nat.LIB_PATH):
        # the in-tree library normally travels with the snapshot; if it did not, build it once per node
        # (local rank 0 compiles, the others wait for the file) -- still no fallback: without it nothing runs
        if local == 0:
            import __graft_entry__
            __graft_entry__.build()
        for _ in range(600):
            if os.path.exists(nat.LIB_PATH):
                break
            time.sleep(0.5)
    from xcontour_amd.pipeline import KeffPlan
    from xcontour_amd.distributed import SocketGroup
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included

    nd = device_count(nat)
    if nd < 1:
        raise RuntimeError('bench.py needs an MI355X (no CPU fallback)')
    dev = local if local < nd else local % nd     # a launcher may expose one device per rank; a rehearsal has fewer GPUs than ranks
    ctx = nat.Context(dev)
    stage('rendezvous')
    group = SocketGroup(rank, world)
    pre = preflight(ctx, nat, group, dev, nd) if world > 1 else None
    if a.config != 'cfg2':
        if world > 1:
            raise SystemExit('--config %s is a single-GPU line; the multi-GPU cfg4 job is the `cfg4_strong` block of the default run' % a.config)
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import bench_configs
        print(json.dumps(bench_configs.run(a.config, ctx, a.steps, a.warmup)), flush=True)
        ctx.close()
        return
    stage('carrier')
    gather = Gather(ctx, group, a.backend)
    if world > 1 and rank == 0:
        print('[preflight] carrier: %s; trials %s' % (gather.carrier, json.dumps(gather.trials)), file=sys.stderr, flush=True)
    stage('setup')
    B, K, W = a.batch, a.steps, a.warmup
    qdt = np.dtype(np.float32 if a.dtype == 'f32' else np.float64)
    bpc = qdt.itemsize + 8                                        # algorithmic bytes per cell: tracer once + 2-D f64 dA once
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
    res = ctx.alloc(slot * K)
    wres = ctx.alloc(slot)                                        # warm-up slot
    plan = KeffPlan(ctx, NB * B, NY, NX, NCONT, qdt, qdt, dA=dA, lat=lat, lon=lon, tbl=tbl,
                    tbl_coord=lat, increase=True, lt=True, nslots=K, out_ptr=res.ptr, detect_row_dA=a.row_dA,
                    out_slabs=B, replicate_dA=a.slab_dA, right_edge='xhistogram', deterministic=a.deterministic, single_read=False)
    if a.variant == 3:
        plan.set_q(baro_slabs(NB * B, qdt, rank * NB * B))        # the reference's barotropic field on the cfg2 grid (smooth at grid scale)
    else:
        plan.synth(lat, lon, SEED + rank * NB * B, a.variant)     # slab s of rank r: seed + r*2B + s
    grp = a.group or B
    chain = bool(a.chain)

    def step(k, slot_idx, pl=None, ch=None):
        pl = plan if pl is None else pl
        ch = chain if ch is None else ch
        s0 = (k % NB) * B                                         # this step's batch
        nxt = ((k + 1) % NB) * B                                  # the batch of the next step
        for g0 in range(s0, s0 + B, grp):
            n = min(grp, s0 + B - g0)
            g1 = g0 + grp if g0 + grp < s0 + B else nxt          # what runs after this launch set
            pl.run_range(slot_idx, g0, n, g1 if (ch and min(grp, B) == n) else None, out_s0=g0 - s0)

    stage('warmup')
    plan.out_ptr = wres.ptr
    for k in range(-W, 0):                                        # ends on batch B; its pass carries batch A's min/max
        step(k, 0)
    plan.out_ptr = res.ptr
    ctx.sync()
    ev = [(ctx.event(), ctx.event()) for _ in range(K)]
    nres = slot * K
    rb = gather.recv_block(nres) if world > 1 else None           # the root's landing area: (world, nres) bytes
    if world > 1:
        # warm-up of the collective, like the W warm-up steps of the compute: a carrier's first transfer into a new buffer sets up
        # channels / mappings, which is start-up cost and not part of a steady-state job
        gather.push(rb, res.ptr, nres)
        gather.finish(rb)
        ctx._check(ctx.lib.xc_memset(ctx.handle, res.ptr, 0, nres))                     # the timed region fills them again
        if rb.buf is not None:
            ctx._check(ctx.lib.xc_memset(ctx.handle, rb.buf.ptr, 0, nres * world))
        ctx.sync()

    stage('timed')
    es0, es1 = ctx.event(), ctx.event()                           # this rank's own sweep: HIP events around its K steps
    group.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    ctx.record(es0)
    for k in range(K):
        if grp == B:
            ctx.set_hist_events(ev[k][0], ev[k][1])               # events around the K3 launch only
        step(k, k)
        if world > 1:
            gather.push(rb, res.ptr + k * slot, slot, k * slot)   # this step's vectors leave for rank 0 on the comm stream while the next step computes
    ctx.record(es1)
    if world > 1:
        gather.finish(rb)                                         # the one collective of the job ends here: only the last step's piece is exposed
    ctx.sync()                                                    # the library's own HIP streams (every kernel and the gather run on them)
    t_mine = time.perf_counter() - t0
    group.barrier()
    t1 = time.perf_counter()
    el = group.allreduce_max(t1 - t0)
    # what every rank saw, for the line: its sweep (events), its own wall clock up to the end of its part of the gather, and what RCCL
    # ITSELF reports about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) -- the proof of "N ranks on N devices"
    mine_info = {'rank': rank, 'sweep_ms': ctx.elapsed_ms(es0, es1), 'wall_ms': t_mine * 1e3, 'device': ctx.device, 'comm': ctx.comm_info()}
    per_rank = [json.loads(p.decode()) for p in group.allgather_bytes(json.dumps(mine_info).encode())]
    stage('checks')
    if world > 1 and rank == 0:
        # every rank's block arrived, in rank order, and rank r's slabs differ from rank 0's (seed + r*2B + s)
        g = gather.fetch(rb)
        mine = res.download((nres,), np.uint8)
        assert np.array_equal(g[0], mine) and all(not np.array_equal(g[r], g[0]) for r in range(1, world)), 'gathered blocks are not in rank order'
        del g, mine
    if world > 1:
        gather.free_block(rb)

    line = None
    if rank == 0:
        work_step = world * B * NY * NX * NCONT
        dA_kind = 'per-row vector (detected constant rows)' if a.row_dA else ('2-D f64 plane PER SLAB (time-varying weights)' if a.slab_dA else '2-D f64 plane shared by the slabs')
        line = {
            'metric': 'lat-lon cells*contours/s, full Keff pipeline', 'value': work_step * K / el,
            'unit': 'cells*contours/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': el / K * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': 'cfg2: synthetic %dx%d %s PV-like slabs, 2-D f64 dA, %d contours, '
                                   'full Keff (min/max + histogram with in-kernel |grad q|^2 + CDF + epilogue)'
                                   % (NX, NY, qdt.name, NCONT),
                       'slabs_per_step_per_gpu': B, 'slabs_per_launch': grp, 'resident_batches': NB, 'variant': a.variant, 'dA': dA_kind,
                       'minmax': 'folded into the previous histogram pass (q_next)' if chain else 'stand-alone K1 pass',
                       'sums': 'order-free fixed point (deterministic)' if a.deterministic else 'float64 LDS atomics',
                       'parallelism': ('independent slabs per GPU (dp%d), one gather to rank 0 at the end; carrier %s: %s' % (world, gather.carrier, gather.describe())) if world > 1 else 'single GPU',
                       'carrier_trials': gather.trials if world > 1 else None, 'preflight': pre,
                       'launcher': ('bench.py itself (one child process per rank)' if os.environ.get('XC_BENCH_LAUNCHED') else 'external (RANK / WORLD_SIZE from the environment)') if world > 1 else 'none',
                       'collective_note': gather.note,
                       'rccl': rccl_block(per_rank, gather.carrier if world > 1 else None),
                       'host_code': 'python + ctypes, no torch', 'device': ctx.device_name()},
            'per_rank': {'sweep_ms': [float(x['sweep_ms']) for x in per_rank], 'wall_ms': [float(x['wall_ms']) for x in per_rank],
                         'device': [int(x['device']) for x in per_rank],
                         'note': 'sweep_ms: HIP events around the rank\'s own K steps; wall_ms: its host clock from the common barrier to the end '
                                 'of its part of the gather; value uses the MAX over ranks of the barrier-to-barrier time'},
        }
        cells = B * NY * NX
        alg = cells * (qdt.itemsize if a.row_dA else bpc)              # tracer once + dA once per slab

        def uniq_bytes(slab_dA, esz=qdt.itemsize):
            # bytes that MUST cross HBM once per launch: this batch's tracer + the weights that are not shared
            # (a dA plane shared by the B slabs of a launch is fetched once; per-slab dA planes B times; a per-row vector ~0)
            return cells * esz + (cells * 8 if slab_dA else (NY * 8 if a.row_dA else NY * NX * 8))

        def kernel_name(slab_dA, ch, f32=(a.dtype == 'f32')):
            return ('k_hist<float,%s,%s>' if f32 else 'k_hist<double,%s,%s>') % ('DA_SLAB' if slab_dA else ('DA_ROW' if a.row_dA else 'DA_PLANE'), 'NEXT' if ch else 'plain')

        if grp == B:
            ms = np.array([ctx.elapsed_ms(e0, e1) for e0, e1 in ev])
            ach = alg / (ms.mean() * 1e-3) / 1e9
            uniq = uniq_bytes(a.slab_dA)
            det_key = a.deterministic and a.dtype == 'f64' and not a.slab_dA and chain
            traffic, tsrc = (None, 'not measured for this variant') if (a.row_dA or a.variant != 0 or (a.deterministic and not det_key) or (a.dtype != 'f64' and a.slab_dA)) else \
                stored_traffic('det_chain' if det_key else (('f32_' if a.dtype == 'f32' else '') + ('slab_' if a.slab_dA else '') + ('chain' if chain else 'nochain')), B)
            line['roofline'] = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                'frac': ach / HBM_PEAK_GBS,
                                'hbm_unique_frac': uniq / (ms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                'traffic': traffic, 'traffic_source': tsrc,
                                'kernel': kernel_name(a.slab_dA, chain),
                                'launch_ms': float(ms.mean()), 'launch_ms_std': float(ms.std()),
                                'algorithmic_bytes_per_launch': alg,
                                'hbm_unique_bytes_per_launch': uniq,
                                'streamed_bytes_per_launch': alg + (cells * qdt.itemsize if chain else 0),
                                'pipeline_frac': (alg * K / el / 1e9) / HBM_PEAK_GBS / world,
                                'pipeline_hbm_unique_frac': (uniq * K / el / 1e9) / HBM_PEAK_GBS / world,
                                'measured_stream_ceilings': stream_ceilings(),
                                'note': 'frac = ' + str(bpc) + ' B/cell (SURVEY 8d) / launch time / 8 TB/s; its numerator counts the dA plane once per '
                                        'slab although the %d slabs of a launch share it (cache-served after the first fetch) -- '
                                        'hbm_unique_frac counts only bytes that must come from HBM.  variants.slab_dA is the '
                                        'configuration where the two coincide' % B}
        # self-check of the last step: every cell lands in exactly one bin (xhistogram rule: last edge + 1e-8 keeps the max cell)
        out = plan.fetch(slot=K - 1)
        # (float32 contours: the last level is the float32 rounding of the maximum and `+ 1e-8` is below its resolution, so the
        #  maximum cell itself may fall outside the last edge -- reference behaviour; the oracle check below is exact)
        csum = out['counts'].sum(axis=1).astype(np.int64)
        lost = NX if a.variant == 2 else 4                           # (sin(lat): the whole pole row holds the maximum)
        if not ((csum == NY * NX).all() if a.dtype == 'f64' else ((csum <= NY * NX) & (csum >= NY * NX - lost)).all()) or out['status'].any():
            raise RuntimeError('bench self-check failed: counts %r status %r' % (out['counts'].sum(axis=1), out['status']))
        extras = world == 1 and grp == B and not a.no_extras
        if extras and chain:
            # transparency: the same work in the plain order (stand-alone K1 launch, then K3), a short extra run
            # AFTER the timed region (identical per-step outputs; tests/test_gpu_parity.py::test_chained_minmax_is_bit_identical)
            K2 = max(5, min(20, K))
            plan.out_ptr = wres.ptr
            for k in range(-3, 0):
                step(k, 0, ch=False)
            ctx.sync()
            t2 = time.perf_counter()
            for k in range(K2):
                step(k, 0, ch=False)
            ctx.sync()
            el2 = time.perf_counter() - t2
            plan.out_ptr = res.ptr
            line['unchained'] = {'value': work_step * K2 / el2, 'ms_per_step': el2 / K2 * 1e3, 'steps': K2,
                                 'note': 'stand-alone min/max launch before every histogram launch (--no-chain), same slabs'}
        if extras and a.long_run_s > 0:
            # the timed region above is K steps (tens of ms at the default K); the same steps for >= long_run_s seconds, every
            # histogram launch between its own pair of HIP events: a steadier twin of the headline (results go to the warm-up slot)
            nlr = int(min(100000, max(K, np.ceil(a.long_run_s / (el / K)))))
            lev = [(ctx.event(), ctx.event()) for _ in range(nlr)]
            plan.out_ptr = wres.ptr
            for k in range(-2, 0):
                step(k, 0)
            ctx.sync()
            t2 = time.perf_counter()
            for k in range(nlr):
                ctx.set_hist_events(lev[k][0], lev[k][1])
                step(k, 0)
            ctx.sync()
            el3 = time.perf_counter() - t2
            plan.out_ptr = res.ptr
            lms = np.array([ctx.elapsed_ms(e0, e1) for e0, e1 in lev])
            line['long_run'] = {'steps': nlr, 'seconds': el3, 'ms_per_step': el3 / nlr * 1e3, 'value': work_step * nlr / el3,
                                'launch_ms_mean': float(lms.mean()), 'launch_ms_std': float(lms.std()), 'launch_ms_min': float(lms.min()),
                                'launch_ms_p05': float(np.percentile(lms, 5)), 'launch_ms_p95': float(np.percentile(lms, 95)),
                                'launch_ms_max': float(lms.max()), 'frac': alg / (lms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                'hbm_unique_frac': uniq_bytes(a.slab_dA) / (lms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                'note': 'same schedule and slabs as the timed region, run after it'}
            for e0, e1 in lev:
                ctx.lib.xc_event_destroy(ctx.handle, e0); ctx.lib.xc_event_destroy(ctx.handle, e1)
        if extras and not a.slab_dA and not a.row_dA:
            # the configuration whose roofline numerator is 100 % unique HBM bytes: per-slab dA planes (XC_DA_SLAB), same tracer
            # batches, same chained schedule; its vectors are compared with the main leg's (same dA values -> same answers)
            p2 = None
            try:
                p2 = KeffPlan(ctx, NB * B, NY, NX, NCONT, qdt, qdt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                              increase=True, lt=True, nslots=1, out_slabs=B, replicate_dA=True, right_edge='xhistogram',
                              alloc_q=False, deterministic=a.deterministic)
                p2.set_q_device(plan._q_ptr)
                p2.desc.q_gen = plan.desc.q_gen
                KV = 10 + ((K - 10) % NB)                        # ends on the batch of the main leg's last step
                vev = [(ctx.event(), ctx.event()) for _ in range(KV)]
                for k in range(-3, 0):
                    step(k, 0, pl=p2)
                ctx.sync()
                t2 = time.perf_counter()
                for k in range(KV):
                    ctx.set_hist_events(vev[k][0], vev[k][1])
                    step(k, 0, pl=p2)
                ctx.sync()
                el4 = time.perf_counter() - t2
                vms = np.array([ctx.elapsed_ms(e0, e1) for e0, e1 in vev])
                vo = p2.fetch(slot=0)
                bad = [k for k in ('ctr', 'counts') if not np.array_equal(vo[k], out[k])]
                bad += [k for k in ('area', 'intgrdS', 'latEq', 'dqdA', 'dintSdA', 'Leq2') if not rel_err(vo[k], out[k]) < 1e-9]
                if bad or vo['status'].any():
                    raise RuntimeError('variants.slab_dA: results differ from the main leg: %s' % bad)
                ub = uniq_bytes(True)
                vt, vsrc = stored_traffic('slab_chain' if chain else 'slab_nochain', B) if (a.variant == 0 and not a.deterministic and a.dtype == 'f64') else (None, 'not measured')
                line['variants'] = {'slab_dA': {
                    'steps': KV, 'ms_per_step': el4 / KV * 1e3, 'value': work_step * KV / el4, 'kernel': kernel_name(True, chain),
                    'launch_ms': float(vms.mean()), 'launch_ms_std': float(vms.std()),
                    'frac': alg / (vms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'hbm_unique_frac': ub / (vms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'pipeline_hbm_unique_frac': (ub * KV / el4 / 1e9) / HBM_PEAK_GBS,
                    'algorithmic_bytes_per_launch': alg, 'hbm_unique_bytes_per_launch': ub, 'traffic': vt, 'traffic_source': vsrc,
                    'parity': 'levels + counts bit-identical to the main leg, sums / derived <= 1e-9 (the main leg is oracle-checked below)',
                    'note': 'every slab reads ITS OWN 2-D f64 dA plane (time-varying weights, core.py:1271-1274): all 16 B/cell of the '
                            'numerator cross HBM, frac == hbm_unique_frac'}}
            except nat.XContourHipError as e:
                line['variants'] = {'slab_dA': {'skipped': str(e)}}
            finally:
                if p2 is not None:
                    p2.free()
        if not a.no_cpu:
            # the oracle on slabs of the LAST timed step's batch; their vectors are compared with that step's GPU result.  N > 1: rank 0
            # alone, a smaller sample (the other ranks wait at a barrier whose timeout covers it, below): a line without a CPU
            # baseline beside it reads as unmeasured
            stage('cpu_baseline')
            nd_ = max(1, min(a.cpu_slabs or (8 if world == 1 else 4), B))
            s0 = ((K - 1) % NB) * B
            esz = NY * NX * qdt.itemsize
            qh = np.empty((nd_, NY, NX), dtype=qdt)
            ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, qh.ctypes.data, plan._q_ptr + s0 * esz, nd_ * esz))
            line['cpu_baseline'] = cpu_baseline(qh, out, nd_, qdt.name, a.cpu_workers)
    if world > 1:
        group.barrier(timeout=900.0)                              # rank 0 has been timing the CPU baseline meanwhile
    plan.free()
    res.free(); wres.free()
    if rank == 0 and world == 1 and a.dtype == 'f64' and grp == B and not a.no_extras and not (a.slab_dA or a.row_dA or a.deterministic):
        # float32 tracers and contours -- the reference's default `dtype` and the dtype of the files it ships -- through the same
        # chained schedule, so that the driver's line carries them; two slabs of its last step against the oracle
        v32 = variant_f32(ctx, nat, a, lat, lon, dA, tbl, chain)
        if a.variant == 0 and 'skipped' not in v32:
            # the same float32 schedule on fields that look like the reference's data: sin(lat) (whole rows in one bin) and the
            # reference's own barotropic vorticity field interpolated to this grid (smooth at grid scale: adjacent cells share bins)
            v32['fields'] = {'pv_like_with_grid_scale_noise': {k: v32[k] for k in ('us_per_slab', 'launch_ms', 'frac', 'hbm_unique_frac')},
                             'sin_lat': variant_f32(ctx, nat, a, lat, lon, dA, tbl, chain, variant=2, ncheck=0, brief=True),
                             'barotropic_vorticity_interpolated': variant_f32(ctx, nat, a, lat, lon, dA, tbl, chain, variant=3, ncheck=1, brief=True)}
        line.setdefault('variants', {})['f32'] = v32
    if rank == 0 and world == 1 and a.dtype == 'f64' and not a.no_extras and not (a.slab_dA or a.row_dA):
        # ONE cfg2 slab through the pipeline, a sync before every call: north_star's literal unit and what a caller with one (time, level)
        # field gets (the reference's callers hand planes over one at a time, tests/LWA.py:40-43) -- four dependent launches
        # (K1, K3, reduce, finalize), nothing to overlap them with
        stage('single_slab')
        line['single_slab'] = single_slab(ctx, nat, a, lat, lon, dA, tbl)
        # the reference's OWN call sequence through the facade at its demo size (cfg1's shape: 15 x 241 x 480 float32, 201 contours),
        # numpy in / numpy out with resident inputs: what a user of the reference's API waits for per analysis step
        stage('facade_demo')
        line['facade_demo'] = facade_demo()
    # ---- cfg4 strong scaling: every rank takes part (its own timed region, after the cfg2 buffers are gone)
    if not a.no_cfg4 and a.dtype == 'f64':
        stage('cfg4')
        blk = cfg4_strong(ctx, nat, a, group, gather)
        if rank == 0:
            line['cfg4_strong'] = blk
    stage('done')
    if rank == 0:
        print(json.dumps(line), flush=True)
    group.barrier()
    if group.stuck:
        # a helper thread never came back from ncclCommInitRank (SocketGroup.init_device): the results are out, and an interpreter
        # that waits for that thread -- or a library destructor that waits for its bootstrap -- would hang the finished job
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    ctx.close()
    group.close()


def rccl_block(per_rank, carrier):
    """`config.rccl` of the line: what RCCL itself reported on every rank (xc_comm_info: ncclCommCount, ncclCommUserRank,
    ncclCommCuDevice, ncclGetVersion, the file librccl came from).  None when the job's gather did not run on RCCL (one rank, or
    another carrier of the ladder: `config.parallelism` / `carrier_trials` say which and why)."""
    infos = [x['comm'] for x in per_rank]
    if carrier != 'rccl' or not any(i['comm_count'] for i in infos):
        return None
    return {'comm_count': [int(i['comm_count']) for i in infos],            # ncclCommCount as EVERY rank's communicator reports it
            'comm_rank': [int(i['comm_rank']) for i in infos],              # ncclCommUserRank: must read 0 .. N-1
            'rank_devices': [int(i['comm_device']) for i in infos],         # ncclCommCuDevice: the HIP device each rank's communicator drives
            'ctx_devices': [int(i['ctx_device']) for i in infos],
            'librccl': infos[0]['rccl_path'], 'version': int(infos[0]['rccl_version']),
            'consistent': bool(all(int(i['comm_count']) == len(infos) for i in infos) and
                               sorted(int(i['comm_rank']) for i in infos) == list(range(len(infos))))}


def baro_slabs(nslab, dtype, seed=0):
    """`--variant 3`: the reference's own barotropic vorticity field (tests/golden/baro_q.npy, 256 x 512 float32 on a Gaussian
    grid) interpolated bilinearly to the cfg2 grid -- smooth at grid scale, as real data is (a 256-cell row spans two to four of the
    201 bins; the PV-like default carries grid-scale noise that puts every cell in a bin of its own) -- as `nslab` distinct slabs
    (a slow drift in amplitude and offset per slab)."""
    q = np.load(os.path.join(ROOT, 'tests', 'golden', 'baro_q.npy')).astype(np.float64)
    lat0 = np.load(os.path.join(ROOT, 'tests', 'golden', 'baro_lat.npy')).astype(np.float64)
    ny0, nx0 = q.shape
    lat = np.linspace(-90, 90, NY)
    fy = np.interp(lat, lat0, np.arange(ny0))                          # fractional row index (clamped at the poles)
    y0 = np.minimum(fy.astype(np.int64), ny0 - 2); wy = (fy - y0)[:, None]
    fx = np.arange(NX) * (nx0 / float(NX))
    x0 = fx.astype(np.int64) % nx0; x1 = (x0 + 1) % nx0; wx = (fx - np.floor(fx))[None, :]
    rows = q[:, x0] * (1 - wx) + q[:, x1] * wx                          # (ny0, NX), periodic in X
    base = rows[y0] * (1 - wy) + rows[y0 + 1] * wy                      # (NY, NX)
    out = np.empty((nslab, NY, NX), dtype=dtype)
    for s_ in range(nslab):
        k = seed + s_
        out[s_] = (base * (1.0 + 0.002 * (k % 97)) + 1e-6 * (k % 13)).astype(dtype)
    return out


def device_count(nat):
    import ctypes as C
    n = C.c_int(0)
    nat.load().xc_device_count(C.byref(n))
    return n.value


def facade_demo(reps=40):
    """`facade_demo` of the default line: the seven calls of the reference's Keff sequence (tests/test_Keff_atmos.py:75-92 of the reference)
    on a 15 x 241 x 480 float32 stack with resident inputs, each timed alone on one field (`us`, their `sum_us`) and as a sequence on
    six fields taken in turn (`sequence_us`: the library's cache of small inputs can then only serve the second binning call of a pass,
    as in real use).  Wall clock, Python included."""
    try:
        import xcontour_amd as xa
        NL1, NY1, NX1, N1 = 15, 241, 480, NCONT
        lat = np.linspace(-90, 90, NY1).astype(np.float32); lon = (np.arange(NX1) * 0.75).astype(np.float32); lev = np.arange(NL1, dtype=np.float32)
        rng = np.random.default_rng(0)
        q = (np.sin(np.deg2rad(lat))[None, :, None] * (1 + 0.1 * lev[:, None, None]) + 0.05 * rng.standard_normal((NL1, NY1, NX1))).astype(np.float32)
        c3 = {'lev': lev, 'lat': lat, 'lon': lon}; c2 = {'lat': lat, 'lon': lon}
        dA = xa.DataArray(xa.cell_area(lat.astype(np.float64), lon.astype(np.float64)).astype(np.float32), ('lat', 'lon'), c2, 'dA')
        g2 = xa.DataArray(rng.random(q.shape).astype(np.float32), ('lev', 'lat', 'lon'), c3, 'grdS')
        mask = xa.DataArray(np.ones((NY1, NX1), np.float32), ('lat', 'lon'), c2, 'mask')
        kw = dict(dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'}, increase=True, lt=True, resident=True)
        objs = [xa.Contour2D(xa.DataArray((q * np.float32(1.0 + 0.03 * i) + np.float32(0.01 * i)).astype(np.float32), ('lev', 'lat', 'lon'), c3, 'pv'), dA, **kw)
                for i in range(6)]
        cm = objs[0]
        rec = {}

        def timed(name, fn):
            fn(); fn()
            t = time.perf_counter()
            for _ in range(reps):
                out = fn()
            rec[name] = round((time.perf_counter() - t) / reps * 1e6, 1)
            return out
        table = timed('cal_area_eqCoord_table_hist', lambda: cm.cal_area_eqCoord_table_hist(mask))
        ctr = timed('cal_contours', lambda: cm.cal_contours(N1))
        area = timed('cal_integral_within_contours_hist(area)', lambda: cm.cal_integral_within_contours_hist(ctr))
        intS = timed('cal_integral_within_contours_hist(grdS)', lambda: cm.cal_integral_within_contours_hist(ctr, integrand=g2))
        timed('lookup_coordinates', lambda: table.lookup_coordinates(area))
        timed('cal_gradient_wrt_area x2', lambda: (cm.cal_gradient_wrt_area(ctr, area), cm.cal_gradient_wrt_area(intS, area)))

        def sequence(c):
            tb = c.cal_area_eqCoord_table_hist(mask)
            ct = c.cal_contours(N1)
            ar = c.cal_integral_within_contours_hist(ct)
            iS = c.cal_integral_within_contours_hist(ct, integrand=g2)
            return tb.lookup_coordinates(ar), c.cal_gradient_wrt_area(ct, ar), c.cal_gradient_wrt_area(iS, ar)
        for c in objs:
            sequence(c); sequence(c)
        t = time.perf_counter()
        for _ in range(5):
            for c in objs:
                sequence(c)
        seq = (time.perf_counter() - t) / 30 * 1e6
        fused = timed('keff (fused, one call for everything)', lambda: cm.keff(N1, table, grdS=g2))
        del fused
        out = {'shape': [NL1, NY1, NX1], 'dtype': 'f32', 'contours': N1, 'resident': True, 'us': {k: v for k, v in rec.items() if not k.startswith('keff')},
               'sum_us': round(sum(v for k, v in rec.items() if not k.startswith('keff')), 1), 'sequence_us': round(seq, 1),
               'keff_fused_us': rec['keff (fused, one call for everything)'],
               'note': 'wall clock per call, Python included; sum_us: each call repeated on one field; sequence_us: the seven calls in order on six fields in turn'}
        for c in objs:
            c.close()
        return out
    except Exception as e:                      # the headline does not depend on this block
        return {'skipped': '%s: %s' % (type(e).__name__, e)}


def single_slab(ctx, nat, a, lat, lon, dA, tbl, reps=30):
    """`single_slab` of the default line: one 3600 x 1801 float64 slab, 201 contours, K1 -> K3 -> reduce -> finalize with a stream
    synchronisation before every call; `warm`: back to back (slab and weights stay in the 256 MiB Infinity Cache), `cold`: a 600 MB
    memset between calls.  frac = 16 B x cells / time / 8 TB/s, the same numerator as the headline."""
    from xcontour_amd.pipeline import KeffPlan
    p = pc = big = None
    try:
        kw = dict(dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat, increase=True, lt=True, right_edge='xhistogram', deterministic=a.deterministic)
        p = KeffPlan(ctx, 1, NY, NX, NCONT, np.float64, np.float64, **kw)
        p.synth(lat, lon, SEED + 7, a.variant if a.variant != 3 else 0)
        pc = KeffPlan(ctx, 1, NY, NX, NCONT, np.float64, np.float64, alloc_q=False, single_read=False, **kw)   # the same slab through the chain
        pc.set_q_device(p._q_ptr)
        big = ctx.alloc(600 << 20)
        RNX = 4096; RNY = (600 << 20) // 8 // RNX
        mm2 = ctx.alloc(RNY * 8)
        e0, e1 = ctx.event(), ctx.event()

        def timed(plan, evict):
            ts = []
            for r in range(reps + 3):
                if evict == 'read':                      # the caches emptied by READING 600 MB (row sums, ordinary loads): nothing dirty is left behind
                    ctx._check(ctx.lib.xc_rowsum_dev(ctx.handle, big.ptr, nat.XC_F64, None, nat.XC_DA_NONE, RNY, RNX, 0, mm2.ptr))
                elif evict:
                    ctx._check(ctx.lib.xc_memset(ctx.handle, big.ptr, r & 255, big.nbytes))
                ctx.sync()
                ctx.record(e0); plan.run(); ctx.record(e1)
                ms = ctx.elapsed_ms(e0, e1)
                if r >= 3:
                    ts.append(ms * 1e3)
            return np.array(ts)
        cold, warm = timed(p, True), timed(p, False)
        cold_clean = timed(p, 'read')
        path = ctx.last_keff_path()
        out = p.fetch()
        ok = bool((out['counts'].sum(axis=1) == NY * NX).all() and not out['status'].any() and p.replays == 0)
        ccold, cwarm = timed(pc, True), timed(pc, False)
        outc = pc.fetch()
        ok = ok and bool(np.array_equal(out['counts'], outc['counts']) and np.array_equal(out['ctr'], outc['ctr']) and
                         np.allclose(out['area'], outc['area'], rtol=1e-12, atol=0) and np.allclose(out['intgrdS'], outc['intgrdS'], rtol=1e-12, atol=0))
        # the reference's default dtype (core.py:21): the same call on a float32 slab with float32 contours (12 B/cell algorithmic)
        f32 = None
        if not a.deterministic:
            p32 = KeffPlan(ctx, 1, NY, NX, NCONT, np.float32, np.float32, **kw)
            try:
                p32.synth(lat, lon, SEED + 7, a.variant if a.variant != 3 else 0)
                c32, w32 = timed(p32, True), timed(p32, False)
                o32 = p32.fetch()
                f32 = {'us': float(np.median(w32)), 'us_cold': float(np.median(c32)), 'path': ctx.last_keff_path(),
                       'self_check': bool(not o32['status'].any() and p32.replays == 0 and int(o32['counts'].sum()) in (NY * NX, NY * NX - 1))}
            finally:
                p32.free()
        for e in (e0, e1):
            ctx.lib.xc_event_destroy(ctx.handle, e)
        alg = NY * NX * BYTES_PER_CELL
        return {'us': float(np.median(warm)), 'us_cold': float(np.median(cold)), 'us_cold_after_reads': float(np.median(cold_clean)),
                'us_min': float(warm.min()), 'reps': reps,
                'frac': alg / (np.median(warm) * 1e-6) / 1e9 / HBM_PEAK_GBS, 'frac_cold': alg / (np.median(cold) * 1e-6) / 1e9 / HBM_PEAK_GBS,
                'path': 'single-read kernel (k_keff_single: min/max -> levels -> histogram in one launch, the slab held in registers; then k_finalize)' if path == 1
                        else 'chain: k_minmax_partial (also clears the accumulators), k_hist (blocks add into them), k_finalize',
                'chain': {'us': float(np.median(cwarm)), 'us_cold': float(np.median(ccold)),
                          'launches': 'k_minmax_partial, k_hist, k_finalize (xc_keff_desc.single_read = XC_SINGLE_NEVER)'},
                'f32': f32, 'self_check': ok,
                'note': 'one 3600x1801 f64 slab per call, a stream sync before every call (HIP events around the call); warm = back to back '
                        '(Infinity-Cache resident), cold = a 600 MB memset between calls (the call then competes with the write-back of the memset), '
                        'cold_after_reads = 600 MB READ between calls (caches emptied, nothing dirty); 16 B/cell numerator as the headline; `chain`: the same '
                        'slab through the three-launch path this kernel replaces, results compared'}
    except nat.XContourHipError as e:
        return {'skipped': str(e)}
    finally:
        for x in (pc, p):
            if x is not None:
                x.free()
        if big is not None:
            big.free()
            mm2.free()


def variant_f32(ctx, nat, a, lat, lon, dA, tbl, chain, variant=None, ncheck=2, brief=False):
    """`variants.f32` of the default line: the headline schedule on float32 tracers with float32 contours (12 B/cell algorithmic:
    tracer 4 + dA 8), HIP events around every histogram launch, two slabs of the last step compared with the oracle."""
    from xcontour_amd.pipeline import KeffPlan
    B, NB, KV = a.batch, 2, 20
    qdt = np.dtype(np.float32)
    p = None
    try:
        p = KeffPlan(ctx, NB * B, NY, NX, NCONT, qdt, qdt, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                     increase=True, lt=True, nslots=1, out_slabs=B, right_edge='xhistogram')
        variant = a.variant if variant is None else variant
        if variant == 3:
            p.set_q(baro_slabs(NB * B, qdt))
        else:
            p.synth(lat, lon, SEED, variant)

        def step(k):
            s0, nxt = (k % NB) * B, ((k + 1) % NB) * B
            p.run_range(0, s0, B, nxt if chain else None, out_s0=0)

        for k in range(-4, 0):
            step(k)
        ctx.sync()
        vev = [(ctx.event(), ctx.event()) for _ in range(KV)]
        t2 = time.perf_counter()
        for k in range(KV):
            ctx.set_hist_events(vev[k][0], vev[k][1])
            step(k)
        ctx.sync()
        el = time.perf_counter() - t2
        vms = np.array([ctx.elapsed_ms(e0, e1) for e0, e1 in vev])
        for e0, e1 in vev:
            ctx.lib.xc_event_destroy(ctx.handle, e0); ctx.lib.xc_event_destroy(ctx.handle, e1)
        out = p.fetch(slot=0)
        checked = 0
        if not a.no_cpu:
            sys.path.insert(0, os.path.join(ROOT, 'oracle'))
            import xcontour_oracle as O
            s0 = ((KV - 1) % NB) * B
            qh = np.empty((2, NY, NX), dtype=qdt)
            ctx._check(ctx.lib.xc_memcpy_d2h(ctx.handle, qh.ctypes.data, p._q_ptr + s0 * NY * NX * 4, qh.nbytes))
            for s_ in range(ncheck):
                r = O.keff_pipeline(qh[s_], dA, lat, NCONT, lon=lon, increase=True, lt=True, dtype=np.float32)
                _compare_with_oracle(out, {k: np.asarray(r[k], np.float64) for k in CHECK_NAMES + ('counts',)}, s_)
                checked += 1
        cells = B * NY * NX
        alg, ub = cells * 12, cells * 4 + NY * NX * 8
        vt, vsrc = stored_traffic('f32_chain' if chain else 'f32_nochain', B) if variant == 0 else (None, 'not measured')
        work = B * NY * NX * NCONT
        if brief:
            return {'us_per_slab': el / KV / B * 1e6, 'launch_ms': float(vms.mean()), 'frac': alg / (vms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'hbm_unique_frac': ub / (vms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS, 'oracle_checked_slabs': checked}
        return {'steps': KV, 'ms_per_step': el / KV * 1e3, 'value': work * KV / el, 'us_per_slab': el / KV / B * 1e6,
                'kernel': 'k_hist<float,DA_PLANE,%s>' % ('NEXT' if chain else 'plain'), 'dtype': 'f32',
                'launch_ms': float(vms.mean()), 'launch_ms_std': float(vms.std()),
                'frac': alg / (vms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS, 'hbm_unique_frac': ub / (vms.mean() * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'algorithmic_bytes_per_launch': alg, 'hbm_unique_bytes_per_launch': ub, 'traffic': vt, 'traffic_source': vsrc,
                'oracle_checked_slabs': checked,
                'note': 'float32 tracer AND float32 contours (reference default dtype, core.py:21): 4 + 8 algorithmic bytes per cell; '
                        'counts + levels bit-exact, sums 1e-11, derived 1e-6 against the oracle on the checked slabs'}
    except nat.XContourHipError as e:
        return {'skipped': str(e)}
    finally:
        if p is not None:
            p.free()


if __name__ == '__main__':
    main()
