#!/usr/bin/env python3
"""Probe: do HIP IPC pushes work between processes on this box?  `python tools/probe/ipc_probe.py [world] [MB]` starts `world` rank
processes on the visible GPUs (rank r -> device r % ndev); the root exports a receive buffer, every rank pushes its block into it on
its comm stream, the root checks the bytes.  Prints one JSON line."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def rank_main():
    from xcontour_amd import _native as nat
    from xcontour_amd.distributed import SocketGroup
    import ctypes as C
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    mb = float(os.environ.get('XC_PROBE_MB', '32'))
    n = C.c_int(0); nat.load().xc_device_count(C.byref(n))
    ctx = nat.Context(rank % max(1, n.value))
    g = SocketGroup(rank, world)
    nbytes = int(mb * (1 << 20)) // 8 * 8
    x = np.full(nbytes // 8, float(rank + 1))
    x[::1024] = np.arange(len(x[::1024])) + 1000.0 * rank
    send = ctx.to_device(x)
    recv = ctx.alloc(world * nbytes) if rank == 0 else None
    h = g.broadcast_bytes(ctx.ipc_export(recv.ptr) if rank == 0 else b'')
    dst = recv.ptr if rank == 0 else ctx.ipc_open(h)
    ts = []
    for it in range(5):
        g.barrier(); ctx.sync()
        t0 = time.perf_counter()
        ctx.comm_wait_compute()
        ctx.comm_memcpy_d2d(dst + rank * nbytes, send.ptr, nbytes)
        ctx.compute_wait_comm()
        ctx.sync()
        g.barrier()
        ts.append(time.perf_counter() - t0)
    ok = None
    if rank == 0:
        got = recv.download((world, nbytes // 8), np.float64)
        ok = True
        for r in range(world):
            e = np.full(nbytes // 8, float(r + 1)); e[::1024] = np.arange(len(e[::1024])) + 1000.0 * r
            ok = ok and bool(np.array_equal(got[r], e))
        print(json.dumps({'ipc_push_ok': ok, 'world': world, 'MB_per_rank': mb, 'ms': [round(t * 1e3, 3) for t in ts], 'ndev': n.value}), flush=True)
    g.barrier()
    if rank != 0:
        ctx.ipc_close(dst)
    g.barrier()
    ctx.close(); g.close()
    sys.exit(0 if (ok is None or ok) else 1)


if __name__ == '__main__':
    if 'RANK' in os.environ:
        rank_main()
    else:
        import secrets
        world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        env = dict(os.environ)
        env.update({'WORLD_SIZE': str(world), 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29611', 'XC_DIST_TOKEN': secrets.token_hex(8),
                    'HSA_ENABLE_IPC_MODE_LEGACY': env.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), 'XC_PROBE_MB': sys.argv[2] if len(sys.argv) > 2 else '32'})
        ps = []
        for r in range(world):
            e = dict(env); e['RANK'] = str(r)
            ps.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=e))
        rc = 0
        for p in ps:
            try:
                rc = rc or p.wait(timeout=120)
            except subprocess.TimeoutExpired:
                p.kill(); rc = 1
        sys.exit(rc)
