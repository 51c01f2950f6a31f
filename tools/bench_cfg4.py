#!/usr/bin/env python3
"""
BASELINE.json configs[3]: a 512 time x 37 level stack (18 944 slabs) of 1440 x 721 float64
tracer slabs, Keff per slab (per-slab levels), the flattened (time, level) index sharded
statically over the GPUs of one node, ONE RCCL gather of the per-slab result vectors at the end
(strong scaling: the total work is fixed).  Not the driver's bench (that is ../bench.py, cfg2);
run it by hand:

    python tools/bench_cfg4.py [--slabs 18944] [--steps 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29533 tools/bench_cfg4.py

The full stack is 157 GB (19.7 GB per GPU on 8 GPUs; it also fits the 288 GB of one MI355X).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NY, NX, NCONT, SEED = 721, 1440, 201, 20241008


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--slabs', type=int, default=512 * 37)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--chunk', type=int, default=256, help='slabs per launch set')
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    local = local if local < torch.cuda.device_count() else 0
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    from xcontour_amd import _native as nat
    from xcontour_amd.pipeline import KeffPlan, shard_slabs, OUT_NAMES
    from xcontour_amd.distributed import all_gather_slabs, chunks_to_slabs
    from xcontour_amd.utils import cell_area, table_from_rowsums, last_row_included

    ctx = nat.Context(local)
    lat = np.linspace(-90, 90, NY)
    lon = np.arange(NX) * 0.25
    dA = cell_area(lat, lon)
    tbl = table_from_rowsums(ctx.rowsum(None, dA, NY, NX), True, last_row_included(lat, 'xhistogram'))
    S = a.slabs
    lo, hi = shard_slabs(S, rank, world)
    n = hi - lo
    C = min(a.chunk, max(n, 1))
    nchunk = -(-n // C) if n else 0
    slab_bytes = NY * NX * 8
    qbuf = ctx.alloc(max(n, 1) * slab_bytes)                       # this rank's block of the stack, resident
    lat_b, lon_b = ctx.to_device(lat), ctx.to_device(lon)
    for c0 in range(0, n, C):                                      # slab s of the stack: seed + s
        m = min(C, n - c0)
        ctx._check(ctx.lib.xc_synth_dev(ctx.handle, qbuf.ptr + c0 * slab_bytes, nat.XC_F64, m, NY, NX,
                                        lat_b.ptr, lon_b.ptr, SEED + lo + c0, 0))
    ctx.sync()
    slot = KeffPlan.out_bytes(C, NCONT)
    res = torch.zeros(max(nchunk, 1) * slot // 8, dtype=torch.float64, device='cuda')
    plan = KeffPlan(ctx, C, NY, NX, NCONT, np.float64, np.float64, dA=dA, lat=lat, lon=lon, tbl=tbl, tbl_coord=lat,
                    increase=True, lt=True, nslots=max(nchunk, 1), out_ptr=res.data_ptr(), alloc_q=False)

    def sweep():
        for ci in range(nchunk):
            c0 = ci * C
            m = min(C, n - c0)
            plan.set_q_device(qbuf.ptr + c0 * slab_bytes)
            nxt = ((ci + 1) % nchunk) * C
            chain = min(C, n - nxt) == m                            # equal-shape launch sets can chain their min/max
            plan._point(ci, 0, m)
            plan.desc.q_next = (qbuf.ptr + nxt * slab_bytes) if chain else None
            ctx._check(ctx.lib.xc_keff_dev(ctx.handle, __import__('ctypes').byref(plan.desc)))

    for _ in range(a.warmup):
        sweep()
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sweep()
    ctx.sync()
    # per-slab vectors of this rank's block -> (n, 9, N), then the one collective
    mine = chunks_to_slabs(res, slot // 8, C, n, NCONT)
    full = all_gather_slabs(mine, S, rank, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    if rank == 0:
        assert full.shape == (S, 9, NCONT)
        nk = full[:, OUT_NAMES.index('nkeff'), :]
        print(json.dumps({
            'metric': 'lat-lon cells*contours/s, full Keff pipeline (cfg4 stack)',
            'value': S * NY * NX * NCONT * a.steps / el, 'unit': 'cells*contours/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': el / a.steps * 1e3, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'cfg4: %d slabs of %dx%d float64, %d contours, per-slab levels, static slab '
                                   'partition, one RCCL gather of (S, 9, N) at the end' % (S, NX, NY, NCONT),
                       'slabs_per_gpu': n, 'slabs_per_launch': C},
            'us_per_slab_per_gpu': el / a.steps / max(n, 1) * 1e6,
            'finite_nkeff_fraction': float(torch.isfinite(nk).double().mean().item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
