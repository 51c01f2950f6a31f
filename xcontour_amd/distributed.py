# -*- coding: utf-8 -*-
"""
Multi-GPU driver of the slab-parallel hot path.

Independent (time, level) slabs are partitioned statically over the ranks (one process
per GPU); there is NO exchange during compute.  The only collective is one all-gather
of the small per-slab result vectors at the end -- RCCL over xGMI when the process group
uses the 'nccl' backend (which IS RCCL on ROCm), gloo in the CPU tests.

torch.distributed is plumbing here (rendezvous + the collective); the package itself
does not depend on torch: import this module only in multi-process jobs.
"""
from .pipeline import shard_slabs


def all_gather_slabs(local, nslab, rank, world):
    """`local`: torch tensor (n_local, ...) holding this rank's block [lo, hi) of the
    flattened slab index.  Returns a tensor (nslab, ...) with every rank's block in slab
    order, on every rank.  Blocks are padded to ceil(S/G) slabs for one equal-size
    all_gather_into_tensor (the last ranks may own fewer or no slabs)."""
    import torch
    import torch.distributed as dist
    per = -(-int(nslab) // int(world))
    lo, hi = shard_slabs(nslab, rank, world)
    assert local.shape[0] == hi - lo, 'local block has %d slabs, expected %d' % (local.shape[0], hi - lo)
    tail = tuple(local.shape[1:])
    send = torch.zeros((per,) + tail, dtype=local.dtype, device=local.device)
    if hi > lo:
        send[:hi - lo] = local
    if world == 1:
        return send[:nslab]
    recv = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send)
    if world * per == nslab:
        return recv
    # strip the padding of short blocks (only trailing ranks can be short with a contiguous partition)
    parts = []
    for r in range(world):
        rlo, rhi = shard_slabs(nslab, r, world)
        parts.append(recv[r * per:r * per + (rhi - rlo)])
    return torch.cat(parts, dim=0)


def chunks_to_slabs(res, slot_elems, chunk, nlocal, N, nvec=9):
    """Result slots of a chunked KeffPlan sweep -> this rank's (nlocal, nvec, N) block.

    A rank works through its `nlocal` slabs in launch sets of `chunk` slabs; launch set c writes result slot c of a flat
    float64 buffer `res` (torch tensor or ndarray; `slot_elems` = KeffPlan.out_bytes(chunk, N) // 8 elements per slot)
    whose head is laid out [nvec][chunk][N] (pipeline.OUT_NAMES order).  The last set may be short (ragged): only its
    first nlocal - c * chunk slabs are real.  Returns the slab-major block in slab order."""
    nchunk = -(-int(nlocal) // int(chunk)) if nlocal else 0
    parts = []
    for c in range(nchunk):
        m = min(chunk, nlocal - c * chunk)
        head = res[c * slot_elems:c * slot_elems + nvec * chunk * N].reshape(nvec, chunk, N)
        head = head[:, :m, :]
        parts.append(head.permute(1, 0, 2) if hasattr(head, 'permute') else head.transpose(1, 0, 2))
    if not parts:
        return res[:0].reshape(0, nvec, N)
    if hasattr(parts[0], 'permute'):
        import torch
        return torch.cat(parts, dim=0).contiguous()
    import numpy as np
    return np.ascontiguousarray(np.concatenate(parts, axis=0))


def run_sharded(process, nslab, rank, world, device=None):
    """Process slabs [lo, hi) on this rank with `process(lo, hi) -> ndarray (hi-lo, ...)`
    (e.g. a KeffPlan over the rank's block) and gather every rank's result."""
    import numpy as np
    import torch
    lo, hi = shard_slabs(nslab, rank, world)
    out = np.asarray(process(lo, hi))
    t = torch.from_numpy(np.ascontiguousarray(out))
    if device is not None:
        t = t.to(device)
    return all_gather_slabs(t, nslab, rank, world)
