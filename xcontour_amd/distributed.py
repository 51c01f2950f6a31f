# -*- coding: utf-8 -*-
"""
Multi-GPU driver of the slab-parallel hot path.

Independent (time, level) slabs are partitioned statically over the ranks (one process
per GPU); there is NO exchange during compute.  The only collective is one all-gather
of the small per-slab result vectors at the end.

Two carriers for that one collective:

* `SocketGroup` (this module, numpy + the standard library, NO torch): a rendezvous over TCP
  on MASTER_ADDR / MASTER_PORT (the variables `torch.distributed.run`, `mpirun` wrappers and
  SLURM scripts export) with `barrier`, `allgather` of host arrays through rank 0, and --
  with a device context -- the library's own RCCL communicator (`xc_comm_*`: ncclAllGather over
  xGMI on the context's stream; the 128-byte unique id travels through the sockets).
* a `torch.distributed` process group (gloo / nccl) if the caller already lives in one --
  torch is imported lazily and only then; nothing else in the package touches it.
"""
import os
import pickle
import socket
import struct
import time

import numpy as np

from .pipeline import shard_slabs


# ----------------------------------------------------------------------------- torch-free process group
def _send(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv(sock):
    n = struct.unpack('<Q', _recv_exact(sock, 8))[0]
    return _recv_exact(sock, n)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError('xcontour_amd.distributed: peer closed the connection')
        got += k
    return bytes(buf)


class SocketGroup(object):
    """One process per rank, a star over TCP through rank 0.  Built for ONE small collective at the end of a job
    (a few hundred MB at most), not for bandwidth: the device path is `allgather_device` (RCCL)."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get('RANK', 0) if rank is None else rank)
        self.world = int(os.environ.get('WORLD_SIZE', 1) if world is None else world)
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = int(port if port is not None else int(os.environ.get('MASTER_PORT', 29500)) + 1)     # + 1: the launcher's own store owns MASTER_PORT
        self._peers, self._up = [], None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                peers[struct.unpack('<i', _recv_exact(c, 4))[0]] = c
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=timeout)
                    break
                except OSError:
                    if time.time() - t0 > timeout:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(struct.pack('<i', self.rank))
            self._up = s

    # -- primitives
    def allgather_bytes(self, payload):
        """every rank contributes `payload` (bytes); returns the list of all contributions in rank order, on every rank"""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv(c) for c in self._peers]
            blob = pickle.dumps(parts, protocol=pickle.HIGHEST_PROTOCOL)
            for c in self._peers:
                _send(c, blob)
            return parts
        _send(self._up, payload)
        return pickle.loads(_recv(self._up))

    def broadcast_bytes(self, payload, src=0):
        return self.allgather_bytes(payload if self.rank == src else b'')[src]

    def barrier(self):
        self.allgather_bytes(b'')

    def allreduce_max(self, x):
        return max(struct.unpack('<d', p)[0] for p in self.allgather_bytes(struct.pack('<d', float(x))))

    def allgather(self, arr):
        """equal-shape host arrays -> (world,) + shape, on every rank"""
        arr = np.ascontiguousarray(arr)
        parts = self.allgather_bytes(arr.tobytes())
        return np.stack([np.frombuffer(p, dtype=arr.dtype).reshape(arr.shape) for p in parts])

    # -- the device path: the library's RCCL communicator over xGMI
    def init_device(self, ctx):
        """create the RCCL communicator of `ctx` (one context = one GPU per rank); the unique id travels through the sockets"""
        uid = self.broadcast_bytes(ctx.comm_unique_id() if self.rank == 0 else b'')
        ctx.comm_init(self.world, self.rank, uid)
        self._ctx = ctx

    def allgather_device(self, send_ptr, recv_ptr, bytes_per_rank):
        """ncclAllGather on the context's stream: `bytes_per_rank` from every rank's `send_ptr` into `recv_ptr` (rank order)"""
        self._ctx.comm_allgather(send_ptr, recv_ptr, bytes_per_rank)

    def close(self):
        for c in self._peers:
            c.close()
        if self._up is not None:
            self._up.close()
        self._peers, self._up = [], None


# ----------------------------------------------------------------------------- the one gather
def all_gather_slabs(local, nslab, rank, world, group=None):
    """`local`: (n_local, ...) holding this rank's block [lo, hi) of the flattened slab index -- a numpy array (with a
    `SocketGroup` as `group`) or a torch tensor (inside a torch.distributed process group).  Returns (nslab, ...) with
    every rank's block in slab order, on every rank.  Blocks are padded to ceil(S/G) slabs for one equal-size all-gather
    (the last ranks may own fewer or no slabs)."""
    per = -(-int(nslab) // int(world))
    lo, hi = shard_slabs(nslab, rank, world)
    assert local.shape[0] == hi - lo, 'local block has %d slabs, expected %d' % (local.shape[0], hi - lo)
    tail = tuple(local.shape[1:])
    if isinstance(local, np.ndarray):
        send = np.zeros((per,) + tail, dtype=local.dtype)
        if hi > lo:
            send[:hi - lo] = local
        if world == 1:
            return send[:nslab]
        if group is None:
            raise Exception('all_gather_slabs: a numpy block needs a SocketGroup (or pass a torch tensor inside a process group)')
        recv = group.allgather(send).reshape((world * per,) + tail)
        if world * per == nslab:
            return recv
        return np.concatenate([recv[r * per:r * per + (shard_slabs(nslab, r, world)[1] - shard_slabs(nslab, r, world)[0])]
                               for r in range(world)], axis=0)
    import torch
    import torch.distributed as dist
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
    return np.ascontiguousarray(np.concatenate(parts, axis=0))


def run_sharded(process, nslab, rank, world, device=None, group=None):
    """Process slabs [lo, hi) on this rank with `process(lo, hi) -> ndarray (hi-lo, ...)`
    (e.g. a KeffPlan over the rank's block) and gather every rank's result: through `group` (a SocketGroup: numpy in,
    numpy out, no torch) or, without one, through the torch.distributed process group the caller initialised."""
    lo, hi = shard_slabs(nslab, rank, world)
    out = np.ascontiguousarray(np.asarray(process(lo, hi)))
    if group is not None or world == 1:
        return all_gather_slabs(out, nslab, rank, world, group)
    import torch
    t = torch.from_numpy(out)
    if device is not None:
        t = t.to(device)
    return all_gather_slabs(t, nslab, rank, world)
