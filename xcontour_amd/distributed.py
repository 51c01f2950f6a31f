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
import hashlib
import hmac
import os
import socket
import struct
import time

import numpy as np

from .pipeline import shard_slabs


# ----------------------------------------------------------------------------- torch-free process group
_MAGIC = b'XCDG1\0\0\0'
MAX_FRAME_BYTES = 8 << 30          # a frame length read off the wire is checked against this before anything is allocated


def _send(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)))
    if len(payload):
        sock.sendall(payload)


def _recv(sock, limit=MAX_FRAME_BYTES):
    n = struct.unpack('<Q', _recv_exact(sock, 8))[0]
    if n > limit:
        raise ConnectionError('xcontour_amd.distributed: frame of %d bytes exceeds the limit of %d' % (n, limit))
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


def _pack_parts(parts):
    """a list of byte strings as ONE frame: count, then (length, bytes) per part -- no pickle: nothing received is executed"""
    return b''.join([struct.pack('<I', len(parts))] + [struct.pack('<Q', len(p)) + p for p in parts])


def _unpack_parts(blob, world):
    n = struct.unpack_from('<I', blob, 0)[0]
    if n != world:
        raise ConnectionError('xcontour_amd.distributed: gathered frame holds %d parts, expected %d' % (n, world))
    off, parts = 4, []
    for _ in range(n):
        k = struct.unpack_from('<Q', blob, off)[0]
        off += 8
        if off + k > len(blob):
            raise ConnectionError('xcontour_amd.distributed: malformed gathered frame')
        parts.append(blob[off:off + k])
        off += k
    return parts


def _is_loopback(addr):
    """True for 127.0.0.0/8, ::1 and names that resolve only to such addresses"""
    import ipaddress
    try:
        return ipaddress.ip_address(addr).is_loopback
    except ValueError:
        pass
    try:
        infos = socket.getaddrinfo(addr, None)
    except OSError:
        return False
    return bool(infos) and all(ipaddress.ip_address(i[4][0].split('%')[0]).is_loopback for i in infos)


class SocketGroup(object):
    """One process per rank, a star over TCP through rank 0.  Built for the rendezvous and ONE small collective at the end of
    a job (a few hundred MB at most), not for bandwidth: the device path is `allgather_device` (RCCL).

    Trust: a peer must present the job's token -- `token` or the environment's XC_DIST_TOKEN (the launcher of bench.py draws a
    random one per job); rank 0 answers with a proof of the same token, so neither side talks to a stranger that merely got to
    the port first.  Frames are length-prefixed byte strings (no pickle); ranks are checked for range and uniqueness and frame
    lengths against `MAX_FRAME_BYTES` before anything is allocated.  WITHOUT a token the proofs would be keyed with the empty
    string and prove nothing: that is accepted only when the rendezvous address is a loopback address (one node, e.g. under
    `torch.distributed.run --master-addr 127.0.0.1`, where only local processes can reach the port); on any other address a
    world > 1 without a token raises -- export XC_DIST_TOKEN (the same secret on every rank) or pass `token=`."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0, token=None):
        self.rank = int(os.environ.get('RANK', 0) if rank is None else rank)
        self.world = int(os.environ.get('WORLD_SIZE', 1) if world is None else world)
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = int(port if port is not None else int(os.environ.get('MASTER_PORT', 29500)) + 1)     # + 1: the launcher's own store owns MASTER_PORT
        tok = token if token is not None else os.environ.get('XC_DIST_TOKEN', '')
        tok = tok.encode() if isinstance(tok, str) else bytes(tok)
        self._peers, self._up, self._ctx = [], None, None
        self.timeout = float(timeout)
        self.stuck = []                  # helper threads that never came back from a blocking library call (init_device)
        if not (0 <= self.rank < self.world):
            raise Exception('SocketGroup: rank %d outside [0, %d)' % (self.rank, self.world))
        if self.world == 1:
            return
        if not tok and not _is_loopback(addr):
            raise Exception('SocketGroup: world %d on %s needs a job token (XC_DIST_TOKEN or token=): without one any process that '
                            'reaches the port passes the handshake; only a loopback rendezvous may go without' % (self.world, addr))

        def proof(side, nonce, r):
            return hmac.new(tok, side + nonce + struct.pack('<i', r), hashlib.sha256).digest()

        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            peers = {}
            t_end = time.time() + timeout
            while len(peers) < self.world - 1:
                if time.time() > t_end:
                    raise ConnectionError('SocketGroup: %d of %d peers arrived within %.0f s' % (len(peers), self.world - 1, timeout))
                c, _ = srv.accept()
                try:
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    c.settimeout(min(timeout, 10.0))
                    nonce = os.urandom(16)
                    c.sendall(_MAGIC + nonce)
                    hello = _recv_exact(c, 8 + 4 + 32)
                    r = struct.unpack('<i', hello[8:12])[0]
                    if hello[:8] != _MAGIC or not (1 <= r < self.world) or r in peers or \
                            not hmac.compare_digest(hello[12:], proof(b'C', nonce, r)):
                        c.close()                                  # a stranger, a bad rank or a duplicate: dropped, the job's peers may still come
                        continue
                    c.sendall(proof(b'S', nonce, r))
                    c.settimeout(timeout)
                    peers[r] = c
                except (OSError, struct.error):
                    c.close()
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
            hello = _recv_exact(s, 8 + 16)
            if hello[:8] != _MAGIC:
                s.close()
                raise ConnectionError('SocketGroup: %s:%d does not speak this protocol' % (addr, port))
            nonce = hello[8:]
            s.sendall(_MAGIC + struct.pack('<i', self.rank) + proof(b'C', nonce, self.rank))
            if not hmac.compare_digest(_recv_exact(s, 32), proof(b'S', nonce, self.rank)):
                s.close()
                raise ConnectionError('SocketGroup: rank 0 at %s:%d does not hold this job\'s token' % (addr, port))
            self._up = s

    # -- primitives
    def allgather_bytes(self, payload):
        """every rank contributes `payload` (bytes); returns the list of all contributions in rank order, on every rank"""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv(c) for c in self._peers]
            blob = _pack_parts(parts)
            for c in self._peers:
                _send(c, blob)
            return parts
        _send(self._up, payload)
        return _unpack_parts(_recv(self._up), self.world)

    def gather_bytes(self, payload):
        """contributions of all ranks in rank order on rank 0, None elsewhere (a gather to the root moves 1/world of an all-gather)"""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv(c) for c in self._peers]
            for c in self._peers:
                _send(c, b'')                                      # the release: nobody runs ahead of the root
            return parts
        _send(self._up, payload)
        _recv(self._up, 0)
        return None

    def broadcast_bytes(self, payload, src=0):
        return self.allgather_bytes(payload if self.rank == src else b'')[src]

    def barrier(self, timeout=None):
        """`timeout`: seconds this one barrier may take (default: the group's) -- e.g. while rank 0 times the CPU baseline"""
        if timeout is None or self.world == 1:
            self.allgather_bytes(b'')
            return
        socks = self._peers if self.rank == 0 else [self._up]
        for c in socks:
            c.settimeout(float(timeout))
        try:
            self.allgather_bytes(b'')
        finally:
            for c in socks:
                c.settimeout(self.timeout)

    def allreduce_max(self, x):
        return max(struct.unpack('<d', p)[0] for p in self.allgather_bytes(struct.pack('<d', float(x))))

    def allreduce_min(self, x):
        return min(struct.unpack('<d', p)[0] for p in self.allgather_bytes(struct.pack('<d', float(x))))

    def allgather(self, arr):
        """equal-shape host arrays -> (world,) + shape, on every rank"""
        arr = np.ascontiguousarray(arr)
        parts = self.allgather_bytes(arr.tobytes())
        if any(len(p) != arr.nbytes for p in parts):
            raise ConnectionError('SocketGroup.allgather: ranks contributed blocks of different sizes')
        return np.stack([np.frombuffer(p, dtype=arr.dtype).reshape(arr.shape) for p in parts])

    # -- the device path: the library's RCCL communicator over xGMI
    def init_device(self, ctx, timeout=None):
        """create the RCCL communicator of `ctx` (one context = one GPU per rank); the unique id travels through the sockets.
        Every rank learns whether EVERY rank succeeded (an exception on all of them otherwise: nobody is left waiting).
        ncclCommInitRank blocks until all ranks have joined its bootstrap -- forever, if one of them never calls it -- so it runs
        in a helper thread that is given `timeout` seconds (default XC_COMM_TIMEOUT_S, 60): a rank whose call has not returned
        by then reports 'timed out' as its verdict, the consensus fails on every rank and the caller moves on to another
        carrier.  The stuck thread is remembered in `self.stuck`: a process that has one must leave through os._exit after
        printing its results (an interpreter waiting for it at exit would hang the job after all)."""
        import threading
        if timeout is None:
            timeout = float(os.environ.get('XC_COMM_TIMEOUT_S', 60))
        err = ''
        try:
            uid = self.broadcast_bytes(ctx.comm_unique_id() if self.rank == 0 else b'')
        except Exception as e:                                     # rank 0 could not create the id (librccl missing ...)
            uid, err = self.broadcast_bytes(b''), str(e)
        if len(uid) == 128 and not err:
            box = {}
            lock = threading.Lock()

            def call():
                # (round 6, advisor) the helper thread creates the communicator WITHOUT touching `ctx` (xc_comm_create): the main thread
                # keeps using the context while this call blocks, and a call that returns after its deadline must not write into it --
                # it finds `abandoned` under the lock and disposes of what it got
                try:
                    if os.environ.get('XC_TEST_SKIP_COMM_INIT_RANK') == str(self.rank):
                        # fault injection (tests only): this rank never joins the bootstrap -- the others then sit in the REAL
                        # ncclCommInitRank until their deadline, which is the failure the deadline exists for
                        raise Exception('comm_init skipped on this rank (XC_TEST_SKIP_COMM_INIT_RANK)')
                    comm = ctx.comm_create(self.world, self.rank, uid)
                    with lock:
                        if box.get('abandoned'):
                            ctx.comm_release(comm)
                        else:
                            box['comm'] = comm
                except Exception as e:                             # noqa: BLE001 -- the verdict travels to every rank
                    with lock:
                        box['err'] = str(e)

            t = threading.Thread(target=call, name='xc-comm-init', daemon=True)
            t.start()
            t.join(timeout)
            with lock:
                if 'comm' in box:
                    ctx.comm_attach(box.pop('comm'), self.world, self.rank)          # by the thread that owns the context
                elif 'err' in box:
                    err = box['err']
                else:
                    box['abandoned'] = True
                    self.stuck.append(t)
                    err = 'ncclCommInitRank did not return within %.0f s' % timeout
        elif not err:
            err = 'rank 0 could not create the RCCL unique id'
        # the slowest rank may sit in its deadline while the others already wait here: give the consensus that long
        socks = self._peers if self.rank == 0 else [self._up]
        for c in socks:
            c.settimeout(self.timeout + timeout)
        try:
            errs = [p.decode('utf-8', 'replace') for p in self.allgather_bytes(err.encode())]
        finally:
            for c in socks:
                c.settimeout(self.timeout)
        if any(errs):
            if not self.stuck:
                try:
                    ctx.comm_finalize()
                except Exception:
                    pass
            raise Exception('RCCL communicator not created: ' + '; '.join('rank %d: %s' % (r, e) for r, e in enumerate(errs) if e))
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


def run_sharded(process, nslab, rank, world, device=None, group=None, as_numpy=None):
    """Process slabs [lo, hi) on this rank with `process(lo, hi) -> ndarray (hi-lo, ...)`
    (e.g. a KeffPlan over the rank's block) and gather every rank's result.  The CARRIER decides the return type, not the
    world size: with `group` (a SocketGroup) numpy in, numpy out, no torch; without one the torch.distributed process group
    the caller initialised, a torch tensor on `device` out.  At world == 1 with neither a group nor an initialised torch process
    group there is nothing to gather through: numpy in, numpy out, torch never imported.  `as_numpy=True` asks for the numpy
    path explicitly (world > 1 then needs a SocketGroup: anything else raises)."""
    lo, hi = shard_slabs(nslab, rank, world)
    out = np.ascontiguousarray(np.asarray(process(lo, hi)))
    if as_numpy is None:
        as_numpy = group is not None
        if not as_numpy and world == 1:
            # a single rank needs no carrier at all: stay in numpy unless the caller lives in an initialised torch process group
            import sys
            dist = sys.modules.get('torch.distributed')            # not imported yet: nobody can have initialised a process group
            as_numpy = not (dist is not None and dist.is_available() and dist.is_initialized())
    if as_numpy:
        if world > 1 and group is None:
            raise Exception('run_sharded: as_numpy=True with world > 1 needs a SocketGroup as `group` (numpy blocks cannot travel '
                            'through torch.distributed)')
        return all_gather_slabs(out, nslab, rank, world, group)
    import torch
    t = torch.from_numpy(out)
    if device is not None:
        t = t.to(device)
    return all_gather_slabs(t, nslab, rank, world)
