# -*- coding: utf-8 -*-
"""
Batched, device-resident Keff pipeline (host side of `xc_keff_dev`).

One `KeffPlan` = one batch of independent (time, level) slabs resident in HBM on
one GPU plus every static input (dA, gradient metrics, A(Yeq) table, prescribed
equivalent coordinates) and pre-allocated outputs.  `run()` enqueues the three
kernels of the reference's call sequence (SURVEY 3.1 steps 2-10:
core.py:205-249, 412-460, 1136-1174, 463-488, 619-637, 945-966, 1050-1100,
utils.py:518-534) with no host round trip; `fetch()` copies the results back.

Independent slabs shard across GPUs with no exchange during compute:
`shard_slabs(S, rank, world)` is the static contiguous partition of SURVEY 8(e).
"""
import ctypes as C

import numpy as np

from . import _native as nat
from .utils import Rearth, grad_metrics

OUT_NAMES = ('ctr', 'area', 'intgrdS', 'latEq', 'dqdA', 'dintSdA', 'Leq2', 'Lmin', 'nkeff')
INTERP_ORDER = ('ctr', 'area', 'intgrdS', 'latEq', 'dintSdA', 'dqdA', 'Leq2', 'Lmin', 'nkeff')


def shard_slabs(nslab, rank, world):
    """Contiguous block [lo, hi) of the flattened (time, level) slab index owned
    by `rank` of `world`: ceil(S/G) slabs each, last ranks may be short."""
    per = -(-int(nslab) // int(world))
    lo = min(int(nslab), rank * per)
    hi = min(int(nslab), lo + per)
    return lo, hi


class KeffPlan(object):
    def __init__(self, ctx, nslab, ny, nx, N, q_dtype=np.float64, ctr_dtype=np.float64,
                 dA=None, lat=None, lon=None, rdx=None, rdy=None, periodic_x=True,
                 tbl=None, tbl_coord=None, preY=None, increase=True, lt=True,
                 right_edge='xhistogram', nkeff_mask=1e5, Rearth=Rearth, grdS_dtype=None,
                 prod_f32=False, alloc_q=True, nslots=1, out_ptr=None, detect_row_dA=False,
                 out_slabs=None, replicate_dA=False, deterministic=False, slab_major=False, counts=True, single_read=True):
        """dA: None | (ny,) | (ny,nx) | (nslab,ny,nx) f64 (the last: weights that change with the leading
        (time, level) index, which the reference allows -- core.py:1271-1274).  Gradient metrics either `rdx, rdy`
        (per-row reciprocals) or derived from `lat, lon` (sphere).  If
        `grdS_dtype` is given the squared gradient is an INPUT (set with
        `set_grdS`) instead of being computed in-kernel.
        `nslots`: number of result slots (`run(slot=k)` writes slot k, so a long job keeps
        every step's vectors on the device until one gather at the end); `out_ptr`: use a
        caller-owned device allocation of `out_bytes(…) * nslots` bytes for them.
        `detect_row_dA`: a 2-D dA whose rows are constant (every regular lat-lon `rA`) is passed
        to the kernels as its first column (identical results, 8 B/cell less traffic).
        `replicate_dA`: store a (ny,nx) dA once PER SLAB on the device and run the per-slab-weights path
        (XC_DA_SLAB) on it -- what a time-varying metric costs, without a (nslab,ny,nx) host array.
        `slab_major`: lay the nine result vectors of a slot out as ONE [out_slabs][9][N] block (pipeline.OUT_NAMES order,
        xc_keff_desc.out_stride = 9 N) instead of nine [out_slabs][N] arrays: the head of the slot IS the block a rank hands to
        the one gather at the end of a job, no repacking pass; `head_bytes` is its size.
        `counts=False`: the per-bin cell counts are not wanted (the reference's Keff sequence never looks at them): the histogram
        pass then skips their LDS adds -- a third of its atomics; `fetch()['counts']` is meaningless.
        `deterministic`: order-free fixed-point sums (xc_keff_desc.deterministic): area / intgrdS and everything derived
        from them are bit-identical between runs, launch-set sizes and ranks, chained (q_next) or not.
        `single_read=False`: launch sets of ONE slab keep the min/max + histogram + finalize chain instead of the
        single-read kernel ('force': launch sets of two slabs take that kernel too -- slower than the chain there, for tests) (xc_keff_desc.single_read; that kernel needs the whole GPU to itself -- `fetch` repeats a launch set
        that came back with status 2 on the chain, so the switch is for measurements, not for safety)."""
        self.ctx = ctx
        self.nslab, self.ny, self.nx, self.N = int(nslab), int(ny), int(nx), int(N)
        self.q_dtype, self.ctr_dtype = np.dtype(q_dtype), np.dtype(ctr_dtype)
        self.npre = 0 if preY is None else len(preY)
        self._keep = []
        d = nat.KeffDesc()
        self.desc = d
        cells = self.nslab * self.ny * self.nx
        self.q_buf = ctx.alloc(cells * self.q_dtype.itemsize) if alloc_q else None
        self._q_ptr = self.q_buf.ptr if alloc_q else 0
        d.q = self._q_ptr if alloc_q else None
        d.q_dtype, d.ctr_dtype = nat.dtype_code(self.q_dtype), nat.dtype_code(self.ctr_dtype)
        d.nslab, d.ny, d.nx, d.N = self.nslab, self.ny, self.nx, self.N
        d.increase, d.lt = int(bool(increase)), int(bool(lt))
        d.right_edge = nat.XC_EDGE_XHISTOGRAM if right_edge == 'xhistogram' else nat.XC_EDGE_NUMPY
        d.deterministic = 1 if deterministic else 0
        d.single_read = nat.XC_SINGLE_FORCE if single_read == 'force' else (nat.XC_SINGLE_AUTO if single_read else nat.XC_SINGLE_NEVER)
        self._runs = {}                                  # slot -> the launch sets enqueued into it since its last fetch
        self.replays = 0                                 # launch sets repeated on the chain after a status 2
        if dA is None:
            d.dA, d.dA_rank = None, nat.XC_DA_NONE
        else:
            dA = np.ascontiguousarray(dA, dtype=np.float64)
            if detect_row_dA and dA.shape == (self.ny, self.nx) and bool((dA == dA[:, :1]).all()):
                dA = np.ascontiguousarray(dA[:, 0])
            if dA.shape == (self.ny,):
                d.dA_rank = nat.XC_DA_ROW
            elif dA.shape == (self.ny, self.nx) and replicate_dA:
                d.dA_rank = nat.XC_DA_SLAB
            elif dA.shape == (self.ny, self.nx):
                d.dA_rank = nat.XC_DA_PLANE
            elif dA.ndim == 3 and dA.shape[1:] == (self.ny, self.nx) and 1 <= dA.shape[0] <= self.nslab:
                d.dA_rank = nat.XC_DA_SLAB                         # (fewer planes than slabs: the rest is uploaded later, batch by batch)
            else:
                raise Exception('dA must be (ny,), (ny,nx) or (nslab,ny,nx)')
            if d.dA_rank == nat.XC_DA_SLAB and dA.ndim == 2:
                self.dA_buf = ctx.alloc(self.nslab * dA.nbytes)
                for s_ in range(self.nslab):
                    ctx._check(ctx.lib.xc_memcpy_h2d(ctx.handle, self.dA_buf.ptr + s_ * dA.nbytes, dA.ctypes.data, dA.nbytes))
            elif d.dA_rank == nat.XC_DA_SLAB:
                self.dA_buf = ctx.alloc(self.nslab * self.ny * self.nx * 8).upload(dA)
            else:
                self.dA_buf = ctx.to_device(dA)
            self._dA_ptr = self.dA_buf.ptr
            d.dA = self._dA_ptr
            d.dA_pos_finite = int(bool(np.isfinite(dA).all() and (dA >= 0).all()))   # static metric: checked once
            fin = np.abs(dA[np.isfinite(dA)])
            d.dA_max = float(fin.max()) if fin.size else 0.0          # (deterministic sums: the window of the dA channel's accumulator)
            if d.dA_rank == nat.XC_DA_SLAB and dA.ndim == 3 and dA.shape[0] < self.nslab:
                d.dA_max = 0.0                                        # planes uploaded later: the library looks at the device array itself
        if grdS_dtype is None:
            if rdx is None:
                rdx, rdy = grad_metrics(lat, lon, Rearth)
            self.rdx_buf = ctx.to_device(np.asarray(rdx, dtype=np.float64))
            self.rdy_buf = ctx.to_device(np.asarray(rdy, dtype=np.float64))
            d.grad, d.rdx, d.rdy = 1, self.rdx_buf.ptr, self.rdy_buf.ptr
            self.grdS_buf = None
            self._g_ptr = 0
        else:
            self.grdS_dtype = np.dtype(grdS_dtype)
            self.grdS_buf = ctx.alloc(cells * self.grdS_dtype.itemsize)
            self._g_ptr = 0
            d.grad, d.grdS, d.grdS_dtype = 0, self.grdS_buf.ptr, nat.dtype_code(self.grdS_dtype)
        d.prod_f32 = int(bool(prod_f32))
        d.periodic_x = int(bool(periodic_x))
        # xc_keff_dev reads exactly ny table entries in ascending-coordinate order (include/xcontour_hip.h)
        tbl = np.asarray(tbl, dtype=np.float64)
        tbl_coord = np.asarray(tbl_coord, dtype=np.float64)
        if tbl.shape != (self.ny,) or tbl_coord.shape != (self.ny,):
            raise Exception('the A(Yeq) table and its coordinate must have length ny = %d (got %r, %r)'
                            % (self.ny, tbl.shape, tbl_coord.shape))
        if not (np.diff(tbl_coord) > 0).all():
            if (np.diff(tbl_coord) < 0).all():
                tbl, tbl_coord = tbl[::-1], tbl_coord[::-1]          # e.g. a table from cal_area_eqCoord_table on a descending coordinate
            else:
                raise Exception('the table coordinate must be strictly monotonic')
        self.tbl_buf = ctx.to_device(tbl)
        self.coord_buf = ctx.to_device(tbl_coord)
        d.tbl, d.tbl_coord = self.tbl_buf.ptr, self.coord_buf.ptr
        d.npre = self.npre
        if self.npre:
            self.pre_buf = ctx.to_device(np.asarray(preY, dtype=np.float64))
            d.preY = self.pre_buf.ptr
        d.nkeff_mask = float(nkeff_mask)
        d.lmin_scale = float(2.0 * np.pi * Rearth)
        # outputs: one allocation, [9][out_slabs][N] f64 | counts | interp | status
        # (out_slabs < nslab: the tracer buffer holds several batches, a slot one batch)
        self.out_slabs = self.nslab if out_slabs is None else int(out_slabs)
        self.slab_major = bool(slab_major)
        self.want_counts = bool(counts)
        nN = self.out_slabs * self.N
        self._off = {}
        off = 0
        for i, name in enumerate(OUT_NAMES):
            self._off[name] = i * self.N * 8 if self.slab_major else off
            off += nN * 8
        self.head_bytes = off                                       # the nine vectors of every slab of the slot
        self._vstep = (9 if self.slab_major else 1) * self.N * 8    # bytes from one slab to the next inside a vector output
        d.out_stride = 9 * self.N if self.slab_major else 0
        self._off['counts'] = off
        off += nN * 8
        self._off['interp'] = off
        off += self.out_slabs * 9 * self.npre * 8
        self._off['status'] = off
        off += self.out_slabs * 4
        self.slot_bytes = (off + 255) & ~255
        self.nslots = int(nslots)
        if out_ptr is None:
            self.out_buf = ctx.alloc(self.slot_bytes * self.nslots)
            self.out_ptr = self.out_buf.ptr
        else:
            self.out_buf = None
            self.out_ptr = int(out_ptr)
        self._point(0, 0, self.nslab)

    @staticmethod
    def out_bytes(nslab, N, npre=0):
        """bytes of one result slot (so that a caller can own the allocation); the same for both layouts"""
        off = 9 * nslab * N * 8 + nslab * N * 8 + nslab * 9 * npre * 8 + nslab * 4
        return (off + 255) & ~255

    def _point(self, slot, s0, n, out_s0=None):
        """aim the descriptor at slabs [s0, s0+n) and result slot `slot` (results land at slab
        index `out_s0` of the slot, default s0)"""
        d = self.desc
        o0 = s0 if out_s0 is None else out_s0
        base = self.out_ptr + slot * self.slot_bytes
        for name in OUT_NAMES:
            setattr(d, name, base + self._off[name] + o0 * self._vstep)
        d.counts = (base + self._off['counts'] + o0 * self.N * 8) if self.want_counts else None
        d.interp = (base + self._off['interp'] + o0 * 9 * self.npre * 8) if self.npre else None
        d.status = base + self._off['status'] + o0 * 4
        d.nslab = n
        d.q = self._q_ptr + s0 * self.ny * self.nx * self.q_dtype.itemsize
        if d.dA_rank == nat.XC_DA_SLAB:
            d.dA = self._dA_ptr + s0 * self.ny * self.nx * 8
        if self.grdS_buf is not None:
            d.grdS = (self._g_ptr or self.grdS_buf.ptr) + s0 * self.ny * self.nx * self.grdS_dtype.itemsize

    # -- inputs
    def touch(self):
        """Tell the library that the tracer bytes changed behind its back (a caller writing through its own device
        pointer): bumps `xc_keff_desc.q_gen`, which invalidates min/max partials chained from an earlier call."""
        self.desc.q_gen = (self.desc.q_gen + 1) & 0x7fffffff

    def set_q(self, q):
        q = np.ascontiguousarray(q, dtype=self.q_dtype).reshape(self.nslab, self.ny, self.nx)
        self.q_buf.upload(q)
        self.touch()

    def set_q_device(self, ptr):
        """Use an existing device pointer ([nslab][ny][nx], q_dtype) as the tracer.  (Chained min/max partials are
        keyed on the pointer: pointing at OTHER resident data needs no `touch()`; new contents behind the SAME
        pointer, written by someone else than this library, do.)"""
        self._q_ptr = int(ptr)
        self.desc.q = self._q_ptr

    def set_grdS_device(self, ptr):
        """Use an existing device pointer ([nslab][ny][nx], grdS_dtype) as the supplied squared gradient; 0 / None: the plan's own buffer again."""
        self._g_ptr = int(ptr or 0)

    def set_dA_device(self, ptr):
        """Use an existing device pointer as dA (same rank and shape as the dA given to the constructor)."""
        self._dA_ptr = int(ptr)
        self.desc.dA = self._dA_ptr
        self.desc.dA_max = 0.0                       # unknown contents: deterministic sums take the bound from the device array

    def set_grdS(self, g):
        g = np.ascontiguousarray(g, dtype=self.grdS_dtype).reshape(self.nslab, self.ny, self.nx)
        self.grdS_buf.upload(g)

    def synth(self, lat, lon, seed, variant=0):
        """Fill the tracer batch on device with the bench's synthetic PV-like slabs."""
        lat_b = self.ctx.to_device(np.asarray(lat, dtype=np.float64))
        lon_b = self.ctx.to_device(np.asarray(lon, dtype=np.float64))
        self.ctx._check(self.ctx.lib.xc_synth_dev(self.ctx.handle, self._q_ptr, self.desc.q_dtype,
                                                  self.nslab, self.ny, self.nx, lat_b.ptr, lon_b.ptr,
                                                  int(seed), int(variant)))
        self.ctx.sync()
        lat_b.free()
        lon_b.free()
        self.touch()

    # -- compute
    def run(self, slot=0, group=None, chain=False):
        """Enqueue min/max -> histogram -> finalize+epilogue on the context's stream for all
        slabs of the batch, `group` slabs per launch set (None: the whole batch at once).
        `chain=True`: tell each launch set which slabs come NEXT (the following group, or this
        batch again on the next call) so that their min/max is accumulated inside this
        histogram pass and the separate K1 pass disappears (xc_keff_desc.q_next); only valid
        while the tracer batch is not modified between calls."""
        g = self.nslab if not group else int(group)
        starts = list(range(0, self.nslab, g))
        esz = self.ny * self.nx * self.q_dtype.itemsize
        self._runs[slot] = []
        for i, s0 in enumerate(starts):
            n = min(g, self.nslab - s0)
            self._point(slot, s0, n)
            nxt = starts[(i + 1) % len(starts)]
            ok = chain and min(g, self.nslab - nxt) == n          # same shape only
            self.desc.q_next = (self._q_ptr + nxt * esz) if ok else None
            self._enqueue(slot, s0, n, None)

    def run_range(self, slot, s0, n, next_s0=None, out_s0=None):
        """One launch set over slabs [s0, s0+n) into result slot `slot`; `next_s0`: first slab of
        the launch set that will run next (its min/max rides along, xc_keff_desc.q_next);
        `out_s0`: slab index inside the slot where the results go (default s0)."""
        self._point(slot, s0, n, out_s0)
        esz = self.ny * self.nx * self.q_dtype.itemsize
        self.desc.q_next = (self._q_ptr + next_s0 * esz) if next_s0 is not None else None
        self._runs.setdefault(slot, [])
        self._enqueue(slot, s0, n, out_s0)

    def _enqueue(self, slot, s0, n, out_s0):
        self.ctx._check(self.ctx.lib.xc_keff_dev(self.ctx.handle, C.byref(self.desc)))
        if n <= 2 and self.desc.single_read != nat.XC_SINGLE_NEVER:
            # a launch set the single-read kernel may have taken: remember where it read and wrote, in case it comes back with status 2
            runs = self._runs[slot]
            runs.append((s0, n, out_s0, self._q_ptr, self._dA_ptr if self.desc.dA_rank != nat.XC_DA_NONE else 0, self._g_ptr))
            del runs[:-64]

    def _replay(self, slot, status):
        """launch sets of `slot` whose slabs came back with status 2 (the single-read kernel gave up waiting for its workgroups:
        something else held compute units): once more, on the min/max + histogram + finalize chain"""
        keep = (self._q_ptr, getattr(self, '_dA_ptr', 0), self._g_ptr, self.desc.single_read)
        done = False
        try:
            self.desc.single_read = nat.XC_SINGLE_NEVER
            for s0, n, out_s0, qp, dp, gp in self._runs.get(slot, []):
                o0 = s0 if out_s0 is None else out_s0
                if not (status[o0:o0 + n] == 2).any():
                    continue
                self._q_ptr = qp
                if dp:
                    self._dA_ptr = dp
                    self.desc.dA = dp                      # (per-slab weights: _point moves it to the launch set's first slab)
                self._g_ptr = gp
                self._point(slot, s0, n, out_s0)
                self.desc.q_next = None
                self.ctx._check(self.ctx.lib.xc_keff_dev(self.ctx.handle, C.byref(self.desc)))
                self.replays += 1
                done = True
        finally:
            self._q_ptr, dp, self._g_ptr, self.desc.single_read = keep
            if dp:
                self._dA_ptr = dp
                self.desc.dA = dp
            self.desc.q = self._q_ptr
        return done

    def unpack(self, raw):
        """one result slot (bytes as a uint8 ndarray) -> dict of arrays"""
        S, N = self.out_slabs, self.N
        out = {}
        if self.slab_major:
            blk = raw[:S * 9 * N * 8].view(np.float64).reshape(S, 9, N)
            for i, name in enumerate(OUT_NAMES):
                out[name] = blk[:, i, :]
        else:
            for name in OUT_NAMES:
                out[name] = raw[self._off[name]:self._off[name] + S * N * 8].view(np.float64).reshape(S, N)
        out['counts'] = raw[self._off['counts']:self._off['counts'] + S * N * 8].view(np.uint64).reshape(S, N)
        if self.npre:
            it = raw[self._off['interp']:self._off['interp'] + S * 9 * self.npre * 8].view(np.float64).reshape(S, 9, self.npre)
            for i, name in enumerate(INTERP_ORDER):
                out[name + '_eq'] = it[:, i, :]
        out['status'] = raw[self._off['status']:self._off['status'] + S * 4].view(np.int32)
        return out

    def fetch(self, check=True, slot=0):
        # (xc_memcpy_d2h enqueues the copy on the compute stream behind the kernels and waits for it: ONE host wait per fetch;
        # a ctx.sync() in front of it was a second one, ~20 us per call at the reference's demo sizes)
        raw = np.empty(self.slot_bytes, dtype=np.uint8)
        self.ctx._check(self.ctx.lib.xc_memcpy_d2h(self.ctx.handle, raw.ctypes.data,
                                                   self.out_ptr + slot * self.slot_bytes, self.slot_bytes))
        out = self.unpack(raw)
        if (out['status'] == 2).any() and self._replay(slot, out['status']):
            self.ctx._check(self.ctx.lib.xc_memcpy_d2h(self.ctx.handle, raw.ctypes.data,
                                                       self.out_ptr + slot * self.slot_bytes, self.slot_bytes))
            out = self.unpack(raw)
        self._runs[slot] = []
        if check and (out['status'] == 2).any():
            raise Exception('xc_keff: status 2 -- the single-read kernel gave up waiting for its workgroups and the launch set is not on record')
        if check and out['status'].any():
            raise Exception('non monotonic bins')          # reference core.py:1233-1251
        return out

    def download_q(self):
        return self.q_buf.download((self.nslab, self.ny, self.nx), self.q_dtype)

    def free(self):
        for name in ('q_buf', 'dA_buf', 'rdx_buf', 'rdy_buf', 'grdS_buf', 'tbl_buf', 'coord_buf', 'pre_buf', 'out_buf'):
            b = getattr(self, name, None)
            if b is not None:
                b.free()
