# -*- coding: utf-8 -*-
"""
ctypes binding of libxcontour_hip.so (C ABI: include/xcontour_hip.h).

There is NO CPU fallback: if the shared library is missing or no gfx950 device
is visible, every compute entry point raises.  `load()` only dlopen()s the
library (works on a GPU-less build box so that symbol / ABI checks can run);
`Context()` is what needs the device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('XC_LIB_PATH') or os.path.join(_HERE, 'libxcontour_hip.so')   # XC_LIB_PATH: diagnostic builds only

XC_OK, XC_EBADARG, XC_EEDGES, XC_EHIP, XC_ENOMEM, XC_ENODEV = 0, -1, -2, -3, -4, -5
XC_F32, XC_F64 = 0, 1
XC_DA_NONE, XC_DA_ROW, XC_DA_PLANE, XC_DA_SLAB = 0, 1, 2, 3
XC_EDGE_NUMPY, XC_EDGE_XHISTOGRAM = 0, 1
XC_SINGLE_AUTO, XC_SINGLE_NEVER, XC_SINGLE_FORCE = 0, 1, 2
XC_MAX_INTEGRANDS = 2
MAX_SLABS_PER_LAUNCH = 65535
XC_PAD_EDGE, XC_PAD_WRAP, XC_PAD_NAN, XC_PAD_REFLECT, XC_PAD_SYMMETRIC = 0, 1, 2, 3, 4
PAD_MODES = {'edge': XC_PAD_EDGE, 'wrap': XC_PAD_WRAP, 'constant': XC_PAD_NAN, 'reflect': XC_PAD_REFLECT,
             'symmetric': XC_PAD_SYMMETRIC}

_vp, _i32, _i64, _u64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double


class HistDesc(C.Structure):
    """struct xc_hist_desc (include/xcontour_hip.h)"""
    _fields_ = [
        ('q', _vp), ('q_dtype', _i32), ('dA_pos_finite', _i32),
        ('nslab', _i64), ('ny', _i64), ('nx', _i64),
        ('edges', _vp), ('nedge', _i64), ('edges_per_slab', _i32), ('last_closed', _i32),
        ('dA', _vp), ('dA_rank', _i32), ('prod_f32', _i32),
        ('nint', _i32), ('grad', _i32),
        ('integrand', _vp * XC_MAX_INTEGRANDS),
        ('integrand_dtype', _i32 * XC_MAX_INTEGRANDS),
        ('rdx', _vp), ('rdy', _vp),
        ('periodic_x', _i32), ('lt', _i32),
        ('reverse', _i32), ('negate', _i32),
        ('pdf', _vp), ('counts', _vp), ('cdf', _vp),
        ('deterministic', _i32), ('reserved0', _i32),
    ]


class CommInfo(C.Structure):
    """struct xc_comm_info_t (include/xcontour_hip.h)"""
    _fields_ = [('comm_count', _i32), ('comm_rank', _i32), ('comm_device', _i32), ('ctx_device', _i32),
                ('rccl_version', _i32), ('reserved0', _i32), ('rccl_path', C.c_char * 256)]


class KeffDesc(C.Structure):
    """struct xc_keff_desc (include/xcontour_hip.h)"""
    _fields_ = [
        ('q', _vp), ('q_dtype', _i32), ('ctr_dtype', _i32),
        ('nslab', _i64), ('ny', _i64), ('nx', _i64),
        ('N', _i32), ('increase', _i32), ('lt', _i32), ('right_edge', _i32),
        ('dA', _vp), ('dA_rank', _i32), ('grad', _i32),
        ('grdS', _vp), ('grdS_dtype', _i32), ('prod_f32', _i32),
        ('rdx', _vp), ('rdy', _vp), ('periodic_x', _i32), ('npre', _i32),
        ('tbl', _vp), ('tbl_coord', _vp), ('preY', _vp),
        ('nkeff_mask', _f64), ('lmin_scale', _f64),
        ('ctr', _vp), ('area', _vp), ('intgrdS', _vp), ('latEq', _vp),
        ('dqdA', _vp), ('dintSdA', _vp), ('Leq2', _vp), ('Lmin', _vp), ('nkeff', _vp),
        ('counts', _vp), ('interp', _vp), ('status', _vp), ('q_next', _vp),
        ('dA_pos_finite', _i32), ('q_gen', _i32),
        ('deterministic', _i32), ('out_stride', _i32),
        ('dA_max', _f64),
        ('single_read', _i32), ('reserved0', _i32),
    ]


# name -> (restype, argtypes): every symbol include/xcontour_hip.h declares
PROTOTYPES = {
    'xc_create': (C.c_int, [C.c_int, C.POINTER(_vp)]),
    'xc_destroy': (C.c_int, [_vp]),
    'xc_last_error': (C.c_char_p, [_vp]),
    'xc_version': (C.c_char_p, []),
    'xc_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'xc_device_name': (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    'xc_device_cus': (C.c_int, [_vp, C.POINTER(C.c_int)]),
    'xc_sync': (C.c_int, [_vp]),
    'xc_stream': (_vp, [_vp]),
    'xc_trace': (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    'xc_malloc': (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    'xc_free': (C.c_int, [_vp, _vp]),
    'xc_memcpy_h2d': (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    'xc_memcpy_d2h': (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    'xc_memset': (C.c_int, [_vp, _vp, C.c_int, C.c_size_t]),
    'xc_memcpy_h2d_async': (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    'xc_stream_wait_copies': (C.c_int, [_vp]),
    'xc_keep_resident': (C.c_int, [_vp, _vp, C.c_size_t]),
    'xc_release_resident': (C.c_int, [_vp, _vp]),
    'xc_resident_lookup': (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(_vp)]),
    'xc_copies_wait_stream': (C.c_int, [_vp]),
    'xc_event_create': (C.c_int, [_vp, C.POINTER(_vp)]),
    'xc_event_destroy': (C.c_int, [_vp, _vp]),
    'xc_event_record': (C.c_int, [_vp, _vp]),
    'xc_event_elapsed_ms': (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_float)]),
    'xc_event_record_copies': (C.c_int, [_vp, _vp]),
    'xc_event_query': (C.c_int, [_vp, _vp, C.POINTER(C.c_int)]),
    'xc_minmax_dev': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _vp]),
    'xc_minmax': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _vp]),
    'xc_levels_dev': (C.c_int, [_vp, _vp, C.c_int, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    'xc_levels': (C.c_int, [_vp, _vp, C.c_int, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    'xc_contours': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    'xc_hist_dev': (C.c_int, [_vp, C.POINTER(HistDesc)]),
    'xc_hist': (C.c_int, [_vp, C.POINTER(HistDesc)]),
    'xc_rowsum_dev': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _i64, _i64, C.c_int, _vp]),
    'xc_rowsum': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _i64, _i64, C.c_int, _vp]),
    'xc_grad2_dev': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, _vp, _vp, C.c_int, _vp]),
    'xc_grad2': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, _vp, _vp, C.c_int, _vp]),
    'xc_lwa_dev': (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, _f64, _vp, C.c_int,
                             _i64, _i64, _i64, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    'xc_lwa': (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, _f64, _vp, C.c_int,
                         _i64, _i64, _i64, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    'xc_crossing_dev': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, C.c_int, C.c_int, _vp, C.c_int, C.c_int,
                                  _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'xc_crossing': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, C.c_int, C.c_int, _vp, C.c_int, C.c_int,
                              _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'xc_sort_profile_dev': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _i64, _i64, C.c_int, _vp, C.c_int,
                                      _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'xc_sort_profile': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _i64, _i64, C.c_int, _vp, C.c_int,
                                  _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'xc_sort_profile_batch_dev': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _i64, _i64, _i64, C.c_int,
                                            _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'xc_sort_profile_batch': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _i64, _i64, _i64, C.c_int,
                                        _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'xc_last_sort_path': (C.c_int, [_vp, C.POINTER(C.c_int)]),
    'xc_last_keff_path': (C.c_int, [_vp, C.POINTER(C.c_int)]),
    'xc_dbg_single_stamps': (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_int)]),
    'xc_set_lwa_exact': (C.c_int, [_vp, C.c_int]),
    'xc_last_lwa_path': (C.c_int, [_vp, C.POINTER(C.c_int)]),
    'xc_keff_dev': (C.c_int, [_vp, C.POINTER(KeffDesc)]),
    'xc_keff_epilogue_dev': (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int,
                                       C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'xc_keff_epilogue': (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int,
                                   C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'xc_host_gradient_wrt_area': (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _i64, _i64, _i64, _i64, _i64, _vp]),
    'xc_host_edges_from_levels': (C.c_int, [_vp, C.c_int, _i64, _i64, C.c_int, _vp, C.POINTER(C.c_int)]),
    'xc_set_kernel_timing': (C.c_int, [_vp, C.c_int]),
    'xc_last_hist_ms': (C.c_int, [_vp, C.POINTER(C.c_float)]),
    'xc_set_hist_events': (C.c_int, [_vp, _vp, _vp]),
    'xc_comm_unique_id': (C.c_int, [_vp, _vp]),
    'xc_comm_init': (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    'xc_comm_create': (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp), C.c_char_p, C.c_size_t]),
    'xc_comm_attach': (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    'xc_comm_release': (C.c_int, [_vp]),
    'xc_comm_info': (C.c_int, [_vp, _vp]),
    'xc_comm_allgather_dev': (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    'xc_comm_finalize': (C.c_int, [_vp]),
    'xc_comm_gather_dev': (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int]),
    'xc_comm_abort': (C.c_int, [_vp]),
    'xc_comm_wait_compute': (C.c_int, [_vp]),
    'xc_compute_wait_comm': (C.c_int, [_vp]),
    'xc_comm_memcpy_d2d': (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    'xc_streams_idle': (C.c_int, [_vp, C.POINTER(C.c_int)]),
    'xc_ipc_export': (C.c_int, [_vp, _vp, _vp]),
    'xc_ipc_open': (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    'xc_ipc_close': (C.c_int, [_vp, _vp]),
    'xc_device_can_access_peer': (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    'xc_synth_dev': (C.c_int, [_vp, _vp, C.c_int, _i64, _i64, _i64, _vp, _vp, _u64, C.c_int]),
}

_lib = None


class XContourHipError(Exception):
    """Raised for every non-zero status of the C ABI (the reference raises bare
    `Exception`, core.py:53-57, 1233-1251; this subclasses it)."""

    def __init__(self, code, msg):
        Exception.__init__(self, msg)
        self.code = code


def load():
    """dlopen the in-tree library and attach prototypes.  Raises (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XContourHipError(
            XC_ENODEV,
            'xcontour_amd: %s not found -- build it with `python -c "import '
            '__graft_entry__ as g; g.build()"` or `make -C xcontour_amd/csrc`. '
            'There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_DT_CODES = {np.dtype(np.float32): XC_F32, np.dtype(np.float64): XC_F64}


def dtype_code(dt):
    c = _DT_CODES.get(dt)                                        # (a np.dtype instance: the usual argument)
    if c is None:
        c = _DT_CODES.get(np.dtype(dt))
        if c is None:
            raise XContourHipError(XC_EBADARG, 'unsupported dtype %s (float32/float64 only)' % np.dtype(dt))
    return c


def _ptr(a):
    """Host address of a C-contiguous ndarray (or None): an int, which ctypes takes for a void* argument or field.  The CALLER keeps the
    array alive until the library call has returned (every entry point here holds its arrays in locals)."""
    if a is None:
        return None
    assert a.flags.c_contiguous
    return a.ctypes.data


def _contig(a, dtype=None):
    """np.ascontiguousarray without the call when there is nothing to do"""
    if type(a) is np.ndarray and a.flags.c_contiguous and (dtype is None or a.dtype == dtype):
        return a
    return np.ascontiguousarray(a, dtype=dtype)


def _is_lazy(q):
    """a labeled.LazyStack (or anything with its protocol): (S, ny, nx) shape / dtype known, slabs read on `q[s0:s1]`"""
    return bool(getattr(q, '_xc_lazy_stack', False))


def _stack_in(q):
    return q if _is_lazy(q) else _contig(q)


def _stack_now(q):
    """the batch is about to be staged on the device: a lazy stack is read now (this one batch, nothing more)"""
    return np.ascontiguousarray(q[:]) if _is_lazy(q) else q


class DeviceBuffer(object):
    """A device allocation owned by a Context (freed with the context or explicitly)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = _vp()
        ctx._check(ctx.lib.xc_malloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value
        ctx._buffers.append(self)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.xc_memcpy_h2d(self.ctx.handle, self.ptr, _ptr(arr), arr.nbytes))
        return self

    def upload_async(self, arr, offset_bytes=0):
        """copy on the context's copy stream (overlaps kernels already enqueued); pair with Context.stream_wait_copies()"""
        arr = np.ascontiguousarray(arr)
        assert offset_bytes + arr.nbytes <= self.nbytes
        # lifetime rule: the source must stay alive until the copy has run.  Pageable sources make hipMemcpyAsync return
        # after the copy today, but that is how the runtime behaves, not a promise (and a pinned source returns at once):
        # the context holds a reference until an event recorded on the copy stream behind this copy has completed
        ctx = self.ctx
        ctx._reap_staged()
        ctx._check(ctx.lib.xc_memcpy_h2d_async(ctx.handle, self.ptr + offset_bytes, _ptr(arr), arr.nbytes))
        ev = ctx._ev_pool.pop() if ctx._ev_pool else ctx.event()
        ctx._check(ctx.lib.xc_event_record_copies(ctx.handle, ev))
        ctx._staged.append((arr, ev))
        return self

    def download(self, shape, dtype, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.xc_memcpy_d2h(self.ctx.handle, _ptr(out), self.ptr + offset_bytes, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.xc_free(self.ctx.handle, self.ptr)
            self.ptr = None
            if self in self.ctx._buffers:
                self.ctx._buffers.remove(self)


class Context(object):
    """One HIP device + stream (xc_create / xc_destroy)."""

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        rc = self.lib.xc_create(int(device), C.byref(h))
        if rc != XC_OK:
            raise XContourHipError(rc, (self.lib.xc_last_error(None) or b'').decode())
        self.handle = h
        self.device = int(device)
        self._buffers = []
        self._resident = {}          # data pointer -> ndarray registered with xc_keep_resident (kept alive here)
        self._staged = []            # (host array, event) of asynchronous uploads in flight (DeviceBuffer.upload_async): dropped once the event has completed
        self._ev_pool = []           # recycled events of completed uploads
        # host-pointer entry points stage at most this many bytes of per-slab data (tracer, integrands, per-slab weights,
        # per-slab outputs) on the device at once: larger stacks go through in batches of whole slabs (the reference's
        # histogram path is lazy / dask-friendly, core.py:158-160, 241-246)
        self.max_batch_bytes = int(os.environ.get('XC_MAX_BATCH_BYTES', 8 << 30))

    # -- plumbing
    def _check(self, rc):
        if rc != XC_OK:
            raise XContourHipError(rc, (self.lib.xc_last_error(self.handle) or b'').decode())

    def close(self):
        if getattr(self, 'handle', None):
            # the per-upload events (in flight and recycled): wait for the copy stream, then destroy them (round-5 advisor: they leaked)
            try:
                self.lib.xc_stream_wait_copies(self.handle)
                self.lib.xc_sync(self.handle)
            except Exception:
                pass
            for ev in [e for _, e in self._staged] + list(self._ev_pool):
                self.lib.xc_event_destroy(self.handle, ev)
            self._staged, self._ev_pool = [], []
            for b in list(self._buffers):
                b.free()
            self.lib.xc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        """wait for the COMPUTE stream.  Uploads in flight on the copy stream are not waited for (a download of batch k must not
        stand behind the upload of batch k + 1): work that needs them says so with stream_wait_copies() before it is enqueued"""
        self._check(self.lib.xc_sync(self.handle))
        self._reap_staged()

    def _reap_staged(self):
        """drop the host arrays of asynchronous uploads whose copy has completed (an event per upload, polled: never blocks)"""
        if not self._staged:
            return
        keep, done = [], C.c_int()
        for arr, ev in self._staged:
            self._check(self.lib.xc_event_query(self.handle, ev, C.byref(done)))
            if done.value:
                self._ev_pool.append(ev)
            else:
                keep.append((arr, ev))
        self._staged = keep

    def stream_wait_copies(self):
        self._check(self.lib.xc_stream_wait_copies(self.handle))

    # -- resident inputs: host arrays with a device mirror (xc_keep_resident)
    def keep_resident(self, arr):
        """upload `arr` (C-contiguous ndarray) once; host-form calls whose input is this array -- or whole leading-index slices of
        it -- read the device mirror from now on.  The context keeps a reference to `arr` while it is registered (its memory
        cannot be recycled under the mirror); do not modify it in place without calling keep_resident again."""
        assert isinstance(arr, np.ndarray) and arr.flags['C_CONTIGUOUS']
        self._check(self.lib.xc_keep_resident(self.handle, _ptr(arr), arr.nbytes))
        self._resident[arr.ctypes.data] = arr
        return arr

    def release_resident(self, arr=None):
        if not getattr(self, 'handle', None):
            return
        if arr is None:
            self._check(self.lib.xc_release_resident(self.handle, None))
            self._resident.clear()
        elif arr.ctypes.data in self._resident:
            self._check(self.lib.xc_release_resident(self.handle, _ptr(arr)))
            del self._resident[arr.ctypes.data]

    def resident_ptr(self, arr):
        """device address of the mirror of `arr` (a C-contiguous ndarray, or a leading-index slice of a registered one), or None"""
        if not self._resident:
            return None
        p = _vp()
        self._check(self.lib.xc_resident_lookup(self.handle, _ptr(arr), arr.nbytes, C.byref(p)))
        return p.value

    def copies_wait_stream(self):
        self._check(self.lib.xc_copies_wait_stream(self.handle))

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.xc_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def device_cus(self):
        n = C.c_int()
        self._check(self.lib.xc_device_cus(self.handle, C.byref(n)))
        return n.value

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 1)).upload(arr)

    def event(self):
        e = _vp()
        self._check(self.lib.xc_event_create(self.handle, C.byref(e)))
        return e

    def record(self, ev):
        self._check(self.lib.xc_event_record(self.handle, ev))

    def elapsed_ms(self, e0, e1):
        ms = C.c_float()
        self._check(self.lib.xc_event_elapsed_ms(self.handle, e0, e1, C.byref(ms)))
        return ms.value

    def last_sort_path(self):
        """K8, last call: 0 key passes only, 1 the three range-key passes sufficed, 2 they failed the check (re-sorted)"""
        p = C.c_int()
        self._check(self.lib.xc_last_sort_path(self.handle, C.byref(p)))
        return p.value

    def last_keff_path(self):
        """last xc_keff_dev call: 0 the min/max + histogram + finalize chain, 1 the single-read kernel (calls of one or two slabs)"""
        p = C.c_int()
        self._check(self.lib.xc_last_keff_path(self.handle, C.byref(p)))
        return p.value

    def single_stamps(self, enable=True):
        """diagnostics: (device pointer, slots) of the single-read kernel's phase stamps; enable=False frees them"""
        ptr, n = _vp(), C.c_int()
        self._check(self.lib.xc_dbg_single_stamps(self.handle, 1 if enable else 0, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def last_lwa_path(self):
        """K7, last call: 0 band walk (bit-exact), 1 the O(ny log ny) interval kernel, 2 its premises failed the check"""
        p = C.c_int()
        self._check(self.lib.xc_last_lwa_path(self.handle, C.byref(p)))
        return p.value

    def set_kernel_timing(self, on):
        self._check(self.lib.xc_set_kernel_timing(self.handle, 1 if on else 0))

    def set_hist_events(self, e0, e1):
        self._check(self.lib.xc_set_hist_events(self.handle, e0, e1))

    def last_hist_ms(self):
        ms = C.c_float()
        self._check(self.lib.xc_last_hist_ms(self.handle, C.byref(ms)))
        return ms.value

    # -- the one collective (RCCL)
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self._check(self.lib.xc_comm_unique_id(self.handle, buf))
        return bytes(buf.raw)

    def comm_init(self, nranks, rank, uid):
        assert len(uid) == 128
        self._check(self.lib.xc_comm_init(self.handle, int(nranks), int(rank), C.create_string_buffer(uid, 128)))

    def comm_create(self, nranks, rank, uid):
        """ncclCommInitRank WITHOUT touching this context (safe in a helper thread under a deadline): returns the communicator handle
        for `comm_attach`, raises with RCCL's text otherwise"""
        assert len(uid) == 128
        h, err = _vp(), C.create_string_buffer(512)
        rc = self.lib.xc_comm_create(self.device, int(nranks), int(rank), C.create_string_buffer(uid, 128), C.byref(h), err, 512)
        if rc != 0:
            raise XContourHipError(rc, err.value.decode('utf-8', 'replace') or 'xc_comm_create failed (%d)' % rc)
        return h.value

    def comm_attach(self, comm, nranks, rank):
        self._check(self.lib.xc_comm_attach(self.handle, comm, int(nranks), int(rank)))

    def comm_release(self, comm):
        self.lib.xc_comm_release(comm)

    def comm_info(self):
        """what RCCL itself reports about this context's communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice /
        ncclGetVersion + the file it was loaded from); comm_count 0: no RCCL communicator here"""
        i = CommInfo()
        self._check(self.lib.xc_comm_info(self.handle, C.byref(i)))
        return {'comm_count': i.comm_count, 'comm_rank': i.comm_rank, 'comm_device': i.comm_device, 'ctx_device': i.ctx_device,
                'rccl_version': i.rccl_version, 'rccl_path': i.rccl_path.decode('utf-8', 'replace')}

    def comm_allgather(self, send_ptr, recv_ptr, bytes_per_rank):
        self._check(self.lib.xc_comm_allgather_dev(self.handle, send_ptr, recv_ptr, int(bytes_per_rank)))

    def comm_finalize(self):
        self._check(self.lib.xc_comm_finalize(self.handle))

    def comm_gather(self, send_ptr, nbytes, recv_ptr, rank_stride, root=0):
        """gather to `root` over RCCL (grouped ncclSend / ncclRecv) on the comm stream: rank r's block lands at recv + r * rank_stride"""
        self._check(self.lib.xc_comm_gather_dev(self.handle, send_ptr, int(nbytes), recv_ptr, int(rank_stride), int(root)))

    def comm_abort(self):
        self._check(self.lib.xc_comm_abort(self.handle))

    # -- the comm stream (blocks leave while the next launch set computes) and the HIP IPC carrier
    def comm_wait_compute(self):
        self._check(self.lib.xc_comm_wait_compute(self.handle))

    def compute_wait_comm(self):
        self._check(self.lib.xc_compute_wait_comm(self.handle))

    def comm_memcpy_d2d(self, dst_ptr, src_ptr, nbytes):
        self._check(self.lib.xc_comm_memcpy_d2d(self.handle, dst_ptr, src_ptr, int(nbytes)))

    def streams_idle(self):
        v = C.c_int()
        self._check(self.lib.xc_streams_idle(self.handle, C.byref(v)))
        return bool(v.value)

    def sync_within(self, seconds, poll=0.002):
        """wait for the compute and comm streams like sync(), but give up after `seconds`: True when they drained"""
        import time
        t_end = time.time() + float(seconds)
        while not self.streams_idle():
            if time.time() > t_end:
                return False
            time.sleep(poll)
        return True

    def ipc_export(self, ptr):
        buf = C.create_string_buffer(64)
        self._check(self.lib.xc_ipc_export(self.handle, ptr, buf))
        return bytes(buf.raw)

    def ipc_open(self, handle):
        assert len(handle) == 64
        p = _vp()
        self._check(self.lib.xc_ipc_open(self.handle, C.create_string_buffer(handle, 64), C.byref(p)))
        return p.value

    def ipc_close(self, ptr):
        self._check(self.lib.xc_ipc_close(self.handle, ptr))

    # -- host-pointer compute entry points (numpy in, numpy out)
    def _batches(self, nslab, per_slab_bytes):
        """[(s0, s1), ...]: the slab axis cut into equal batches of whole slabs whose staged bytes stay below
        `max_batch_bytes` (and below the 65535 slabs one launch takes); one batch if everything fits"""
        b = max(1, min(int(nslab), MAX_SLABS_PER_LAUNCH, int(self.max_batch_bytes) // max(1, int(per_slab_bytes))))
        return [(s0, min(int(nslab), s0 + b)) for s0 in range(0, int(nslab), b)]

    def minmax(self, q):
        """q: (nslab, ny, nx) or (nslab, ncell) f32/f64 -> (nslab, 2) f64"""
        q = _stack_in(q)
        nslab = q.shape[0]
        bt = self._batches(nslab, q.nbytes // max(1, nslab))
        if len(bt) > 1:
            return np.concatenate([self.minmax(q[s0:s1]) for s0, s1 in bt])
        q = _stack_now(q)
        out = np.empty((nslab, 2), dtype=np.float64)
        self._check(self.lib.xc_minmax(self.handle, _ptr(q), dtype_code(q.dtype), nslab,
                                       int(q.size // nslab), _ptr(out)))
        return out

    def contours(self, q, N, increase, ctr_dtype, right_edge=XC_EDGE_XHISTOGRAM, want_minmax=False):
        """cal_contours(int) in one call (xc_contours): q (nslab, ny, nx) -> levels (nslab, N) float64 [, minmax (nslab, 2)]"""
        q = _stack_in(q)
        nslab = q.shape[0]
        bt = self._batches(nslab, q.nbytes // max(1, nslab))
        if len(bt) > 1 or _is_lazy(q):
            mm = self.minmax(q)
            ctr = self.levels(mm, q.dtype, N, increase, ctr_dtype, right_edge)[0]
            return (ctr, mm) if want_minmax else ctr
        ctr = np.empty((nslab, N), dtype=np.float64)
        mm = np.empty((nslab, 2), dtype=np.float64) if want_minmax else None
        self._check(self.lib.xc_contours(self.handle, _ptr(q), dtype_code(q.dtype), nslab, int(q.size // nslab), int(N),
                                         int(bool(increase)), dtype_code(ctr_dtype), int(right_edge), _ptr(mm), _ptr(ctr), None, None))
        return (ctr, mm) if want_minmax else ctr

    def trace(self, reset=True):
        """seconds the host-form calls spent staging inputs / handing results over / waiting for the stream since the last reset"""
        out = (C.c_double * 3)()
        self._check(self.lib.xc_trace(self.handle, 1 if reset else 0, out))
        return {'stage_in_s': out[0], 'hand_over_s': out[1], 'sync_wait_s': out[2]}

    def levels(self, minmax, q_dtype, N, increase, ctr_dtype, right_edge=XC_EDGE_XHISTOGRAM):
        minmax = np.ascontiguousarray(minmax, dtype=np.float64)
        nslab = minmax.shape[0]
        ctr = np.empty((nslab, N), dtype=np.float64)
        edges = np.empty((nslab, N + 1), dtype=np.float64)
        status = np.empty(nslab, dtype=np.int32)
        self._check(self.lib.xc_levels(self.handle, _ptr(minmax), dtype_code(q_dtype), nslab, int(N),
                                       int(bool(increase)), dtype_code(ctr_dtype), int(right_edge),
                                       _ptr(ctr), _ptr(edges), _ptr(status)))
        return ctr, edges, status

    def hist(self, q, edges, dA=None, integrands=(), grad=None, last_closed=True, lt=True,
             reverse=False, prod_f32=False, negate=False, want=('pdf', 'counts', 'cdf'), deterministic=False):
        """q: (nslab, ny, nx); edges: (nedge,) or (nslab, nedge) ascending f64.
        dA: None | (ny,) | (ny,nx) | (nslab,ny,nx) (converted to f64).
        grad: None or (rdx, rdy, periodic_x).  deterministic: order-free fixed-point sums (bit-reproducible).
        Returns dict of requested outputs."""
        q = _stack_in(q)
        assert q.ndim == 3
        nslab, ny, nx = q.shape
        integrands = [v if _is_lazy(v) else np.asarray(v) for v in integrands]     # (nested lists are fine, as for q)
        per = ny * nx * (q.dtype.itemsize + sum(np.dtype(v.dtype).itemsize for v in integrands) + (8 if dA is not None and np.ndim(dA) == 3 else 0))
        bt = self._batches(nslab, per)
        if len(bt) > 1:                                      # more than one launch / one arena takes: batches of whole slabs
            parts = []
            for s0, s1 in bt:
                sl = slice(s0, s1)
                e = np.asarray(edges)
                d3 = dA is not None and np.ndim(dA) == 3
                parts.append(self.hist(q[sl], e[sl] if e.ndim == 2 else e, dA[sl] if d3 else dA,
                                       [v[sl] for v in integrands], grad, last_closed, lt, reverse,
                                       prod_f32, negate, want, deterministic))
            return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        q = _stack_now(q)
        integrands = [_stack_now(v) for v in integrands]
        edges = _contig(edges, np.float64)
        d = HistDesc()
        keep = [q, edges]
        d.q, d.q_dtype = _ptr(q), dtype_code(q.dtype)
        d.nslab, d.ny, d.nx = nslab, ny, nx
        d.edges, d.nedge = _ptr(edges), edges.shape[-1]
        d.edges_per_slab = 1 if edges.ndim == 2 else 0
        if edges.ndim == 2 and edges.shape[0] != nslab:
            raise XContourHipError(XC_EBADARG, 'edges must be (nedge,) or (nslab, nedge)')
        d.last_closed = 1 if last_closed else 0
        if dA is None:
            d.dA, d.dA_rank = None, XC_DA_NONE
        else:
            dA = _contig(dA, np.float64)
            keep.append(dA)
            if dA.shape == (ny,):
                d.dA_rank = XC_DA_ROW
            elif dA.shape == (ny, nx):
                d.dA_rank = XC_DA_PLANE
            elif dA.shape == (nslab, ny, nx):
                d.dA_rank = XC_DA_SLAB
            else:
                raise XContourHipError(XC_EBADARG, 'dA must be (ny,), (ny,nx) or (nslab,ny,nx)')
            d.dA = _ptr(dA)
        d.prod_f32 = 1 if prod_f32 else 0
        d.nint = len(integrands)
        if d.nint > XC_MAX_INTEGRANDS:
            raise XContourHipError(XC_EBADARG, 'at most %d integrands per pass' % XC_MAX_INTEGRANDS)
        for i, v in enumerate(integrands):
            v = _contig(v)
            if v.shape != q.shape:
                raise XContourHipError(XC_EBADARG, 'integrand shape must equal tracer shape')
            keep.append(v)
            d.integrand[i], d.integrand_dtype[i] = v.ctypes.data, dtype_code(v.dtype)
        if grad is not None:
            rdx = np.ascontiguousarray(grad[0], dtype=np.float64)
            rdy = np.ascontiguousarray(grad[1], dtype=np.float64)
            assert rdx.shape == (ny,) and rdy.shape == (ny,)
            keep += [rdx, rdy]
            d.grad, d.rdx, d.rdy, d.periodic_x = 1, _ptr(rdx), _ptr(rdy), 1 if grad[2] else 0
        d.lt, d.reverse, d.negate = 1 if lt else 0, 1 if reverse else 0, 1 if negate else 0
        d.deterministic = 1 if deterministic else 0
        nch, nbin = 1 + d.nint + d.grad, d.nedge - 1
        out = {}
        if 'pdf' in want:
            out['pdf'] = np.empty((nslab, nch, nbin), dtype=np.float64)
            d.pdf = _ptr(out['pdf'])
        if 'cdf' in want:
            out['cdf'] = np.empty((nslab, nch, nbin), dtype=np.float64)
            d.cdf = _ptr(out['cdf'])
        if 'counts' in want:
            out['counts'] = np.empty((nslab, nbin), dtype=np.uint64)
            d.counts = _ptr(out['counts'])
        self._check(self.lib.xc_hist(self.handle, C.byref(d)))
        return out

    def rowsum(self, mask, dA, ny, nx, multiply=False):
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            if mask.dtype not in (np.float32, np.float64):
                mask = mask.astype(np.float64)
            assert mask.shape == (ny, nx)
        rank = XC_DA_NONE
        if dA is not None:
            dA = np.ascontiguousarray(dA, dtype=np.float64)
            rank = XC_DA_ROW if dA.shape == (ny,) else XC_DA_PLANE
            assert dA.shape in ((ny,), (ny, nx))
        out = np.empty(ny, dtype=np.float64)
        self._check(self.lib.xc_rowsum(self.handle, _ptr(mask), dtype_code(mask.dtype) if mask is not None else XC_F64,
                                       _ptr(dA), rank, ny, nx, 1 if multiply else 0, _ptr(out)))
        return out

    def keff_epilogue(self, pdf, ctr, tbl, tbl_coord, increase=True, lt=True, ctr_dtype=np.float64, preY=None,
                      nkeff_mask=1e5, lmin_scale=2.0 * np.pi * 6371200.0):
        """K5 / K6 alone (xc_keff_epilogue): pdf (nslab, 2, N) per-bin sums of dA and integrand * dA in ascending-value bin
        order, ctr (nslab, N) levels in level order -> dict of the Keff vectors (nslab, N) (+ 'interp' (nslab, 9, npre))."""
        pdf = np.ascontiguousarray(pdf, dtype=np.float64); ctr = np.ascontiguousarray(ctr, dtype=np.float64)
        assert pdf.ndim == 3 and pdf.shape[1] == 2 and ctr.shape == (pdf.shape[0], pdf.shape[2])
        nslab, _, N = pdf.shape
        tbl = np.ascontiguousarray(tbl, dtype=np.float64); crd = np.ascontiguousarray(tbl_coord, dtype=np.float64)
        assert tbl.shape == crd.shape and tbl.ndim == 1
        names = ('area', 'intgrdS', 'latEq', 'dqdA', 'dintSdA', 'Leq2', 'Lmin', 'nkeff')
        out = {k: np.empty((nslab, N), dtype=np.float64) for k in names}
        pre = None if preY is None else np.ascontiguousarray(preY, dtype=np.float64)
        interp = None if pre is None else np.empty((nslab, 9, len(pre)), dtype=np.float64)
        self._check(self.lib.xc_keff_epilogue(self.handle, _ptr(pdf), _ptr(ctr), dtype_code(np.dtype(ctr_dtype)), nslab, N,
                                              1 if increase else 0, 1 if lt else 0, _ptr(tbl), _ptr(crd), len(tbl),
                                              _ptr(pre), 0 if pre is None else len(pre), float(nkeff_mask), float(lmin_scale),
                                              *[_ptr(out[k]) for k in names], _ptr(interp)))
        if interp is not None:
            out['interp'] = interp
        return out

    def grad2(self, q, rdx, rdy, periodic_x=True):
        q = _stack_in(q)
        assert q.ndim == 3
        nslab, ny, nx = q.shape
        bt = self._batches(nslab, ny * nx * (q.dtype.itemsize + 8))
        if len(bt) > 1:
            return np.concatenate([self.grad2(q[s0:s1], rdx, rdy, periodic_x) for s0, s1 in bt])
        q = _stack_now(q)
        rdx = np.ascontiguousarray(rdx, dtype=np.float64)
        rdy = np.ascontiguousarray(rdy, dtype=np.float64)
        out = np.empty(q.shape, dtype=np.float64)
        self._check(self.lib.xc_grad2(self.handle, _ptr(q), dtype_code(q.dtype), nslab, ny, nx,
                                      _ptr(rdx), _ptr(rdy), 1 if periodic_x else 0, _ptr(out)))
        return out

    def crossing(self, q, contours, area, stride=1, pad_x=0, pad_mode='edge', full_width=False):
        """Box-counting contour crossing (xc_crossing).  q (nslab, ny, nx) f32/f64; contours (N,) or
        (nslab, N) ASCENDING f64; area (ny, nx) or (nslab, ny, nx) f32/f64.
        Returns (lengths f64 (nslab, N), box counts uint64 (nslab, N))."""
        q = _stack_in(q)
        assert q.ndim == 3
        nslab, ny, nx = q.shape
        contours = np.ascontiguousarray(contours, dtype=np.float64)
        per_slab = contours.ndim == 2
        if per_slab and contours.shape[0] != nslab:
            raise XContourHipError(XC_EBADARG, 'contours must be (N,) or (nslab, N)')
        area = np.ascontiguousarray(area)
        if area.dtype not in (np.float32, np.float64):
            area = area.astype(np.float64)
        if area.shape not in ((ny, nx), (nslab, ny, nx)):
            raise XContourHipError(XC_EBADARG, 'area must be (ny, nx) or (nslab, ny, nx)')
        if pad_mode not in PAD_MODES:
            raise XContourHipError(XC_EBADARG, 'pad mode must be one of %s' % sorted(PAD_MODES))
        N = contours.shape[-1]
        bt = self._batches(nslab, ny * nx * (q.dtype.itemsize + (area.dtype.itemsize if area.ndim == 3 else 0)))
        if len(bt) > 1:
            parts = [self.crossing(q[s0:s1], contours[s0:s1] if per_slab else contours, area[s0:s1] if area.ndim == 3 else area,
                                   stride, pad_x, pad_mode, full_width) for s0, s1 in bt]
            if np.ndim(stride) > 0:
                return [(np.concatenate([p[i][0] for p in parts]), np.concatenate([p[i][1] for p in parts])) for i in range(len(parts[0]))]
            return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        q = _stack_now(q)
        if np.ndim(stride) > 0:
            # several strides on the same padded slab: one upload, one device call per stride
            # (`stride` may be a list; returns lists of results in the same order)
            for t in stride:
                if int(t) < 1:
                    raise XContourHipError(XC_EBADARG, 'stride must be >= 1')
            # the host entry point validates the contours; the device one trusts its caller
            c2 = contours.reshape(-1, N)
            if np.isnan(c2).any() or (np.diff(c2, axis=1) < 0).any():
                raise XContourHipError(XC_EEDGES, 'xc_crossing: contours must be ascending without NaN')
            bufs = [self.to_device(q), self.to_device(contours), self.to_device(area),
                    self.alloc(nslab * N * 8), self.alloc(nslab * N * 8)]
            try:
                out = []
                for t in stride:
                    self._check(self.lib.xc_crossing_dev(self.handle, bufs[0].ptr, dtype_code(q.dtype), nslab, ny, nx, int(pad_x),
                                                         PAD_MODES[pad_mode], bufs[1].ptr, N, 1 if per_slab else 0,
                                                         bufs[2].ptr, dtype_code(area.dtype), 1 if area.ndim == 3 else 0,
                                                         int(t), 1 if full_width else 0, bufs[3].ptr, bufs[4].ptr))
                    out.append((bufs[3].download((nslab, N), np.float64), bufs[4].download((nslab, N), np.uint64)))
            finally:
                for b in bufs:
                    b.free()
            return out
        lens = np.empty((nslab, N), dtype=np.float64)
        cnts = np.empty((nslab, N), dtype=np.uint64)
        self._check(self.lib.xc_crossing(self.handle, _ptr(q), dtype_code(q.dtype), nslab, ny, nx, int(pad_x),
                                         PAD_MODES[pad_mode], _ptr(contours), N, 1 if per_slab else 0,
                                         _ptr(area), dtype_code(area.dtype), 1 if area.ndim == 3 else 0,
                                         int(stride), 1 if full_width else 0, _ptr(lens), _ptr(cnts)))
        return lens, cnts

    def lwa(self, q, Q, coord, dA, dA_max, M=None, increase=True, part=0, mask_idx=None, variant=0, exact=None):
        """`exact`: None (default) -- planes of up to 512 rows are summed in numpy's own order (bit-exact band walk), larger ones by
        the O(ny log ny) interval kernel when the reference state is monotone (checked on the device); True -- the band walk for
        every plane; False -- the interval kernel for every plane whose premises hold: they are checked HERE, on the host copy of
        Q, the coordinate and the tracer (Q finite, s Q non-decreasing, the coordinate strictly monotone, no infinite tracer
        cell -- NaN cells are fine), and vouched for to the library,
        so the call is one launch (xc_set_lwa_exact modes 0 / 1 / 3); agreement ~1e-13 of the plane's largest value."""
        q = _stack_in(q)
        assert q.ndim == 3
        nslab, ny, nx = q.shape
        Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(nslab, ny)
        bt = self._batches(nslab, ny * nx * (q.dtype.itemsize + 8 + (0 if mask_idx is None else len(mask_idx))))
        if len(bt) > 1:
            parts = [self.lwa(q[s0:s1], Q[s0:s1], coord, dA, dA_max, M, increase, part, mask_idx, variant, exact) for s0, s1 in bt]
            return (np.concatenate([p[0] for p in parts]),
                    None if parts[0][1] is None else np.concatenate([p[1] for p in parts]))
        q = _stack_now(q)
        coord = np.ascontiguousarray(coord, dtype=np.float64)
        dA = np.ascontiguousarray(dA, dtype=np.float64)
        dr = XC_DA_ROW if dA.shape == (ny,) else XC_DA_PLANE
        assert dA.shape in ((ny,), (ny, nx))
        mr = XC_DA_NONE
        if M is not None:
            M = np.ascontiguousarray(M, dtype=np.float64)
            mr = XC_DA_ROW if M.shape == (ny,) else XC_DA_PLANE
            assert M.shape in ((ny,), (ny, nx))
        out = np.empty(q.shape, dtype=np.float64)
        nmask = 0 if mask_idx is None else len(mask_idx)
        mi = np.ascontiguousarray(mask_idx, dtype=np.int32) if nmask else None
        mo = np.empty((nslab, nmask, ny, nx), dtype=np.int8) if nmask else None
        mode = 0 if exact is None else (1 if exact else 0)
        if exact is not None and not exact and variant == 0:
            sg = 1.0 if increase else -1.0
            c64 = np.asarray(coord, dtype=np.float64)
            dq, dc = np.diff(sg * Q, axis=1), np.diff(c64)
            # FINITE, not only NaN-free: an infinite Q_j would make (Q'_j - c) * S0 = inf * 0 = NaN where the reference sums to 0
            ok = bool(np.isfinite(Q).all()) and bool((dq >= 0).all()) and bool((dc > 0).all() or (dc < 0).all()) and not bool(np.isinf(q).any())
            mode = 3 if ok else 1
        self._check(self.lib.xc_set_lwa_exact(self.handle, mode))
        try:
            self._check(self.lib.xc_lwa(self.handle, _ptr(q), dtype_code(q.dtype), _ptr(Q), _ptr(coord),
                                        _ptr(dA), dr, float(dA_max), _ptr(M), mr, nslab, ny, nx,
                                        1 if increase else 0, int(part), int(variant), _ptr(mi), nmask, _ptr(out), _ptr(mo)))
        finally:
            self.lib.xc_set_lwa_exact(self.handle, 0)
        return out, mo



    def sort_profile(self, q, dA=None, mask=None, targets=None, tbl=None, coord=None,
                     want_sorted=False, want_acum=False, negate=False):
        """Exact adiabatic rearrangement (xc_sort_profile_batch).  q (ny, nx): one plane -> scalars / 1-D arrays
        as before; q (nslab, ny, nx): a stack sorted by ONE set of launches -> leading slab dim on every
        output ('nvalid' (nslab,), 'Q' (nslab, J), 'q_sorted' / 'acum' (nslab, ny*nx), 'bpe' (nslab,)).
        dA: None | (ny,) | (ny, nx) | (nslab, ny, nx); mask: (ny, nx) or (nslab, ny, nx)."""
        q = _stack_in(q)
        single = q.ndim == 2
        if single:
            q = q[None]
        assert q.ndim == 3
        nslab, ny, nx = q.shape
        # staged per slab: the tracer, per-slab mask / dA, the requested full-length outputs and the sort's own four work arrays
        per = ny * nx * (q.dtype.itemsize + 32 + (8 if want_sorted else 0) + (8 if want_acum else 0) +
                         (8 if dA is not None and np.ndim(dA) == 3 else 0) + (8 if mask is not None and np.ndim(mask) == 3 else 0))
        bt = self._batches(nslab, per)
        if len(bt) > 1:
            parts = [self.sort_profile(q[s0:s1], dA[s0:s1] if dA is not None and np.ndim(dA) == 3 else dA,
                                       mask[s0:s1] if mask is not None and np.ndim(mask) == 3 else mask,
                                       targets, tbl, coord, want_sorted, want_acum, negate) for s0, s1 in bt]
            return {k: np.concatenate([np.atleast_1d(p[k]) for p in parts]) for k in parts[0]}
        q = _stack_now(q)
        rank = XC_DA_NONE
        if dA is not None:
            dA = np.ascontiguousarray(dA, dtype=np.float64)
            if dA.shape == (ny,):
                rank = XC_DA_ROW
            elif dA.shape == (ny, nx):
                rank = XC_DA_PLANE
            elif dA.shape == (nslab, ny, nx):
                rank = XC_DA_SLAB
            else:
                raise XContourHipError(XC_EBADARG, 'dA must be (ny,), (ny, nx) or (nslab, ny, nx)')
        per_slab = 0
        if mask is not None:
            mask = np.ascontiguousarray(mask)
            if mask.dtype not in (np.float32, np.float64):
                mask = mask.astype(np.float64)
            if mask.shape == (nslab, ny, nx) and not (single and mask.ndim == 2):
                per_slab = 1
            elif mask.shape != (ny, nx):
                raise XContourHipError(XC_EBADARG, 'mask must be (ny, nx) or (nslab, ny, nx)')
        out = {}
        J = 0
        Q = None
        if targets is not None:
            targets = np.ascontiguousarray(targets, dtype=np.float64)
            J = len(targets)
            Q = np.empty((nslab, J), dtype=np.float64)
        qs = np.empty((nslab, ny * nx), dtype=np.float64) if want_sorted else None
        ac = np.empty((nslab, ny * nx), dtype=np.float64) if want_acum else None
        nv = np.zeros(nslab, dtype=np.uint32)
        bpe = None
        ntbl = 0
        if tbl is not None:
            tbl = np.ascontiguousarray(tbl, dtype=np.float64)
            coord = np.ascontiguousarray(coord, dtype=np.float64)
            ntbl = len(tbl)
            bpe = np.zeros(nslab, dtype=np.float64)
        self._check(self.lib.xc_sort_profile_batch(self.handle, _ptr(q), dtype_code(q.dtype), _ptr(mask),
                                                   dtype_code(mask.dtype) if mask is not None else XC_F64, per_slab,
                                                   _ptr(dA), rank, nslab, ny, nx, 1 if negate else 0, _ptr(targets), J,
                                                   _ptr(tbl), _ptr(coord), ntbl,
                                                   _ptr(Q), _ptr(qs), _ptr(ac), _ptr(nv), _ptr(bpe)))
        out['nvalid'] = int(nv[0]) if single else nv.astype(np.int64)
        if Q is not None:
            out['Q'] = Q[0] if single else Q
        if qs is not None:
            out['q_sorted'] = qs[0] if single else qs
        if ac is not None:
            out['acum'] = ac[0] if single else ac
        if bpe is not None:
            out['bpe'] = float(bpe[0]) if single else bpe
        return out


_default_ctx = {}


def default_context(device=0):
    """Process-wide context per device (created on first use)."""
    ctx = _default_ctx.get(device)
    if ctx is None or ctx.handle is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx
