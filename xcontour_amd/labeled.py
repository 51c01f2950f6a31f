# -*- coding: utf-8 -*-
"""
A minimal labelled array, API-compatible with the small part of
`xarray.DataArray` that the reference's Contour2D / Table use and return
(dims, coords, name, values, isel, rename, squeeze, ...).

xarray is the reference's in/out type (core.py:10) but is not installed in the
build or GPU images.  `xcontour_amd` therefore accepts, in order of preference,
  * a real `xarray.DataArray` (if xarray is importable) -> results are returned
    as `xarray.DataArray`,
  * this `DataArray`,
and always returns the same kind it was given.
"""
import numpy as np

try:                                    # optional
    import xarray as _xr
except Exception:                       # pragma: no cover - not installed here
    _xr = None


def is_lazy_data(x):
    """not an ndarray (nor a list / scalar) but array-like enough to be read piecewise: shape, dtype, __getitem__"""
    return (not isinstance(x, (np.ndarray, np.generic, list, tuple, int, float)) and hasattr(x, 'shape') and hasattr(x, 'dtype')
            and hasattr(x, '__getitem__') and not isinstance(x, DataArray) and len(getattr(x, 'shape', ())) > 0)


class LazyStack(object):
    """The (S, ny, nx) face of a LAZY (..., eq, x) stack: shape and dtype are known, the slabs are read when a batch of
    them is asked for (`stack[s0:s1]`, `stack[[3, 7]]` -> C-contiguous ndarray in (slab, eq, x) order, float32 / float64).
    The reference's histogram API is lazy too (`dask='allowed'`, core.py:242, 258; docstring 158-160 'memory-friendly');
    the native entry points cut such a stack into batches below `Context.max_batch_bytes` and read one batch at a time."""
    _xc_lazy_stack = True

    def __init__(self, data, lead_axes, eq_axis, x_axis):
        self.src = data
        self.lead_axes, self.eq_axis, self.x_axis = tuple(lead_axes), int(eq_axis), int(x_axis)
        shp = tuple(int(n) for n in data.shape)
        self.lshape = tuple(shp[a] for a in self.lead_axes)
        self.shape = (int(np.prod(self.lshape, dtype=np.int64)), shp[self.eq_axis], shp[self.x_axis])
        dt = np.dtype(data.dtype)
        self.dtype = dt if dt in (np.dtype(np.float32), np.dtype(np.float64)) else np.dtype(np.float64)
        self.ndim = 3
        self.largest_request_bytes = 0
        # consecutive slabs are consecutive along the LAST leading axis when the leading axes come first and in order:
        # a run of them is then ONE read of the source
        self._runs = self.lead_axes == tuple(range(len(self.lead_axes))) and len(self.lead_axes) > 0

    @property
    def size(self):
        return self.shape[0] * self.shape[1] * self.shape[2]

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def __len__(self):
        return self.shape[0]

    def _read(self, key, nslab):
        a = np.asarray(self.src[tuple(key)])
        self.largest_request_bytes = max(self.largest_request_bytes, a.nbytes)
        if self.eq_axis > self.x_axis:
            a = np.swapaxes(a, -1, -2)
        return a.reshape((nslab,) + self.shape[1:])

    def take(self, idx):
        """slabs `idx` (flat indices over the leading dims) -> (len(idx), ny, nx) C-contiguous in the stack's float dtype"""
        idx = [int(i) for i in idx]
        out = np.empty((len(idx),) + self.shape[1:], dtype=self.dtype)
        nd = len(self.src.shape)
        i = 0
        while i < len(idx):
            lead = np.unravel_index(idx[i], self.lshape) if self.lshape else ()
            run = 1
            if self._runs:                                           # extend over consecutive slabs inside the last leading axis
                last = self.lshape[-1]
                while i + run < len(idx) and idx[i + run] == idx[i] + run and lead[-1] + run < last:
                    run += 1
            key = [slice(None)] * nd
            for a, l in zip(self.lead_axes, lead):
                key[a] = int(l)
            if run > 1 or (self._runs and len(self.lead_axes) >= 1):
                key[self.lead_axes[-1]] = slice(int(lead[-1]), int(lead[-1]) + run)
            out[i:i + run] = self._read(key, run)
            i += run
        return out

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self.take([key + self.shape[0] if key < 0 else key])[0]
        if isinstance(key, slice):
            return self.take(range(*key.indices(self.shape[0])))
        return self.take(list(np.asarray(key).reshape(-1)))

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self[:], dtype=dtype)


class LazyProduct(object):
    """a * b of two equally shaped sources, at least one of them lazy, evaluated piece by piece (the `integrand * grdm` of
    cal_contour_mean, reference core.py:568-570 / 599-601, which xarray + dask also leave lazy)"""

    def __init__(self, a, b):
        if tuple(a.shape) != tuple(b.shape):
            raise ValueError('LazyProduct needs equal shapes')
        self.a, self.b, self.shape = a, b, tuple(int(n) for n in a.shape)
        self.dtype = np.result_type(np.dtype(a.dtype), np.dtype(b.dtype))

    def __getitem__(self, key):
        return np.asarray(self.a[key]) * np.asarray(self.b[key])


class DataArray(object):
    """values + dims + 1-D coords + name.  Only what the hot path needs."""

    def __init__(self, data, dims=None, coords=None, name=None):
        # `data` may be LAZY: anything that is not an ndarray but has shape, dtype and __getitem__ (a dask array, an
        # `ncio` variable opened with lazy=True, an h5py / zarr dataset).  It is kept as it is -- `values` materialises it,
        # `data` hands it out untouched -- and the hot path pulls it slab batch by slab batch (core._plane_of, LazyStack).
        self._data = data if is_lazy_data(data) else np.asarray(data)
        if dims is None:
            dims = tuple('dim_%d' % i for i in range(self.ndim))
        if isinstance(dims, str):
            dims = (dims,)
        self.dims = tuple(dims)
        if len(self.dims) != self.ndim:
            raise ValueError('dims %r do not match data of shape %r' % (self.dims, self.shape))
        self.coords = {}
        for k, v in (coords or {}).items():
            v = np.asarray(v.values if isinstance(v, DataArray) else v)
            if k in self.dims and v.shape != (self.shape[self.dims.index(k)],):
                raise ValueError('coordinate %r has wrong length' % k)
            self.coords[k] = v
        self.name = name
        self.attrs = {}

    @classmethod
    def _trusted(cls, data, dims, coords, name):
        """internal fast path for results the package builds itself (an ndarray, a dims tuple that matches it, 1-D ndarray coords): no
        validation, the coords dict is shared, not copied -- nine of these per keff() call were a third of its Python time"""
        self = cls.__new__(cls)
        self._data, self.dims, self.coords, self.name, self.attrs = data, dims, coords, name, {}
        return self

    # -- numpy-ish
    @property
    def values(self):
        """the data as an ndarray (a lazy source is read in full: use `data` / slices to stay lazy)"""
        return self._data if isinstance(self._data, np.ndarray) else np.asarray(self._data[(slice(None),) * len(self._data.shape)])

    @values.setter
    def values(self, v):
        self._data = np.asarray(v)

    @property
    def data(self):
        """the underlying array, lazy or not (xarray's name for it)"""
        return self._data

    @property
    def shape(self):
        return tuple(self._data.shape)

    @property
    def dtype(self):
        return np.dtype(self._data.dtype)

    @property
    def ndim(self):
        return len(self._data.shape)

    @property
    def size(self):
        return int(np.prod(self._data.shape, dtype=np.int64))

    def __len__(self):
        return int(self._data.shape[0])

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __repr__(self):
        return '<xcontour_amd.DataArray %r %s %s>\n%r' % (
            self.name, dict(zip(self.dims, self.shape)), self.dtype,
            self._data if isinstance(self._data, np.ndarray) else '(lazy %s)' % type(self._data).__name__)

    # -- xarray-ish
    def copy(self, data=None):
        return DataArray(self.values.copy() if data is None else data, self.dims,
                         dict(self.coords), self.name)

    def load(self):
        """read a lazy source into memory (in place, like xarray)"""
        self._data = self.values
        return self

    def astype(self, dtype):
        return self.copy(data=self.values.astype(dtype))

    def rename(self, new):
        if isinstance(new, dict):
            dims = tuple(new.get(d, d) for d in self.dims)
            coords = {new.get(k, k): v for k, v in self.coords.items()}
            return DataArray(self._data, dims, coords, self.name)
        return DataArray(self._data, self.dims, dict(self.coords), new)

    def squeeze(self):
        keep = [i for i, n in enumerate(self.shape) if n != 1]
        dims = tuple(self.dims[i] for i in keep)
        coords = {k: v for k, v in self.coords.items() if k in dims}
        return DataArray(self.values.reshape([self.shape[i] for i in keep]), dims, coords, self.name)

    def transpose(self, *dims):
        order = [self.dims.index(d) for d in dims]
        return DataArray(self.values.transpose(order), dims, dict(self.coords), self.name)

    def isel(self, indexers):
        out, dims = self.values, list(self.dims)
        coords = dict(self.coords)
        for d, ix in indexers.items():
            ax = dims.index(d)
            out = np.take(out, ix, axis=ax) if not isinstance(ix, slice) else out[(slice(None),) * ax + (ix,)]
            if d in coords:
                coords[d] = coords[d][ix]
            if np.ndim(ix) == 0 and not isinstance(ix, slice):
                dims.pop(ax)
                coords.pop(d, None)
        return DataArray(out, dims, coords, self.name)

    def __getitem__(self, key):
        if isinstance(key, str):                      # coordinate access, like xarray
            return DataArray(self.coords[key], (key,), {key: self.coords[key]}, key)
        if isinstance(key, dict):
            return self.isel(key)
        return self.isel({self.dims[0]: key})

    def assign_coords(self, coords):
        c = dict(self.coords)
        c.update({k: np.asarray(v) for k, v in coords.items()})
        return DataArray(self._data, self.dims, c, self.name)


class Dataset(dict):
    """`xr.merge([...])` stand-in: name -> DataArray, attribute access included."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    # the little of xarray.Dataset the reference's scripts use on the merged results
    # (tests/test_Keff_ocean.py:76: `cm.interp_to_dataset(preY, Yeq, origin).rename({'new': 'latitude'})`)
    def rename(self, new):
        """rename dimensions / coordinates (in every member) and variables, like xarray.Dataset.rename"""
        out = Dataset()
        for k, v in self.items():
            a = v.rename(new)
            nk = new.get(k, k)
            out[nk] = a.rename(nk) if a.name == k else a
        return out

    @property
    def data_vars(self):
        return dict(self)

    def load(self):
        return self

    def __repr__(self):
        return '<xcontour_amd.Dataset %s>' % ', '.join('%s%r' % (k, tuple(v.dims)) for k, v in self.items())


# ---------------------------------------------------------------------------
# unwrap / rewrap helpers used by the facade
# ---------------------------------------------------------------------------
def is_xarray(x):
    return _xr is not None and isinstance(x, _xr.DataArray)


def is_labeled(x):
    return isinstance(x, DataArray) or is_xarray(x)


def unwrap(x, lazy=False):
    """-> (values ndarray, dims tuple, coords dict of ndarrays, name).  `lazy=True`: a lazy source (dask array behind an
    xarray.DataArray, a lazy `ncio` variable, ...) is handed out as it is instead of being read in full."""
    if isinstance(x, DataArray):
        return (x.data if lazy else x.values), x.dims, dict(x.coords), x.name
    if is_xarray(x):
        coords = {k: np.asarray(v.values) for k, v in x.coords.items() if v.ndim == 1 and k in x.dims}
        raw = getattr(x, 'data', None)
        if lazy and raw is not None and is_lazy_data(raw):
            return raw, tuple(x.dims), coords, x.name
        return np.asarray(x.values), tuple(x.dims), coords, x.name
    raise TypeError('expected a DataArray (xcontour_amd.DataArray or xarray.DataArray), got %r' % type(x))


def wrap(values, dims, coords, name, like, trusted=False):
    """Build the same kind of labelled array as `like`.  `trusted`: `coords` already holds exactly the 1-D ndarray coordinates of `dims`
    (the caller built it for this result): the in-house DataArray then takes everything as it is."""
    if is_xarray(like):
        return _xr.DataArray(values, dims=dims, coords={k: v for k, v in coords.items() if k in dims}, name=name)
    if trusted:
        return DataArray._trusted(values, tuple(dims), coords, name)
    return DataArray(values, dims, {k: v for k, v in coords.items() if k in dims}, name)


def merge(arrays, like):
    if is_xarray(like):
        return _xr.merge(arrays)
    return Dataset((a.name, a) for a in arrays)
