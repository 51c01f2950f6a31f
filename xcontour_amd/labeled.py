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


class DataArray(object):
    """values + dims + 1-D coords + name.  Only what the hot path needs."""

    def __init__(self, data, dims=None, coords=None, name=None):
        self.values = np.asarray(data)
        if dims is None:
            dims = tuple('dim_%d' % i for i in range(self.values.ndim))
        if isinstance(dims, str):
            dims = (dims,)
        self.dims = tuple(dims)
        if len(self.dims) != self.values.ndim:
            raise ValueError('dims %r do not match data of shape %r' % (self.dims, self.values.shape))
        self.coords = {}
        for k, v in (coords or {}).items():
            v = np.asarray(v.values if isinstance(v, DataArray) else v)
            if k in self.dims and v.shape != (self.values.shape[self.dims.index(k)],):
                raise ValueError('coordinate %r has wrong length' % k)
            self.coords[k] = v
        self.name = name
        self.attrs = {}

    # -- numpy-ish
    @property
    def shape(self):
        return self.values.shape

    @property
    def dtype(self):
        return self.values.dtype

    @property
    def ndim(self):
        return self.values.ndim

    @property
    def size(self):
        return self.values.size

    def __len__(self):
        return len(self.values)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __repr__(self):
        return '<xcontour_amd.DataArray %r %s %s>\n%r' % (
            self.name, dict(zip(self.dims, self.shape)), self.dtype, self.values)

    # -- xarray-ish
    def copy(self, data=None):
        return DataArray(self.values.copy() if data is None else data, self.dims,
                         dict(self.coords), self.name)

    def load(self):
        return self

    def astype(self, dtype):
        return self.copy(data=self.values.astype(dtype))

    def rename(self, new):
        if isinstance(new, dict):
            dims = tuple(new.get(d, d) for d in self.dims)
            coords = {new.get(k, k): v for k, v in self.coords.items()}
            return DataArray(self.values, dims, coords, self.name)
        return DataArray(self.values, self.dims, dict(self.coords), new)

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
        return DataArray(self.values, self.dims, c, self.name)


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


def unwrap(x):
    """-> (values ndarray, dims tuple, coords dict of ndarrays, name)"""
    if isinstance(x, DataArray):
        return x.values, x.dims, dict(x.coords), x.name
    if is_xarray(x):
        coords = {k: np.asarray(v.values) for k, v in x.coords.items() if v.ndim == 1 and k in x.dims}
        return np.asarray(x.values), tuple(x.dims), coords, x.name
    raise TypeError('expected a DataArray (xcontour_amd.DataArray or xarray.DataArray), got %r' % type(x))


def wrap(values, dims, coords, name, like):
    """Build the same kind of labelled array as `like`."""
    coords = {k: v for k, v in coords.items() if k in dims}
    if is_xarray(like):
        return _xr.DataArray(values, dims=dims, coords=coords, name=name)
    return DataArray(values, dims, coords, name)


def merge(arrays, like):
    if is_xarray(like):
        return _xr.merge(arrays)
    return Dataset((a.name, a) for a in arrays)
