"""A TEST DOUBLE of the small part of xarray the package touches (xarray itself is not installed on the build or GPU boxes):
put this directory on sys.path BEFORE importing xcontour_amd and `labeled.is_xarray` becomes true, so the `xarray in,
xarray out` branches of labeled.py (unwrap / wrap / merge) and every facade method that returns labelled arrays run for real.
Signatures follow xarray: DataArray(data, coords=None, dims=None, name=None, attrs=None); coords behave like a mapping of
name -> DataArray; rename(str | dict); squeeze(); merge([...]) -> Dataset."""
import numpy as np

__version__ = '0.0-test-double'


class _Coords(dict):
    pass


class DataArray(object):
    def __init__(self, data=None, coords=None, dims=None, name=None, attrs=None):
        self._data = data if (hasattr(data, 'shape') and hasattr(data, 'dtype') and not isinstance(data, np.ndarray)
                              and not isinstance(data, DataArray)) else np.asarray(data)
        nd = len(self._data.shape)
        if dims is None:
            dims = tuple('dim_%d' % i for i in range(nd))
        if isinstance(dims, str):
            dims = (dims,)
        self.dims = tuple(dims)
        if len(self.dims) != nd:
            raise ValueError('different number of dimensions on data and dims: %d vs %d' % (nd, len(self.dims)))
        self.name = name
        self.attrs = dict(attrs or {})
        self.coords = _Coords()
        for k, v in (coords or {}).items():
            if isinstance(v, DataArray):
                self.coords[k] = v
            else:
                v = np.asarray(v)
                if v.ndim == 1 and k in self.dims and v.shape[0] != self._data.shape[self.dims.index(k)]:
                    raise ValueError('conflicting sizes for dimension %r' % k)
                c = DataArray.__new__(DataArray)
                c._data, c.dims, c.name, c.attrs, c.coords = v, ((k,) if v.ndim == 1 else ()), k, {}, _Coords()
                self.coords[k] = c

    values = property(lambda self: np.asarray(self._data[(slice(None),) * len(self._data.shape)]) if not isinstance(self._data, np.ndarray) else self._data)
    data = property(lambda self: self._data)
    shape = property(lambda self: tuple(self._data.shape))
    dtype = property(lambda self: np.dtype(self._data.dtype))
    ndim = property(lambda self: len(self._data.shape))
    size = property(lambda self: int(np.prod(self._data.shape, dtype=np.int64)))

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __len__(self):
        return self.shape[0]

    def _coord_dict(self):
        return {k: v.values for k, v in self.coords.items()}

    def rename(self, new=None):
        if isinstance(new, dict):
            dims = tuple(new.get(d, d) for d in self.dims)
            coords = {new.get(k, k): v.values for k, v in self.coords.items()}
            return DataArray(self._data, coords, dims, self.name, self.attrs)
        return DataArray(self._data, self._coord_dict(), self.dims, new, self.attrs)

    def squeeze(self):
        keep = [i for i, n in enumerate(self.shape) if n != 1]
        dims = tuple(self.dims[i] for i in keep)
        coords = {k: v.values for k, v in self.coords.items() if k in dims}
        return DataArray(self.values.reshape([self.shape[i] for i in keep]), coords, dims, self.name, self.attrs)

    def load(self):
        self._data = self.values
        return self

    def copy(self, deep=True, data=None):
        return DataArray(self.values.copy() if data is None else data, self._coord_dict(), self.dims, self.name, self.attrs)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        raise NotImplementedError('the test double indexes by coordinate name only')

    def __repr__(self):
        return '<fake xarray.DataArray %r %s>' % (self.name, dict(zip(self.dims, self.shape)))


class Dataset(object):
    def __init__(self, data_vars=None):
        self.data_vars = dict(data_vars or {})

    def __getitem__(self, k):
        return self.data_vars[k]

    def __getattr__(self, k):
        try:
            return self.__dict__['data_vars'][k]
        except KeyError:
            raise AttributeError(k)

    def __contains__(self, k):
        return k in self.data_vars

    def __iter__(self):
        return iter(self.data_vars)

    def __len__(self):
        return len(self.data_vars)

    def rename(self, new):
        out = {}
        for k, v in self.data_vars.items():
            a = v.rename(new)
            nk = new.get(k, k)
            out[nk] = a.rename(nk) if a.name == k else a
        return Dataset(out)


def merge(objects):
    out = {}
    for o in objects:
        if o.name is None:
            raise ValueError('cannot merge an unnamed DataArray')
        out[o.name] = o
    return Dataset(out)
