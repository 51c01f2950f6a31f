"""xcontour_amd.ncio: the numpy-only NetCDF-3 / NetCDF-4 (HDF5) reader against (i) the data file the
reference's own demo scripts open, (ii) h5py-written fixtures (tests/golden/make_nc_fixtures.py) and
(iii) classic files written on the spot with scipy.io.netcdf_file."""
import os

import numpy as np
import pytest

from xcontour_amd import ncio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NC = os.path.join(GOLD, 'nc')


def test_reference_data_file_matches_h5py_extraction():
    """barotropic_vorticity.nc (reference tests/test_LWA.py:14): superblock v2, contiguous float32, dimension
    scales -- read here without h5py, compared with the .npy files extract_barotropic.py made WITH h5py."""
    ds = ncio.open_dataset(os.path.join(GOLD, 'barotropic_vorticity.nc'))
    assert sorted(ds) == ['absolute_vorticity', 'latitude', 'longitude'] and ds.unreadable == {}
    q = ds.absolute_vorticity
    assert q.dims == ('latitude', 'longitude') and q.dtype == np.float32 and q.attrs == {'units': '1/s'}
    assert np.array_equal(q.values, np.load(os.path.join(GOLD, 'baro_q.npy')))
    assert np.array_equal(q.coords['latitude'], np.load(os.path.join(GOLD, 'baro_lat.npy')))
    assert np.array_equal(ds.longitude.values, np.load(os.path.join(GOLD, 'baro_lon.npy')))
    # and it plugs straight into the facade (constructor only: no GPU here)
    import xcontour_amd as xa
    dA = xa.DataArray(xa.cell_area(ds.latitude.values, ds.longitude.values), q.dims, q.coords)
    cm = xa.Contour2D(q, dA, dims={'X': 'longitude', 'Y': 'latitude'}, dimEq={'Y': 'latitude'})
    assert cm.dimEqV == 'latitude'


def test_hdf5_old_style_groups_chunked_filters_packed():
    E = np.load(os.path.join(NC, 'expected.npz'))
    raw = ncio.open_dataset(os.path.join(NC, 'old_groups.nc'), mask_and_scale=False)
    assert raw.attrs == {'title': 'old-style groups'}
    assert raw.t.dims == ('time', 'lat', 'lon') and raw.packed.dims == ('lat', 'lon')
    assert np.array_equal(raw.t.values, E['og_t'])                       # (1,3,5) chunks, shuffle + deflate + fletcher32
    assert np.array_equal(raw.lat.values, E['og_lat']) and raw.lat.dtype == np.dtype('float32')   # stored big-endian
    assert np.array_equal(raw.time.values, E['og_time']) and np.array_equal(raw.packed.values, E['og_packed'])
    assert np.array_equal(raw.t.coords['lat'], E['og_lat'])
    ds = ncio.open_dataset(os.path.join(NC, 'old_groups.nc'))
    p = E['og_packed']
    want = p.astype(np.float64) * 0.01 + 273.15                          # offset present -> float64 (xarray's rule)
    want[p == -32767] = np.nan
    assert ds.packed.dtype == np.float64 and np.array_equal(ds.packed.values, want, equal_nan=True)
    assert ds.t.dtype == np.float32 and np.array_equal(ds.t.values, E['og_t'])    # _FillValue never hit


def test_hdf5_dense_links_dense_attributes_fill_values():
    E = np.load(os.path.join(NC, 'expected.npz'))
    ds = ncio.open_dataset(os.path.join(NC, 'v18.nc'))
    assert len(ds) == 14 and ds.unreadable == {}                        # 14 links: fractal heap + v2 B-tree
    for k in range(9):
        v = ds['var%02d' % k]
        assert v.dims == ('y', 'x') and v.dtype == E['v18_var%02d' % k].dtype and np.array_equal(v.values, E['v18_var%02d' % k])
    a = ds.var03.attrs                                                   # 15 attributes: dense storage
    assert len(a) == 15 and a['long_name'] == 'a variable-length string'
    assert all(a['att%02d' % k] == 1.5 * k for k in range(14))
    assert np.array_equal(ds.chunky.values, E['v18_chunky']) and ds.chunky.dtype == np.int32    # big-endian, edge chunks
    assert ds.never_written.shape == (7, 11) and np.all(ds.never_written.values == -5.0)      # no chunk allocated
    assert ds.scalar.shape == () and float(ds.scalar.values) == 3.25
    assert np.array_equal(ds.y.values, E['v18_y']) and np.array_equal(ds.var00.coords['x'], E['v18_x'])


def test_hdf5_unsupported_layout_is_reported_not_fatal():
    E = np.load(os.path.join(NC, 'expected.npz'))
    ds = ncio.open_dataset(os.path.join(NC, 'latest.nc'))
    assert np.array_equal(ds.plain.values, E['latest_plain'])
    assert list(ds.unreadable) == ['chunked_v4'] and isinstance(ds.unreadable['chunked_v4'], NotImplementedError)


@pytest.mark.parametrize('version', [1, 2])
def test_classic_netcdf3(tmp_path, version):
    from scipy.io import netcdf_file
    path = str(tmp_path / ('c%d.nc' % version))
    rng = np.random.default_rng(version)
    lat, lon = np.linspace(-80, 80, 9).astype(np.float32), np.arange(12, dtype=np.float64) * 30
    pv = rng.standard_normal((4, 9, 12)).astype(np.float32)
    ps = rng.integers(-20000, 20000, (4, 9)).astype(np.int16)
    mask = rng.integers(0, 2, (9, 12)).astype(np.int32)
    with netcdf_file(path, 'w', version=version) as f:
        f.title = 'classic'
        f.createDimension('time', None); f.createDimension('lat', 9); f.createDimension('lon', 12)
        v = f.createVariable('lat', 'f4', ('lat',)); v[:] = lat; v.units = 'degrees_north'
        v = f.createVariable('lon', 'f8', ('lon',)); v[:] = lon
        v = f.createVariable('time', 'f8', ('time',)); v[:] = np.arange(4.0); v.units = 'hours since 1985-08-01 06:00:00'
        v = f.createVariable('pv', 'f4', ('time', 'lat', 'lon')); v[:] = pv; v.missing_value = np.float32(-999.0)
        v = f.createVariable('ps', 'i2', ('time', 'lat')); v[:] = ps; v.scale_factor = np.float32(0.5)
        v = f.createVariable('mask', 'i4', ('lat', 'lon')); v[:] = mask
    ds = ncio.open_dataset(path)
    assert ds.attrs['title'] == 'classic' and ds.pv.dims == ('time', 'lat', 'lon')
    assert np.array_equal(ds.pv.values, pv) and ds.pv.dtype == np.float32           # interleaved record variables
    assert np.array_equal(ds.ps.values, ps.astype(np.float32) * np.float32(0.5)) and ds.ps.dtype == np.float32
    assert np.array_equal(ds.mask.values, mask) and ds.mask.dims == ('lat', 'lon')
    assert np.array_equal(ds.pv.coords['lat'], lat) and ds.lat.attrs['units'] == 'degrees_north'
    t0 = np.datetime64('1985-08-01T06:00:00', 'ns')
    assert ds.time.dtype == np.dtype('datetime64[ns]') and np.array_equal(ds.time.values, t0 + np.arange(4) * np.timedelta64(3600, 's'))
    assert np.array_equal(ds.pv.coords['time'], ds.time.values)
    raw = ncio.open_dataset(path, decode_times=False)
    assert np.array_equal(raw.time.values, np.arange(4.0))


def test_not_a_netcdf_file(tmp_path):
    p = tmp_path / 'junk.nc'
    p.write_bytes(b'definitely not netcdf' * 100)
    with pytest.raises(ncio.NetCDFError):
        ncio.open_dataset(str(p))


# ---------------------------------------------------------------- lazy variables (open_dataset(lazy=...))
def test_lazy_variables_read_only_what_is_asked_for(tmp_path):
    """lazy=True / lazy=<min bytes>: stacks of planes stay in the (memory-mapped) file; slices along the leading axis read
    only those rows -- records of a classic record variable, rows of a fixed-size one, the chunks of an HDF5 chunked dataset
    that overlap the range (shuffle + deflate + fletcher32, partial edge chunks), a never-written dataset -- and every piece
    equals the eager read, CF decoding and byte order included"""
    from scipy.io import netcdf_file
    E = np.load(os.path.join(NC, 'expected.npz'))
    # HDF5, chunked (1, 3, 5) with filters
    lz = ncio.open_dataset(os.path.join(NC, 'old_groups.nc'), lazy=16)
    v = lz.t.data
    assert isinstance(v, ncio.LazyVariable) and v.shape == (3, 5, 8) and v.dtype == np.float32 and lz.t.dims == ('time', 'lat', 'lon')
    assert isinstance(lz.packed.data, np.ndarray)                        # two dims: read eagerly
    assert np.array_equal(v[1], E['og_t'][1]) and v.rows_read == 1
    assert np.array_equal(v[0:2, 1:4, ::2], E['og_t'][0:2, 1:4, ::2]) and np.array_equal(v[::2], E['og_t'][::2])
    assert np.array_equal(v[-1, ..., 3], E['og_t'][-1, ..., 3])
    assert np.array_equal(lz.t.values, E['og_t']) and np.array_equal(np.asarray(v), E['og_t'])
    # classic: a record variable interleaved with another one, and a fixed-size 3-D variable; masks and scales per piece
    path = str(tmp_path / 'stack.nc')
    rng = np.random.default_rng(3)
    pv = rng.standard_normal((6, 9, 12)).astype(np.float32)
    pv[2, 3, 4] = -999.0
    fx = rng.integers(-30000, 30000, (5, 9, 12)).astype(np.int16)
    with netcdf_file(path, 'w', version=2) as f:
        f.createDimension('time', None); f.createDimension('lev', 5); f.createDimension('lat', 9); f.createDimension('lon', 12)
        w = f.createVariable('time', 'f8', ('time',)); w[:] = np.arange(6.0)
        w = f.createVariable('pv', 'f4', ('time', 'lat', 'lon')); w[:] = pv; w.missing_value = np.float32(-999.0)
        w = f.createVariable('ps', 'i2', ('time', 'lat')); w[:] = rng.integers(0, 9, (6, 9)).astype(np.int16)
        w = f.createVariable('fx', 'i2', ('lev', 'lat', 'lon')); w[:] = fx; w.scale_factor = np.float32(0.25)
    eager, lz = ncio.open_dataset(path), ncio.open_dataset(path, lazy=64)
    for name in ('pv', 'fx'):
        a, b = eager[name].values, lz[name].data
        assert isinstance(b, ncio.LazyVariable) and b.shape == a.shape and b.dtype == a.dtype
        assert np.array_equal(b[1:3], a[1:3], equal_nan=True) and b.rows_read == 2
        assert np.array_equal(b[4], a[4], equal_nan=True) and np.array_equal(np.asarray(b), a, equal_nan=True)
    assert np.isnan(lz.pv.data[2, 3, 4]) and lz.pv.coords['time'].shape == (6,)
    # a lazy DataArray keeps its labels and goes through the facade's plumbing without being read
    import xcontour_amd as xa
    from xcontour_amd import labeled as lb
    d = lz.pv
    assert d.shape == (6, 9, 12) and d.rename({'lat': 'latitude'}).dims == ('time', 'latitude', 'lon')
    n0 = d.data.rows_read
    dA = xa.DataArray(np.ones((9, 12)), ('lat', 'lon'))
    cm = xa.Contour2D(d, dA, dims={'X': 'lon', 'Y': 'lat'}, dimEq={'Y': 'lat'})
    st, lead, lshape, coords = cm._plane(cm.tracer)
    assert isinstance(st, lb.LazyStack) and st.shape == (6, 9, 12) and lead == ('time',) and d.data.rows_read == n0
    assert np.array_equal(st[2:5], eager.pv.values[2:5], equal_nan=True) and d.data.rows_read == n0 + 3
    assert st.largest_request_bytes == 3 * 9 * 12 * 4


def test_lazy_stack_orders_and_runs():
    """LazyStack: any dim order of the source (leading dims anywhere, eq / x swapped), integer sources converted to float64,
    consecutive slabs fetched as ONE read when the leading dims come first"""
    from xcontour_amd.labeled import LazyStack, DataArray

    class Src(object):
        def __init__(self, a):
            self.a, self.shape, self.dtype, self.reads = a, a.shape, a.dtype, []

        def __getitem__(self, k):
            r = self.a[k]
            self.reads.append(r.shape)
            return r

    a = np.arange(2 * 3 * 4 * 5, dtype=np.int32).reshape(2, 3, 4, 5)
    s = Src(a)
    st = LazyStack(s, (0, 1), 2, 3)
    assert st.shape == (6, 4, 5) and st.dtype == np.float64 and st.nbytes == 6 * 20 * 8
    assert np.array_equal(st[1:5], a.reshape(6, 4, 5)[1:5]) and s.reads == [(2, 4, 5), (2, 4, 5)]   # slabs 1-2 | 3-4: two runs
    assert np.array_equal(st[5], a[1, 2]) and np.array_equal(st[[0, 5]], a.reshape(6, 4, 5)[[0, 5]])
    t = Src(np.ascontiguousarray(a.transpose(2, 0, 3, 1)))            # (eq, time, x, level)
    st2 = LazyStack(t, (1, 3), 0, 2)
    assert np.array_equal(np.asarray(st2), a.reshape(6, 4, 5))
    u = Src(np.ascontiguousarray(a.transpose(0, 1, 3, 2)).astype(np.float32))   # (..., x, eq): swapped plane axes
    st3 = LazyStack(u, (0, 1), 3, 2)
    assert st3.dtype == np.float32 and np.array_equal(st3[2:4], a.reshape(6, 4, 5)[2:4])
    d = DataArray(s, ('t', 'z', 'y', 'x'), {'y': np.arange(4.0)}, 'q')
    assert d.shape == (2, 3, 4, 5) and d.data is s and np.array_equal(d.values, a) and 'lazy' in repr(d)
    assert d.load().data is not s and isinstance(d.data, np.ndarray)
