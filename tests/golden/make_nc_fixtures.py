#!/opt/conda/bin/python3.9
"""Writes the small HDF5 / NetCDF-4 style fixtures of tests/golden/nc/ with h5py (only
/opt/conda/bin/python3.9 has it in the build container) plus the expected raw arrays as .npz:

    /opt/conda/bin/python3.9 tests/golden/make_nc_fixtures.py

  old_groups.nc   libver 'earliest': superblock v0, v1 object headers, symbol-table groups; a (3,5,8) f32
                  variable chunked (1,3,5) with shuffle + deflate + fletcher32, big-endian latitude, an int16
                  variable packed with scale_factor / add_offset / _FillValue, dimension scales attached
  v18.nc          libver ('v108','v108'): superblock v2, v2 object headers, 12 variables (dense link storage
                  in a fractal heap), one variable with 14 attributes (dense attribute storage), a chunked +
                  deflate variable with partial edge chunks, a never-written (fill-value) variable, a scalar
  latest.nc       libver 'latest' (data layout v4 chunk index: reported as unreadable, the rest readable)
barotropic_vorticity.nc (next to the .npy fixtures) is the data file the reference's own demo scripts read.
"""
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nc')
rng = np.random.default_rng(7)
exp = {}


def scales(f, names_sizes):
    out = {}
    for i, (n, v) in enumerate(names_sizes):
        d = f.create_dataset(n, data=v)
        d.make_scale(n)
        d.attrs['_Netcdf4Dimid'] = np.int32(i)
        out[n] = d
    return out


with h5py.File(os.path.join(HERE, 'old_groups.nc'), 'w', libver='earliest') as f:
    time = np.arange(3, dtype=np.float64) * 6
    lat = np.linspace(-60, 60, 5).astype('>f4')
    lon = (np.arange(8) * 45).astype(np.float32)
    sc = scales(f, [('time', time), ('lat', lat), ('lon', lon)])
    t = rng.standard_normal((3, 5, 8)).astype(np.float32)
    d = f.create_dataset('t', data=t, chunks=(1, 3, 5), compression='gzip', compression_opts=4, shuffle=True, fletcher32=True)
    for i, n in enumerate(('time', 'lat', 'lon')):
        d.dims[i].attach_scale(sc[n])
    d.attrs['units'] = np.string_('K')
    d.attrs['_FillValue'] = np.float32(9.96921e36)
    p = rng.integers(-30000, 30000, (5, 8)).astype(np.int16)
    p[1, 2] = -32767
    dp = f.create_dataset('packed', data=p)
    dp.dims[0].attach_scale(sc['lat']); dp.dims[1].attach_scale(sc['lon'])
    dp.attrs['scale_factor'] = np.float64(0.01)
    dp.attrs['add_offset'] = np.float64(273.15)
    dp.attrs['_FillValue'] = np.int16(-32767)
    f.attrs['title'] = np.string_('old-style groups')
    exp.update(og_time=time, og_lat=lat.astype('<f4'), og_lon=lon, og_t=t, og_packed=p)

with h5py.File(os.path.join(HERE, 'v18.nc'), 'w', libver=('v108', 'v108')) as f:
    y = np.arange(7, dtype=np.float64)
    x = np.arange(11, dtype=np.float32)
    sc = scales(f, [('y', y), ('x', x)])
    for k in range(9):                                   # > 8 links: dense link storage
        v = rng.standard_normal((7, 11)).astype(np.float64 if k % 2 else np.float32)
        d = f.create_dataset('var%02d' % k, data=v)
        d.dims[0].attach_scale(sc['y']); d.dims[1].attach_scale(sc['x'])
        exp['v18_var%02d' % k] = v
    d = f['var03']
    for k in range(14):                                  # > 8 attributes: dense attribute storage
        d.attrs['att%02d' % k] = np.float64(k) * 1.5
    d.attrs['long_name'] = 'a variable-length string'
    c = rng.integers(0, 1000, (7, 11)).astype('>i4')
    dc = f.create_dataset('chunky', data=c, chunks=(4, 4), compression='gzip')
    dc.dims[0].attach_scale(sc['y']); dc.dims[1].attach_scale(sc['x'])
    f.create_dataset('never_written', shape=(7, 11), dtype=np.float32, chunks=(4, 4), fillvalue=-5.0)
    f.create_dataset('scalar', data=np.float64(3.25))
    exp.update(v18_y=y, v18_x=x, v18_chunky=c.astype('<i4'))

with h5py.File(os.path.join(HERE, 'latest.nc'), 'w', libver='latest') as f:
    a = rng.standard_normal((6, 6))
    f.create_dataset('plain', data=a)
    f.create_dataset('chunked_v4', data=a, chunks=(3, 3), compression='gzip')
    exp.update(latest_plain=a)

np.savez_compressed(os.path.join(HERE, 'expected.npz'), **exp)
print('wrote', sorted(os.listdir(HERE)))
