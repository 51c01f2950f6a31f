#!/opt/conda/bin/python3.9
"""Extract the reference's bundled test input into plain .npy fixtures.

Run ONCE in the build container (needs h5py, which only /opt/conda/bin/python3.9
has here):

    /opt/conda/bin/python3.9 tests/golden/extract_barotropic.py

Input : /root/reference/Data/barotropic_vorticity.nc  (the data file the
        reference's own demo scripts load: tests/test_LWA.py:14,
        tests/test_hist.py:108)
Output: tests/golden/baro_q.npy    (256, 512) float32  absolute_vorticity [1/s]
        tests/golden/baro_lat.npy  (256,)     float32  Gaussian latitudes
        tests/golden/baro_lon.npy  (512,)     float32  longitudes

These are DATA (inputs), not reference source.  sha256 prefixes of the raw
bytes are asserted so a silent re-extraction mismatch is caught.
"""
import hashlib
import os

import h5py
import numpy as np

SRC = '/root/reference/Data/barotropic_vorticity.nc'
HERE = os.path.dirname(os.path.abspath(__file__))

with h5py.File(SRC, 'r') as f:
    q = np.ascontiguousarray(f['absolute_vorticity'][...])
    lat = np.ascontiguousarray(f['latitude'][...])
    lon = np.ascontiguousarray(f['longitude'][...])

assert q.dtype == np.float32 and q.shape == (256, 512)
assert hashlib.sha256(q.tobytes()).hexdigest().startswith('c8d30d7acd84c77a')
assert hashlib.sha256(lat.tobytes()).hexdigest().startswith('49888387fcbea5e6')

np.save(os.path.join(HERE, 'baro_q.npy'), q)
np.save(os.path.join(HERE, 'baro_lat.npy'), lat)
np.save(os.path.join(HERE, 'baro_lon.npy'), lon)
print('wrote baro_q.npy baro_lat.npy baro_lon.npy')
