import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the oracle is test infrastructure: importable from tests only
if os.path.join(ROOT, 'oracle') not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def baro():
    import numpy as np
    g = os.path.join(ROOT, 'tests', 'golden')
    return (np.load(os.path.join(g, 'baro_q.npy')), np.load(os.path.join(g, 'baro_lat.npy')),
            np.load(os.path.join(g, 'baro_lon.npy')))


@pytest.fixture(scope='session')
def ctx():
    """A device context; only gpu-marked tests may request it."""
    from xcontour_amd import _native
    c = _native.Context(0)
    yield c
    c.close()
