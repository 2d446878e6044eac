import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


H2O = 'O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587'


@pytest.fixture(scope='session')
def h2o_dz():
    from pyscf_amd import gto
    return gto.M(atom=H2O, basis='cc-pvdz'), gto.M(atom=H2O, basis='weigend')
