import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """A skipped whole-tensor config-5 test (the only N3 parity gate; its one skip path is host memory) is named in the summary."""
    for rep in terminalreporter.stats.get('skipped', []):
        if 'config5_whole_tensor' in getattr(rep, 'nodeid', ''):
            reason = rep.longrepr[2] if isinstance(rep.longrepr, tuple) else str(rep.longrepr)
            terminalreporter.write_line('CONFIG5_SKIPPED %s :: %s' % (rep.nodeid, reason))


H2O = 'O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587'


@pytest.fixture(scope='session')
def h2o_dz():
    from pyscf_amd import gto
    return gto.M(atom=H2O, basis='cc-pvdz'), gto.M(atom=H2O, basis='weigend')
