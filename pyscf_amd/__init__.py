"""pyscf_amd: MI355X-native density-fitted SCF Fock-build engine behind PySCF's plug-in
surface (mf.with_df / DF.get_jk / DF.loop).  See DESIGN.md and INTEGRATION.md."""
__version__ = '0.1.0'
from . import lib, gto    # noqa: F401


def __getattr__(name):
    if name in ('df', 'scf', 'dft'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
