"""`pyscf.amd` - the MI355X engine as a plug-in of the `pyscf` namespace package.

PySCF discovers extension modules through the environment variable PYSCF_EXT_PATH (pyscf/__init__.py:42-60: a directory that
contains a `pyscf/` folder is appended to `pyscf.__path__`).  With

    export PYSCF_EXT_PATH=/path/to/this/repo/plugin

a stock PySCF install finds this module, and an unmodified script reaches the engine with two attribute assignments - the
reference's own extension points (`mf.with_df`, pyscf/df/df_jk.py:31,77-105; `mf._numint` / `mf.grids`, pyscf/dft/rks.py:318-330):

    import pyscf.amd
    mf = scf.RHF(mol).density_fit()
    mf.with_df = pyscf.amd.DF(mol, auxbasis='cc-pvtz-jkfit', devices=range(8))     # J/K: PAMD_df_get_jk
    mf = pyscf.amd.density_fit(dft.RKS(mol, xc='b3lyp'), devices=range(8))         # or both legs at once
    mf.kernel()

Everything here is the numpy-only host-array binding (pyscf_amd.df.native / pyscf_amd.dft.native over the C ABI of
include/pyscf_amd.h): no torch in the process, the caller owns every array.  `lib.to_gpu()` is NOT the hook - it is hard-wired to
the external gpu4pyscf / CuPy package (pyscf/lib/misc.py:1618-1645).
"""
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _REPO not in sys.path:                       # the engine package lives beside the plug-in directory (in-tree build)
    sys.path.insert(0, _REPO)

from pyscf_amd.df.native import NativeDF as _NativeDF                  # noqa: E402
from pyscf_amd.dft.native import NativeNumInt, NativeGrids             # noqa: E402


def _is_stock_mole(mol):
    return type(mol).__module__.split('.')[0] == 'pyscf'


class DF(_NativeDF):
    """`pyscf.df.DF` stand-in (build / reset / loop / get_naoaux / get_jk / range_coulomb, pyscf/df/df.py:147-333) backed by the
    HIP handle.  For a stock `pyscf.gto.Mole` the auxiliary molecule comes from PySCF's own `df.addons.make_auxmol`
    (pyscf/df/addons.py:113-160), so basis parsing stays the reference's; the handle only reads `_atm / _bas / _env`."""

    def build(self):
        if self.auxmol is None and _is_stock_mole(self.mol):
            from pyscf.df import addons
            self.auxmol = addons.make_auxmol(self.mol, self.auxbasis)
        return super().build()
    kernel = build


NumInt = NativeNumInt
Grids = NativeGrids


def density_fit(mf, auxbasis=None, devices=None):
    """Route an SCF object onto the engine: `mf.with_df` (J/K) and - for Kohn-Sham objects - `mf._numint` (the XC quadrature) over
    the same device list.  Returns `mf` (what `mf.density_fit()` returns, pyscf/df/df_jk.py:31-105)."""
    devs = None if devices is None else [int(d) for d in devices]
    if not hasattr(mf, 'with_df'):
        mf = mf.density_fit(auxbasis=auxbasis)
    mf.with_df = DF(mf.mol, auxbasis, devices=devs)
    if hasattr(mf, '_numint'):
        mf._numint = NativeNumInt(devices=devs)
        from pyscf_amd.dft.numint import estimate_ao_image_bytes
        mf.with_df.xc_image_hint = estimate_ao_image_bytes(mf.mol)        # one HBM budget: the tensor leaves room for the AO cache
    return mf


__all__ = ['DF', 'NumInt', 'Grids', 'density_fit']
