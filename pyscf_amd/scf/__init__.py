from . import hf
from .hf import RHF, SCF


def density_fit(mf, auxbasis=None, with_df=None):
    return mf.density_fit(auxbasis, with_df)
