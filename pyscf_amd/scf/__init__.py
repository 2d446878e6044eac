from . import hf
from .hf import RHF, SCF
from .uhf import UHF
from .rohf import ROHF


def density_fit(mf, auxbasis=None, with_df=None):
    return mf.density_fit(auxbasis, with_df)
