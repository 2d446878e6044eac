from . import gen_grid, libxc, numint, radi, rks, uks
from .gen_grid import Grids
from .numint import NumInt
from .rks import RKS
from .uks import UKS
