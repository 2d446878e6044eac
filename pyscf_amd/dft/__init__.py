from . import gen_grid, libxc, numint, radi, rks
from .gen_grid import Grids
from .numint import NumInt
from .rks import RKS
