"""Nuclear gradients of the density-fitted SCF path."""
from . import rhf, rks, uks
from .rhf import Gradients, grad_nuc
