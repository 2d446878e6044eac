"""Nuclear gradients of the density-fitted SCF path."""
from . import rhf
from .rhf import Gradients, grad_nuc
