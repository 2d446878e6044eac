"""DF-RHF / DF-UHF gradient probe against the reference goldens (pyscf/df/test/test_df_grad.py:57-75,95-121)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_amd import gto, scf
from oracle import ref
mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='6-31g')
mf = scf.RHF(mol).density_fit(auxbasis='ccpvdz-jkfit')
mf.conv_tol = 1e-12
mf.kernel()
print('E', mf.e_tot)
g0 = mf.Gradients().set(auxbasis_response=False).kernel()
print(g0, ref.fp(g0), 'golden 0.005466630382488041')
g1 = mf.Gradients().kernel()
print(g1, ref.fp(g1), 'golden 0.005516638190173352')
print('sum over atoms', g1.sum(0))
