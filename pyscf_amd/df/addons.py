"""Auxiliary-basis selection (pyscf/df/addons.py:42-72 DEFAULT_AUXBASIS, :230-299 make_auxmol,
:335-358 predefined_auxbasis)."""
from ..gto import mole as _mole

# AO basis -> JK-fit set (pyscf/df/addons.py:42-72)
DEFAULT_AUXBASIS = {
    'ccpvdz': 'cc-pvdz-jkfit', 'ccpvtz': 'cc-pvtz-jkfit',
    'def2svp': 'def2-svp-jkfit', 'def2tzvp': 'def2-tzvp-jkfit',
    'sto3g': 'def2-svp-jkfit', '631g': 'cc-pvdz-jkfit',
}


def predefined_auxbasis(mol, basis, xc='HF'):
    if not isinstance(basis, str):
        return None
    return DEFAULT_AUXBASIS.get(_mole._format_basis_name(basis))


def make_auxbasis(mol, xc='HF'):
    if isinstance(mol.basis, str):
        aux = predefined_auxbasis(mol, mol.basis, xc)
        if aux is not None:
            return aux
    raise NotImplementedError('even-tempered auxiliary basis generation is out of scope; '
                              'pass auxbasis explicitly')


def make_auxmol(mol, auxbasis=None):
    """Fake Mole that carries the fitting basis on the same atoms; env[:20] copied
    (pyscf/df/addons.py:230-299)."""
    if auxbasis is None:
        auxbasis = make_auxbasis(mol)
    pmol = _mole.Mole()
    pmol.atom = mol.atom if hasattr(mol, 'atom') else None
    pmol._atom = mol._atom
    pmol.basis = auxbasis
    pmol.charge = getattr(mol, 'charge', 0)
    pmol.spin = getattr(mol, 'spin', 0)
    pmol._basis = pmol.format_basis(auxbasis)
    pmol._atm, pmol._bas, pmol._env = _mole.make_env(
        mol._atom, pmol._basis, mol._env[:_mole.PTR_ENV_START])
    pmol._built = True
    return pmol
