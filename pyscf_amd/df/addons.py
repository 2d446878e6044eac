"""Auxiliary-basis selection (pyscf/df/addons.py:42-72 DEFAULT_AUXBASIS, :230-299 make_auxmol,
:335-358 predefined_auxbasis)."""
from ..gto import mole as _mole

# AO basis -> JK-fit set (pyscf/df/addons.py:42-72)
DEFAULT_AUXBASIS = {
    'ccpvdz': 'cc-pvdz-jkfit', 'ccpvtz': 'cc-pvtz-jkfit',
    'def2svp': 'def2-svp-jkfit', 'def2tzvp': 'def2-tzvp-jkfit',
    'sto3g': 'def2-svp-jkfit', '631g': 'cc-pvdz-jkfit', '321g': 'def2-svp-jkfit', '6311g': 'cc-pvtz-jkfit',
    'augccpvdz': 'aug-cc-pvdz-jkfit', 'augccpvtz': 'aug-cc-pvtz-jkfit',
    'ccpvqz': 'cc-pvqz-jkfit', 'augccpvqz': 'aug-cc-pvqz-jkfit',
    'def2svpd': 'def2-svp-jkfit', 'def2tzvpd': 'def2-tzvp-jkfit', 'def2tzvpp': 'def2-tzvpp-jkfit',
    'def2qzvp': 'def2-qzvp-jkfit', 'def2qzvpp': 'def2-qzvpp-jkfit',
}


def predefined_auxbasis(mol, basis, xc='HF'):
    if not isinstance(basis, str):
        return None
    return DEFAULT_AUXBASIS.get(_mole._format_basis_name(basis))


# electrons per l of the ground-state atom, H-Kr (pyscf/data/elements.py:457-468 CONFIGURATION)
_CONFIGURATION = [[0, 0, 0, 0], [1, 0, 0, 0], [2, 0, 0, 0], [3, 0, 0, 0], [4, 0, 0, 0], [4, 1, 0, 0],
                  [4, 2, 0, 0], [4, 3, 0, 0], [4, 4, 0, 0], [4, 5, 0, 0], [4, 6, 0, 0],
                  [5, 6, 0, 0], [6, 6, 0, 0], [6, 7, 0, 0], [6, 8, 0, 0], [6, 9, 0, 0], [6, 10, 0, 0], [6, 11, 0, 0], [6, 12, 0, 0],
                  [7, 12, 0, 0], [8, 12, 0, 0], [8, 12, 1, 0], [8, 12, 2, 0], [8, 12, 3, 0], [7, 12, 5, 0], [8, 12, 5, 0], [8, 12, 6, 0],
                  [8, 12, 7, 0], [8, 12, 8, 0], [7, 12, 10, 0], [8, 12, 10, 0], [8, 13, 10, 0], [8, 14, 10, 0], [8, 15, 10, 0],
                  [8, 16, 10, 0], [8, 17, 10, 0], [8, 18, 10, 0]]
ETB_BETA = 2.0          # pyscf/df/addons.py:33


def _aug_etb_element(nuc_charge, basis, beta):
    """Even-tempered fitting shells [(l, n, e_min, beta)] for one element from its AO primitives: the default
    (USE_VERSION_26_AUXBASIS = True) branch of pyscf/df/addons.py:84-135 - exponent ranges of the AO pair products
    by geometric means, n = ceil(log((e_max + e_min)/e_min) / log(beta)) shells per l <= 2 min(l_max, occupied shells)."""
    import numpy as np
    l_max = max(b[0] for b in basis)
    emin_by_l = [1e99] * (l_max + 1)
    emax_by_l = [0] * (l_max + 1)
    for b in basis:
        l = b[0]
        e_c = np.array(b[2:] if isinstance(b[1], (int, np.integer)) else b[1:])
        es = e_c[:, 0]
        cs = e_c[:, 1:]
        es = es[abs(cs).max(axis=1) > 1e-3]
        emax_by_l[l] = max(es.max(), emax_by_l[l])
        emin_by_l[l] = min(es.min(), emin_by_l[l])
    max_shells = 4 - _CONFIGURATION[nuc_charge].count(0)
    l_max = min(l_max, max_shells)
    l_max_aux = l_max * 2
    l_max1 = l_max + 1
    emin_by_l = np.array(emin_by_l[:l_max1])
    emax_by_l = np.array(emax_by_l[:l_max1])
    emax = (emax_by_l[:, None] * emax_by_l) ** .5 * 2
    emin = (emin_by_l[:, None] * emin_by_l) ** .5 * 2
    liljsum = np.arange(l_max1)[:, None] + np.arange(l_max1)
    emax_by_l = np.array([emax[liljsum == ll].max() for ll in range(l_max_aux + 1)])
    emin_by_l = np.array([emin[liljsum == ll].min() for ll in range(l_max_aux + 1)])
    ns = np.log((emax_by_l + emin_by_l) / emin_by_l) / np.log(beta)
    return [(l, int(n), float(emin_by_l[l]), beta) for l, n in enumerate(np.ceil(ns).astype(int)) if n > 0]


def expand_etbs(etbs):
    """pyscf/gto/mole.py:766-801: [(l, n, alpha, beta)] -> [[l, [alpha beta^i, 1]] for i = n-1 .. 0]."""
    return [[l, [alpha * beta ** i, 1]] for l, n, alpha, beta in etbs for i in reversed(range(n))]


def aug_etb(mol, beta=ETB_BETA):
    """Even-tempered auxiliary basis for every element of mol (pyscf/df/addons.py:137-168 with start_at = 0)."""
    out = {}
    for symb, basis in mol._basis.items():
        etb = _aug_etb_element(_mole.element_charge(symb), basis, beta)
        if not etb:
            raise RuntimeError('Failed to generate even-tempered auxbasis for %s' % symb)
        out[symb] = expand_etbs(etb)
    return out


def make_auxbasis(mol, xc='HF'):
    """{atom label: fitting basis}: the predefined JK-fit set of the atom's AO basis where one exists, even-tempered
    Gaussians otherwise (pyscf/df/addons.py:170-227)."""
    uniq = []
    for a in mol._atom:
        if a[0] not in uniq:
            uniq.append(a[0])
    if isinstance(mol.basis, str):
        names = {a: mol.basis for a in uniq}
    elif isinstance(mol.basis, dict):
        names = {a: mol.basis.get(a, mol.basis.get(_mole.std_symbol_without_ghost(a), mol.basis.get('default')))
                 for a in uniq}
    else:
        names = {a: None for a in uniq}
    auxbasis = {}
    for a, obs in names.items():
        if isinstance(obs, str):
            balias = _mole._format_basis_name(obs)
            aux = predefined_auxbasis(mol, balias, xc)
            if aux is not None:
                try:
                    _mole.load_basis(aux, a)
                    auxbasis[a] = aux
                except KeyError:
                    pass
    if len(auxbasis) != len(uniq):
        etb = aug_etb(mol)
        for a in uniq:
            auxbasis.setdefault(a, etb[a])
    return auxbasis


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
