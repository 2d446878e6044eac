"""Molecule / basis container producing the libcint-format ``_atm/_bas/_env`` tables.

This is the input format of the hot path.  It mirrors the layout
that the reference builds in pyscf/gto/mole.py:

* slot constants                       mole.py:58-88
* ``gto_norm`` / ``gaussian_int``      mole.py:122-157
* ``make_atm_env``                     mole.py:963-982
* ``make_bas_env``                     mole.py:986-1018  (primitives sorted by exponent
  descending, coefficients stored contraction-major and pre-multiplied by the radial
  normalisation)
* ``_nomalize_contracted_ao``          mole.py:1020-1029
* ``make_env``                         mole.py:1031-1107
* ``conc_env``                         mole.py:805-838

When real PySCF is importable a ``pyscf.gto.Mole`` can be used instead: every routine
downstream only reads ``_atm/_bas/_env`` (+ ``nelectron``/``_atom``).
"""
import json
import math
import os
import re

import numpy as np
from scipy.special import gamma

from .basis import parse_nwchem

BOHR = 0.52917721092  # pyscf/data/nist.py:24

# _atm slots (mole.py:58-66)
CHARGE_OF, PTR_COORD, NUC_MOD_OF, PTR_ZETA, PTR_FRAC_CHARGE, RESERVE_ATMSLOT = range(6)
ATM_SLOTS = 6
# _bas slots (mole.py:67-75)
ATOM_OF, ANG_OF, NPRIM_OF, NCTR_OF, KAPPA_OF, PTR_EXP, PTR_COEFF, RESERVE_BASLOT = range(8)
BAS_SLOTS = 8
# _env slots (mole.py:76-88)
PTR_EXPCUTOFF = 0
PTR_RANGE_OMEGA = 8
PTR_ENV_START = 20
NUC_POINT = 1

ELEMENTS = ['X', 'H', 'He', 'Li', 'Be', 'B', 'C', 'N', 'O', 'F', 'Ne', 'Na', 'Mg', 'Al',
            'Si', 'P', 'S', 'Cl', 'Ar', 'K', 'Ca', 'Sc', 'Ti', 'V', 'Cr', 'Mn', 'Fe', 'Co',
            'Ni', 'Cu', 'Zn', 'Ga', 'Ge', 'As', 'Se', 'Br', 'Kr']
_CHARGE = {s.upper(): z for z, s in enumerate(ELEMENTS)}

_BASIS_DATA = None
# alias -> key in data.json (pyscf/gto/basis/__init__.py:49-208)
_ALIAS = {
    'ano': 'ano', 'sto3g': 'sto3g', '631g': '631g', 'ccpvdz': 'ccpvdz', 'ccpvtz': 'ccpvtz',
    'def2svp': 'def2svp', 'def2tzvp': 'def2tzvp',
    'ccpvdzjkfit': 'ccpvdzjkfit', 'ccpvtzjkfit': 'ccpvtzjkfit', 'ccpvdzri': 'ccpvdzri',
    'def2universaljkfit': 'def2universaljkfit', 'def2universaljfit': 'def2universaljfit',
    'def2svpjkfit': 'def2universaljkfit', 'def2tzvpjkfit': 'def2universaljkfit',
    'def2tzvppjkfit': 'def2universaljkfit', 'def2qzvpjkfit': 'def2universaljkfit',
    'def2svpjfit': 'def2universaljfit', 'def2tzvpjfit': 'def2universaljfit',
    'weigend': 'def2universaljfit', 'weigendjfit': 'def2universaljfit',
    'weigendcfit': 'def2universaljfit', 'weigendjkfit': 'def2universaljkfit',
    'ccpvqz': 'ccpvqz', 'augccpvdz': 'augccpvdz', 'augccpvtz': 'augccpvtz',
    'augccpvdzjkfit': 'augccpvdzjkfit', 'augccpvtzjkfit': 'augccpvtzjkfit', 'ccpvtzri': 'ccpvtzri',
    '321g': '321g', '631gs': '631gs', '631g*': '631gs', '631g(d)': '631gs',
    '631gss': '631gss', '631g**': '631gss', '631g(d,p)': '631gss',
    '6311g': '6311g', '6311gss': '6311gss', '6311g**': '6311gss', '6311g(d,p)': '6311gss',
    'ccpvqzjkfit': 'ccpvqzjkfit', 'augccpvqz': 'augccpvqz', 'augccpvqzjkfit': 'augccpvqzjkfit', 'ccpvqzri': 'ccpvqzri',
    'def2qzvp': 'def2qzvp', 'def2qzvpp': 'def2qzvpp', 'def2tzvpp': 'def2tzvpp', 'def2svpd': 'def2svpd',
    'def2tzvpd': 'def2tzvpd', 'def2qzvppjkfit': 'def2universaljkfit', 'def2svpdjkfit': 'def2universaljkfit',
    'def2tzvpdjkfit': 'def2universaljkfit',
}


def _format_basis_name(name):
    """pyscf/gto/basis/__init__.py:853 - lower-case and strip '-', '_', ' ', '*' kept out."""
    return name.lower().replace('-', '').replace('_', '').replace(' ', '')


def _rm_digit(symb):
    return ''.join(c for c in symb if c.isalpha())


def is_ghost_atom(symb):
    """'ghost:H', 'GHOST-H', 'ghost_H', 'X-H', 'X:H', 'GHOST' (pyscf/data/elements.py:is_ghost_atom)."""
    u = str(symb).upper()
    return u.startswith('GHOST') or u.startswith(('X-', 'X:', 'X_'))


def std_symbol_without_ghost(symb):
    """Element symbol of an atom label with the ghost prefix and numeric suffix removed
    (pyscf/data/elements.py:_std_symbol_without_ghost)."""
    u = str(symb)
    up = u.upper()
    if up.startswith('GHOST'):
        u = u[5:].lstrip(':-_ ')
    elif up.startswith(('X-', 'X:', 'X_')):
        u = u[2:]
    u = _rm_digit(u) or 'X'
    return u[0].upper() + u[1:].lower()


def charge(symb):
    """Nuclear charge of an atom label; ghost atoms carry basis functions and grids but no charge."""
    if is_ghost_atom(symb):
        return 0
    return _CHARGE[_rm_digit(symb).upper()]


def element_charge(symb):
    """Proton number of the element behind a label, ghost or not (what grids and guesses are sized by)."""
    return _CHARGE[std_symbol_without_ghost(symb).upper()]


def load_basis(name, symb):
    """Return the internal basis list for element `symb` from the packaged table."""
    global _BASIS_DATA
    if _BASIS_DATA is None:
        fn = os.path.join(os.path.dirname(__file__), 'basis', 'data.json')
        with open(fn) as f:
            _BASIS_DATA = json.load(f)
    key = _ALIAS.get(_format_basis_name(name))
    el = std_symbol_without_ghost(symb)
    if key is None or el not in _BASIS_DATA[key]:
        raise KeyError('Basis %s not found for %s (packaged table covers H-Kr for: %s)'
                       % (name, symb, ', '.join(sorted(set(_ALIAS)))))
    return [[sh[0]] + [list(ec) for ec in sh[1:]] for sh in _BASIS_DATA[key][el]]


def gaussian_int(n, alpha):
    """int_0^inf x^n exp(-alpha x^2) dx  (mole.py:122-125)"""
    n1 = (n + 1) * .5
    return gamma(n1) / (2. * np.asarray(alpha, dtype=float) ** n1)


def gto_norm(l, expnt):
    """1/sqrt(int r^(2l+2) exp(-2 a r^2) dr)  (mole.py:127-157)"""
    return 1. / np.sqrt(gaussian_int(l * 2 + 2, 2 * np.asarray(expnt, dtype=float)))


def _normalize_contracted_ao(l, es, cs):
    ee = es.reshape(-1, 1) + es.reshape(1, -1)
    ee = gaussian_int(l * 2 + 2, ee)
    s1 = 1. / np.sqrt(np.einsum('pi,pq,qi->i', cs, ee, cs))
    return np.einsum('pi,i->pi', cs, s1)


def make_bas_env(basis_add, atom_id=0, ptr=0):
    _bas, _env = [], []
    for b in basis_add:
        angl = b[0]
        b_coeff = np.array(sorted(b[1:], reverse=True))
        es = b_coeff[:, 0]
        cs = b_coeff[:, 1:]
        nprim, nctr = cs.shape
        cs = np.einsum('pi,p->pi', cs, gto_norm(angl, es))
        cs = _normalize_contracted_ao(angl, es, cs)
        _env.append(es)
        _env.append(cs.T.reshape(-1))
        ptr_exp = ptr
        ptr_coeff = ptr_exp + nprim
        ptr = ptr_coeff + nprim * nctr
        _bas.append([atom_id, angl, nprim, nctr, 0, ptr_exp, ptr_coeff, 0])
    env = np.hstack(_env) if _env else np.zeros(0)
    return np.array(_bas, np.int32).reshape(-1, BAS_SLOTS), env


def make_env(atoms, basis, pre_env):
    _atm, _bas = [], []
    _env = [np.asarray(pre_env, dtype=float)]
    ptr_env = len(pre_env)
    for ia, atom in enumerate(atoms):
        a = np.zeros(ATM_SLOTS, np.int32)
        a[CHARGE_OF] = charge(atom[0])
        a[PTR_COORD] = ptr_env
        a[NUC_MOD_OF] = NUC_POINT
        a[PTR_ZETA] = ptr_env + 3
        _atm.append(a)
        _env.append(np.hstack((atom[1], 0.)))
        ptr_env += 4
    basdic = {}
    for symb, basis_add in basis.items():
        bas0, env0 = make_bas_env(basis_add, 0, ptr_env)
        ptr_env += len(env0)
        basdic[symb] = bas0
        _env.append(env0)
    for ia, atom in enumerate(atoms):
        symb = atom[0]
        if symb in basdic:
            b = basdic[symb].copy()
        elif _rm_digit(symb) in basdic:
            b = basdic[_rm_digit(symb)].copy()
        else:
            b = basdic[std_symbol_without_ghost(symb)].copy()
        b[:, ATOM_OF] = ia
        _bas.append(b)
    atm = np.asarray(np.vstack(_atm), np.int32).reshape(-1, ATM_SLOTS)
    bas = (np.asarray(np.vstack(_bas), np.int32).reshape(-1, BAS_SLOTS)
           if _bas else np.zeros((0, BAS_SLOTS), np.int32))
    return atm, bas, np.asarray(np.hstack(_env), dtype=np.float64)


def conc_env(atm1, bas1, env1, atm2, bas2, env2):
    """Concatenate two (atm, bas, env) sets (mole.py:805-838)."""
    off = len(env1)
    natm_off = len(atm1)
    atm2 = np.array(atm2, copy=True)
    bas2 = np.array(bas2, copy=True)
    atm2[:, PTR_COORD] += off
    atm2[:, PTR_ZETA] += off
    bas2[:, ATOM_OF] += natm_off
    bas2[:, PTR_EXP] += off
    bas2[:, PTR_COEFF] += off
    return (np.asarray(np.vstack((atm1, atm2)), np.int32),
            np.asarray(np.vstack((bas1, bas2)), np.int32),
            np.hstack((env1, env2)))


def format_atom(atom, unit='angstrom'):
    """-> list of (symbol, np.array([x,y,z]) in Bohr)  (mole.py:400-500 semantics)."""
    if isinstance(atom, str):
        atom = atom.replace(';', '\n').replace(',', ' ')
        rows = []
        for line in atom.split('\n'):
            line = line.strip()
            if not line or line.startswith('#'):
                continue
            t = line.split()
            if len(t) == 1 and t[0][0].isalpha():  # a bare symbol: one atom at the origin (mole.py:424-426)
                rows.append((t[0], [0.0, 0.0, 0.0]))
                continue
            if len(t) < 4:
                raise ValueError('atom line %r: expected "symbol x y z"' % line)
            rows.append((t[0], [float(x) for x in t[1:4]]))
    else:
        rows = []
        for a in atom:
            if len(a) == 2:
                rows.append((a[0], list(a[1])))
            else:
                rows.append((a[0], [float(x) for x in a[1:4]]))
    u = unit.lower()
    scale = 1.0 if u.startswith(('b', 'au')) else 1.0 / BOHR
    out = []
    for s, xyz in rows:
        s = str(s)
        if s.isdigit():
            s = ELEMENTS[int(s)]
        out.append((s, np.array(xyz, dtype=float) * scale))
    return out


class Mole:
    """Minimal stand-in for ``pyscf.gto.Mole`` (attributes the DF/SCF path reads)."""

    def __init__(self, **kw):
        self.atom = []
        self.basis = 'sto-3g'
        self.unit = 'angstrom'
        self.charge = 0
        self.spin = 0
        self.verbose = 0
        self.max_memory = 4000
        self.cart = False
        self.stdout = None
        self._atm = np.zeros((0, ATM_SLOTS), np.int32)
        self._bas = np.zeros((0, BAS_SLOTS), np.int32)
        self._env = np.zeros(PTR_ENV_START)
        self._atom = []
        self._basis = {}
        self._built = False
        for k, v in kw.items():
            setattr(self, k, v)

    # --- construction -------------------------------------------------------------
    def build(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
        self._atom = format_atom(self.atom, self.unit)
        self._basis = self.format_basis(self.basis)
        pre_env = np.zeros(PTR_ENV_START)
        self._atm, self._bas, self._env = make_env(self._atom, self._basis, pre_env)
        self._built = True
        if (self.nelectron - self.spin) % 2 != 0:
            raise RuntimeError('Electron number %d and spin %d are not consistent'
                               % (self.nelectron, self.spin))
        return self

    def format_basis(self, basis):
        uniq = []
        for a in self._atom:
            if a[0] not in uniq:
                uniq.append(a[0])
        if isinstance(basis, (str, list, tuple)):
            basis = {a: basis for a in uniq}
        else:
            basis = dict(basis)
            if 'default' in basis:
                d = basis.pop('default')
                full = {a: d for a in uniq}
                full.update(basis)
                basis = full
        def one(b, symb):
            """name | NWChem text | [[l, [e, c...], ...], ...] | tuple of those (concatenated, mole.py:440-470)."""
            if isinstance(b, str):
                if '\n' in b or re.search(r'\b[SPDFGHI]\b', b):
                    el = std_symbol_without_ghost(symb)
                    return parse_nwchem.parse(b, el) if el in b else parse_nwchem.parse(b)
                return load_basis(b, symb)
            b = list(b)
            if b and isinstance(b[0], (int, np.integer)):          # a single raw shell [l, (e, c), ...]
                return [[b[0]] + [list(x) for x in b[1:]]]
            if b and all(isinstance(x, (list, tuple)) and x and isinstance(x[0], (int, np.integer)) for x in b):
                return [[x[0]] + [list(y) if not isinstance(y, (int, np.integer)) else y for y in x[1:]] for x in b]
            shells = []
            for part in b:
                shells += one(part, symb)
            return shells
        out = {}
        for symb, b in basis.items():
            # sorted by l, stable (mole.py:463-469)
            out[symb] = sorted(one(b, symb), key=lambda x: x[0])
        return out

    def copy(self, deep=True):
        import copy
        return copy.deepcopy(self) if deep else copy.copy(self)

    # --- sizes / tables -----------------------------------------------------------
    @property
    def natm(self):
        return len(self._atm)

    @property
    def nbas(self):
        return len(self._bas)

    @property
    def nelectron(self):
        return int(self._atm[:, CHARGE_OF].sum()) - self.charge

    @property
    def nelec(self):
        ne = self.nelectron
        nalpha = (ne + self.spin) // 2
        return nalpha, ne - nalpha

    def ao_loc_nr(self):
        """moleintor.make_loc (pyscf/gto/moleintor.py:805-821), spherical."""
        l = self._bas[:, ANG_OF]
        dims = (l * 2 + 1) * self._bas[:, NCTR_OF]
        loc = np.zeros(len(dims) + 1, np.int32)
        np.cumsum(dims, out=loc[1:])
        return loc

    def nao_nr(self):
        return int(self.ao_loc_nr()[-1])

    def aoslice_by_atom(self, ao_loc=None):
        """(natm, 4): shell start, shell end, AO start, AO end of every atom (pyscf/gto/mole.py:1600-1630);
        atoms own contiguous bas rows (make_env order)."""
        if ao_loc is None:
            ao_loc = self.ao_loc_nr()
        atom_of = self._bas[:, ATOM_OF]
        out = np.zeros((self.natm, 4), dtype=np.int64)
        for ia in range(self.natm):
            idx = np.nonzero(atom_of == ia)[0]
            if len(idx):
                out[ia] = idx[0], idx[-1] + 1, ao_loc[idx[0]], ao_loc[idx[-1] + 1]
            else:
                prev = out[ia - 1] if ia else out[ia]
                out[ia] = prev[1], prev[1], prev[3], prev[3]
        return out

    @property
    def nao(self):
        return self.nao_nr()

    def atom_coords(self):
        ptr = self._atm[:, PTR_COORD]
        return np.array([self._env[p:p + 3] for p in ptr]).reshape(-1, 3)

    def atom_charges(self):
        return self._atm[:, CHARGE_OF].astype(np.int64)

    def atom_symbol(self, ia):
        return self._atom[ia][0]

    def atom_pure_symbol(self, ia):
        s = self._atom[ia][0]
        if is_ghost_atom(s):
            return 'Ghost-' + std_symbol_without_ghost(s)
        s = _rm_digit(s)
        return s[0].upper() + s[1:].lower()

    def energy_nuc(self):
        """E_nuc = sum_{i<j} Z_i Z_j / r_ij  (mole.py:1524-1548)."""
        q = self.atom_charges().astype(float)
        r = self.atom_coords()
        real = q != 0                                  # ghost atoms may sit on top of anything
        q, r = q[real], r[real]
        if len(q) < 2:
            return 0.0
        d = np.sqrt(((r[:, None, :] - r[None, :, :]) ** 2).sum(axis=2))
        i, j = np.tril_indices(len(q), -1)
        return float((q[i] * q[j] / d[i, j]).sum())

    def tot_electrons(self):
        return self.nelectron

    def bas_angular(self, ib):
        return int(self._bas[ib, ANG_OF])

    def bas_atom(self, ib):
        return int(self._bas[ib, ATOM_OF])

    def bas_exp(self, ib):
        p = self._bas[ib, PTR_EXP]
        return self._env[p:p + self._bas[ib, NPRIM_OF]].copy()

    def bas_ctr_coeff_raw(self, ib):
        """Stored (normalised) coefficients, shape (nprim, nctr)."""
        nprim, nctr = self._bas[ib, NPRIM_OF], self._bas[ib, NCTR_OF]
        p = self._bas[ib, PTR_COEFF]
        return self._env[p:p + nprim * nctr].reshape(nctr, nprim).T.copy()

    def intor(self, name, comp=None, hermi=0, aosym='s1', shls_slice=None):
        """Integrals evaluated by the HIP integral engine (mirrors Mole.intor)."""
        from . import moleintor
        return moleintor.getints(name, self._atm, self._bas, self._env,
                                 shls_slice=shls_slice, hermi=hermi, aosym=aosym)


def M(**kw):
    mol = Mole()
    mol.build(**kw)
    return mol
