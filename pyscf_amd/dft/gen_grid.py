"""Becke/Lebedev molecular integration grids.

Mirror of ``pyscf/dft/gen_grid.py``: ``nwchem_prune`` (:90-134), ``treutler_prune`` (:136-158),
``sg1_prune`` (:50-88), ``gen_atomic_grids`` (:254-338), ``get_partition`` (:341-419),
``arg_group_grids`` (:449-471), ``Grids`` (:487-744), level tables (:747-785).  Atomic grids
are assembled on the host (one-time, small); the Becke partition - O(ngrids * natm^2), the only
heavy step - runs on the device (``PAMD_grid_partition``, the analogue of ``VXCgen_grid`` /
``VXCgen_grid_lko``, pyscf/lib/dft/grid_basis.c:32-101,266-384, and of the generic Stratmann branch :388-404).  Lebedev tables: pyscf_amd/dft/lebedev.npz.
"""
import ctypes
import os

import numpy as np

from .. import lib as _lib_mod
from ..gto.mole import element_charge as _charge        # ghost atoms get the grid of their element (gen_grid.py:297)
from . import radi

GROUP_BOX_SIZE = 1.2
GROUP_BOUNDARY_PENALTY = 4.2
ALIGNMENT_UNIT = 8
CUTOFF = 1e-15

_leb = None


def _lebedev():
    global _leb
    if _leb is None:
        _leb = np.load(os.path.join(os.path.dirname(__file__), 'lebedev.npz'))
    return _leb


def _order_table():
    return {int(k): int(v) for k, v in _lebedev()['order']}


LEBEDEV_NGRID = np.array([1, 6, 14, 26, 38, 50, 74, 86, 110, 146, 170, 194, 230, 266, 302, 350, 434, 590])


def MakeAngularGrid(n):
    return _lebedev()['n%d' % n]


_PRUNE_ALPHAS = np.array(((0.25, 0.5, 1.0, 4.5), (0.1667, 0.5, 0.9, 3.5), (0.1, 0.4, 0.8, 2.5)))


def _prune_region(nuc, rads, radii):
    """Index 0..4 of the radial region each shell falls in (inner core ... outer tail) for H-He / Li-Ne / heavier."""
    row = 0 if nuc <= 2 else (1 if nuc <= 10 else 2)
    return ((rads / (radii[nuc] + 1e-200)).reshape(-1, 1) > _PRUNE_ALPHAS[row]).sum(axis=1)


def sg1_prune(nuc, rads, n_ang, radii=radi.SG1RADII):
    """SG-1 angular orders 6/38/86/194/86 over the five regions, whatever n_ang is (gen_grid.py:53-88)."""
    return np.array([6, 38, 86, 194, 86])[_prune_region(nuc, rads, radii)]


def nwchem_prune(nuc, rads, n_ang, radii=radi.BRAGG_RADII):
    leb_ngrid = LEBEDEV_NGRID[4:]
    if n_ang < 50:
        return np.repeat(n_ang, len(rads))
    elif n_ang == 50:
        leb_l = np.array([1, 2, 2, 2, 1])
    else:
        idx = np.where(leb_ngrid == n_ang)[0][0]
        leb_l = np.array([1, 3, idx - 1, idx, idx - 1])
    return leb_ngrid[leb_l[_prune_region(nuc, rads, radii)]]


def treutler_prune(nuc, rads, n_ang, radii=None):
    nr = len(rads)
    leb_ngrid = np.empty(nr, dtype=int)
    leb_ngrid[:nr // 3] = 14
    leb_ngrid[nr // 3:nr // 2] = 50
    leb_ngrid[nr // 2:] = n_ang
    return leb_ngrid


RAD_GRIDS = np.array(((10, 15, 20, 30, 35, 40, 50), (30, 40, 50, 60, 65, 70, 75),
                      (40, 60, 65, 75, 80, 85, 90), (50, 75, 80, 90, 95, 100, 105),
                      (60, 90, 95, 105, 110, 115, 120), (70, 105, 110, 120, 125, 130, 135),
                      (80, 120, 125, 135, 140, 145, 150), (90, 135, 140, 150, 155, 160, 165),
                      (100, 150, 155, 165, 170, 175, 180), (200, 200, 200, 200, 200, 200, 200)))
ANG_ORDER = np.array(((11, 15, 17, 17, 17, 17, 17), (17, 23, 23, 23, 23, 23, 23),
                      (23, 29, 29, 29, 29, 29, 29), (29, 29, 35, 35, 35, 35, 35),
                      (35, 41, 41, 41, 41, 41, 41), (41, 47, 47, 47, 47, 47, 47),
                      (47, 53, 53, 53, 53, 53, 53), (53, 59, 59, 59, 59, 59, 59),
                      (59, 59, 59, 59, 59, 59, 59), (65, 65, 65, 65, 65, 65, 65)))
_PERIOD_TAB = np.array((2, 10, 18, 36, 54, 86, 118))


def _default_rad(nuc, level=3):
    return int(RAD_GRIDS[level, (nuc > _PERIOD_TAB).sum()])


def _default_ang(nuc, level=3):
    return _order_table()[int(ANG_ORDER[level, (nuc > _PERIOD_TAB).sum()])]


def gen_atomic_grids(mol, atom_grid={}, radi_method=radi.gauss_chebyshev, level=3, prune=nwchem_prune):
    """{symbol: (coords relative to the atom, volume weights)}  (gen_grid.py:254-338)."""
    if isinstance(atom_grid, (list, tuple)):
        atom_grid = {mol.atom_symbol(ia): atom_grid for ia in range(mol.natm)}
    default = atom_grid.get('default', None)
    tab = {}
    for ia in range(mol.natm):
        symb = mol.atom_symbol(ia)
        if symb in tab:
            continue
        chg = _charge(symb)
        conf = atom_grid.get(symb, default)
        if conf is not None:
            n_rad, n_ang = conf
            if n_ang not in LEBEDEV_NGRID:
                if n_ang not in _order_table():
                    raise ValueError('Unsupported angular grids %d' % n_ang)
                n_ang = _order_table()[n_ang]           # a Lebedev order was given: use its point count (gen_grid.py:301-306)
        else:
            n_rad, n_ang = _default_rad(chg, level), _default_ang(chg, level)
        rad, dr = radi_method(n_rad, chg, ia)
        rad_weight = 4 * np.pi * rad ** 2 * dr
        angs = np.array(prune(chg, rad, n_ang) if callable(prune) else [n_ang] * n_rad)
        coords, vol = [], []
        for n in sorted(set(angs)):
            grid = MakeAngularGrid(n)
            idx = np.where(angs == n)[0]
            for i0 in range(0, len(idx), 12):          # 12 radial shells per group
                sel = idx[i0:i0 + 12]
                coords.append(np.einsum('i,jk->jik', rad[sel], grid[:, :3]).reshape(-1, 3))
                vol.append(np.einsum('i,j->ji', rad_weight[sel], grid[:, 3]).ravel())
        tab[symb] = (np.vstack(coords), np.hstack(vol))
    return tab


def original_becke(g):
    """Marker for Becke's thrice-iterated cell polynomial (gen_grid.py:214-221); evaluated in the partition kernel."""
    raise NotImplementedError('original_becke selects a device kernel; it is not evaluated on the host')


def stratmann(g):
    """Marker for the Stratmann-Scuseria-Frisch cell function, a = 0.64 (gen_grid.py:203-212)."""
    raise NotImplementedError('stratmann selects a device kernel; it is not evaluated on the host')


def becke_lko(g):
    """Marker for the Laqua-Kussmann-Ochsenfeld partition (gen_grid.py:223-237, grid_basis.c:266-384)."""
    raise NotImplementedError('becke_lko selects a device kernel; it is not evaluated on the host')


_SCHEME_ID = {original_becke: 0, 'original_becke': 0, 'becke': 0, stratmann: 1, 'stratmann': 1, becke_lko: 2, 'becke_lko': 2,
              'lko': 2}


def scheme_id(becke_scheme):
    try:
        return _SCHEME_ID[becke_scheme]
    except (KeyError, TypeError):
        raise NotImplementedError('becke_scheme %r: original_becke, stratmann and becke_lko are built' % (becke_scheme,))


def becke_partition_gpu(coords, atm_coords, radii_table, device, scheme=0):
    """pbecke[natm][ngrids] on the device (VXCgen_grid / VXCgen_grid_lko / generic get_partition analogue)."""
    import torch
    lib = _lib_mod.load_library()
    ngrids, natm = len(coords), len(atm_coords)
    c = torch.from_numpy(np.ascontiguousarray(coords)).to(device)
    a = torch.from_numpy(np.ascontiguousarray(atm_coords)).to(device)
    out = torch.empty((natm, ngrids), dtype=torch.float64, device=device)
    if radii_table is None:
        rt_ptr = ctypes.c_void_p(0)
    else:
        rt = torch.from_numpy(np.ascontiguousarray(radii_table)).to(device)
        rt_ptr = ctypes.c_void_p(rt.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib_mod.check(lib.PAMD_grid_partition(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(c.data_ptr()),
                                           ctypes.c_void_p(a.data_ptr()), rt_ptr, ctypes.c_int(natm),
                                           ctypes.c_long(ngrids), ctypes.c_int(scheme), st))
    return out


def arg_group_grids(mol, coords, box_size=GROUP_BOX_SIZE):
    atom_coords = mol.atom_coords()
    boundary = [atom_coords.min(axis=0) - GROUP_BOUNDARY_PENALTY, atom_coords.max(axis=0) + GROUP_BOUNDARY_PENALTY]
    boxes = ((boundary[1] - boundary[0]) * (1. / box_size)).round().astype(int)
    box_size = (boundary[1] - boundary[0]) / boxes
    frac = (coords - boundary[0]) * (1. / box_size)
    box_ids = np.floor(frac).astype(int)
    box_ids[box_ids < -1] = -1
    for k in range(3):
        box_ids[box_ids[:, k] > boxes[k], k] = boxes[k]
    # the reference ranks the boxes with numpy.unique(box_ids, axis=0) (lexicographic in the three box indices) and sorts the
    # points stably by that rank (gen_grid.py:386-388).  One scalar key per point that is monotone in the same lexicographic
    # order gives the SAME permutation from a single stable sort (r06: 0.6 s -> 0.1 s of a config-3 SCF's set-up)
    n1, n2 = int(boxes[1]) + 2, int(boxes[2]) + 2
    key = ((box_ids[:, 0] + 1).astype(np.int64) * n1 + (box_ids[:, 1] + 1)) * n2 + (box_ids[:, 2] + 1)
    return key.argsort(kind='stable')


def make_mask(mol, coords, relativity=0, shls_slice=None, cutoff=1e-15, verbose=None):
    """pyscf/dft/gen_grid.py:422-447: the shell mask of the reference (uint8 screen index per block of 56 grid points and shell;
    0 = ignorable).  Not used by the product's own XC path, which screens on AO values (dft/sparse_grid.py)."""
    from ..gto.eval_gto import make_screen_index
    return make_screen_index(mol, coords, shls_slice, cutoff)


class Grids:
    """pyscf/dft/gen_grid.py:487-744 (defaults :565-576)."""

    def __init__(self, mol):
        self.mol = mol
        self.atomic_radii = radi.BRAGG_RADII
        self.radii_adjust = radi.treutler_atomic_radii_adjust
        self.radi_method = radi.treutler
        self.becke_scheme = original_becke
        self.prune = nwchem_prune
        self.level = 3
        self.alignment = ALIGNMENT_UNIT
        self.atom_grid = {}
        self.device = None
        self.coords = self.weights = self.atm_idx = self.quadrature_weights = None
        self.non0tab = self.screen_index = None
        self._build_id = 0          # bumped by build() / reset(): device-side caches of coords / weights key on it

    @property
    def size(self):
        return 0 if self.weights is None else self.weights.size

    def reset(self, mol=None):
        if mol is not None:
            self.mol = mol
        self.coords = self.weights = self.atm_idx = self.quadrature_weights = None
        self._build_id = getattr(self, '_build_id', 0) + 1
        return self

    def _device(self):
        import torch
        if self.device is not None:
            return torch.device(self.device)
        if not torch.cuda.is_available():
            raise RuntimeError('Grids.build: the Becke partition runs on the HIP device; none is visible')
        return torch.device('cuda', torch.cuda.current_device())

    def get_partition(self, mol, atom_grids_tab):
        scheme = scheme_id(self.becke_scheme)
        table = None
        if callable(self.radii_adjust) and self.atomic_radii is not None:
            table = self.radii_adjust(mol, self.atomic_radii)
        atm_coords = mol.atom_coords()
        dev = self._device()
        coords_all, weights_all = [], []
        for ia in range(mol.natm):
            c, vol = atom_grids_tab[mol.atom_symbol(ia)]
            c = c + atm_coords[ia]
            pb = becke_partition_gpu(c, atm_coords, table, dev, scheme)
            w = (pb[ia] / pb.sum(dim=0)).cpu().numpy() * vol
            coords_all.append(c)
            weights_all.append(w)
        return np.vstack(coords_all), np.hstack(weights_all)

    def build(self, mol=None, with_non0tab=False, sort_grids=True):
        if mol is None:
            mol = self.mol
        self._build_id = getattr(self, '_build_id', 0) + 1
        tab = gen_atomic_grids(mol, self.atom_grid, self.radi_method, self.level, self.prune)
        self.coords, self.weights = self.get_partition(mol, tab)
        atm_idx = np.empty(len(self.weights), np.int32)
        qw = np.empty(len(self.weights))
        p1 = 0
        for ia in range(mol.natm):
            vol = tab[mol.atom_symbol(ia)][1]
            p0, p1 = p1, p1 + vol.size
            atm_idx[p0:p1] = ia
            qw[p0:p1] = vol
        self.atm_idx, self.quadrature_weights = atm_idx, qw
        if sort_grids:
            idx = arg_group_grids(mol, self.coords)
            self.coords, self.weights = self.coords[idx], self.weights[idx]
            self.atm_idx, self.quadrature_weights = self.atm_idx[idx], self.quadrature_weights[idx]
        if self.alignment > 1:
            pad = (self.size + self.alignment - 1) // self.alignment * self.alignment - self.size
            if pad > 0:
                self.coords = np.vstack([self.coords, np.repeat([[1e-4] * 3], pad, axis=0)])
                self.weights = np.hstack([self.weights, np.zeros(pad)])
                self.atm_idx = np.hstack([self.atm_idx, np.full(pad, -1, np.int32)])
                self.quadrature_weights = np.hstack([self.quadrature_weights, np.zeros(pad)])
        if with_non0tab:                            # gen_grid.py:615-620: screen_index = non0tab = make_mask(...)
            self.non0tab = self.screen_index = make_mask(mol, self.coords, cutoff=getattr(self, 'cutoff', 1e-15))
        else:
            self.non0tab = self.screen_index = None
        return self

    kernel = build
