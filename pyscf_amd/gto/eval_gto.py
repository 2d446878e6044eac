"""Shell screening index on grid blocks: host mirror of ``pyscf/gto/eval_gto.py:155-206`` (``make_screen_index``) ->
``GTO_screen_index`` (``pyscf/lib/gto/grid_ao_drv.c:32-123``).

The product's XC path does NOT use this estimate: its block-sparse plan screens on the AO VALUES themselves per (grid tile,
shell) (dft/sparse_grid.py).  The function exists so that the reference's mask - and its golden fingerprint
``lib.fp(non0tab) = -83.54934301013405`` (pyscf/dft/test/test_grids.py:132-140) - is available to callers that ask for it
(``Grids.build(with_non0tab=True)``, ``gen_grid.make_mask``) and pinned.

screen_index[block][shell] (uint8): 0 = the shell is negligible on all ``blksize`` points of the block, otherwise
``min(255, nbins - scale * arr + 1)`` with ``arr = alpha_min r_min^2 - (l/2) ln r_min^2 - ln c_max`` at the closest point of the
block (the value of the most diffuse primitive's exponent argument), ``scale = -nbins / ln(min(cutoff, 0.1))``.
"""
import numpy as np

from . import mole as _mole

BLKSIZE = 56      # lib/gto/grid_ao_drv.h:30-36
NBINS = 100
CUTOFF = 1e-15


def make_screen_index(mol, coords, shls_slice=None, cutoff=CUTOFF, blksize=BLKSIZE):
    assert NBINS < 120
    coords = np.asarray(coords, dtype=np.float64)
    ngrids = len(coords)
    atm, bas, env = np.asarray(mol._atm), np.asarray(mol._bas), np.asarray(mol._env)
    sh0, sh1 = (0, len(bas)) if shls_slice is None else shls_slice
    nbas = sh1 - sh0
    nblk = (ngrids + blksize - 1) // blksize
    scale = -NBINS / np.log(min(cutoff, .1))
    out = np.zeros((nblk, nbas), dtype=np.uint8)
    # closest squared distance of every block to every atom
    pad = nblk * blksize - ngrids
    for ib_, b in enumerate(bas[sh0:sh1]):
        npr, nc, l = int(b[_mole.NPRIM_OF]), int(b[_mole.NCTR_OF]), int(b[_mole.ANG_OF])
        exps = env[b[_mole.PTR_EXP]:b[_mole.PTR_EXP] + npr]
        coef = env[b[_mole.PTR_COEFF]:b[_mole.PTR_COEFF] + npr * nc]
        r = env[atm[b[_mole.ATOM_OF], _mole.PTR_COORD]:atm[b[_mole.ATOM_OF], _mole.PTR_COORD] + 3]
        min_exp = exps.min()
        log_coeff = np.log(np.abs(coef).max())
        rr = ((coords - r) ** 2).sum(axis=1)
        if pad:
            rr = np.concatenate([rr, np.full(pad, 1e9)])
        rr_min = np.minimum(rr.reshape(nblk, blksize).min(axis=1), 1e9)
        if l == 0:
            arr = min_exp * rr_min - log_coeff
        else:
            r2sup = l / (2. * min_exp)
            arr_min = min_exp * r2sup - .5 * np.log(r2sup) * l - log_coeff
            with np.errstate(divide='ignore'):
                far = min_exp * rr_min - .5 * np.log(rr_min) * l - log_coeff
            arr = np.where(rr_min < r2sup, arr_min, far)
        si = NBINS - arr * scale
        col = np.where(si <= 0, 0, np.where(si > 254, 255, np.floor(si + 1))).astype(np.uint8)     # (uint8_t)(si + 1) truncates
        out[:, ib_] = col
    return out
