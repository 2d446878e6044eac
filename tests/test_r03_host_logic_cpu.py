"""CPU tests of the round-3 host logic: the device SCF loop's DIIS against the host CDIIS (pyscf/scf/diis.py:40-96 +
pyscf/lib/diis.py:225-290 restated in scf/hf.py), the SYRK plan, the collective switch, the numpy-only client's conc_env."""
import numpy as np
import pytest


def test_device_diis_equals_host_cdiis():
    import torch
    from pyscf_amd.scf import hf
    from pyscf_amd.scf.device_scf import DeviceDIIS
    rng = np.random.default_rng(2)
    n = 24
    a = rng.standard_normal((n, n))
    s = a.dot(a.T) / n + np.eye(n)
    w, v = np.linalg.eigh(s)
    x = v / np.sqrt(w)
    host = hf.CDIIS(4)
    host.Corth = x
    host.device_linalg = False
    dev = DeviceDIIS(4, torch.from_numpy(x))
    for it in range(7):                                   # more updates than the subspace holds
        d = rng.standard_normal((n, n))
        d = d + d.T
        f = rng.standard_normal((n, n))
        f = f + f.T
        out_h = host.update(s, d, f)
        out_d = dev.update(torch.from_numpy(s), torch.from_numpy(d), torch.from_numpy(f)).numpy()
        assert np.abs(out_h - out_d).max() < 1e-9 * max(1.0, np.abs(out_h).max()), it


def test_syrk_plan_and_items():
    from pyscf_amd.df.df_jk import syrk_items, syrk_plan
    # nao = 1856: 29 blocks of 64 -> 91 off-diagonal 2 x 2 items + 14 diagonal combos + 5 items for the rest of the last row
    assert syrk_items(1856) == 110 and syrk_items(2228) == 159
    assert syrk_items(3072) == 0 and syrk_items(128) == 0            # even block counts / tiny: the 2 x 2 tiling stays
    # every live 64-block is covered exactly once: count them
    for nao in (1856, 2228, 700, 300):
        nb = -(-nao // 64)
        nt = nb // 2
        live = 4 * (nt * (nt - 1) // 2) + 4 * nt + nt + 1          # 2x2 items, diagonal combos (3 + 1), rest blocks + corner
        assert live == nb * (nb + 1) // 2
    flags, nsplit = syrk_plan(1856)
    assert flags == 1 | 2 | 4 | 8 and nsplit == 5                     # 4 full pieces + one half piece: 440 + 55 <= 512 slots
    assert 110 * 4 + -(-110 // 2) <= 512 < 110 * 4 + 110
    assert syrk_plan(1856, None, 0) == (3, 4) and syrk_plan(464) == (3, 4)
    assert syrk_plan(1856, 7)[1] == 7                                  # explicit split count wins


def test_collective_switch():
    from pyscf_amd.lib import comm
    comm.force(None)
    assert not comm.active(1) and comm.active(2)
    comm.force(True)
    try:
        assert comm.active(1) == comm.initialized()                   # forced, but only when a process group exists
    finally:
        comm.force(None)
    assert comm.all_reduce([], world=1) is False


def test_native_client_conc_env_matches_gto():
    from pyscf_amd import gto
    from pyscf_amd.df import addons, native
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='cc-pvdz')
    aux = addons.make_auxmol(mol, 'weigend')
    a1 = native._conc_env(np.asarray(mol._atm), np.asarray(mol._bas), np.asarray(mol._env), np.asarray(aux._atm),
                          np.asarray(aux._bas), np.asarray(aux._env))
    a0 = gto.conc_env(mol._atm, mol._bas, mol._env, aux._atm, aux._bas, aux._env)
    for x, y in zip(a0, a1):
        assert np.array_equal(np.asarray(x), y)
    assert a1[0].dtype == np.int32 and a1[1].dtype == np.int32 and a1[2].dtype == np.float64
