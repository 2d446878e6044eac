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
    # r06: the rule lives in the library (PAMD_syrk_plan) - the Python layer and the C handle call the same function.  The old
    # Python transcription, kept here as the checker, over shapes and reserves
    def old_plan(nao, nsplit=None, flags=None, reserve=0):
        nb = -(-nao // 64)
        nt2 = nb // 2
        items = 0 if (nb % 2 == 0 or nb < 5) else nt2 * (nt2 - 1) // 2 + nt2 + -(-nt2 // 3)
        base = 1 | 2
        if flags is None:
            flags = 12 if items else 0
        base |= flags
        if nsplit:
            return base, nsplit
        if flags & 4:
            nt = -(-nao // 128)
            units = items if (flags & 8) and items else nt * (nt + 1) // 2
            best = None
            for n in range(1, 8):
                for m in range(1, 9):
                    if units * n + -(-units // m) <= 512 - reserve and (best is None or n + 1.0 / m > best[0]):
                        best = (n + 1.0 / m, n)
            if best is not None and units >= 32:
                return base, best[1] + 1
        return base & ~4, 4
    for nao in (61, 130, 257, 320, 700, 1856, 2228, 2496, 3072, 4000):
        for reserve in (0, 16, 48):
            for flags in (None, 0, 4, 12):
                assert syrk_plan(nao, None, flags, reserve) == old_plan(nao, None, flags, reserve), (nao, reserve, flags)
        assert syrk_plan(nao, 3, None, 0) == old_plan(nao, 3, None, 0)


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
