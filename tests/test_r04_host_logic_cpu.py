"""Host-side logic added in r04 that needs no GPU: the work list of PAMD_sub_vmat_sym, the options struct of PAMD_df_create_ex as
ctypes sees it, the device-list plumbing of density_fit (no handle is built), the XC golden generator's energy bookkeeping."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from pyscf_amd.df import native
    return native.load()


def test_sub_vmat_work_list_covers_every_block_pair_once_and_balances_the_xcd_queues():
    so = _lib()
    so.PAMD_sub_vmat_work.restype = C.c_long
    rng = np.random.RandomState(4)
    ld = np.ascontiguousarray(rng.randint(1, 60, size=257) * 16, dtype=np.int32)
    ld[5] = 16
    ld[9] = 8 * 16 * 3 + 16                       # 25 groups: pieces 7 + 6 + 6 + 6
    n = so.PAMD_sub_vmat_work(ld.ctypes.data_as(C.c_void_p), len(ld), None)
    w = np.zeros(n * 6, np.int32)
    assert so.PAMD_sub_vmat_work(ld.ctypes.data_as(C.c_void_p), len(ld), w.ctypes.data_as(C.c_void_p)) == n
    it = w.reshape(-1, 6)
    live = it[it[:, 2] > 0]
    assert n % 8 == 0 and np.all(live[:, 2] <= 8) and np.all(live[:, 4] <= 8) and np.all(live[:, 4] > 0)
    for t in range(len(ld)):
        mine = live[live[:, 0] == t]
        g = ld[t] // 16
        diag = mine[mine[:, 5] == 1]
        pieces = sorted((int(p0), int(gp)) for p0, gp in diag[:, 1:3])
        # the pieces tile [0, ld) without gaps; r06: sizes are even (at most ONE odd piece, for an odd group count) and differ by
        # at most two groups - an even piece splits evenly between the two wave rows / columns of a workgroup
        pos = 0
        for p0, gp in pieces:
            assert p0 == pos
            pos += gp * 16
        assert pos == ld[t] and max(gp for _, gp in pieces) - min(gp for _, gp in pieces) <= 2
        assert sum(gp & 1 for _, gp in pieces) == (g & 1)
        npc = len(pieces)
        assert npc == -(-g // 8) and len(mine) == npc * (npc + 1) // 2
        # every pair (i >= j) exactly once, none above the diagonal
        pairs = {(int(r[1]), int(r[3])) for r in mine}
        assert len(pairs) == len(mine) and all(a >= b for a, b in pairs)
        # all items of a tile sit in ONE dispatch queue (position % 8)
        where = np.nonzero((it[:, 0] == t) & (it[:, 2] > 0))[0]
        assert len(set(where % 8)) == 1
    cost = ((it[:, 2] + 1) // 2) * ((it[:, 4] + 1) // 2) * np.where(it[:, 5] == 1, 1, 2)
    per = np.array([cost[k::8].sum() for k in range(8)], dtype=float)
    assert per.max() - per.min() <= 0.02 * per.mean() + 32, per
    # no tiles at all
    assert so.PAMD_sub_vmat_work(ld.ctypes.data_as(C.c_void_p), 0, None) == 0


def test_df_options_struct_matches_the_header():
    from pyscf_amd.df import native
    hdr = open(os.path.join(ROOT, 'include', 'pyscf_amd.h')).read()
    body = re.search(r'typedef struct PAMD_df_options \{(.*?)\} PAMD_df_options;', hdr, re.S).group(1)
    names = re.findall(r'(\w+);\s*(?:/\*.*?\*/)?\s*$', body, re.M)
    assert names == [f[0] for f in native._Options._fields_], (names, native._Options._fields_)
    assert C.sizeof(native._Options) == 8 + 8 + 8 + 4 + 4 + 8 + 4 + 4 + 8          # r06: + reserve_bytes


def test_density_fit_device_list_plumbing_without_a_device():
    from pyscf_amd import gto, scf, dft
    from pyscf_amd.df.native import NativeDF
    from pyscf_amd.dft.native import NativeNumInt
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='sto-3g')
    mf = scf.RHF(mol).density_fit(devices=range(4))
    assert isinstance(mf.with_df, NativeDF) and mf.with_df.devices == [0, 1, 2, 3] and mf.with_df._h is None
    ks = dft.RKS(mol, xc='b3lyp').density_fit(auxbasis='weigend', devices=[2, 3])
    assert isinstance(ks.with_df, NativeDF) and isinstance(ks._numint, NativeNumInt) and ks._numint.devices == [2, 3]
    assert ks._numint.rsh_and_hybrid_coeff('b3lyp')[2] == 0.2 and ks._numint._xc_type('b3lyp') == 'GGA'
    os.environ['PAMD_DEVICES'] = '1, 0'
    try:
        m2 = scf.RHF(mol).density_fit()
    finally:
        del os.environ['PAMD_DEVICES']
    assert isinstance(m2.with_df, NativeDF) and m2.with_df.devices == [1, 0]
    m3 = scf.RHF(mol).density_fit()
    assert not isinstance(m3.with_df, NativeDF)
    # range_coulomb keeps one handle object per omega, reset drops them
    a = mf.with_df.range_coulomb(0.3)
    assert a is mf.with_df.range_coulomb(0.3) and a.omega == 0.3 and a.devices == [0, 1, 2, 3] and mf.with_df.range_coulomb(0) is mf.with_df
    mf.with_df.reset()
    assert mf.with_df._rsh_df == {}


def test_xc_golden_generator_combines_the_oracle_energy_functional(tmp_path):
    """tools/gen_golden_xc.py --combine-only: E_RKS[D] = E_RHF[D] + (1 - hyb)/4 Tr(D K) + E_xc from two oracle-only files."""
    g = {'conv_e_rhf_functional': -10.0, 'conv_tr_d_vk': 4.0, 'conv_e_tot_of_the_orbital_source': -11.5}
    jk = tmp_path / 'jk.json'
    jk.write_text(json.dumps(g))
    tag = '_unit_test_combine'
    out = os.path.join(ROOT, 'tests', 'golden', tag + '_oracle.json')
    try:
        with open(out, 'w') as f:
            json.dump({'xc_x_exc': -2.0}, f)
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_golden_xc.py'), '--molecule', 'water', '--nwater', '1', '--xc', 'b3lyp',
                            '--orbitals', 'x', '--combine-only', '--jk-json', 'x=%s' % jk, '--tag', tag], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        r = json.load(open(out))
        assert abs(r['xc_x_e_rks_functional'] - (-10.0 + 0.25 * 0.8 * 4.0 - 2.0)) < 1e-12
    finally:
        if os.path.exists(out):
            os.remove(out)


def test_orbital_leading_dimension_covers_every_half_transform_tiling():
    """PAMD_e2_orb_ld (host function): whole chunks of the exact-tile kernels (32 x wave-row tiles, <= 10 tiles per chunk), of the
    uniform 160 / 128-orbital tilings and of the 128-orbital chunks with a narrower last chunk (r04, df_jk.hip::v2_wide)."""
    so = _lib()
    ld = lambda n: int(so.PAMD_e2_orb_ld(C.c_int(n)))
    assert ld(0) == 0
    for n in list(range(1, 700, 7)) + [16, 128, 160, 226, 240, 256, 1856, 2240]:
        v, p16 = ld(n), -(-n // 16) * 16
        assert v >= p16 and v % 16 == 0
        assert v >= min(-(-p16 // 160) * 160, -(-p16 // 128) * 128)
        t = p16 // 16
        nch = -(-t // 8)
        r = t - 8 * (nch - 1)
        cost = 8 * (nch - 1) + max(r, 4)
        uniform = min(-(-p16 // 160) * 10, -(-p16 // 128) * 8)
        if r < 8 and cost < uniform and cost * 16 <= 1.30 * p16:
            assert v >= nch * 128, n                     # room for the wide last chunk's 128-column panel
    assert ld(160) == 160 and ld(16) == 128              # config 3 unchanged; tiny operands already took one 128-column chunk
    assert ld(226) == ld(240) == 320 and ld(400) == 512


def test_namespace_plugin_is_found_through_PYSCF_EXT_PATH(tmp_path):
    """`PYSCF_EXT_PATH=<repo>/plugin` makes a `pyscf` namespace package find `pyscf.amd`
    (pyscf/__init__.py:42-60 appends <dir>/pyscf to pyscf.__path__ when <dir> contains a `pyscf` folder).  PySCF itself cannot be
    imported in this container, so a stub package that applies the same rule stands in for it."""
    import subprocess
    import sys
    stub = tmp_path / 'site' / 'pyscf'
    stub.mkdir(parents=True)
    (stub / '__init__.py').write_text(
        "import os\n"
        "for p in (os.getenv('PYSCF_EXT_PATH') or '').split(':'):\n"
        "    if os.path.isdir(p) and 'pyscf' in os.listdir(p):\n"
        "        __path__.append(os.path.join(p, 'pyscf'))\n")
    code = ("import sys; sys.path.insert(0, %r); import pyscf.amd as amd; "
            "from pyscf_amd.df.native import NativeDF; from pyscf_amd.dft.native import NativeNumInt; "
            "assert issubclass(amd.DF, NativeDF) and amd.NumInt is NativeNumInt and callable(amd.density_fit); "
            "from pyscf_amd import gto, scf, dft; "
            "mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='sto-3g'); "
            "mf = amd.density_fit(dft.RKS(mol, xc='b3lyp'), auxbasis='weigend', devices=[0, 1]); "
            "assert isinstance(mf.with_df, amd.DF) and mf.with_df.devices == [0, 1] and isinstance(mf._numint, NativeNumInt); "
            "print('PLUGIN_OK')" % str(tmp_path / 'site'))
    env = dict(os.environ, PYSCF_EXT_PATH=os.path.join(ROOT, 'plugin'))
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and 'PLUGIN_OK' in p.stdout, p.stdout + p.stderr[-2000:]
