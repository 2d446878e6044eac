"""CPU-only: the C-ABI library loads and exports every symbol declared in include/pyscf_amd.h;
host-side logic (Mole tables, aux selection, shard ranges, pair tables, c2s matrices)."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.conftest import H2O, ROOT


def test_header_symbols_exported():
    from pyscf_amd import lib
    so = lib.load_library()
    hdr = open(os.path.join(ROOT, 'include', 'pyscf_amd.h')).read()
    names = sorted(set(re.findall(r'\b(PAMD_[a-z0-9_]+)\s*\(', hdr)))
    assert len(names) >= 18
    for n in names:
        assert hasattr(so, n), 'missing symbol %s' % n
    assert so.PAMD_version() >= 100
    assert so.PAMD_device_count() >= 0            # no compute call without a GPU
    assert so.PAMD_rys_table_len() == 38976


def test_host_side_size_functions():
    """Pure host arithmetic of the C ABI (no device call): workspace sizes a caller allocates before the launches."""
    from pyscf_amd import lib
    so = lib.load_library()
    so.PAMD_e2_diag_size.restype = ctypes.c_long
    # diag[nL][ceil(ldx / 128)][128][128] doubles: BASELINE config 3 (ldx = 1856 -> 15 blocks) is 8.7 GB, 14 % of the packed tensor
    assert so.PAMD_e2_diag_size(4448, 1856) == 4448 * 15 * 128 * 128
    assert abs(so.PAMD_e2_diag_size(4448, 1856) / (4448 * (1856 * 1857 // 2)) - 0.1426) < 1e-3
    assert so.PAMD_e2_diag_size(3, 129) == 3 * 2 * 128 * 128 and so.PAMD_e2_diag_size(0, 500) == 0
    # one partial per wave of every (aux row, column tile, orbital chunk) workgroup; monotone in every argument
    w = so.PAMD_nr_e2_rho_worksize
    assert w(4448, 1856, 160) >= 4448 * 15 * 1 * 4 and w(10, 1856, 160) <= w(11, 1856, 160) <= w(11, 1857, 160) <= w(11, 1857, 336)
    so.PAMD_df_vj_pass1_worksize.restype = ctypes.c_long
    assert so.PAMD_df_vj_pass1_worksize(ctypes.c_long(1856 * 1857 // 2), 4448, 1) > 0


def test_args_struct_matches_header():
    from pyscf_amd.gto import moleintor
    hdr = open(os.path.join(ROOT, 'include', 'pyscf_amd.h')).read()
    body = hdr[hdr.index('typedef struct PAMD_int3c2e_args {'):hdr.index('} PAMD_int3c2e_args;')]
    fields = re.findall(r'\b(\w+);', re.sub(r'/\*.*?\*/', '', body, flags=re.S))
    assert fields == [f[0] for f in moleintor._Args._fields_]


def test_no_cpu_fallback_without_device():
    """The product path must fail loudly when no HIP device is present."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from pyscf_amd import gto, df
    mol = gto.M(atom=H2O, basis='sto-3g')
    with pytest.raises(RuntimeError):
        df.DF(mol, 'weigend').build()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'pyscf_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(d, f)).read()
                assert 'oracle' not in src.replace('test oracle', ''), os.path.join(d, f)


def test_mole_tables_and_aux_selection():
    from pyscf_amd import gto, df
    mol = gto.M(atom=H2O, basis='cc-pvdz')
    assert (mol.nao, mol.nbas, mol.nelectron) == (24, 11, 10)
    assert abs(mol.energy_nuc() - 9.188258417746113) < 1e-12
    assert abs(gto.gto_norm(0, 1.0) - 2.5264751109842591) < 1e-14        # mole.py:127-157 docstring
    # general contraction row kept as one bas row with nctr=2 (mole.py:986-1018)
    assert tuple(mol._bas[0][[1, 2, 3]]) == (0, 8, 2)
    assert df.make_auxmol(mol).nao == 116                                 # df/test/test_df.py:53
    assert df.make_auxmol(mol, 'weigend').nao == 71
    tz = gto.M(atom=H2O, basis='cc-pvtz')
    assert tz.nao == 58 and df.make_auxmol(tz).nao == 139


def test_shard_ranges_cover_and_balance():
    from pyscf_amd.df import DF
    for naux, world in [(4448, 8), (14848, 8), (71, 2), (5, 8), (4448, 1)]:
        r = [DF.shard_range(naux, k, world) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == naux
        assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1


def test_c2s_matrices_match_oracle_and_are_orthonormal():
    from oracle import ref
    from pyscf_amd.gto.moleintor import c2s_matrix
    for l in range(5):
        m = c2s_matrix(l)
        nc = (l + 1) * (l + 2) // 2
        o = np.zeros((2 * l + 1, nc))
        ref.lib().oracle_c2s_matrix(ctypes.c_int(l), o.ctypes.data_as(ctypes.c_void_p))
        assert np.abs(m - o).max() < 1e-14
    d = c2s_matrix(2)           # d_xy = sqrt(15/4pi) xy, p order x,y,z
    assert abs(d[0, 1] - 1.092548430592079070) < 1e-15
    assert np.allclose(c2s_matrix(1), np.eye(3) * 0.488602511902919921)


def test_pair_tables_host_side():
    """Pair classes partition all significant shell pairs; row-shell slabs are contiguous."""
    from pyscf_amd import gto, df
    from pyscf_amd.gto.moleintor import IntEngine
    mol = gto.M(atom=H2O, basis='cc-pvtz')
    aux = df.make_auxmol(mol)
    eng = IntEngine(mol, aux, 'cpu')
    nsh = eng.ao.n
    assert eng.ao.nao == 58 and nsh == 22         # O: 2 contracted s split -> segmented shells
    tot = sum(pc.n for pc in eng.pair_classes())
    assert tot == nsh * (nsh + 1) // 2            # nothing screened in one water molecule
    for pc in eng.pair_classes():
        assert np.all(np.diff(pc.rowshell) >= 0)
        i0, i1 = pc.subrange(0, nsh)
        assert (i0, i1) == (0, pc.n)
        mid = nsh // 2
        a0, a1 = pc.subrange(0, mid)
        b0, b1 = pc.subrange(mid, nsh)
        assert a0 == 0 and a1 == b0 and b1 == pc.n
    r0, r1 = eng.slab_rows(0, nsh)
    assert (r0, r1) == (0, 58 * 59 // 2)


def test_grid_block_assignment():
    """nr_rks deals grid blocks round-robin over the ranks: every grid point is covered exactly once and
    the block count is a multiple of the world size (balanced)."""
    from pyscf_amd.dft.numint import grid_block_size
    for ngrids, max_rows, world in [(1078336, 100000, 1), (1078336, 100000, 8), (1078336, 100000, 3),
                                    (5000, 100000, 4), (300, 1 << 20, 2)]:
        blk = grid_block_size(ngrids, max_rows, world)
        assert blk % 256 == 0 and 256 <= blk <= max(256, max_rows // 256 * 256)
        starts = list(range(0, ngrids, blk))
        covered = sum(min(blk, ngrids - g0) for g0 in starts)
        assert covered == ngrids
        owners = [b % world for b in range(len(starts))]
        counts = [owners.count(r) for r in range(world)]
        assert max(counts) - min(counts) <= 1


def test_grad_host_pieces():
    """Host-side pieces of the gradient path: nuclear-repulsion gradient (pyscf/grad/rhf.py:148-166) against finite
    differences of Mole.energy_nuc, aoslice_by_atom (gto/mole.py), the ctypes mirror of PAMD_int3c2e_grad_args."""
    import ctypes
    from pyscf_amd import gto
    from pyscf_amd.grad import grad_nuc
    from pyscf_amd.gto import moleintor
    atoms = [('O', (0.03, -0.02, 0.01)), ('H', (0.1, -0.757, 0.587)), ('H', (-0.2, 0.8, 0.5))]
    mol = gto.M(atom=atoms, basis='cc-pvdz')
    g = grad_nuc(mol)
    r = mol.atom_coords()
    h = 1e-5
    for ia in range(3):
        for x in range(3):
            es = []
            for d in (h, -h):
                rr = r.copy()
                rr[ia, x] += d
                m2 = gto.M(atom=[(s, tuple(rr[i])) for i, (s, _) in enumerate(atoms)], basis='sto-3g', unit='Bohr')
                es.append(m2.energy_nuc())
            assert abs((es[0] - es[1]) / (2 * h) - g[ia, x]) < 1e-7
    sl = mol.aoslice_by_atom()
    assert sl[:, 2:].tolist() == [[0, 14], [14, 19], [19, 24]] and sl[-1, 1] == mol.nbas
    # struct layout: base block + 4 pointers + 3 ints, as declared in include/pyscf_amd.h
    assert ctypes.sizeof(moleintor._GradArgs) == ctypes.sizeof(moleintor._Args) + 4 * 8 + 16
    hdr = open(os.path.join(ROOT, 'include', 'pyscf_amd.h')).read()
    body = hdr[hdr.index('typedef struct PAMD_int3c2e_grad_args'):hdr.index('} PAMD_int3c2e_grad_args;')]
    names = [f[0] for f in moleintor._GradArgs._fields_]
    import re
    pos = [re.search(r'[ \*]%s;' % n, body).start() for n in names]
    assert pos == sorted(pos), 'field order of _GradArgs differs from the header'


def test_aux_basis_selection_and_even_tempered_generation():
    """pyscf/df/test/test_addons.py:28-41,73-101: aug_etb shell counts (USE_VERSION_26_AUXBASIS branch), make_auxbasis per
    atom label incl. ghost atoms and mixed AO bases, even-tempered fallback when no JK-fit set is tabulated."""
    from pyscf_amd import gto, df
    from pyscf_amd.df import addons
    mol = gto.M(atom='O 0 0 0; 1 0 -0.757 0.587; 1 0 0.757 0.587', basis='cc-pvdz')
    etb = addons.aug_etb(mol)
    assert len(etb['O']) == 36 and len(etb['H']) == 12
    assert addons.expand_etbs([(1, 3, 1.5, 2)]) == [[1, [6.0, 1]], [1, [3.0, 1]], [1, [1.5, 1]]]
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587; GHOST-H 0 0 0.587', basis='cc-pvdz')
    ab = addons.make_auxbasis(mol)
    assert ab == {'O': 'cc-pvdz-jkfit', 'H': 'cc-pvdz-jkfit', 'GHOST-H': 'cc-pvdz-jkfit'}
    assert df.make_auxmol(mol).nao_nr() == 116 + 23 and mol.nelectron == 10
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis={'O': 'cc-pvdz', 'H': 'cc-pvtz'})
    assert addons.make_auxbasis(mol) == {'O': 'cc-pvdz-jkfit', 'H': 'cc-pvtz-jkfit'}
    mol = gto.M(atom='O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587', basis='ano')      # no tabulated fitting set
    ab = addons.make_auxbasis(mol)
    assert all(isinstance(v, list) for v in ab.values())
    aux = df.make_auxmol(mol)
    assert aux.nao_nr() > mol.nao_nr() and int(aux._bas[:, 1].max()) <= 4


def test_xc_description_parser():
    """Host-side functional parser (subset of pyscf/dft/libxc.py:parse_xc): component weights in the kernel's order
    {Slater, VWN5, VWN_RPA, B88, LYP, PBE_X, PBE_C}, hybrid and range-separation coefficients."""
    from pyscf_amd.dft import libxc
    hyb, fac = libxc.parse_xc('b3lyp')                      # id 402: VWN_RPA flavour (libxc.py:175)
    assert hyb == 0.2 and np.allclose(fac[:7], [0.08, 0, 0.19, 0.72, 0.81, 0, 0]) and len(fac) == libxc.NFAC == 10
    assert np.allclose(libxc.parse_xc('b3lyp5')[1][:7], [0.08, 0.19, 0, 0.72, 0.81, 0, 0])
    assert np.allclose(libxc.parse_xc('lda,vwn')[1], libxc.parse_xc('SLATER , VWN5')[1])
    assert np.allclose(libxc.parse_xc('LDA,VWN')[1], [1, 1, 0, 0, 0, 0, 0, 0, 0, 0]) and libxc.xc_type('lda,vwn') == 'LDA'
    assert libxc.xc_type('b88,lyp') == 'GGA' and libxc.xc_type('hf') == 'HF'
    assert np.allclose(libxc.parse_xc('pbe0')[1][:7], [0, 0, 0, 0, 0, 0.75, 1]) and libxc.parse_xc('pbe0')[0] == 0.25
    assert np.allclose(libxc.parse_xc('0.5*b88+0.5*lda,lyp')[1][:7], [0.5, 0, 0, 0.5, 1, 0, 0])
    assert libxc.parse_xc_rsh('lda+0.5*SR_HF(0.3)')[:3] == (0.5, 0.0, 0.3)
    assert libxc.rsh_coeff('lda+0.5*SR_HF(0.3)') == (0.3, 0.0, 0.5)
    assert libxc.rsh_coeff('b3lyp') == (0.0, 0.0, 0.0) and libxc.is_hybrid_xc('b3lyp') and not libxc.is_hybrid_xc('pbe')
    assert libxc.parse_xc_rsh('0.2*LR_HF(0.4)+b88,lyp')[:3] == (0.0, 0.2, 0.4) and libxc.is_hybrid_xc('0.2*LR_HF(0.4)+b88,lyp')
    # CAM-B3LYP (hyb_gga_xc_cam_b3lyp): 0.35 B88 + 0.46 ITYH(omega 0.33) + 0.19 VWN5 + 0.81 LYP, exact exchange 0.19 SR / 0.65 LR;
    # the spelled-out form of pyscf/dft/test/test_h2o.py:574 parses to the same thing with its own omega
    hyb, alpha, omega, fac = libxc.parse_xc_rsh('cam-b3lyp')
    assert (hyb, alpha, omega) == (0.19, 0.65, 0.33) and np.allclose(fac, [0, 0.19, 0, 0.35, 0.81, 0, 0, 0.46, 0, 0.33])
    assert np.allclose(libxc.rsh_coeff('camb3lyp'), (0.33, 0.65, -0.46)) and libxc.xc_type('camb3lyp') == 'GGA'
    hyb, alpha, omega, fac = libxc.parse_xc_rsh('RSH(.15,0.65,-0.46) + 0.46*ITYH + .35*B88 + VWN5*0.19, LYP*0.81')
    assert abs(hyb - 0.19) < 1e-15 and (alpha, omega) == (0.65, 0.15)
    assert np.allclose(fac, [0, 0.19, 0, 0.35, 0.81, 0, 0, 0.46, 0, 0.15])
    # omega-B97: the whole functional is one device component; no short-range, full long-range exact exchange at omega 0.4
    hyb, alpha, omega, fac = libxc.parse_xc_rsh('wb97')
    assert (hyb, alpha, omega) == (0.0, 1.0, 0.4) and np.allclose(fac, [0] * 8 + [1, 0.4]) and libxc.is_hybrid_xc('wb97')
    with pytest.raises(ValueError):
        libxc.parse_xc('0.5*ITYH,lyp')                      # attenuated exchange without a range-separation parameter
    with pytest.raises(NotImplementedError):
        libxc.parse_xc('scan')


def test_host_eigensolvers_of_the_response_drivers():
    """The host-side iterative solvers of tdscf (block Davidson, symmetric and product-form) and soscf (augmented-Hessian
    step) on model matrices with known answers."""
    import numpy as np
    from pyscf_amd.tdscf import _davidson
    from pyscf_amd.soscf import _augmented_hessian_step
    rng = np.random.default_rng(0)
    n = 300
    d = np.sort(rng.random(n) * 10 + 1)
    off = rng.standard_normal((n, n)) * 0.05
    a = np.diag(d) + off + off.T
    th, x, conv = _davidson(lambda v: v.dot(a.T), np.diag(a).copy(), 4, 1e-8, 100, 40, True)
    assert conv.all() and np.abs(th - np.linalg.eigvalsh(a)[:4]).max() < 1e-10
    assert np.abs(x.dot(a) - th[:, None] * x).max() < 1e-6
    b = rng.standard_normal((n, n)) * 0.03
    c = rng.standard_normal((n, n)) * 0.03
    m = (np.diag(d) + b + b.T).dot(np.diag(d) + c + c.T)            # (A - B)(A + B): real positive spectrum
    th, x, conv = _davidson(lambda v: v.dot(m.T), np.diag(m).copy(), 4, 1e-8, 200, 40, False)
    assert conv.all() and np.abs(th - np.sort(np.linalg.eigvals(m).real)[:4]).max() < 1e-8
    n = 60
    h = np.diag(rng.random(n) * 2 + 0.5)
    o = rng.standard_normal((n, n)) * 0.02
    h = h + o + o.T
    g = rng.standard_normal(n) * 0.05
    step, w, nhop = _augmented_hessian_step(g, lambda v: h.dot(v), np.diag(h).copy(), 1e-9, 50)
    aug = np.block([[np.zeros((1, 1)), g[None]], [g[:, None], h]])
    wa, va = np.linalg.eigh(aug)
    assert abs(w - wa[0]) < 1e-12 and np.abs(step - va[1:, 0] / va[0, 0]).max() < 1e-8
    assert g.dot(step) < 0                                           # a descent direction


def test_hdf5_j3c_roundtrip_through_libhdf5(tmp_path):
    """lib/hdf5.py (ctypes on libhdf5, no h5py): the 'j3c' dataset written in row blocks reads back exactly, also by
    arbitrary row ranges - the access pattern of DF.loop and of the aux-row shards."""
    from pyscf_amd.lib import hdf5
    if not hdf5.available():
        pytest.skip('libhdf5 not found')
    rng = np.random.default_rng(0)
    a = rng.standard_normal((53, 301))
    path = str(tmp_path / 'cderi.h5')
    with hdf5.File(path, 'w') as f:
        d = f.create_dataset('j3c', a.shape)
        for r0 in range(0, 53, 16):
            d.write_rows(r0, a[r0:r0 + 16])
    assert hdf5.is_hdf5(path) and not hdf5.is_hdf5(__file__)
    with hdf5.File(path) as f:
        d = f['j3c']
        assert d.shape == a.shape
        assert np.array_equal(d.read_rows(0, 53), a) and np.array_equal(d.read_rows(17, 40), a[17:40])
    with hdf5.File(path, 'r+') as f:
        f['j3c'].write_rows(10, a[:5] * 2)
    with hdf5.File(path) as f:
        assert np.array_equal(f['j3c'].read_rows(10, 15), a[:5] * 2)


def test_hdf5_interoperates_with_h5py(tmp_path):
    """Files written by lib/hdf5.py open in h5py (what stock PySCF uses for `_cderi`) and the other way round.  h5py is not
    importable by this interpreter; the image's conda python3.9 has it - skipped where that is missing."""
    import subprocess
    from pyscf_amd.lib import hdf5
    py = '/opt/conda/bin/python3.9'
    if not hdf5.available() or not os.path.exists(py) or subprocess.run([py, '-c', 'import h5py']).returncode != 0:
        pytest.skip('no interpreter with h5py')
    a = np.arange(84.0).reshape(12, 7) / 7
    ours, theirs = str(tmp_path / 'ours.h5'), str(tmp_path / 'theirs.h5')
    with hdf5.File(ours, 'w') as f:
        f.create_dataset('j3c', a.shape).write_rows(0, a)
    code = '\n'.join(["import h5py, numpy as np",
                      "f = h5py.File(%r, 'r'); d = f['j3c']" % ours,
                      "assert d.shape == (12, 7) and d.dtype == np.float64 and abs(d[5, 3] - (5 * 7 + 3) / 7) < 1e-15",
                      "g = h5py.File(%r, 'w'); g['j3c'] = d[()] * 2; g.close()" % theirs])
    assert subprocess.run([py, '-c', code]).returncode == 0
    with hdf5.File(theirs) as f:
        assert np.array_equal(f['j3c'].read_rows(0, 12), a * 2)
