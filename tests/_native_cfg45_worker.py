"""Worker of tests/test_gpu_native_r04.py: BASELINE configs 4 and 5 through the host-array C handle on ONE GPU, numpy only.

  config4   taxol def2-TZVP (111 GB tensor) as the configuration is meant to run - the aux index sharded EIGHT ways - with all eight
            parts on the one test GPU (PAMD_df_create_multi, devices = [0] * 8): every part builds its 700 rows, eight host threads
            contract them concurrently, the partial [J~ | K] are gathered and summed - against the oracle-only golden
            tests/golden/taxol_def2tzvp_oracle.json (J / K of the seeded rank-32 density: fp, norms, 4096 samples, 1e-9).
  config5   (H2O)_128 cc-pVDZ, nao 3072, naux 14 848: the WHOLE 560 GB tensor on one GPU - what fits in HBM resident, the rest
            (about 330 GB) in page-locked host memory, streamed under the kernels (the out-of-core path at full size) - J / K of the
            seeded local density against the SUM of the two oracle-only goldens that cover all aux rows
            (h2o128_ccpvdz_rows0-7424 / rows7424-14848_local_oracle.json: fp and samples are linear in the rows).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _golden(name):
    with open(os.path.join(ROOT, 'tests', 'golden', name)) as f:
        return json.load(f)


def config4():
    from oracle import golden_util
    from pyscf_amd import gto, lib
    from pyscf_amd.data import clusters
    from pyscf_amd.df import native
    g = _golden('taxol_def2tzvp_oracle.json')
    mol = gto.M(atom=clusters.taxol(), basis='def2-tzvp')
    nao = mol.nao
    nsyn = int(g['syn_density'].split(',')[-1].strip(' )'))
    c = golden_util.synthetic_orbitals(nao, nsyn) * np.sqrt(2.0)
    dm = c.dot(c.T)
    occ = np.zeros(nao)
    occ[:nsyn] = 1.0
    cfull = np.zeros((nao, nao))
    cfull[:, :nsyn] = c
    t0 = time.perf_counter()
    obj = native.NativeDF(mol, devices=[0] * 8).build()
    tb = time.perf_counter() - t0
    lay = obj.layout()
    assert obj.get_naoaux() == g['naux'] == 5598 and lay['parts'] == 8 and sum(lay['part_rows']) == 5598, lay
    tagged = lib.tag_array(dm, mo_coeff=cfull, mo_occ=occ)
    vj, vk = obj.get_jk(tagged, hermi=1)
    t0 = time.perf_counter()
    vj, vk = obj.get_jk(tagged, hermi=1)
    tj = time.perf_counter() - t0
    ri, ci = golden_util.sample_positions(nao, len(g['syn_vk_sample']))
    for name, m in (('vj', vj), ('vk', vk)):
        scale = g['syn_%s_absmax' % name]
        assert abs(np.linalg.norm(m) - g['syn_%s_norm' % name]) < 1e-9 * g['syn_%s_norm' % name], name
        assert abs(golden_util.fp(m) - g['syn_%s_fp' % name]) < 1e-8 * g['syn_%s_norm' % name], name
        assert np.abs(m[ri, ci] - np.array(g['syn_%s_sample' % name])).max() < 1e-9 * scale, name
    print('config 4, eight parts on one GPU: build %.1f s, get_jk %.1f ms (rank-32 density), layout %s' % (tb, tj * 1e3, lay), flush=True)
    obj.reset()


def config5():
    from oracle import golden_util
    from pyscf_amd import gto, lib
    from pyscf_amd.data import clusters
    from pyscf_amd.df import native
    ga = _golden('h2o128_ccpvdz_rows0-7424_local_oracle.json')
    gb = _golden('h2o128_ccpvdz_rows7424-14848_local_oracle.json')
    assert ga['aux_rows'] == [0, 7424] and gb['aux_rows'] == [7424, 14848] and ga['support_ao_range'] == gb['support_ao_range']
    mol = gto.M(atom=clusters.water_cluster(128), basis='cc-pvdz')
    nao = mol.nao
    (a0, a1), nsyn = ga['support_ao_range'], ga['nsyn']
    ns = a1 - a0
    c = np.zeros((nao, nsyn))
    c[a0:a1] = golden_util.synthetic_orbitals(ns, nsyn) * np.sqrt(2.0)
    dm = c.dot(c.T)
    occ = np.zeros(nao)
    occ[:nsyn] = 1.0
    cfull = np.zeros((nao, nao))
    cfull[:, :nsyn] = c
    t0 = time.perf_counter()
    obj = native.NativeDF(mol).build()
    tb = time.perf_counter() - t0
    lay = obj.layout()
    npair = nao * (nao + 1) // 2
    assert obj.get_naoaux() == 14848 and lay['rows_resident'] + lay['rows_host'] == 14848 and lay['rows_host'] > 0, lay
    print('config 5 on one GPU: build %.1f s, %d rows (%.0f GB) in HBM, %d rows (%.0f GB) in page-locked host memory' % (
        tb, lay['rows_resident'], lay['rows_resident'] * npair * 8e-9, lay['rows_host'], lay['rows_host'] * npair * 8e-9), flush=True)
    tagged = lib.tag_array(dm, mo_coeff=cfull, mo_occ=occ)
    t0 = time.perf_counter()
    vj, vk = obj.get_jk(tagged, hermi=1)
    tj = time.perf_counter() - t0
    ri, ci = golden_util.sample_positions(nao, len(ga['vk_sample']))
    vk_s = np.array(ga['vk_sample']) + np.array(gb['vk_sample'])
    scale_k = max(ga['vk_absmax'], gb['vk_absmax'])
    assert np.abs(vk[ri, ci] - vk_s).max() < 1e-9 * scale_k, np.abs(vk[ri, ci] - vk_s).max() / scale_k
    assert abs(golden_util.fp(vk) - (ga['vk_fp'] + gb['vk_fp'])) < 1e-8 * (ga['vk_norm'] + gb['vk_norm'])
    rect = vj[:, a0:a1]
    rj = np.array(ga['vj_rect_sample']) + np.array(gb['vj_rect_sample'])
    scale_j = max(ga['vj_rect_absmax'], gb['vj_rect_absmax'])
    assert np.abs(rect[ri, ci % ns] - rj).max() < 1e-9 * scale_j, np.abs(rect[ri, ci % ns] - rj).max() / scale_j
    assert abs(golden_util.fp(rect) - (ga['vj_rect_fp'] + gb['vj_rect_fp'])) < 1e-8 * (ga['vj_rect_norm'] + gb['vj_rect_norm'])
    print('    get_jk (rank-32 local density) %.2f s: %.0f GB streamed -> %.1f GB/s incl. compute' % (
        tj, lay['rows_host'] * npair * 8e-9, lay['rows_host'] * npair * 8e-9 / tj), flush=True)
    config5_energy(mol, obj)
    obj.reset()


def config5_energy(mol, obj):
    """r05: the ENERGY of config 5 against an oracle-only golden, on the handle that already holds the whole
    560 GB tensor.  tests/golden/h2o128_ccpvdz_energy_oracle.json is the oracle's DF-RHF energy functional at the density of
    tests/golden/h2o128_ccpvdz_rhf_orbitals.npz (tools/gen_golden_energy_sweep.py: one sweep of the oracle's McMurchie-Davidson
    integrals, the oracle's own Roothaan residual on the AO rows of two molecules).  Checked here:
      (a) J rows and (K C) rows of the product's J/K build at those orbitals, element by element (1e-9);
      (b) the product's energy functional at those orbitals (1e-8 Eh);
      (c) the product's OWN SCF from its minao guess through the out-of-core handle converges to the same energy (1e-8 Eh) -
          the north-star gate, pyscf/df/test/test_df_jk.py:57-59 at config-5 size."""
    from oracle import golden_util
    from pyscf_amd import lib, scf
    gpath = os.path.join(ROOT, 'tests', 'golden', 'h2o128_ccpvdz_energy_oracle.json')
    opath = os.path.join(ROOT, 'tests', 'golden', 'h2o128_ccpvdz_rhf_orbitals.npz')
    if not (os.path.exists(gpath) and os.path.exists(opath)):
        print('    config 5 energy golden not generated (tools/gen_golden_energy_sweep.py): energy checks skipped', flush=True)
        print('NATIVE_CONFIG5_ENERGY_SKIPPED', flush=True)
        return
    ge = _golden('h2o128_ccpvdz_energy_oracle.json')
    orbo = np.ascontiguousarray(np.load(opath)['orbo'])
    nao, nocc = orbo.shape
    assert (nao, nocc) == (ge['nao'], ge['nocc']) == (3072, 640)
    dm = lib.tag_array(orbo.dot(orbo.T), mo_coeff=orbo / np.sqrt(2.0), mo_occ=np.full(nocc, 2.0))
    t0 = time.perf_counter()
    vj, vk = obj.get_jk(dm, hermi=1)
    tj = time.perf_counter() - t0
    rr, cc = golden_util.sample_positions(1 << 20, len(ge['row_samples'][0]['vj_rows_sample']), seed=ge['sample_seed'])
    for rs in ge['row_samples']:
        p0, p1 = rs['ao_rows']
        npp = p1 - p0
        jrows = np.ascontiguousarray(vj[p0:p1])
        kc = vk[p0:p1].dot(orbo)
        for name, m, cols in (('vj_rows', jrows, nao), ('vkc_rows', kc, nocc)):
            scale = rs[name + '_absmax']
            assert abs(np.linalg.norm(m) - rs[name + '_norm']) < 1e-9 * rs[name + '_norm'], (name, rs['molecule'])
            assert abs(golden_util.fp(m) - rs[name + '_fp']) < 1e-8 * rs[name + '_norm'], (name, rs['molecule'])
            err = np.abs(m[rr % npp, cc % cols] - np.array(rs[name + '_sample'])).max()
            assert err < 1e-9 * scale, (name, rs['molecule'], err / scale)
    mf = scf.RHF(mol).density_fit(with_df=obj)
    h1e = mf.get_hcore()
    e_fun = float(np.einsum('ij,ji', h1e, dm) + .5 * np.einsum('ij,ji', vj - .5 * vk, dm) + mol.energy_nuc())
    print('    J/K at the golden orbitals %.2f s (nocc 640); E[D] product %.10f, oracle %.10f, diff %.2e; oracle Roothaan residual on '
          'its sampled rows %.2e' % (tj, e_fun, ge['e_tot'], e_fun - ge['e_tot'], ge.get('roothaan_residual_norm_sampled_rows', -1)), flush=True)
    assert abs(e_fun - ge['e_tot']) < 1e-8, (e_fun, ge['e_tot'])
    mf.conv_tol = 1e-10
    t0 = time.perf_counter()
    e = mf.kernel()
    print('    converged SCF from the minao guess through the out-of-core handle: %d cycles, %.1f s, E %.10f, oracle %.10f, diff %.2e'
          % (mf.cycles, time.perf_counter() - t0, e, ge['e_tot'], e - ge['e_tot']), flush=True)
    assert mf.converged and abs(e - ge['e_tot']) < 1e-8, (e, ge['e_tot'])
    print('NATIVE_CONFIG5_ENERGY_OK', flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'config4'
    {'config4': config4, 'config5': config5}[which]()
    assert which == 'config5' or 'torch' not in sys.modules      # (config 5 also drives the SCF loop, which keeps F / S on the device)
    print('NATIVE_%s_OK' % which.upper(), flush=True)
