"""Worker of tests/test_gpu_native_r04.py: BASELINE config 3, (H2O)_32 cc-pVTZ (nao 1856, naux 4448, 61 GB tensor), through the
host-array C handle (PAMD_df_create_ex / PAMD_df_get_jk) against the oracle-only golden tests/golden/h2o32_ccpvtz_oracle.json
(tools/gen_golden_fullsize.py: the oracle's own McMurchie-Davidson tensor, J/K of a seeded density).  numpy + ctypes, no torch.
Three layouts: one shard in HBM; two shards (device list [0, 0]: sharding + gather + sum); a device-memory cap that leaves ~40 %
of the rows in page-locked host memory (streamed: the out-of-core path at full size, its PCIe-bound rate is printed)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import golden_util
    from pyscf_amd import gto, lib
    from pyscf_amd.data import clusters
    from pyscf_amd.df import native
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'h2o32_ccpvtz_oracle.json')))
    mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz')
    nao, nocc = mol.nao, mol.nelectron // 2
    assert (g['nao'], g['nocc']) == (nao, nocc)
    c = golden_util.synthetic_orbitals(nao, nocc)
    dm = 2 * c.dot(c.T)
    occ = np.zeros(nao)
    mo = np.zeros((nao, nao))
    mo[:, :nocc] = c
    occ[:nocc] = 2
    ri, ci = golden_util.sample_positions(nao, 4096)
    vj_s, vk_s = np.array(g['vj_sample']), np.array(g['vk_sample'])

    def check(vj, vk, tag):
        assert np.abs(vj[ri, ci] - vj_s).max() < 1e-9 * g['vj_absmax'], tag
        assert np.abs(vk[ri, ci] - vk_s).max() < 1e-9 * g['vk_absmax'], tag
        assert abs(np.linalg.norm(vj) - g['vj_norm']) < 1e-9 * g['vj_norm'], tag
        assert abs(np.linalg.norm(vk) - g['vk_norm']) < 1e-9 * g['vk_norm'], tag
        assert abs(lib.fp(vj) - g['vj_fp']) < 1e-8 * g['vj_norm'] and abs(lib.fp(vk) - g['vk_fp']) < 1e-8 * g['vk_norm'], tag

    tagged = lib.tag_array(dm, mo_coeff=mo, mo_occ=occ)
    npair = nao * (nao + 1) // 2
    for tag, kw in (('one shard in HBM', dict()), ('one shard in HBM, SQUARE rows only (r06)', dict(square=True)),
                    ('two parts on device 0', dict(devices=[0, 0])),
                    ('two parts on device 0, SQUARE rows only (r06)', dict(devices=[0, 0], square=True)),
                    ('a 69 GB device-memory cap (about 40 %% of the rows in host memory)', dict(max_device_bytes=int(69e9)))):
        t0 = time.perf_counter()
        want_square = kw.pop('square', False)
        # (the handle keeps packed rows + a full image while 3x fits the budget; PAMD_DF_PREFER_IMAGE=0 = the square layout, the
        # one a tensor whose 3x does not fit - taxol on one GPU - gets by itself)
        os.environ['PAMD_DF_PREFER_IMAGE'] = '0' if want_square else '1'
        obj = native.NativeDF(mol, **kw).build()
        tb = time.perf_counter() - t0
        assert obj.get_naoaux() == g['naux'] == 4448
        lay = obj.layout()
        assert lay['tensor_layout'] == ('square' if want_square else 'packed'), lay
        vj, vk = obj.get_jk(tagged, hermi=1)                       # first call: schedule timing included
        check(vj, vk, tag)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            vj, vk = obj.get_jk(tagged, hermi=1)
            ts.append(time.perf_counter() - t0)
        check(vj, vk, tag + ' (repeat)')
        print('%s: build %.1f s, get_jk %.1f ms (numpy in / out), layout %s' % (tag, tb, min(ts) * 1e3, lay), flush=True)
        if lay['rows_host']:
            print('    streamed %.1f GB per build -> %.1f GB/s over PCIe incl. compute' % (
                lay['rows_host'] * npair * 8e-9, lay['rows_host'] * npair * 8e-9 / min(ts)), flush=True)
        obj.reset()
    assert 'torch' not in sys.modules
    print('NATIVE_CFG3_OK', flush=True)


if __name__ == '__main__':
    main()
