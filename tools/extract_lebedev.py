"""Build pyscf_amd/dft/lebedev.npz: Lebedev-Laikov angular quadrature points/weights for the
orders the DFT grid levels use (6 ... 590 points).  The numbers are the published Lebedev-Laikov
tables (Dokl. Math. 59, 477 (1999)); they are read here from the reference's data module
pyscf/dft/LebedevGrid.py (pure numpy, loaded by file path) in the authoring container only.
    python tools/extract_lebedev.py
"""
import importlib.util, os
import numpy as np
spec = importlib.util.spec_from_file_location('LebedevGrid', '/root/reference/pyscf/dft/LebedevGrid.py')
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
out = {}
for n in m.LEBEDEV_NGRID:
    if 6 <= n <= 590:
        g = np.asarray(m.MakeAngularGrid(int(n)))
        assert g.shape == (n, 4) and abs(g[:, 3].sum() - 1) < 1e-12
        out['n%d' % n] = g
out['order'] = np.array([[k, v] for k, v in m.LEBEDEV_ORDER.items() if v <= 590])
dst = os.path.join(os.path.dirname(__file__), '..', 'pyscf_amd', 'dft', 'lebedev.npz')
np.savez_compressed(dst, **out)
print(sorted(out), os.path.getsize(dst))
