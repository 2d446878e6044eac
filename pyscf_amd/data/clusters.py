"""Deterministic synthetic water clusters: the reference ships no (H2O)_n
geometry, so bench and parity use this generator.  Monomer = the reference's test water
(O 0 0 0; H 0 -+0.757 0.587 Angstrom); molecules on a simple-cubic lattice with 3.1 Angstrom O-O
spacing, x fastest, each rotated by a fixed-seed random rotation."""
import numpy as np

MONOMER = np.array([[0.0, 0.0, 0.0], [0.0, -0.757, 0.587], [0.0, 0.757, 0.587]])
SYMBOLS = ('O', 'H', 'H')
SPACING = 3.1


def _rotation(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def water_cluster(n):
    """-> list of (symbol, (x, y, z)) in Angstrom for (H2O)_n."""
    rng = np.random.default_rng(20240601)
    side = 1
    while side ** 3 < n:
        side += 1
    atoms = []
    for m in range(n):
        ix, iy, iz = m % side, (m // side) % side, m // (side * side)
        rot = _rotation(rng)
        xyz = MONOMER.dot(rot.T) + SPACING * np.array([ix, iy, iz])
        for s, r in zip(SYMBOLS, xyz):
            atoms.append((s, tuple(float(v) for v in r)))
    return atoms


# benzene geometry of the reference benchmark (examples/2-benchmark/bz.py:7-19), Angstrom
BENZENE = '''
C   1.217739890298750 -0.703062453466927  0.000000000000000
H   2.172991468538160 -1.254577209307266  0.000000000000000
C   1.217739890298750  0.703062453466927  0.000000000000000
H   2.172991468538160  1.254577209307266  0.000000000000000
C   0.000000000000000  1.406124906933854  0.000000000000000
H   0.000000000000000  2.509154418614532  0.000000000000000
C  -1.217739890298750  0.703062453466927  0.000000000000000
H  -2.172991468538160  1.254577209307266  0.000000000000000
C  -1.217739890298750 -0.703062453466927  0.000000000000000
H  -2.172991468538160 -1.254577209307266  0.000000000000000
C   0.000000000000000 -1.406124906933854  0.000000000000000
H   0.000000000000000 -2.509154418614532  0.000000000000000
'''


def taxol():
    """-> list of (symbol, (x, y, z)) in Angstrom for paclitaxel C47H51NO14 (BASELINE config 4): the molecule's connectivity
    embedded in 3-D by tools/make_taxol_xyz.py (data/taxol.xyz).  Right atoms, bonds and rings - not an optimised geometry."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'taxol.xyz')
    with open(path) as f:
        lines = f.read().splitlines()
    n = int(lines[0])
    atoms = []
    for ln in lines[2:2 + n]:
        s, x, y, z = ln.split()
        atoms.append((s, (float(x), float(y), float(z))))
    return atoms
