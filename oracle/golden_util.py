"""ORACLE - TEST INFRASTRUCTURE ONLY.  Helpers shared by the golden-vector generator
(tools/gen_golden_fullsize.py, run on the CPU) and the GPU parity tests that read tests/golden/*.json."""
import numpy as np


def synthetic_orbitals(nao, nocc, seed=7):
    """Deterministic, LAPACK-free 'occupied orbital' block for full-size J/K parity: uniform random entries in
    [-0.5, 0.5) / sqrt(nao) from numpy's legacy RandomState (bit-stable across numpy versions).  The density 2 C C^T is not
    idempotent - irrelevant for a contraction check - but has the rank (nocc) and the scale of a real one."""
    rng = np.random.RandomState(seed)
    return (rng.random_sample((nao, nocc)) - 0.5) / np.sqrt(nao)


def sample_positions(n, count, seed=11):
    """`count` (row, col) positions of an n x n matrix, seeded."""
    rng = np.random.RandomState(seed)
    return rng.randint(0, n, size=count), rng.randint(0, n, size=count)


def fp(a):
    a = np.asarray(a)
    return float(np.dot(np.cos(np.arange(a.size)), a.ravel()))
