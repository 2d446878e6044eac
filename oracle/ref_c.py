"""TEST INFRASTRUCTURE ONLY (oracle/): ctypes driver of oracle/_ref/libref_dfjk.so - the REFERENCE'S OWN compiled C for the
DF J/K hot path (built by `make -C oracle ref` from the sources where they lie under /root/reference, never copied).

`get_jk` below performs the calls of ``pyscf/df/df_jk.py:329-381`` one for one:

  * ``dmtril = lib.pack_tril(dm + dm.T)``; diagonal halved            (:329-332)  -> NPdpack_tril_2d   (np_helper/pack_tril.c:245)
  * per block of `blksize` aux rows (``dfobj.loop(blksize)``, :362)
      ``vj += dmtril.dot(eri1.T).dot(eri1)``                             (:367, "uses numpy.matmul")
      ``fdrv(ftrans, fmmm, buf1, eri1, orbo, naux, nao, (0,nocc,0,nao), null, 0)``   (:373-379)
            fdrv = AO2MOnr_e2_drv (ao2mo/nr_ao2mo.c:1240-1266, OpenMP over aux rows), ftrans = AO2MOtranse2_nr_s2 (:1026-1031,
            NPdunpack_tril per row), fmmm = AO2MOmmm_bra_nr_s2 (:399-419, dsymm per row)
      ``vk += lib.dot(buf1.T, buf1)``                                    (:380)      -> NPdgemm (np_helper/npdot.c:32, OpenMP over k)
  * ``vj = lib.unpack_tril(vj, 1)``                                      (:410)      -> NPdunpack_tril_2d (pack_tril.c:214)

Threading as in a stock PySCF build (OpenMP-threaded C, BLAS serial inside the parallel regions): the OpenBLAS that ships in
scipy's wheel is pthread-based and compiled for at most 64 concurrent callers, so it is pinned to ONE thread per call and the
reference's own `#pragma omp` loops run on min(host cores, 64) threads (more callers crashed it on the 256-core GPU host);
`numpy.matmul` of the J line keeps numpy's own BLAS threading.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_ref', 'libref_dfjk.so')
_lib = None


def available():
    return os.path.exists(SO)


def build():
    """(re)build oracle/_ref when the reference tree is present (this container); elsewhere use the prebuilt file."""
    if os.path.isdir('/root/reference/pyscf/lib'):
        import subprocess
        subprocess.check_call(['make', '-s', '-C', HERE, 'ref'])
    return available()


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError('oracle/_ref/libref_dfjk.so is missing: run `make -C oracle ref` where /root/reference exists')
        _lib = ctypes.CDLL(SO)
        _lib.scipy_openblas_set_num_threads.argtypes = [ctypes.c_int]
        _lib.omp_get_max_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


MAX_BLAS_CALLERS = 64      # scipy's OpenBLAS is compiled for 64 threads: more CONCURRENT callers corrupt its buffer table
                           # ("precompiled NUM_THREADS exceeded ... double free or corruption" on a 256-core host, r03)


def set_threads(n):
    """OpenMP threads of the reference's C (lib.num_threads(), pyscf/lib/misc.py:195-224), capped at MAX_BLAS_CALLERS; BLAS
    stays at one thread per call.  Returns the thread count really used (bench.py reports it as `cores`)."""
    l = lib()
    l.scipy_openblas_set_num_threads(1)
    n = min(int(n or os.cpu_count() or 1), MAX_BLAS_CALLERS)
    l.omp_set_num_threads(ctypes.c_int(n))
    return l.omp_get_max_threads()


def pack_tril(mats):
    l = lib()
    mats = np.ascontiguousarray(mats, dtype=np.float64)
    count, nd = mats.shape[0], mats.shape[-1]
    out = np.empty((count, nd * (nd + 1) // 2))
    l.NPdpack_tril_2d(ctypes.c_int(count), ctypes.c_int(nd), _p(out), _p(mats))
    return out


def unpack_tril(tril, filltriu=1):
    l = lib()
    tril = np.ascontiguousarray(tril, dtype=np.float64)
    count = tril.shape[0]
    nd = int(round((np.sqrt(8 * tril.shape[1] + 1) - 1) / 2))
    out = np.empty((count, nd, nd))
    l.NPdunpack_tril_2d(ctypes.c_int(count), ctypes.c_int(nd), _p(tril), _p(out), ctypes.c_int(filltriu))
    return out


def lib_dot_tn(buf1):
    """lib.dot(buf1.T, buf1): ddot sees a = buf1.T (F-contiguous -> trans_a 'T', a = buf1) and b = buf1 (C-contiguous, 'N')
    and calls NPdgemm(trans_b, trans_a, n, m, k, ldb, lda, ldc, ...)  (pyscf/lib/numpy_helper.py:825-858,980-1004)."""
    l = lib()
    k, n = buf1.shape
    c = np.empty((n, n))
    l.NPdgemm(ctypes.c_char(b'N'), ctypes.c_char(b'T'), ctypes.c_int(n), ctypes.c_int(n), ctypes.c_int(k),
              ctypes.c_int(n), ctypes.c_int(n), ctypes.c_int(n), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0),
              _p(buf1), _p(buf1), _p(c), ctypes.c_double(1.0), ctypes.c_double(0.0))
    return c


def get_jk(cderi, dm, mo_coeff, mo_occ, blockdim=240, nthreads=None):
    """J and K of ONE density (nset = 1, hermi = 1, MO branch) over the rows of `cderi` (naux_rows, nao_pair), exactly the call
    sequence of pyscf/df/df_jk.py:329-381.  Returns (vj, vk, flops, phases)."""
    l = lib()
    nth = set_threads(nthreads)
    nao = dm.shape[-1]
    naux = cderi.shape[0]
    dms = np.ascontiguousarray(dm.reshape(1, nao, nao), dtype=np.float64)
    ph = {'pack': 0.0, 'vj': 0.0, 'e2_drv': 0.0, 'dot': 0.0, 'unpack': 0.0}
    t = time.perf_counter()
    dmtril = pack_tril(dms + dms.transpose(0, 2, 1))
    idx = np.arange(nao)
    dmtril[:, idx * (idx + 1) // 2 + idx] *= .5
    ph['pack'] = time.perf_counter() - t
    occ = np.asarray(mo_occ)
    orbo = np.asarray(np.einsum('pi,i->pi', np.asarray(mo_coeff)[:, occ > 0], np.sqrt(occ[occ > 0])), order='F')
    nocc = orbo.shape[1]
    blksize = max(4, int(blockdim))
    buf = np.empty((blksize * nao, nao))
    vj = np.zeros((1, nao * (nao + 1) // 2))
    vk = np.zeros((nao, nao))
    fdrv = l.AO2MOnr_e2_drv
    ftrans = l.AO2MOtranse2_nr_s2
    fmmm = l.AO2MOmmm_bra_nr_s2
    null = ctypes.c_void_p(0)
    for b0 in range(0, naux, blksize):
        eri1 = cderi[b0:b0 + blksize]
        if not eri1.flags.c_contiguous:
            eri1 = np.ascontiguousarray(eri1)
        nb = eri1.shape[0]
        t = time.perf_counter()
        vj += dmtril.dot(eri1.T).dot(eri1)
        ph['vj'] += time.perf_counter() - t
        t = time.perf_counter()
        buf1 = buf[:nb * nocc]
        fdrv(ftrans, fmmm, _p(buf1), _p(eri1), _p(orbo), ctypes.c_int(nb), ctypes.c_int(nao),
             (ctypes.c_int * 4)(0, nocc, 0, nao), null, ctypes.c_int(0))
        ph['e2_drv'] += time.perf_counter() - t
        t = time.perf_counter()
        vk += lib_dot_tn(buf1)
        ph['dot'] += time.perf_counter() - t
    t = time.perf_counter()
    vj = unpack_tril(vj, 1)[0]
    ph['unpack'] = time.perf_counter() - t
    flops = 4.0 * naux * nao * nao * nocc + 4.0 * naux * nao * (nao + 1) / 2
    get_jk.last_phases = {k: round(v, 3) for k, v in ph.items()}
    get_jk.last_threads = nth
    return vj, vk, flops
