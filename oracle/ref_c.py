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

Threading as in a stock PySCF build (OpenMP-threaded C, BLAS serial inside the parallel regions).  Two builds of the same sources:
`libref_dfjk.so` on the OpenBLAS that ships in scipy's wheel (pthread-based, compiled for at most 64 concurrent callers: the
reference's `#pragma omp` loops then run on min(host cores, 64) threads - more callers crashed it on the 256-core GPU host), and
- r05 - `libref_dfjk_mkl.so` on the MKL inside libtorch_cpu.so (no cap: all host cores; chosen by `variant()` after a self-test in
a child process).  `numpy.matmul` of the J line keeps numpy's own BLAS threading in both.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_ref', 'libref_dfjk.so')
SO_MKL = os.path.join(HERE, '_ref', 'libref_dfjk_mkl.so')
_lib = None
_variant = None            # 'openblas' (scipy's, <= 64 concurrent callers) | 'mkl' (libtorch_cpu's, no cap: all host cores)


def available():
    return os.path.exists(SO)


def build():
    """(re)build oracle/_ref when the reference tree is present (this container); elsewhere use the prebuilt files."""
    if os.path.isdir('/root/reference/pyscf/lib'):
        import subprocess
        subprocess.check_call(['make', '-s', '-C', HERE, 'ref'])
        try:
            subprocess.check_call(['make', '-s', '-C', HERE, 'ref_mkl'])
        except Exception:                       # no torch install to link against: the OpenBLAS variant alone
            pass
    return available()


calibration = None         # filled by calibrate(): {'openblas': {'threads', 'ms_per_row'}, 'mkl': {...}, 'chosen'}


def mkl_usable():
    """The MKL variant exists, the host has more cores than OpenBLAS admits callers, and the variant passed its self-test in a
    CHILD process (all host cores calling into it at once on a small case, checked against numpy) - a BLAS that misbehaves takes
    down the child, not the benchmark."""
    global _mkl_ok
    if _mkl_ok is None:
        _mkl_ok = False
        if os.path.exists(SO_MKL) and (os.cpu_count() or 1) > MAX_BLAS_CALLERS and os.environ.get('PAMD_REF_BLAS', '') != 'openblas':
            import subprocess
            import sys
            code = ("import sys; sys.path.insert(0, %r); import os; os.environ['PAMD_REF_BLAS'] = 'mkl'; "
                    "from oracle import ref_c; ref_c.selftest()" % os.path.dirname(HERE))
            try:
                p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
                _mkl_ok = p.returncode == 0 and 'REF_C_SELFTEST_OK' in p.stdout
            except Exception:
                pass
    return _mkl_ok


_mkl_ok = None


def calibrate(cderi, dm, mo_coeff, mo_occ, nthreads=None, rows=240):
    """r05: which build of the reference's C is the honest baseline on THIS host?  Both are timed on the first `rows` aux rows
    (second of two runs each): scipy's OpenBLAS on min(cores, 64) threads, and the MKL of libtorch_cpu.so on all cores.  The FASTER
    one is used for the timed run (on the AMD hosts of this pool MKL takes its generic code path and 256 threads of it are ~4x
    slower than 64 threads of OpenBLAS - measured, r05); both figures are reported (`calibration`)."""
    global _variant, calibration
    import time as _t
    want = os.environ.get('PAMD_REF_BLAS', '')
    cands = ['openblas'] + (['mkl'] if mkl_usable() else [])
    if want in cands:
        cands = [want]
    out = {}
    sub = cderi[:min(rows, len(cderi))]
    for v in cands:
        get_jk(sub, dm, mo_coeff, mo_occ, nthreads=nthreads, which=v)
        t0 = _t.perf_counter()
        get_jk(sub, dm, mo_coeff, mo_occ, nthreads=nthreads, which=v)
        out[v] = {'threads': get_jk.last_threads, 'ms_per_row': round((_t.perf_counter() - t0) / len(sub) * 1e3, 3)}
    _variant = min(out, key=lambda k: out[k]['ms_per_row'])
    out['chosen'] = _variant
    calibration = out
    return out


def variant():
    """Which BLAS the reference's C runs on: PAMD_REF_BLAS=openblas | mkl, else what calibrate() chose, else OpenBLAS."""
    global _variant
    if _variant is not None:
        return _variant
    want = os.environ.get('PAMD_REF_BLAS', '')
    if want in ('openblas', 'mkl'):
        _variant = want if (want == 'openblas' or os.path.exists(SO_MKL)) else 'openblas'
        return _variant
    _variant = 'openblas'
    return _variant


def selftest():
    """Small J/K through the loaded variant on ALL host cores against numpy; prints REF_C_SELFTEST_OK."""
    rng = np.random.RandomState(3)
    nao, naux, nocc = 96, 3 * (os.cpu_count() or 1) + 5, 24
    npair = nao * (nao + 1) // 2
    cderi = rng.rand(naux, npair) - .5
    c = np.linalg.qr(rng.rand(nao, nao))[0]
    occ = np.zeros(nao)
    occ[:nocc] = 2
    dm = (c * occ).dot(c.T)
    for _ in range(3):                           # repeated: a caller-table overflow shows up under sustained load
        vj, vk, _f = get_jk(cderi, dm, c, occ, blockdim=2 * (os.cpu_count() or 1), nthreads=os.cpu_count())
    full = np.zeros((naux, nao, nao))
    idx = np.tril_indices(nao)
    full[:, idx[0], idx[1]] = cderi
    full[:, idx[1], idx[0]] = cderi
    vj0 = np.einsum('Lpq,L->pq', full, np.einsum('Lpq,pq->L', full, dm))
    vk0 = np.einsum('Lpr,rs,Lqs->pq', full, dm, full)
    assert np.abs(vj - vj0).max() < 1e-10 and np.abs(vk - vk0).max() < 1e-10, (np.abs(vj - vj0).max(), np.abs(vk - vk0).max())
    print('REF_C_SELFTEST_OK %s threads %d' % (variant(), get_jk.last_threads), flush=True)


def describe():
    extra = ''
    if calibration and len(calibration) > 2:
        extra = '; calibration on 240 rows: ' + ', '.join('%s %d threads %.2f ms/row' % (k, v['threads'], v['ms_per_row'])
                                                          for k, v in calibration.items() if k != 'chosen') + ' -> the faster one is timed'
    return _describe() + extra


def _describe():
    if variant() == 'mkl':
        return ('BLAS = the MKL linked into libtorch_cpu.so (dgemm_; dsymm_ = triangle completion + dgemm_, oracle/ref_build/'
                'dsymm_via_dgemm.c), one BLAS thread per call inside the reference\'s OpenMP regions, %d OpenMP threads = all host '
                'cores (self-tested in a child process first)' % get_jk.last_threads)
    return ("BLAS = scipy's OpenBLAS, one BLAS thread per call inside the reference's OpenMP regions; that build admits 64 concurrent "
            "callers, so the omp loops run on min(host cores, 64) = %d threads (the host has %d)" % (get_jk.last_threads, os.cpu_count()))


_libs = {}


def lib(which=None):
    which = which or variant()
    if which not in _libs:
        if not available():
            raise RuntimeError('oracle/_ref/libref_dfjk.so is missing: run `make -C oracle ref` where /root/reference exists')
        if which == 'mkl':
            import torch                                   # libtorch_cpu.so (and its OpenMP runtime) first: ONE libgomp in the process
            torch.set_num_threads(1)                       # MKL: one thread per dgemm_ call
            l = ctypes.CDLL(SO_MKL)
        else:
            l = ctypes.CDLL(SO)
            l.scipy_openblas_set_num_threads.argtypes = [ctypes.c_int]
        l.omp_get_max_threads.restype = ctypes.c_int
        _libs[which] = l
    return _libs[which]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


MAX_BLAS_CALLERS = 64      # scipy's OpenBLAS is compiled for 64 threads: more CONCURRENT callers corrupt its buffer table
                           # ("precompiled NUM_THREADS exceeded ... double free or corruption" on a 256-core host, r03)


def set_threads(n, which=None):
    """OpenMP threads of the reference's C (lib.num_threads(), pyscf/lib/misc.py:195-224), capped at MAX_BLAS_CALLERS; BLAS
    stays at one thread per call.  Returns the thread count really used (bench.py reports it as `cores`)."""
    which = which or variant()
    l = lib(which)
    if which == 'mkl':
        import torch
        torch.set_num_threads(1)
        n = int(n or os.cpu_count() or 1)                  # no cap on concurrent callers
    else:
        l.scipy_openblas_set_num_threads(1)
        n = min(int(n or os.cpu_count() or 1), MAX_BLAS_CALLERS)
    l.omp_set_num_threads(ctypes.c_int(n))
    return l.omp_get_max_threads()


_current = None            # the variant get_jk is running on (the helpers below follow it)


def pack_tril(mats):
    l = lib(_current)
    mats = np.ascontiguousarray(mats, dtype=np.float64)
    count, nd = mats.shape[0], mats.shape[-1]
    out = np.empty((count, nd * (nd + 1) // 2))
    l.NPdpack_tril_2d(ctypes.c_int(count), ctypes.c_int(nd), _p(out), _p(mats))
    return out


def unpack_tril(tril, filltriu=1):
    l = lib(_current)
    tril = np.ascontiguousarray(tril, dtype=np.float64)
    count = tril.shape[0]
    nd = int(round((np.sqrt(8 * tril.shape[1] + 1) - 1) / 2))
    out = np.empty((count, nd, nd))
    l.NPdunpack_tril_2d(ctypes.c_int(count), ctypes.c_int(nd), _p(tril), _p(out), ctypes.c_int(filltriu))
    return out


def lib_dot_tn(buf1):
    """lib.dot(buf1.T, buf1): ddot sees a = buf1.T (F-contiguous -> trans_a 'T', a = buf1) and b = buf1 (C-contiguous, 'N')
    and calls NPdgemm(trans_b, trans_a, n, m, k, ldb, lda, ldc, ...)  (pyscf/lib/numpy_helper.py:825-858,980-1004)."""
    l = lib(_current)
    k, n = buf1.shape
    c = np.empty((n, n))
    l.NPdgemm(ctypes.c_char(b'N'), ctypes.c_char(b'T'), ctypes.c_int(n), ctypes.c_int(n), ctypes.c_int(k),
              ctypes.c_int(n), ctypes.c_int(n), ctypes.c_int(n), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0),
              _p(buf1), _p(buf1), _p(c), ctypes.c_double(1.0), ctypes.c_double(0.0))
    return c


def get_jk(cderi, dm, mo_coeff, mo_occ, blockdim=240, nthreads=None, which=None):
    """J and K of ONE density (nset = 1, hermi = 1, MO branch) over the rows of `cderi` (naux_rows, nao_pair), exactly the call
    sequence of pyscf/df/df_jk.py:329-381.  Returns (vj, vk, flops, phases)."""
    global _current
    which = which or variant()
    _current = which
    l = lib(which)
    nth = set_threads(nthreads, which)
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
