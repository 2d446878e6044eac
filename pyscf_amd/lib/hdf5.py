"""Minimal HDF5 access through the C library (ctypes on libhdf5; h5py is not part of this image).

Just enough for the one on-disk object of the DF path: the dataset ``'j3c'`` of shape (naux, nao_pair), float64, that
``pyscf.df.DF`` keeps in its ``_cderi`` file (pyscf/df/df.py:97-99,185-199; written by pyscf/df/outcore.py:217-221 and
read back row block by row block in ``DF.loop``, df.py:214-242).  Files written here (one 2-d dataset 'j3c') open in h5py /
stock PySCF (``DF._cderi = 'file.h5'``); in the other direction both layouts stock PySCF writes are read: the single
dataset (``_compatible_format = True``) and the group ``'j3c/0..N'`` of column blocks (the default of outcore.cholesky_eri_b).
"""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

_hid = ctypes.c_int64
_lib = None
_H5F_ACC_RDONLY, _H5F_ACC_RDWR, _H5F_ACC_TRUNC = 0, 1, 2
_H5S_SELECT_SET = 0
SIGNATURE = b'\x89HDF\r\n\x1a\n'


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = []
    if os.environ.get('PAMD_LIBHDF5'):
        cands.append(os.environ['PAMD_LIBHDF5'])
    found = ctypes.util.find_library('hdf5')
    if found:
        cands.append(found)
    for d in ('/opt/conda/lib', '/usr/lib/x86_64-linux-gnu', '/usr/lib/x86_64-linux-gnu/hdf5/serial', '/usr/lib64', '/usr/local/lib'):
        cands += sorted(glob.glob(os.path.join(d, 'libhdf5.so*')) + glob.glob(os.path.join(d, 'libhdf5_serial.so*')))
    for c in cands:
        try:
            lib = ctypes.CDLL(c)
            if lib.H5open() < 0:
                continue
        except OSError:
            continue
        for f in ('H5Fcreate', 'H5Fopen', 'H5Screate_simple', 'H5Dcreate2', 'H5Dopen2', 'H5Dget_space'):
            getattr(lib, f).restype = _hid
        lib._f64 = _hid.in_dll(lib, 'H5T_NATIVE_DOUBLE_g').value
        _lib = lib
        return lib
    raise ImportError('libhdf5 not found (set PAMD_LIBHDF5 to the shared library)')


def available():
    try:
        _load()
        return True
    except ImportError:
        return False


def is_hdf5(path):
    try:
        with open(path, 'rb') as f:
            return f.read(8) == SIGNATURE
    except OSError:
        return False


def _chk(rc, what):
    if rc < 0:
        raise IOError('HDF5: %s failed' % what)
    return rc


class Dataset:
    def __init__(self, lib, dset):
        self._lib, self._id = lib, dset
        sp = _chk(lib.H5Dget_space(_hid(dset)), 'H5Dget_space')
        dims = (ctypes.c_uint64 * 8)()
        nd = _chk(lib.H5Sget_simple_extent_dims(_hid(sp), dims, None), 'H5Sget_simple_extent_dims')
        lib.H5Sclose(_hid(sp))
        self.shape = tuple(int(dims[i]) for i in range(nd))

    def _io(self, fn, r0, arr):
        lib = self._lib
        assert len(self.shape) == 2 and arr.ndim == 2 and arr.shape[1] == self.shape[1] and arr.flags.c_contiguous
        assert arr.dtype == np.float64 and 0 <= r0 and r0 + arr.shape[0] <= self.shape[0]
        if arr.shape[0] == 0:
            return
        fsp = _chk(lib.H5Dget_space(_hid(self._id)), 'H5Dget_space')
        start = (ctypes.c_uint64 * 2)(r0, 0)
        count = (ctypes.c_uint64 * 2)(arr.shape[0], arr.shape[1])
        _chk(lib.H5Sselect_hyperslab(_hid(fsp), _H5S_SELECT_SET, start, None, count, None), 'H5Sselect_hyperslab')
        msp = _chk(lib.H5Screate_simple(2, count, None), 'H5Screate_simple')
        rc = fn(_hid(self._id), _hid(lib._f64), _hid(msp), _hid(fsp), _hid(0), arr.ctypes.data_as(ctypes.c_void_p))
        lib.H5Sclose(_hid(msp))
        lib.H5Sclose(_hid(fsp))
        _chk(rc, 'dataset transfer')

    def write_rows(self, r0, arr):
        self._io(self._lib.H5Dwrite, r0, np.ascontiguousarray(arr, dtype=np.float64))

    def read_rows(self, r0, r1):
        out = np.empty((r1 - r0, self.shape[1]))
        self._io(self._lib.H5Dread, r0, out)
        return out

    def is_native_f64_le(self):
        """True when the stored datatype is IEEE float, 8 bytes, little-endian - the only layout a raw np.memmap('<f8') of the
        dataset's bytes reads correctly (r06, ADVICE r05: a float32 or big-endian 'j3c' used to be streamed as garbage).
        H5Dget_type / H5Tget_class (H5T_FLOAT = 1) / H5Tget_size / H5Tget_order (H5T_ORDER_LE = 0)."""
        lib = self._lib
        lib.H5Dget_type.restype = _hid
        lib.H5Tget_size.restype = ctypes.c_size_t
        t = lib.H5Dget_type(_hid(self._id))
        if t < 0:
            return False
        try:
            return lib.H5Tget_class(_hid(t)) == 1 and lib.H5Tget_size(_hid(t)) == 8 and lib.H5Tget_order(_hid(t)) == 0
        finally:
            lib.H5Tclose(_hid(t))

    def file_offset(self):
        """Byte offset of the raw data in the file when the dataset is stored CONTIGUOUSLY (what `create_dataset` here and
        h5py's default write), else None (chunked / compressed layouts have no single offset) - H5Dget_offset."""
        fn = self._lib.H5Dget_offset
        fn.restype = ctypes.c_uint64
        off = fn(_hid(self._id))
        return None if off == 0xffffffffffffffff else int(off)

    def close(self):
        if self._id is not None:
            self._lib.H5Dclose(_hid(self._id))
            self._id = None


class File:
    """``with File(path, 'w') as f: d = f.create_dataset('j3c', (naux, npair)); d.write_rows(0, block)``"""

    def __init__(self, path, mode='r'):
        lib = self._lib = _load()
        p = os.fsencode(path)
        if mode == 'w':
            self._id = _chk(lib.H5Fcreate(p, ctypes.c_uint(_H5F_ACC_TRUNC), _hid(0), _hid(0)), 'H5Fcreate ' + path)
        else:
            self._id = _chk(lib.H5Fopen(p, ctypes.c_uint(_H5F_ACC_RDWR if mode == 'r+' else _H5F_ACC_RDONLY), _hid(0)),
                            'H5Fopen ' + path)
        self._open = []

    def create_dataset(self, name, shape):
        lib = self._lib
        dims = (ctypes.c_uint64 * len(shape))(*shape)
        sp = _chk(lib.H5Screate_simple(len(shape), dims, None), 'H5Screate_simple')
        ds = _chk(lib.H5Dcreate2(_hid(self._id), name.encode(), _hid(lib._f64), _hid(sp), _hid(0), _hid(0), _hid(0)),
                  'H5Dcreate2 ' + name)
        lib.H5Sclose(_hid(sp))
        d = Dataset(lib, ds)
        self._open.append(d)
        return d

    def exists(self, name):
        """True when the link `name` (every component of it) exists in the file."""
        parts, cur = name.strip('/').split('/'), ''
        for comp in parts:
            cur = comp if not cur else cur + '/' + comp
            if self._lib.H5Lexists(_hid(self._id), cur.encode(), _hid(0)) <= 0:
                return False
        return True

    def column_blocks(self, name):
        """The datasets 'name/0', 'name/1', ... of a group of column blocks - the layout stock PySCF's
        outcore.cholesky_eri_b writes when DF._compatible_format is False (pyscf/df/outcore.py:215-221, df.py:227-241) -
        or [] when `name` is a plain dataset."""
        out, i = [], 0
        while self.exists('%s/%d' % (name, i)):
            out.append(self['%s/%d' % (name, i)])
            i += 1
        return out

    def __getitem__(self, name):
        d = Dataset(self._lib, _chk(self._lib.H5Dopen2(_hid(self._id), name.encode(), _hid(0)), 'H5Dopen2 ' + name))
        self._open.append(d)
        return d

    def close(self):
        for d in self._open:
            d.close()
        self._open = []
        if self._id is not None:
            self._lib.H5Fclose(_hid(self._id))
            self._id = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
