"""Recycling pool of page-locked host blocks for RESULT arrays (J, K, vxc handed to the caller as numpy arrays).

A device -> host copy into page-locked memory runs at the PCIe rate (~50 GB/s) instead of the pageable ~10 GB/s plus first-touch
page faults; allocating such a block costs milliseconds, so blocks are reused.  Liveness is tracked EXPLICITLY (the
earlier pools asked `sys.getrefcount`, a CPython implementation detail that changes with 3.14's borrowed references): every
hand-out wraps the block in a fresh ctypes buffer object; the numpy array given to the caller - and every view anybody takes of
it - keeps that buffer object alive through `.base`; a `weakref.finalize` on the buffer object returns the block to the free
list when the LAST of them is gone.  A lock makes the pool safe for callers on several Python threads.  numpy + ctypes only.
"""
import ctypes
import threading
import weakref

import numpy as np


class Block:
    """One page-locked allocation: `address` (int), `nbytes`, and whatever object owns the memory (`owner`)."""

    def __init__(self, address, nbytes, owner):
        self.address = int(address)
        self.nbytes = int(nbytes)
        self.owner = owner
        self.busy = False


class PinnedPool:
    def __init__(self, allocate, max_idle=8):
        """allocate(nbytes) -> (address, owner) of a page-locked block, or raises."""
        self._allocate = allocate
        self._blocks = []
        self._lock = threading.RLock()      # re-entrant: a finalizer (_release) may fire on THIS thread inside take() when the
                                            # cyclic GC runs there - a plain Lock would deadlock
        self._max_idle = max_idle

    def _release(self, blk):
        with self._lock:
            blk.busy = False

    def take(self, n):
        """(float64 numpy array of n elements in a page-locked block, block); None when the allocation is refused."""
        n = max(int(n), 1)
        with self._lock:
            blk = None
            for b in self._blocks:
                if not b.busy and b.nbytes >= 8 * n and b.nbytes <= 32 * n:
                    blk = b
                    break
            if blk is None:
                if sum(1 for b in self._blocks if not b.busy) >= self._max_idle:      # drop idle blocks before growing without bound
                    self._blocks = [b for b in self._blocks if b.busy]
                try:
                    address, owner = self._allocate(8 * n)
                except Exception:
                    return None
                blk = Block(address, 8 * n, owner)
                self._blocks.append(blk)
            blk.busy = True
        buf = (ctypes.c_double * n).from_address(blk.address)
        weakref.finalize(buf, self._release, blk)            # fires when the array below and all its views are gone
        return np.frombuffer(buf, dtype=np.float64), blk

    def stats(self):
        with self._lock:
            return {'blocks': len(self._blocks), 'busy': sum(1 for b in self._blocks if b.busy)}
