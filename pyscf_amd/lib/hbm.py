"""One HBM budget per device: who holds what, so that the components of one calculation - the DF tensor
(df/df.py), its half-transform block, the compact AO image of the XC leg (dft/sparse_grid.py) - size themselves against the SAME
numbers instead of each asking hipMemGetInfo with a private reserve.  The order of placement is: tensor, X block, XC compact image,
work space; an OPTIONAL buffer (the partial K image of the packed layout) only ever takes what that leaves.

This is book-keeping, not an allocator: `hold` records bytes a component has allocated (by name), `held` lets another component
subtract them from what it was told to leave room for (a Kohn-Sham SCF builds its XC plan BEFORE the tensor, bench.py after it: the
decision must not depend on the order)."""
import threading

_lock = threading.Lock()
_held = {}


def _key(dev):
    if isinstance(dev, str):
        return int(dev.split(':')[1]) if ':' in dev else 0
    idx = getattr(dev, 'index', dev)
    return 0 if idx is None else int(idx)


def hold(dev, name, nbytes):
    with _lock:
        _held[(_key(dev), name)] = int(nbytes)


def drop(dev, name):
    with _lock:
        _held.pop((_key(dev), name), None)


def held(dev, name):
    with _lock:
        return _held.get((_key(dev), name), 0)


def free_bytes(dev):
    """Bytes a new allocation of this process can still get on `dev`: the driver's free memory plus what torch's caching allocator
    holds without using."""
    import torch
    return torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
