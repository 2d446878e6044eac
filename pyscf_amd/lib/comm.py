"""The collectives of the aux-shard / grid-tile decomposition: one place that decides whether a partial
result is all-reduced (torch.distributed; backend 'nccl' = RCCL over xGMI, 'gloo' in the CPU tests) and that can time it.

The serial decomposition this replaces is the `vj += ...; vk[k] += ...` accumulation over `dfobj.loop()` blocks of
pyscf/df/df_jk.py:362-381: with the aux index sharded over ranks the same sum runs over the ranks instead.

A collective is issued when the group has more than one rank - or, with PAMD_FORCE_COLLECTIVE=1 / `force(True)`, whenever a
process group is initialised, so that the RCCL path can be exercised on a single GPU (world_size = 1; tests/test_gpu_rccl.py).
"""
import os

_force = None
_timer = None


def force(flag=True):
    global _force
    _force = flag


def forced():
    if _force is not None:
        return _force
    return os.environ.get('PAMD_FORCE_COLLECTIVE', '0') not in ('', '0')


def initialized():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except ImportError:
        return False


def active(world):
    return world > 1 or (forced() and initialized())


class CommTimer:
    """HIP events on the launch stream around every collective (ProcessGroupNCCL makes the current stream wait for the
    collective, so the pair brackets it); bench.py reports the sum as comm_ms."""

    def __init__(self):
        self.records = []

    def total_ms(self):
        import torch
        torch.cuda.synchronize()
        return sum(e0.elapsed_time(e1) for e0, e1, _ in self.records), sum(n for _, _, n in self.records)

    def reset(self):
        self.records = []


def set_timer(t):
    global _timer
    _timer = t


def all_reduce(tensors, group=None, world=None, op=None):
    """Sum (or `op`) every tensor of `tensors` over the ranks of `group`, in place; no-op when no collective is active."""
    if world is None:
        import torch.distributed as dist
        world = dist.get_world_size(group) if initialized() else 1
    if not active(world):
        return False
    import torch.distributed as dist
    tensors = [t for t in tensors if t is not None]
    t = _timer
    if t is not None and tensors and tensors[0].is_cuda:
        import torch
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    for x in tensors:
        dist.all_reduce(x, op=op if op is not None else dist.ReduceOp.SUM, group=group)
    if t is not None and tensors and tensors[0].is_cuda:
        e1.record()
        t.records.append((e0, e1, sum(x.numel() * x.element_size() for x in tensors)))
    return True


def backend_name(group=None):
    """'nccl' (RCCL), 'gloo', ... of the group's process group; '' when none is initialised."""
    if not initialized():
        return ''
    import torch.distributed as dist
    try:
        return str(dist.get_backend(group))
    except Exception:
        return ''
