"""Multi-GPU sharding of a render: one process per GPU, sample-partitioned, one sum-reduce.

Path samples are independent and the scene is read-only, so a render of `total_samples` per
pixel splits into disjoint sample-index ranges, one per rank; every rank renders its range
into its own full-frame film and the films are summed once at the end.  That sum is the
counterpart of the reference's ``RenderEngine::rebuild_master_canvas`` /
``Canvas::add_from`` (integrator.cpp:469-487, canvas.cpp:47-52); with the ``nccl`` backend it
is an RCCL reduce over xGMI, with ``gloo`` (CPU tests) the same call on host tensors.
There is no other exchange step on this path.
"""
from __future__ import annotations


def partition_samples(total_samples: int, world: int, rank: int):
    """Sample-index range [first, first + count) of `rank`: ranges are disjoint, cover
    [0, total_samples) in rank order, and the remainder goes to the low ranks."""
    if world <= 0 or not (0 <= rank < world) or total_samples < 0:
        raise ValueError("bad partition request")
    base, rem = divmod(total_samples, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def reduce_film(film_tensor, dst: int = 0):
    """Sums every rank's film {r, g, b, passes} into rank `dst` (in place there).

    `film_tensor` is the flat float32 tensor backing the film (W*H*4).  Pass counts are small
    integers stored as floats, so their sum is exact; colour sums depend on the reduction
    order in the last bits, as any float sum over ranks does."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():   # (a group of one is a valid, if trivial, reduce)
        dist.reduce(film_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return film_tensor
