"""Scan-parallel multi-GPU driver: one process per GPU, no data-path collective, one gather.

Every output scan of the reference has its own mesh, BVH and image -- the loop body of
``lidar_deform.py:393-462`` carries no state from one iteration to the next -- so the scan index list
is block-partitioned over the ranks of a ``torch.distributed`` job (backend ``nccl`` = RCCL over xGMI
on the GPU box, ``gloo`` in CPU tests) and the rendered images are gathered ONCE at the end.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence


def scan_indices(n_scan_files: int, nscans: int = 1, offset: int = 0, batch_interval: int = 1) -> List[int]:
    """Indices the reference's batch loop visits (lidar_deform.py:385-390, :457-459): start at
    ``max(offset, nscans // 2)``, step ``batch_interval``, stop before ``n_scan_files - (nscans - 1)``."""
    idx = offset
    prev = nscans // 2
    if prev > idx:
        idx += prev - idx
    end = n_scan_files - (nscans - 1)
    return list(range(idx, max(end, idx), batch_interval)) if idx < end else []


def partition(items: Sequence, world_size: int, rank: int) -> List:
    """Contiguous block partition, sizes differ by at most one (lower ranks get the extra item)."""
    n = len(items)
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return list(items[start:start + base + (1 if rank < extra else 0)])


def render_scans(indices: Sequence[int], render_fn: Callable[[int], Dict[str, "object"]], keys: Sequence[str],
                 group=None, dst: int = 0):
    """Render ``indices`` scan-parallel and gather the images on rank ``dst``.

    ``render_fn(idx)`` returns a dict of equally shaped ``torch`` tensors per scan (e.g. ``range`` [H*W]
    f32, ``label`` [H*W] i32) on this rank's device.  Each rank renders its block; local results are
    stacked, padded to the longest block and exchanged with ONE ``all_gather_into_tensor`` per key (7
    concurrent peer-to-peer xGMI transfers into every rank with RCCL).  Returns on ``dst`` a dict
    ``key -> tensor [len(indices), ...]`` in the order of ``indices`` (``None`` on other ranks).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = partition(list(indices), world, rank)
    local = [render_fn(i) for i in mine]
    counts = [len(partition(list(indices), world, r)) for r in range(world)]
    longest = max(counts) if counts else 0
    out = {}
    for k in keys:
        if local:
            stack = torch.stack([d[k] for d in local])
        else:
            stack = None
        if world == 1:
            out[k] = stack
            continue
        # shape/dtype of one scan's tensor: ranks with an empty block learn it from rank 0's metadata
        meta = [None]
        if rank == 0:
            meta = [(tuple(stack.shape[1:]), stack.dtype, stack.device.type)]
        dist.broadcast_object_list(meta, src=0, group=group)
        shape, dtype, devtype = meta[0]
        dev = stack.device if stack is not None else torch.device(
            "cuda", torch.cuda.current_device()) if devtype == "cuda" else torch.device("cpu")
        padded = torch.zeros((longest,) + shape, dtype=dtype, device=dev)
        if stack is not None:
            padded[:stack.shape[0]] = stack
        gathered = torch.empty((world * longest,) + shape, dtype=dtype, device=dev)
        dist.all_gather_into_tensor(gathered, padded, group=group)
        if rank == dst:
            parts = [gathered[r * longest:r * longest + counts[r]] for r in range(world)]
            out[k] = torch.cat(parts)
        else:
            out[k] = None
    return out
