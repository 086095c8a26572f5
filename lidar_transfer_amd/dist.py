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


def gather_to_root(src, recv=None, dst: int = 0, group=None, copy_self: bool = True):
    """ONE gather, issued the way RCCL implements it: a group of point-to-point transfers.

    Every rank other than ``dst`` sends ``src``; ``dst`` receives rank r's tensor into ``recv[r]`` (``recv``:
    a list of tensors, or one tensor whose first dimension is the world size; the parts may differ in size)
    and copies its own part with a plain device copy -- RCCL would move the root's send-to-itself through its
    channel kernels at a fraction of the copy rate.  With 8 GPUs the root receives from 7 peers over 7
    separate xGMI links concurrently.  ``copy_self=False`` leaves the root's own part where it is (``recv[dst]``
    is not touched): a driver that already holds its images does not need a second copy of them.  Returns the
    list of outstanding works (``w.wait()``); an empty ``src`` (0 elements) is skipped on both sides.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ops = []
    if rank == dst:
        if recv is None:
            raise ValueError("gather_to_root: the destination rank needs receive buffers")
        if copy_self and src is not None and src.numel() > 0:
            recv[rank].copy_(src, non_blocking=True)
        ops = [dist.P2POp(dist.irecv, recv[r], r, group) for r in range(world) if r != dst and recv[r].numel() > 0]
    elif src is not None and src.numel() > 0:
        ops = [dist.P2POp(dist.isend, src, dst, group)]
    return list(dist.batch_isend_irecv(ops)) if ops else []


def render_scans(indices: Sequence[int], render_fn: Callable[[int], Dict[str, "object"]], keys: Sequence[str],
                 group=None, dst: int = 0):
    """Render ``indices`` scan-parallel and gather the images on rank ``dst``.

    ``render_fn(idx)`` returns a dict of equally shaped ``torch`` tensors per scan (e.g. ``range`` [H*W]
    f32, ``label`` [H*W] i32) on this rank's device.  Each rank renders its block; the stacked local results
    go to ``dst`` with ONE gather per key (``gather_to_root``: no padding, every rank sends exactly its
    block).  Returns on ``dst`` a dict ``key -> tensor [len(indices), ...]`` in the order of ``indices``
    (``None`` on other ranks).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = partition(list(indices), world, rank)
    local = [render_fn(i) for i in mine]
    counts = [len(partition(list(indices), world, r)) for r in range(world)]
    out = {}
    for k in keys:
        stack = torch.stack([d[k] for d in local]) if local else None
        if world == 1:
            out[k] = stack
            continue
        # shape/dtype of one scan's tensor: a root with an empty block learns it from the first rank's metadata
        # (block partition: rank 0 holds a scan whenever anybody does)
        meta = [None]
        if rank == 0:
            meta = [(tuple(stack.shape[1:]), stack.dtype, stack.device.type) if stack is not None else None]
        dist.broadcast_object_list(meta, src=0, group=group)
        if meta[0] is None:  # no scans at all
            out[k] = None
            continue
        shape, dtype, devtype = meta[0]
        if rank == dst:
            dev = stack.device if stack is not None else (
                torch.device("cuda", torch.cuda.current_device()) if devtype == "cuda" else torch.device("cpu"))
            full = torch.empty((sum(counts),) + shape, dtype=dtype, device=dev)
            offs = [sum(counts[:r]) for r in range(world)]
            recv = [full[offs[r]:offs[r] + counts[r]] for r in range(world)]
            works = gather_to_root(stack, recv, dst, group)
        else:
            full = None
            works = gather_to_root(stack, None, dst, group)
        for w in works:
            w.wait()
        out[k] = full
    return out
