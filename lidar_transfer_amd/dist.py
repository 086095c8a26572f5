"""Scan-parallel multi-GPU driver: one process per GPU, no data-path collective, one gather.

Every output scan of the reference has its own mesh, BVH and image, and for the adaptions ``cp`` and ``mesh`` the loop
body of ``lidar_deform.py:393-462`` carries no state from one iteration to the next -- so the scan index list is
block-partitioned over the ranks of a ``torch.distributed`` job (backend ``nccl`` = RCCL over xGMI on the GPU box,
``gloo`` in CPU tests) and the rendered images are gathered ONCE at the end.

``mergemesh`` -- the adaption the reference's shipped config selects -- is the exception: it clips ONE ``voxel_bounds``
array scan after scan (laserscan.py:957-962, fusion_lidar.py:33-37), so a scan's volume lattice depends on every EARLIER
scan of its sequence, and the reference starts every sequence with the configured bounds (one process per sequence,
experiments/run_lidar_deform.sh).  A rank whose block starts in the middle of a sequence must therefore first REPLAY the
bounds of the scans before it (projection + bounds statements only: ``DeviceDeform.mergemesh_bounds``), and every rank
resets the bounds where its block crosses into a new sequence: :func:`mergemesh_plan`.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

#: xGMI: every pair of the 8 GPUs of a node has its own link, ~64 GB/s per direction (7 links x ~153 GB/s bidirectional
#: minus protocol).  A gather to ONE root moves each peer's images over that peer's own link into
#: the root: the per-link rate bounds it, and the root's HBM takes the sum.  NOMINAL figure, used only when nothing was
#: measured: a job with peers times a gather before its clock starts (:func:`measure_link_gbs`, bench.py) and decides on that.
XGMI_LINK_GBS = 64.0


#: scan files of SemanticKITTI sequences 00-07 (configuration C5 of BASELINE.json)
C5_SEQUENCES = [("00", 4541), ("01", 1101), ("02", 4661), ("03", 801), ("04", 271), ("05", 2761), ("06", 1101), ("07", 1101)]


class _StagedWork:
    """A transfer of device tensors over a backend that only moves host memory (gloo): the payload is staged through
    host buffers, `wait()` completes the transport and -- on the receiving side -- copies into the device tensors.  This
    is how the whole of this module runs on ONE GPU with several ranks (tests/test_multigpu_gpu.py): everything but the
    RCCL transport itself."""

    def __init__(self, works, copies):
        self._works, self._copies = works, copies

    def wait(self):
        for w in self._works:
            w.wait()
        for dst, host in self._copies:
            dst.copy_(host, non_blocking=False)
        self._works, self._copies = [], []
        return True


def _host_transport(group) -> bool:
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend(group) == "gloo"


def measure_link_gbs(device, megabytes: int = 64, reps: int = 3, dst: int = 0, group=None):
    """GB/s each peer's link into the root sustains while ALL peers send at once (what the gather of the images does):
    every rank sends `megabytes` to `dst` with :func:`gather_to_root`, the slowest rank's time counts (all-reduced, so that
    every rank holds the same figure).  ``None`` at world size 1 (no link)."""
    import time

    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world <= 1:
        return None
    rank = dist.get_rank(group)
    n = megabytes * (1 << 20) // 4
    src = torch.zeros(n, dtype=torch.float32, device=device)
    recv = [torch.empty(n, dtype=torch.float32, device=device) for _ in range(world)] if rank == dst else None
    best = None
    for i in range(reps + 1):
        if src.is_cuda:
            torch.cuda.synchronize(device)
        dist.barrier(group)
        t0 = time.perf_counter()
        for w in gather_to_root(src, recv, dst, group, copy_self=False):
            w.wait()
        if src.is_cuda:
            torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=device if not _host_transport(group) else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        if i:  # (the first round sets the channels up)
            best = float(t.item()) if best is None else min(best, float(t.item()))
    return n * 4 / best / 1e9


def choose_gather(world_size: int, bytes_per_scan: float, scans_per_s_per_rank: float,
                  link_gbs: Optional[float] = None, headroom: float = 0.9) -> str:
    """``"root"`` or ``"sharded"``: can the images of a job be gathered on one rank while they are rendered?

    Each peer produces ``bytes_per_scan * scans_per_s_per_rank`` bytes per second for the root, over its own
    point-to-point xGMI link.  Beyond ``headroom`` of the link rate the gather, not the renderer, would set the job's
    throughput -- then every rank keeps (and writes) the scans it rendered, as the reference's loop would if it were
    started once per shard (lidar_deform.py:385-390 with ``--offset`` / ``--batch_interval``), and only per-scan
    metadata travels (:func:`render_scans` with ``gather="sharded"``)."""
    if world_size <= 1:
        return "root"
    per_link = float(bytes_per_scan) * float(scans_per_s_per_rank) / 1e9
    return "sharded" if per_link > headroom * float(link_gbs or XGMI_LINK_GBS) else "root"


def scan_indices(n_scan_files: int, nscans: int = 1, offset: int = 0, batch_interval: int = 1) -> List[int]:
    """Indices the reference's batch loop visits (lidar_deform.py:385-390, :457-459): start at
    ``max(offset, nscans // 2)``, step ``batch_interval``, stop before ``n_scan_files - (nscans - 1)``."""
    idx = offset
    prev = nscans // 2
    if prev > idx:
        idx += prev - idx
    end = n_scan_files - (nscans - 1)
    return list(range(idx, max(end, idx), batch_interval)) if idx < end else []


def job_scan_list(sequences, nscans: int = 1, offset: int = 0, batch_interval: int = 1) -> List:
    """The output scans of a multi-sequence job, in the order the reference would produce them when its loop is run over the
    sequences one after the other (experiments/run_lidar_deform.sh:13-23): ``[(sequence, scan index), ...]`` with the
    indices of :func:`scan_indices` per sequence.  ``sequences``: ``[(name, number of scan files), ...]``.  Block-partitioning
    THIS list (:func:`partition`) -- not the sequences -- is what balances SemanticKITTI 00-07's unequal lengths
    (4541 / 1101 / 4661 / 801 / 271 / 2761 / 1101 / 1101 files) over 8 ranks to within one scan."""
    out = []
    for name, n_files in sequences:
        out += [(name, i) for i in scan_indices(int(n_files), nscans, offset, batch_interval)]
    return out


def mergemesh_plan(job: Sequence, world_size: int, rank: int) -> Dict[str, List]:
    """This rank's share of a ``mergemesh`` job (``job``: the list of :func:`job_scan_list`), with what its ORDER-DEPENDENT
    bounds need: ``block`` -- the rank's contiguous block of output scans (:func:`partition`); ``replay`` -- the scans of
    the block's first sequence that come BEFORE the block: their clouds go through ``DeviceDeform.mergemesh_bounds``
    (projection + the bounds statements, no fusion) so that the block's first scan finds the bounds the single-process run
    would have left; ``resets`` -- the positions in ``block`` at which a new sequence starts: ``reset_bounds(configured
    bounds)`` before that scan.  Position 0 is in ``resets`` exactly when nothing is to be replayed."""
    job = list(job)
    block = partition(job, world_size, rank)
    if not block:
        return {"block": [], "replay": [], "resets": []}
    first_seq = block[0][0]
    base, extra = divmod(len(job), world_size)
    start = rank * base + min(rank, extra)   # (where partition() starts this rank's block)
    replay = []
    i = start - 1
    while i >= 0 and job[i][0] == first_seq:
        replay.append(job[i])
        i -= 1
    replay.reverse()
    resets = [k for k in range(len(block)) if (k == 0 and not replay) or (k > 0 and block[k][0] != block[k - 1][0])]
    return {"block": block, "replay": replay, "resets": resets}


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
    staged = []  # (device tensor, host buffer) pairs to fill once the transport is done (host-memory backends only)
    host = _host_transport(group)
    if rank == dst:
        if recv is None:
            raise ValueError("gather_to_root: the destination rank needs receive buffers")
        if copy_self and src is not None and src.numel() > 0:
            recv[rank].copy_(src, non_blocking=True)
        for r in range(world):
            if r == dst or recv[r].numel() == 0:
                continue
            buf = recv[r]
            if host and buf.is_cuda:
                import torch
                buf = torch.empty(recv[r].shape, dtype=recv[r].dtype, device="cpu")
                staged.append((recv[r], buf))
            ops.append(dist.P2POp(dist.irecv, buf, r, group))
    elif src is not None and src.numel() > 0:
        ops = [dist.P2POp(dist.isend, src.cpu() if (host and src.is_cuda) else src, dst, group)]
    works = list(dist.batch_isend_irecv(ops)) if ops else []
    return [_StagedWork(works, staged)] if staged else works


def render_scans(indices: Sequence[int], render_fn: Callable[[int], Dict[str, "object"]], keys: Sequence[str],
                 group=None, dst: int = 0, gather: str = "root",
                 meta_fn: Optional[Callable[[Dict[str, "object"]], "object"]] = None):
    """Render ``indices`` scan-parallel and gather the images on rank ``dst``.

    ``render_fn(idx)`` returns a dict of equally shaped ``torch`` tensors per scan (e.g. ``range`` [H*W]
    f32, ``label`` [H*W] i32) on this rank's device.  Each rank renders its block; the stacked local results
    go to ``dst`` with ONE gather per key (``gather_to_root``: no padding, every rank sends exactly its
    block).  Returns on ``dst`` a dict ``key -> tensor [len(indices), ...]`` in the order of ``indices``
    (``None`` on other ranks).

    ``gather="sharded"`` (see :func:`choose_gather`): the images STAY on the rank that rendered them -- every rank
    gets ``{"sharded": True, "indices": its block, "local": {key: stack}, "counts": block sizes of all ranks}`` --
    and only ``meta_fn(result)`` (a small 1-D tensor per scan, e.g. hit count and checksum; default: the number of
    non-zero elements of the first key) is gathered: ``"meta"`` on ``dst`` is ``[len(indices), M]`` in index order.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if gather not in ("root", "sharded"):
        raise ValueError("render_scans: gather must be 'root' or 'sharded'")
    mine = partition(list(indices), world, rank)
    local = [render_fn(i) for i in mine]
    counts = [len(partition(list(indices), world, r)) for r in range(world)]
    if gather == "sharded":
        if meta_fn is None:
            def meta_fn(d):  # noqa: E731
                return (d[keys[0]] != 0).sum().reshape(1).to(torch.int64)
        stacks = {k: (torch.stack([d[k] for d in local]) if local else None) for k in keys}
        metas = [meta_fn(d).reshape(-1).to(torch.int64) for d in local]
        width = [metas[0].numel() if metas else 0]
        if world > 1:  # (block partition: rank 0 holds a scan whenever anybody does)
            dist.broadcast_object_list(width, src=0, group=group)
        out = {"sharded": True, "indices": mine, "local": stacks, "counts": counts, "meta": None}
        if width[0] == 0:
            return out
        dev = metas[0].device if metas else (local[0][keys[0]].device if local else torch.device("cpu"))
        mstack = torch.stack(metas) if metas else torch.empty((0, width[0]), dtype=torch.int64, device=dev)
        if world == 1:
            out["meta"] = mstack
            return out
        if rank == dst:
            if not metas:  # a root with an empty block: its device
                dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() and \
                    dist.get_backend(group) == "nccl" else torch.device("cpu")
            full = torch.empty((sum(counts), width[0]), dtype=torch.int64, device=dev)
            offs = [sum(counts[:r]) for r in range(world)]
            works = gather_to_root(mstack, [full[offs[r]:offs[r] + counts[r]] for r in range(world)], dst, group)
            out["meta"] = full
        else:
            works = gather_to_root(mstack, None, dst, group)
        for w in works:
            w.wait()
        return out
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
