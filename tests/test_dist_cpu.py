"""Multi-process (world_size 2, gloo, CPU) test of the scan-parallel driver: partitioning and the
single gather reproduce the single-process result byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lidar_transfer_amd.dist import partition, render_scans, scan_indices


def test_scan_indices_follow_the_reference_batch_loop():
    assert scan_indices(10) == list(range(10))
    assert scan_indices(10, nscans=5) == [2, 3, 4, 5]          # automatic offset nscans//2, stop at n-(nscans-1)
    assert scan_indices(4541, nscans=1, offset=0, batch_interval=10)[-1] == 4540
    assert scan_indices(100, nscans=3, offset=7, batch_interval=10) == list(range(7, 98, 10))
    assert scan_indices(3, nscans=5) == []


def test_partition_blocks():
    items = list(range(11))
    parts = [partition(items, 4, r) for r in range(4)]
    assert sum(parts, []) == items and [len(p) for p in parts] == [3, 3, 3, 2]
    assert partition([], 2, 1) == [] and partition([5], 2, 1) == [] and partition([5], 2, 0) == [5]


def _fake_render(idx):
    g = torch.Generator().manual_seed(idx)
    return {"range": torch.rand(8 * 16, generator=g), "label": torch.randint(0, 260, (8 * 16, 3), generator=g,
                                                                              dtype=torch.int32)}


def _worker(rank, world, port, indices, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = render_scans(indices, _fake_render, ("range", "label"))
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    else:
        assert out["range"] is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 2, 1])
def test_render_scans_two_ranks_equals_one(n):
    indices = list(range(3, 3 + n))
    single = render_scans(indices, _fake_render, ("range", "label"))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, indices, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(got["range"], single["range"].numpy())
    assert np.array_equal(got["label"], single["label"].numpy())


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lidar_transfer_amd.dist import gather_to_root
    # the pattern bench.py uses: two images per chunk (range f32, label i32), several chunks in flight
    works, recvs = [], []
    for chunk in range(3):
        rng = torch.full((2 + chunk, 16), float(10 * rank + chunk))
        lab = torch.full((2 + chunk, 16), 100 * rank + chunk, dtype=torch.int32)
        rr = [torch.empty((world, 2 + chunk, 16)), torch.empty((world, 2 + chunk, 16), dtype=torch.int32)] \
            if rank == 0 else [None, None]
        works += gather_to_root(rng, rr[0]) + gather_to_root(lab, rr[1])
        recvs.append(rr)
    for w in works:
        w.wait()
    if rank == 0:
        q.put([[t.numpy() for t in rr] for rr in recvs])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_to_root_grouped_send_recv(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for chunk, (rng, lab) in enumerate(got):
        for r in range(world):
            assert np.all(rng[r] == 10 * r + chunk) and np.all(lab[r] == 100 * r + chunk)


def test_bench_cli_parses_without_a_gpu():
    """`bench.py --help` must work on a box without a GPU (the driver's contract flags are all there)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert res.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in res.stdout
