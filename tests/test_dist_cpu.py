"""Multi-process (world_size 2, gloo, CPU) test of the scan-parallel driver: partitioning and the
single gather reproduce the single-process result byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lidar_transfer_amd.dist import choose_gather, partition, render_scans, scan_indices


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


def _meta(d):
    return torch.stack([(d["range"] > 0.5).sum(), d["label"].to(torch.int64).sum()])


def _sharded_worker(rank, world, port, indices, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = render_scans(indices, _fake_render, ("range", "label"), gather="sharded", meta_fn=_meta)
    q.put((rank, out["indices"], out["counts"], {k: (v.numpy() if v is not None else None) for k, v in out["local"].items()},
           out["meta"].numpy() if out["meta"] is not None else None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 7), (3, 4), (2, 1)])
def test_render_scans_sharded_keeps_images_local_and_gathers_metadata(world, n):
    """The fallback for jobs whose image rate exceeds the xGMI links into one root: every rank keeps the scans it
    rendered (the same bytes the root gather would have delivered), only the per-scan metadata is gathered."""
    indices = list(range(10, 10 + n))
    single = render_scans(indices, _fake_render, ("range", "label"))
    want_meta = torch.stack([_meta(_fake_render(i)) for i in indices]).numpy()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, indices, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    off = 0
    for rank, mine, counts, local, meta in got:
        assert mine == partition(indices, world, rank) and counts == [len(partition(indices, world, r)) for r in range(world)]
        if mine:
            assert np.array_equal(local["range"], single["range"].numpy()[off:off + len(mine)])
            assert np.array_equal(local["label"], single["label"].numpy()[off:off + len(mine)])
        else:
            assert local["range"] is None
        off += len(mine)
        if rank == 0:
            assert np.array_equal(meta, want_meta)
        else:
            assert meta is None
    # single process: same structure
    one = render_scans(indices, _fake_render, ("range", "label"), gather="sharded", meta_fn=_meta)
    assert one["sharded"] and np.array_equal(one["meta"].numpy(), want_meta)


def test_choose_gather_rule():
    """Root gather while each peer's image stream fits its own xGMI link (with headroom), sharded beyond."""
    R = 64 * 2048
    assert choose_gather(1, 6 * R, 1e6) == "root"                      # nothing to gather from
    assert choose_gather(8, 6 * R, 10_000) == "root"                   # 7.9 GB/s per link
    assert choose_gather(8, 6 * R, 80_000) == "sharded"                # 63 GB/s per link: at the wire
    assert choose_gather(8, 6 * R, 50_000, headroom=0.7) == "root"     # 39 GB/s < 44.8
    assert choose_gather(8, 6 * R, 60_000, headroom=0.7) == "sharded"  # 47 GB/s > 44.8


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


KITTI_00_07 = [("00", 4541), ("01", 1101), ("02", 4661), ("03", 801), ("04", 271), ("05", 2761), ("06", 1101), ("07", 1101)]


def _c5_render(item):
    seq, idx = item
    g = torch.Generator().manual_seed(int(seq) * 100000 + idx)
    return {"range": torch.rand(4 * 8, generator=g), "label": torch.randint(0, 260, (4 * 8,), generator=g, dtype=torch.int32)}


def _c5_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lidar_transfer_amd.dist import job_scan_list
    items = job_scan_list(KITTI_00_07, nscans=1, offset=0, batch_interval=10)
    out = render_scans(items, _c5_render, ("range", "label"))
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_c5_eight_ranks_over_the_real_sequence_lengths():
    """Config C5 (BASELINE.json configs[4]): SemanticKITTI sequences 00-07 with `batch_interval: 10`, 8 ranks.  The job's scan
    list is the reference's per-sequence loop (lidar_deform.py:385-390, :457-459) run over the sequences in turn; its block
    partition differs by at most ONE scan between ranks although the sequences differ 17-fold in length, and the one gather
    delivers every image at the position the single-process run has it -- byte for byte."""
    from lidar_transfer_amd.dist import job_scan_list
    items = job_scan_list(KITTI_00_07, nscans=1, offset=0, batch_interval=10)
    per_seq = [len(scan_indices(n, 1, 0, 10)) for _, n in KITTI_00_07]
    assert per_seq == [455, 111, 467, 81, 28, 277, 111, 111] and len(items) == sum(per_seq) == 1641
    blocks = [partition(items, 8, r) for r in range(8)]
    sizes = [len(b) for b in blocks]
    assert max(sizes) - min(sizes) <= 1 and sum(blocks, []) == items
    assert max(per_seq) / min(per_seq) > 16              # what a sequence-per-rank mapping would leave idle
    single = render_scans(items, _c5_render, ("range", "label"))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_c5_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert got["range"].shape == (1641, 32)
    assert got["range"].tobytes() == single["range"].numpy().tobytes()
    assert got["label"].tobytes() == single["label"].numpy().tobytes()


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


def test_mergemesh_plan_replays_the_sequence_prefix_and_resets_at_sequence_boundaries():
    """mergemesh clips ONE bounds array scan after scan (laserscan.py:957-962): a block that starts inside a sequence replays
    the scans before it, a block that crosses into a new sequence resets there; the blocks still tile the job."""
    from lidar_transfer_amd.dist import C5_SEQUENCES, job_scan_list, mergemesh_plan, partition
    job = job_scan_list([("00", 10), ("01", 3), ("02", 7)])
    cover = []
    for r in range(4):
        p = mergemesh_plan(job, 4, r)
        assert p["block"] == partition(job, 4, r)
        cover += p["block"]
        first = p["block"][0]
        assert p["replay"] == [it for it in job if it[0] == first[0] and it[1] < first[1]]
        for k in p["resets"]:
            assert p["block"][k][1] == 0 or (k == 0 and not p["replay"])
        assert (0 in p["resets"]) == (not p["replay"])
    assert cover == job
    assert mergemesh_plan(job, 4, 2) == {"block": job[10:15], "replay": [], "resets": [0, 3]}
    assert mergemesh_plan([], 4, 0) == {"block": [], "replay": [], "resets": []}
    big = job_scan_list(C5_SEQUENCES)
    assert sum(len(mergemesh_plan(big, 8, r)["block"]) for r in range(8)) == len(big) == 16338
