"""Scan-parallel rendering over RCCL: needs >= 2 MI355X on the node (skipped otherwise -- the single-GPU test box).

SURVEY.md section 4 item 4 / section 8e: the same scans rendered by 1 rank and by 2 ranks (block partition of the scan
list, lidar_deform.py:385-390, :457-459, one gather of the images to rank 0) are byte-identical; and
`python bench.py --gpus 2` really becomes a 2-rank job."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _render_fn_factory(dev_index):
    import torch
    from lidar_transfer_amd.laserscan import create_rays
    from lidar_transfer_amd.raytracer import RaySet, Scene
    from lidar_transfer_amd.synth import synth_scene
    dev = torch.device("cuda", dev_index)
    H, W = 16, 256
    rays = torch.from_numpy(create_rays(3.0, -25.0, H, W)).to(dev)
    rs = RaySet(rays, H)
    sc = Scene(dev_index)
    keep = []

    def render(idx):
        mesh = tuple(torch.from_numpy(x).to(dev) for x in synth_scene(100 + idx, 20000))
        keep.append(mesh)
        sc.set_mesh(*mesh)
        o = sc.render(rs, (0.1 * idx, 0.0, 0.0), label_image=True)
        torch.cuda.synchronize(dev)
        return {"range": o["range"].clone(), "label": o["endcolors"].clone(), "tri": o["tri"].clone()}
    return render


def _worker(rank, world, port, indices, q):
    import torch
    import torch.distributed as dist
    from lidar_transfer_amd.dist import render_scans
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    out = render_scans(indices, _render_fn_factory(rank), ("range", "label", "tri"))
    if rank == 0:
        q.put({k: v.cpu().numpy() for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs on the node")
def test_two_ranks_gather_equals_one_rank_bytewise():
    import torch.multiprocessing as mp
    from lidar_transfer_amd.dist import render_scans
    indices = list(range(5))
    single = render_scans(indices, _render_fn_factory(0), ("range", "label", "tri"))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, indices, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for k in ("range", "label", "tri"):
        a, b = single[k].cpu().numpy(), got[k]
        assert a.shape == b.shape and a.tobytes() == b.tobytes(), k
    assert (got["range"] > 0).any()


def _worker_one_gpu(rank, world, port, indices, gather, q):
    """A rank of a job whose ranks SHARE cuda:0: backend gloo (RCCL refuses two ranks on one device), device tensors --
    lidar_transfer_amd.dist stages them through host buffers for a host-memory transport; everything else is the code the
    RCCL job runs: partition, the real HIP render per scan, gather_to_root with real peers, the sharded mode's metadata."""
    import torch
    import torch.distributed as dist
    from lidar_transfer_amd.dist import choose_gather, measure_link_gbs, render_scans
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = render_scans(indices, _render_fn_factory(0), ("range", "label", "tri"), gather=gather)
    link = measure_link_gbs(torch.device("cuda", 0), megabytes=4, reps=1)
    # the all-reduced decision of the bench (every rank must come to the same answer from the same measured figures)
    modes = [choose_gather(world, 786432.0, 80000.0, link_gbs=link), choose_gather(world, 786432.0, 10.0, link_gbs=link)]
    gathered = [None] * world
    dist.all_gather_object(gathered, (modes, round(link, 6)))
    assert all(g == gathered[0] for g in gathered), gathered
    if gather == "root":
        if rank == 0:
            q.put({k: v.cpu().numpy() for k, v in out.items()})
        else:
            assert all(out[k] is None for k in out)
    else:
        q.put((rank, {k: v.cpu().numpy() for k, v in out["local"].items() if v is not None}, out["indices"],
               out["meta"].cpu().numpy() if out["meta"] is not None else None, out["counts"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,gather", [(2, "root"), (3, "root"), (2, "sharded")])
def test_ranks_sharing_the_one_gpu_equal_the_single_process_run(world, gather):
    """N > 1 on the hardware that exists: `world` processes on cuda:0, the real HIP render per scan, lidar_transfer_amd.dist
    with real peers -- byte-identical to the single-process run, in both gather modes."""
    import torch.multiprocessing as mp
    from lidar_transfer_amd.dist import partition, render_scans
    indices = list(range(7))
    single = render_scans(indices, _render_fn_factory(0), ("range", "label", "tri"))
    single = {k: v.cpu().numpy() for k, v in single.items()}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_one_gpu, args=(r, world, port, indices, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(1 if gather == "root" else world)]
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    if gather == "root":
        for k in ("range", "label", "tri"):
            assert single[k].shape == got[0][k].shape and single[k].tobytes() == got[0][k].tobytes(), k
    else:
        meta0 = None
        for rank, local, mine, meta, counts in got:
            assert mine == partition(indices, world, rank) and counts == [len(partition(indices, world, r)) for r in range(world)]
            for k in ("range", "label", "tri"):
                assert local[k].tobytes() == single[k][mine[0]:mine[-1] + 1].tobytes(), (rank, k)
            if rank == 0:
                meta0 = meta
        assert meta0 is not None and meta0.shape == (len(indices), 1)
        assert np.array_equal(meta0[:, 0], (single["range"] != 0).sum(axis=1))


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs on the node")
def test_bench_gpus_2_prints_two_ranks():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                          "--no-cpu-baseline", "--no-other"], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "RCCL" in out["config"]["parallelism"]
    assert np.isfinite(out["value"]) and out["value"] > 0


def test_bench_gpus_more_than_present_fails_loudly():
    n = _n_gpus()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert res.returncode != 0 and f"--gpus {n + 1}" in res.stderr and not res.stdout.strip()


# ---- RCCL at the only world size the single-GPU test box has ------------------------------------------------------------
_NCCL1_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from lidar_transfer_amd.dist import gather_to_root, render_scans
from test_multigpu_gpu import _render_fn_factory
mode = sys.argv[1]
indices = list(range(4))
keys = ("range", "label", "tri")
if mode == "nccl":
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.barrier()                                     # the communicator really exists
    t = torch.arange(8, device="cuda", dtype=torch.float32)
    dist.all_reduce(t)
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
out = render_scans(indices, _render_fn_factory(0), keys)
sh = render_scans(indices, _render_fn_factory(0), keys, gather="sharded")
if mode == "nccl":
    # the explicit gather path of bench.py at world size 1: root part by device copy, no peers
    recv = torch.empty((1,) + tuple(out["range"].shape), dtype=torch.float32, device="cuda")
    for w in gather_to_root(out["range"], recv, dst=0):
        w.wait()
    torch.cuda.synchronize()
    assert torch.equal(recv[0], out["range"])
    dist.barrier()
    dist.destroy_process_group()
for k in keys:
    assert torch.equal(sh["local"][k], out[k]), k
assert sh["meta"].shape == (4, 1) and int(sh["meta"].min()) > 0
np.savez(sys.argv[2], **{k: out[k].cpu().numpy() for k in keys})
"""


def test_rccl_world_size_1_gather_equals_no_dist(tmp_path):
    """RCCL communicator creation, a collective, `render_scans` and `gather_to_root` under backend nccl at world size 1 --
    the only size the single-GPU test box offers -- byte-equal to the run without torch.distributed; the sharded mode
    keeps the same images local."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = {}
    for mode in ("nodist", "nccl"):
        path = str(tmp_path / f"{mode}.npz")
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, "-c", _NCCL1_SCRIPT % (ROOT, ROOT), mode, path, str(port)], capture_output=True,
                           text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        res[mode] = np.load(path)
    for k in ("range", "label", "tri"):
        assert res["nodist"][k].tobytes() == res["nccl"][k].tobytes(), k
    assert (res["nccl"]["range"] > 0).any()


@pytest.mark.parametrize("gather", ["root", "sharded"])
def test_bench_under_torchrun_one_rank_rccl(gather):
    """`torchrun --nproc-per-node 1 bench.py`: backend nccl (= RCCL) is initialised, the gather path of the timed region
    runs (root: images; sharded: metadata), and the line says the timed scans were verified."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(LT_BENCH_GATHER=gather, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps",
                          "2", "--warmup", "1", "--workload", "C1", "--scenes", "4", "--no-cpu-baseline", "--no-other", "--no-e2e", "--no-chain"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and "RCCL" in out["config"]["parallelism"] and out["verified"] is True
    assert out["config"]["gather"]["mode"] == gather
    assert ("sharded" in out["config"]["parallelism"]) == (gather == "sharded")
    assert np.isfinite(out["value"]) and out["value"] > 0


def test_bench_line_is_compact_and_complete(tmp_path):
    """the default command's shape on a small configuration: ONE stdout line < 8 KB with the contract keys, a compact
    roofline and cpu_baseline"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload", "C1",
                          "--scenes", "4", "--cpu-reps", "2"], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) < 8000, (len(lines), len(res.stdout))
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    rl, cb = out["roofline"], out["cpu_baseline"]
    assert rl["kernel"] == "k_sc_tris" and rl["bound"] == "hbm" and 0 < rl["frac"] < 1 and rl["avg_kernel_ms"] > 0
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3 and rl["step_clock"]["frac_on_step_clock"] > 0
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert out["verified"] is True and out["dtype"] == "f32" and out["scaling"] == "weak"


def test_bench_job_c5_eight_ranks_sharing_the_gpu_equals_single_process(tmp_path):
    """Configuration C5 reduced (SemanticKITTI 00-07's real scan-count ratios / 100 = 169 output scans, 64 x 2048 images,
    20 k-triangle meshes) through bench.py's OWN chunked gather: 8 ranks (gloo transport, all on cuda:0 -- the one-GPU
    box) block-partition the job, render their unequal blocks and gather range + label images on rank 0 in pieces inside
    the timed region; what rank 0 holds equals the single-process run byte for byte."""
    common = ["--job", "c5/100", "--tris", "20000", "--scenes", "8", "--streams", "8", "--warmup", "1", "--no-cpu-baseline"]
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    one, eight = str(tmp_path / "one.npz"), str(tmp_path / "eight.npz")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common], capture_output=True, text=True, timeout=900,
                        env=dict(base, LT_BENCH_DUMP=one))
    assert r1.returncode == 0, r1.stderr[-3000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r8 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", *common],
                        capture_output=True, text=True, timeout=1500,
                        env=dict(base, LT_BENCH_DUMP=eight, LT_BENCH_SHARE_GPU="1", LT_BENCH_BACKEND="gloo", LT_BENCH_GATHER="root",
                                 OMP_NUM_THREADS="4"))
    assert r8.returncode == 0, r8.stderr[-3000:]
    l1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    l8 = json.loads([l for l in r8.stdout.splitlines() if l.startswith("{")][0])
    assert l1["n_gpus"] == 1 and l8["n_gpus"] == 8 and l8["scaling"] == "strong" and l8["config"]["gather"]["mode"] == "root"
    assert l1["verified"] is True and l8["verified"] is True
    a, b = np.load(one), np.load(eight)
    assert a["range"].shape == (169, 64 * 2048) and b["range"].shape == a["range"].shape
    assert a["range"].tobytes() == b["range"].tobytes() and a["label"].tobytes() == b["label"].tobytes()
    assert (a["range"] > 0).mean() > 0.5
