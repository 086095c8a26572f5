#!/usr/bin/env python3
"""bench.py -- virtual-LiDAR synthesis throughput on MI355X (one process per GPU).

A *step* is one pass of the hot path over one batch of input: 64 scans (--calls-per-step 8 x --batch 8), each with a NEW
triangle mesh (the mesh changes every scan; the reference rebuilds its BVH per call, RayTracer.cpp:54) -> closest hit
of one ray per (beam, azimuth) cell -> range / colour(label) / remission / end point / triangle images, with
meshes, rays and images resident in HBM; the scatter strategy submits 8 scans per lt_scene_render_batch_dev call.  Workload at N=1: BASELINE.json configs[1] ("C2": ~1 M-triangle scene, 64x2048
HDL-64E target, fov +3/-25), synthetic.

Two MI355X-native strategies produce bit-identical images (tests/test_trace_gpu.py):
  scatter (default)  single-origin triangle scatter: stream the mesh once, atomic z-min per ray
                     (lt_scatter.hip); the ray set of the sensor model is binned once, before the clock,
                     like the rays are uploaded once
  lbvh               Morton/radix-sort/Karras LBVH build + quad traversal per scan (lt_build/lt_trace.hip)
`value` is measured on --strategy (default scatter); the other one is reported beside it.

    python bench.py [--gpus N --steps K --warmup W]            # N=1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Scans shard over ranks (weak scaling: every rank renders K scans), no data-path collective; the rendered
range and label images are gathered to rank 0 over RCCL inside the timed region (one logical gather, issued in
pieces -- eighths, the last eighth in quarters -- so that it overlaps the rendering of later scans and only
1/32 of the images is still to be moved when the last scan is done).  Rank 0 prints
one JSON line with `roofline` for the dominant kernel (HIP events around every launch of it inside the
timed region) and `cpu_baseline` = the real reference raytracer (oracle/_ref, prebuilt from
/root/reference) timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth, 8 XCDs x 4 MiB (MI355X_MICROARCH.md, L2 section)
RANDOM_REQ_CEILING_G = 52.0   # G random 64-byte read requests per second the memory side delivers (tools/tlb_probe.hip, profiles/r03)
ATOMIC_CEILING_G = 27.0       # G memory-side atomic REQUESTS per second, random cells (tools/atomic_probe.hip: "run 1", profiles/r05)
# ... and in LANE atomics per second when neighbouring lanes hit neighbouring cells (runs of >= 8 lanes per 64-byte line: the
# firing order of a spinning LiDAR, what k_pb_project sees): 187-212 G/s measured (profiles/r05/atomic_probe.txt, "run 8..64")
ATOMIC_LANE_CEILING_ORDERED_G = 200.0
# FETCH_SIZE calibration for this kernel's access patterns (tools/fetch_calib.hip -> profiles/r05/fetch_calib.txt): 12-byte
# coalesced triples, 12-byte windowed gathers and 12-byte random gathers all count requests x 64 B while a request moves a
# 128-byte line -> known / counted = 2.0, the same factor as the guide's 16-byte streaming case
FETCH_SIZE_FACTOR = 2.0
N_SIMD, SHADER_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; peak engine clock (MI355X_MICROARCH.md)
# Calibrated with tools/valu_calib.hip (profiles/r03/valu_calib.txt, 4096 straight-line v_fma_f32 per lane):
#   * SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU in every regime (1.000 per instruction): it counts issued instructions, it is NOT
#     a duration -- round 2 multiplied it by 4 cycles ("quad-cycles") and called k_sc_tris 0.79 VALU-busy: wrong by 2x;
#   * with >= 3 waves resident a SIMD issues one wave64 VALU instruction every ~2 cycles (0.89 SIMD-cycles per wave-inst
#     measured per wave with ~3.6 waves co-resident = 2 x 3.6 / 8; 102 TFLOP/s non-packed fp32 over the launch), which is
#     MI355X_MICROARCH.md's "v_fma_f32 (wave64): 2 cyc"; ONE wave alone issues only every 4.67 cycles (dependent or not);
#   * the shader clock under full VALU load is 2.0-2.2 GHz, not the 2.4 GHz peak (s_memtime ticks / 100 MHz wall clock).
# VALU-busy time of a launch = SQ_INSTS_VALU x 2 cycles / 1024 SIMDs / clock.
SQ_CYCLES_PER_COUNT = 2.0
VALU_CYCLES_PER_WAVE_INST = 2.0
PCIE_PEAK_GBS = 63.0   # PCIe Gen5 x16 per direction: 32 GT/s x 16 lanes x 128/130
PCIE_WIRE_GBS = 56.3   # what a pinned hipMemcpyAsync of 26 MB reaches on this box (tools/pcie_probe.hip)
PROBE_EVERY = int(os.environ.get("LT_BENCH_PROBE_EVERY", "8"))  # HIP-event pair around every n-th dominant launch

# ---- algorithmic bytes per unit ----------------------------------------------------
# scatter, kernel k_sc_tris: per triangle 3 indices (12 B); per VERTEX of the mesh 12 B once (an indexed mesh shares
# a vertex between ~6 triangles: the gathers of a block hit the same lines, so a vertex is charged once, not once
# per incident triangle); per Moller-Trumbore test one 16-B grid entry (normalised direction + ray index); per
# accepted hit one 8-B atomic
SC_B_TRI, SC_B_VERT, SC_B_TEST, SC_B_HIT = 12, 12, 16, 8
# lbvh, kernel k_trace4: one 4-wide node 128 B, one triangle record 48 B; per ray 12 B direction in + 44 B out
# (range 4, rem 4, xyz 12, colour 12, tri 4) + 40 B hit gather
LB_B_NODE, LB_B_TRI, LB_B_RAY = 128, 48, 12 + 44 + 40


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step = one batch of input = --calls-per-step x --batch = 64 scans, each with a new mesh, ~0.84 ms: filling and
    # draining the pipeline of 16 scans in flight costs ~1 ms per timed region, so that even a 20-step run (17 ms)
    # measures the steady state to ~5 %; the default 64 steps = 4096 scans = 54 ms
    ap.add_argument("--steps", type=int, default=64, help="timed steps; a step = --calls-per-step batches of --batch scans")
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps before the clock")
    ap.add_argument("--calls-per-step", type=int, default=int(os.environ.get("LT_BENCH_CALLS_PER_STEP", "8")),
                    help="lt_scene_render_batch_dev calls (of --batch scans each) that make up one step")
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--target", default="", help="target sensor YAML (lidar_deform.py --target: name, fov_up, fov_down, "
                                                  "beams, angle_res_hor, fov_hor); overrides the workload's sensor model")
    ap.add_argument("--strategy", default=os.environ.get("LT_BENCH_STRATEGY", "scatter"), choices=["scatter", "lbvh"])
    # what k_sc_tris streams per scene is faces + vertices = 18.3 MB on C2 (colours / remissions are only gathered for hit
    # triangles), so 12 scenes = 220 MB FIT the 256 MiB Infinity Cache: round 3's default was MALL-assisted by ~6 %
    # (profiles/r04/scenes_sweep.jsonl: 4 / 12 / 24 / 48 / 64 scenes -> 11.75 / 10.54 / 10.01 / 9.85 / 9.93 Grays/s; FETCH_SIZE
    # does not move, it counts Infinity-Cache hits).  48 x 18.3 MB = 878 MB > 2 x 256 MiB: the value no longer changes.
    ap.add_argument("--scenes", type=int, default=int(os.environ.get("LT_BENCH_SCENES", "48")),
                    help="distinct scenes cycled through per rank; faces + vertices of all of them (18.3 MB each on C2) must "
                         "exceed twice the 256 MiB Infinity Cache so that no scan finds its mesh cached from the last time "
                         "round (profiles/r04/scenes_sweep.jsonl)")
    # 24 = three batch calls of 8 scans in flight: with round 4's 448-triangle workgroups three launches fill each other's
    # ramps and tails better than two (profiles/r04/streams_sweep.txt: 16 / 24 / 32 -> 10.2 / 10.9 / 9.7 Grays/s; round 3's
    # kernel: 8.24 / 8.28 / 7.74)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("LT_BENCH_STREAMS", "24")),
                    help="scans in flight per GPU (HIP streams)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("LT_BENCH_BATCH", "8")),
                    help="scans per lt_scene_render_batch_dev call (scatter strategy, at most 8; 1 = one call per scan); "
                         "--streams / --batch batches are in flight")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive clock (host buffers in and out)")
    ap.add_argument("--no-chain", action="store_true", help="skip the fusion -> marching cubes -> render sub-record")
    ap.add_argument("--no-other", action="store_true", help="skip the short run of the other strategy")
    ap.add_argument("--cpu-reps", type=int, default=0, help="reference runs for the CPU baseline (0 = auto)")
    ap.add_argument("--probe-only", action="store_true",
                    help="only the serial probe of the dominant kernel (launches of the timed region's shape, back to back "
                         "on one stream): the command tools/r03_profile.sh runs under rocprofv3 --kernel-trace --stats so "
                         "that roofline.avg_kernel_ms can be recomputed from a CSV under profiles/")
    return ap.parse_args()


def cpu_baseline(workload: dict, seed: int, reps: int):
    """Time the real reference (oracle/_ref) in subprocesses on this host; bounded sample.  TWO clocks (SURVEY.md section
    8d) at TWO thread counts: end-to-end `ctrace` (triangle set-up RayTracer.cpp:32-51 + BVH build BVH.cpp:143-243 + trace
    RayTracer.cpp:62-92) and trace-only, at OMP_NUM_THREADS = 1 and = nproc.  The split comes from the reference ITSELF: it
    prints "[Statistic] Built BVH ... in N ms" (BVH.cpp:125) and "Rendering image ..." (RayTracer.cpp:60) on C stdout between
    its phases; the subprocess unbuffers C stdout, routes fd 1 into a pipe and timestamps every line on arrival."""
    code = r"""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, %r)
import numpy as np
from oracle import binding as ob
from lidar_transfer_amd.synth import synth_scene
from lidar_transfer_amd.laserscan import create_rays
wl = json.loads(sys.argv[1]); seed = int(sys.argv[2]); reps = int(sys.argv[3]); kind = sys.argv[4]; budget = float(sys.argv[5])
v, f, c, r = synth_scene(seed, wl["tris"])
rays = create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"]); org = np.zeros(3, np.float32)
libc = C.CDLL(None)
libc.setvbuf(C.c_void_p.in_dll(libc, "stdout"), None, 2, 0)          # _IONBF: a printf is written when it is made
r_fd, w_fd = os.pipe()
saved = os.dup(1); os.dup2(w_fd, 1); os.close(w_fd)
stamps = []
def reader():
    buf = b""
    while True:
        chunk = os.read(r_fd, 65536)
        if not chunk:
            break
        now = time.perf_counter()
        buf += chunk
        while b"\n" in buf:
            line, buf = buf.split(b"\n", 1)
            stamps.append((now, line.decode(errors="replace")))
th = threading.Thread(target=reader, daemon=True); th.start()
runs = []
t_all = time.perf_counter()
for i in range(reps):
    del stamps[:]
    t0 = time.perf_counter()
    if kind == "port":
        o = ob.oracle_trace(rays, org, v, f, c, r, wl["H"], mode=ob.MODE_REF_BVH, norm=ob.NORM_SSE)
        t1 = time.perf_counter()
        st = o["stats"]
        runs.append({"e2e_s": t1 - t0, "setup_build_s": (st["t_setup_ms"] + st["t_build_ms"]) * 1e-3,
                     "build_ms_printed": st["t_build_ms"], "trace_only_s": st["t_trace_ms"] * 1e-3})
    else:
        ob.ref_trace(rays, org, v, f, c, r, wl["H"], kind=kind)
        t1 = time.perf_counter()
        time.sleep(0.002)                                           # let the reader drain the pipe
        built = [(t, l) for t, l in stamps if "Built BVH" in l]
        rend = [(t, l) for t, l in stamps if "Rendering image" in l]
        run = {"e2e_s": t1 - t0}
        if built:
            try:
                run["build_ms_printed"] = float(built[-1][1].rsplit(" in ", 1)[1].split()[0])
            except (IndexError, ValueError):
                pass
        if rend or built:
            t_split = (rend or built)[-1][0]                       # the trace loop starts right after this line
            run["setup_build_s"] = t_split - t0
            run["trace_only_s"] = t1 - t_split
        runs.append(run)
    if time.perf_counter() - t_all > budget:
        break
os.dup2(saved, 1)
sys.stderr.write("LTBASE " + json.dumps({"runs": runs, "threads": ob.num_threads(), "faces": int(f.shape[0])}) + "\n")
""" % ROOT
    n_rays = workload["H"] * workload["W"]
    nproc = os.cpu_count() or 1
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        model = "unknown"

    def one(kind, threads, reps_, budget):
        env = dict(os.environ)
        env["OMP_NUM_THREADS"] = str(threads)
        try:
            res = subprocess.run([sys.executable, "-c", code, json.dumps(workload), str(seed), str(reps_), kind, str(budget)],
                                 capture_output=True, text=True, timeout=300, env=env)
        except subprocess.TimeoutExpired:
            return None
        line = [l for l in res.stderr.splitlines() if l.startswith("LTBASE ")]
        if res.returncode != 0 or not line:
            return None
        info = json.loads(line[0][7:])
        runs = info["runs"]
        best = min(runs, key=lambda r: r["e2e_s"])
        d = {"threads": info["threads"], "runs": len(runs), "faces": info["faces"], "e2e_s": round(best["e2e_s"], 4),
             "e2e_Mrays_s": round(n_rays / best["e2e_s"] / 1e6, 4)}
        tr = [r["trace_only_s"] for r in runs if "trace_only_s" in r]
        if tr:
            d.update(trace_only_s=round(min(tr), 5), trace_only_Mrays_s=round(n_rays / min(tr) / 1e6, 3),
                     setup_build_s=round(min(r["setup_build_s"] for r in runs if "setup_build_s" in r), 4))
        bm = [r["build_ms_printed"] for r in runs if "build_ms_printed" in r]
        if bm:
            d["bvh_build_ms_printed_by_reference"] = min(bm)
        return d

    for kind in ("fast", "strict", "port"):
        if kind != "port" and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", f"libref_{kind}.so")):
            continue
        many = one(kind, nproc, reps, 15.0)
        if not many:
            continue
        single = one(kind, 1, max(2, reps // 3), 12.0)
        lib = f"oracle/_ref/libref_{kind}.so" if kind != "port" else "oracle restatement"
        return {"value": many["e2e_Mrays_s"], "unit": "Mrays/s", "cores": many["threads"],
                "kind": "port" if kind == "port" else "reference",
                "sample": f"{many['runs']} end-to-end ctrace calls (triangle set-up + BVH build + trace) on one "
                          f"{workload['H']}x{workload['W']} scan vs {many['faces']} triangles, min of runs, {lib}, "
                          f"OpenMP threads={many['threads']}; and {single['runs'] if single else 0} calls at 1 thread",
                "s_per_scan": many["e2e_s"], "scans_per_s": round(1.0 / many["e2e_s"], 3), "cpu_model": model,
                "nproc": nproc,
                "clocks": {"all_threads": many, "one_thread": single,
                           "note": "e2e = the whole ctrace call; trace_only = from the reference's own 'Rendering image' "
                                   "line (RayTracer.cpp:60, timestamped on arrival) to the return of the call, i.e. the "
                                   "loop RayTracer.cpp:62-92; the BVH build (BVH.cpp:143-243) is single-threaded at any "
                                   "thread count and dominates e2e"}}
    return None


def launch_command(n_gpus: int, argv, port: int):
    """The one-process-per-GPU launch of this script on one node (what the driver itself would type)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def relaunch_multi_gpu(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become N ranks (one process per GPU)
    under torch.distributed.run on this node -- the job the reference's batch loop (lidar_deform.py:385-390,
    :457-459) is sharded into.  Rank 0 prints the one JSON line on the inherited stdout."""
    import socket
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but this node shows {n_dev} GPU(s)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's peer-to-peer transport needs it here
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    os.execve(sys.executable, launch_command(args.gpus, sys.argv[1:], port), env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_multi_gpu(args)  # does not return
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {os.environ['WORLD_SIZE']} ranks; "
                         f"reporting n_gpus = {os.environ['WORLD_SIZE']}\n")
    # stdout carries exactly one JSON line: RCCL prints a version banner to stdout when its communicator is created
    # (and libraries may print what they like) -- everything else that is written to fd 1 goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from lidar_transfer_amd.dist import gather_to_root
    from lidar_transfer_amd.laserscan import create_rays
    from lidar_transfer_amd.raytracer import RaySet, Scene
    from lidar_transfer_amd.synth import WORKLOADS, synth_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:  # also under `torchrun --nproc-per-node 1` (exercises the gather path)
        dist.init_process_group("nccl", device_id=dev)
    wl = dict(WORKLOADS[args.workload])
    if args.target:  # the target scanner as the reference reads it (lidar_deform.py:302-315)
        from lidar_transfer_amd.config import load_sensor
        sensor = load_sensor(args.target)
        wl.update(H=sensor.H, W=sensor.W, fov_up=float(sensor.fov_up), fov_down=float(sensor.fov_down))
    H, W = wl["H"], wl["W"]
    R = H * W
    S = max(1, args.streams)
    args.batch = min(max(1, args.batch), int(os.environ.get("LT_BENCH_MAX_BATCH", "8")))  # lt_scene_render_batch_dev takes at most 8 scans
    S = (S + args.batch - 1) // args.batch * args.batch  # whole batches of workers
    # A STEP = one pass of the hot path over one batch of input = `--batch` scans, each with its own new mesh (one
    # lt_scene_render_batch_dev call for the scatter strategy).  Everything below counts scans: K timed, Wm untimed.
    SPS = args.batch * max(1, args.calls_per_step)
    K, Wm = args.steps * SPS, args.warmup * SPS
    # every timed scan keeps its range + label image (8 B per ray) and rank 0 also holds the peers' (6 B per ray)
    need = K * R * 8 + (K * R * 6 * (world - 1) if rank == 0 else 0)
    if need > 0.7 * torch.cuda.get_device_properties(dev).total_memory:
        raise SystemExit(f"bench.py: --steps {args.steps} ({K} scans) keeps {need / 2**30:.0f} GiB of images on this "
                         f"rank; use fewer steps")

    # ---- synthetic inputs, resident in HBM before the clock starts -----------------------------------
    scenes = []
    for i in range(args.scenes):
        v, f, c, r = synth_scene(1000 * rank + i, wl["tris"])
        scenes.append(tuple(torch.from_numpy(x).to(dev) for x in (v, f, c, r)))
    n_faces = int(scenes[0][1].shape[0])
    n_verts = int(np.mean([int(sc_[0].shape[0]) for sc_ in scenes]))
    # semantic labels travel as int16 when every label of this rank's scenes fits (decided before the clock;
    # SemanticKITTI labels are < 260): 6 instead of 8 bytes per ray on the xGMI links into the root
    lab_lo = min(int(sc_[2][:, 2].min()) for sc_ in scenes)
    lab_hi = max(int(sc_[2][:, 2].max()) for sc_ in scenes)
    if dist.is_initialized():  # every rank must pick the same width
        lohi = torch.tensor([-lab_lo, lab_hi], dtype=torch.int64, device=dev)
        dist.all_reduce(lohi, op=dist.ReduceOp.MAX)
        lab_lo, lab_hi = -int(lohi[0].item()), int(lohi[1].item())
    label_dtype = torch.int16 if (-32768 <= lab_lo and lab_hi <= 32767) else torch.int32
    rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], H, W)).to(dev)
    origin = (0.0, 0.0, 0.0)
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    workers = [Scene(local_rank) for _ in range(S)]
    # ONE ray set for all workers, as in a sequence (one target sensor model, laserscan.py:1092-1119): it is
    # read-only during renders (the z-min image of a scan in flight belongs to its scene), so its 2 MB bin grid
    # stays resident in every XCD's L2 instead of one grid per in-flight scan cycling through them
    shared_rays = RaySet(rays, H)
    torch.cuda.synchronize()  # (created on the current stream, used on the workers' streams)
    raysets = [shared_rays] * S
    scratch = [workers[0].alloc_outputs(R) for _ in range(S)]

    gather_info = {}

    def run(strategy, K, Wm, keep, groups=None, probe_every=PROBE_EVERY):
        """Timed region for one strategy; returns (seconds, mean dominant-kernel ms, hits of the last scan).
        `groups`: batches in flight (default: all --streams / --batch of them); `probe_every`: HIP-event pair around
        every n-th launch of the dominant kernel."""
        PROBE_EVERY = max(1, probe_every)  # noqa: N806  (shadows the module default inside this timed region)
        dist_on = dist.is_initialized()
        range_all = torch.zeros((K, R), dtype=torch.float32, device=dev) if keep else None
        # the kept colour output is the semantic-label image itself (LT_TRACE_LABEL_IMAGE = deform's unpack
        # label_image = ray_colors[:, :, 2], laserscan.py:912, fused into the write-back)
        color_all = torch.zeros((K, R), dtype=torch.int32, device=dev) if keep else None
        # HIP events around the launches of the dominant kernel in the timed region (created and materialised
        # before the clock starts; recorded by the library on the launch stream).  A marker pair costs the
        # stream ~2.3 us (tools/host_floor.py: 16.6 -> 18.9 us per scan), a tenth of a step, so every
        # PROBE_EVERY-th launch is sampled instead of all of them.
        probes = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range((K + PROBE_EVERY - 1) // PROBE_EVERY)]
        for e0, e1 in probes:
            e0.record()
            e1.record()
        # the gather of the rendered images (range f32 + label i32 = 8 B per ray) to rank 0 is ONE logical
        # collective, issued in `n_chunks` pieces so that it overlaps the rendering of the following scans
        # (grouped send/recv: 7 peers -> root over 7 separate xGMI links)
        n_chunks = int(os.environ.get("LT_BENCH_GATHER_CHUNKS", "8")) if (dist_on and keep) else 1
        do_gather = dist_on and keep and n_chunks > 0
        n_chunks = max(n_chunks, 1)
        bounds = [K * c // n_chunks for c in range(n_chunks + 1)]
        if do_gather and n_chunks > 1:
            # the piece that cannot overlap anything is the LAST one (its scans are the last to finish): split the
            # last chunk into quarters, so that 1/32 of the images is exposed after the last scan instead of 1/8,
            # without paying the per-chunk host cost 32 times
            lo, hi = bounds[-2], bounds[-1]
            bounds = sorted(set(bounds[:-1] + [lo + (hi - lo) * q // 4 for q in (1, 2, 3)] + [hi]))
            n_chunks = len(bounds) - 1
        recv = None
        if do_gather and rank == 0:
            recv = [(torch.empty((world, bounds[c + 1] - bounds[c], R), dtype=torch.float32, device=dev),
                     torch.empty((world, bounds[c + 1] - bounds[c], R), dtype=label_dtype, device=dev))
                    for c in range(n_chunks)]
        works = []
        sharded_meta = False
        coll = os.environ.get("LT_BENCH_COLLECTIVE", "p2p")  # p2p (grouped send/recv) | gather | allgather
        use_allgather = coll == "allgather"

        # Host side of one step, kept as thin as a C++ driver would be: three calls into liblidarhip.so with
        # precomputed handles and pointers (mesh pointer swap, probe events, one render / build+trace).
        import ctypes as C
        from lidar_transfer_amd import _lib
        lib = _lib.load()
        vp = C.c_void_p
        org = (C.c_float * 3)(*origin)
        sh = [vp(st.cuda_stream) for st in streams]
        wh = [w._h for w in workers]
        rh = [r._h for r in raysets]
        mesh_args = [(vp(v.data_ptr()), vp(f.data_ptr()), vp(c.data_ptr()), vp(r.data_ptr()), v.numel() // 3,
                      f.numel() // 3) for v, f, c, r in scenes]
        pr = [(vp(a.cuda_event), vp(b.cuda_event)) for a, b in probes]
        sp = [{k: vp(t.data_ptr()) for k, t in scratch[s].items()} for s in range(S)]
        rng_p = [vp(range_all[k].data_ptr()) for k in range(K)] if keep else None
        col_p = [vp(color_all[k].data_ptr()) for k in range(K)] if keep else None
        rays_p = vp(rays.data_ptr())
        FL = _lib.LT_TRACE_WRITE_MISSES | _lib.LT_TRACE_LABEL_IMAGE

        def step(i, slot=None, timed=False):
            s = i % S
            h = wh[s]
            o = sp[s]
            p_rng = rng_p[slot] if (keep and slot is not None) else o["range"]
            p_col = col_p[slot] if (keep and slot is not None) else o["endcolors"]
            rc = lib.lt_scene_set_mesh_dev(h, *mesh_args[i % len(scenes)])
            if timed and slot % PROBE_EVERY == 0:
                rc |= lib.lt_scene_set_probe(h, *pr[slot // PROBE_EVERY])
            if strategy == "lbvh":
                rc |= lib.lt_scene_build(h, sh[s], None)
                rc |= lib.lt_scene_trace_dev(h, rays_p, org, R, H, o["endpoints"], p_col, p_rng, o["endrem"], o["tri"],
                                             FL, sh[s], None)
            else:
                rc |= lib.lt_scene_render_dev(h, rh[s], org, o["endpoints"], p_col, p_rng, o["endrem"], o["tri"], FL,
                                              sh[s], None)
            if rc:
                _lib.check(rc, "bench step")

        # Scatter strategy, batched: BATCH consecutive scans (each its own mesh, worker and images) go to the GPU with
        # ONE call = three kernel launches for all of them (lt_scene_render_batch_dev); worker w always runs on
        # stream (w // BATCH) % len(streams), so a worker is never in two batches at a time.
        BATCH = args.batch if strategy == "scatter" else 1
        if BATCH > 1:
            assert S % BATCH == 0, "--streams must be a multiple of --batch"
            n_groups = min(groups, S // BATCH) if groups else S // BATCH
            arr = lambda vals: (vp * BATCH)(*vals)  # noqa: E731
            grp_scenes = [arr([wh[g * BATCH + j] for j in range(BATCH)]) for g in range(n_groups)]
            grp_rays = [arr([rh[g * BATCH + j] for j in range(BATCH)]) for g in range(n_groups)]
            grp_out = [{k: arr([sp[g * BATCH + j][k] for j in range(BATCH)]) for k in ("endpoints", "endrem", "tri",
                                                                                       "range", "endcolors")}
                       for g in range(n_groups)]
            org_b = (C.c_float * (3 * BATCH))(*(list(origin) * BATCH))
            if keep:  # the range / colour images of the timed scans go to their slots
                slot_rng = [arr([rng_p[min(b * BATCH + j, K - 1)] for j in range(BATCH)]) for b in range((K + BATCH - 1) // BATCH)]
                slot_col = [arr([col_p[min(b * BATCH + j, K - 1)] for j in range(BATCH)]) for b in range((K + BATCH - 1) // BATCH)]

        def step_batch(i0, nb, slot0=None, timed=False):
            """Scans i0 .. i0 + nb - 1 (nb <= BATCH) as one batch."""
            g = (i0 // BATCH) % n_groups
            rc = 0
            for j in range(nb):
                rc |= lib.lt_scene_set_mesh_dev(wh[g * BATCH + j], *mesh_args[(i0 + j) % len(scenes)])
            b = i0 // BATCH
            if timed and b % PROBE_EVERY == 0:
                rc |= lib.lt_scene_set_probe(wh[g * BATCH], *pr[b // PROBE_EVERY])
            o = grp_out[g]
            use_slots = keep and slot0 is not None
            rc |= lib.lt_scene_render_batch_dev(nb, grp_scenes[g], grp_rays[g], org_b, o["endpoints"],
                                                slot_col[b] if use_slots else o["endcolors"],
                                                slot_rng[b] if use_slots else o["range"], o["endrem"], o["tri"], FL,
                                                sh[g])
            if rc:
                _lib.check(rc, "bench step (batch)")

        def gather_chunk(c):
            c0, c1 = bounds[c], bounds[c + 1]
            cur = torch.cuda.current_stream(dev)
            # (the batch calls only use the first n_groups streams)
            for st in (streams[:n_groups] if BATCH > 1 else streams):  # the collective starts when this chunk's scans are done; later scans keep running
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
            # deform's unpack for the whole chunk: label_image = ray_colors[:, :, 2] (laserscan.py:912)
            label_chunk = color_all[c0:c1].to(label_dtype)
            if coll == "p2p":
                # the gather as RCCL implements it -- one group of send/recv, 7 peers -> root over 7 separate
                # xGMI links -- minus the root's send to itself (a plain device copy instead: RCCL moves the
                # self-part through its channel kernels at ~15 GB/s, which at 48 k scans/s would dominate)
                for k, src in enumerate((range_all[c0:c1], label_chunk)):
                    works.extend(gather_to_root(src, recv[c][k] if rank == 0 else None, dst=0, copy_self=False))
                return
            for k, src in enumerate((range_all[c0:c1], label_chunk)):
                if use_allgather:  # LT_BENCH_COLLECTIVE=allgather: every rank receives everything (ring-bound)
                    ag = recv[c][k] if rank == 0 else torch.empty((world,) + tuple(src.shape), dtype=src.dtype,
                                                                  device=dev)
                    works.append(dist.all_gather_into_tensor(ag, src.contiguous(), async_op=True))
                else:
                    lst = [recv[c][k][r] for r in range(world)] if rank == 0 else None
                    works.append(dist.gather(src, gather_list=lst, dst=0, async_op=True))

        torch.cuda.synchronize()
        t_w = time.perf_counter()
        if BATCH > 1:
            for i in range(0, Wm, BATCH):
                step_batch(i, min(BATCH, Wm - i))
        else:
            for i in range(Wm):
                step(i)
        torch.cuda.synchronize()
        t_w = time.perf_counter() - t_w
        if do_gather:
            # The north star's job: ONE RCCL gather of the rendered images on rank 0 -- the default (`root`; issued in pieces
            # that overlap the rendering).  Each peer's images travel over its own xGMI link into the root, so the link rate
            # bounds a rank at link / bytes-per-scan scans per second: the line reports what the links sustain (MEASURED here
            # with all peers sending at once, lidar_transfer_amd.dist.measure_link_gbs, before the clock) and what the ranks
            # produce (the slowest rank's warm-up rate, all-reduced), so a link-bound job says so.  LT_BENCH_GATHER=auto
            # leaves the images on the ranks that rendered them (only per-scan metadata is gathered) when the images would
            # exceed 0.9 of the measured link rate; =sharded forces that mode.
            from lidar_transfer_amd.dist import XGMI_LINK_GBS, choose_gather, measure_link_gbs
            tw = torch.tensor([t_w], dtype=torch.float64, device=dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            rate = Wm / max(float(tw.item()), 1e-9)
            per_scan = R * (4 + (2 if label_dtype == torch.int16 else 4))
            link = measure_link_gbs(dev) if world > 1 else None
            mode = os.environ.get("LT_BENCH_GATHER", "root")
            if mode not in ("root", "sharded"):
                mode = choose_gather(world, per_scan, rate, link_gbs=link)
            need = per_scan * rate / 1e9
            gather_info.update(mode=mode, warmup_scans_per_s_per_rank=round(rate, 1), bytes_per_scan=per_scan,
                               per_link_GBs_needed=round(need, 2),
                               per_link_GBs_measured=round(link, 2) if link else None,
                               per_link_GBs_nominal=XGMI_LINK_GBS,
                               link_bound_scans_per_s_per_rank=round((link or XGMI_LINK_GBS) * 1e9 / per_scan, 1),
                               link_bound=bool(world > 1 and mode == "root" and need > (link or XGMI_LINK_GBS)))
            gather_info["predicted_job_scans_per_s"] = round(
                world * (min(rate, gather_info["link_bound_scans_per_s_per_rank"]) if (world > 1 and mode == "root") else rate), 1)
            if rank == 0:   # before the clock, on stderr (stdout carries the one JSON line): a first N-GPU run explains itself
                print(f"[bench] gather over {world} ranks: mode={mode}; per-link GB/s measured (all peers sending)="
                      f"{gather_info['per_link_GBs_measured']} nominal={XGMI_LINK_GBS}; each peer needs {need:.2f} GB/s for "
                      f"{rate:.0f} scans/s x {per_scan} B; a link carries {gather_info['link_bound_scans_per_s_per_rank']:.0f} scans/s "
                      f"per rank -> {'LINK-BOUND' if gather_info['link_bound'] else 'not link-bound'}; predicted job rate "
                      f"{gather_info['predicted_job_scans_per_s']:.0f} scans/s", file=sys.stderr, flush=True)
            if mode == "sharded":
                do_gather = False
                sharded_meta = True
                recv = None
        if do_gather:  # warm-up of the collective too: RCCL sets up its peer-to-peer channels lazily
            # ... and of the exact torch ops gather_chunk uses (the first strided-gather / copy kernel of a process
            # costs ~50 ms of module loading, which must not land in the timed region)
            w_lab = color_all[0:2].to(label_dtype)
            torch.empty_like(range_all[0:2]).copy_(range_all[0:2], non_blocking=True)
            torch.empty_like(w_lab).copy_(w_lab, non_blocking=True)
            wbuf = torch.zeros((4, R), dtype=torch.float32, device=dev)
            wl_ = [torch.empty_like(wbuf) for _ in range(world)] if rank == 0 else None
            for _ in range(2):
                if use_allgather:
                    dist.all_gather_into_tensor(torch.empty((world * 4, R), dtype=torch.float32, device=dev), wbuf)
                elif coll == "p2p":
                    for wk in gather_to_root(wbuf, wl_, dst=0, copy_self=False):
                        wk.wait()
                else:
                    dist.gather(wbuf, gather_list=wl_, dst=0)
            torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        chunk = 0
        if BATCH > 1:
            for i in range(0, K, BATCH):
                nb = min(BATCH, K - i)
                step_batch(i, nb, slot0=i, timed=True)
                while do_gather and chunk < n_chunks and i + nb >= bounds[chunk + 1]:
                    gather_chunk(chunk)
                    chunk += 1
            n_probed = ((K + BATCH - 1) // BATCH + PROBE_EVERY - 1) // PROBE_EVERY
        else:
            for i in range(K):
                step(i, slot=i, timed=True)
                if do_gather and i + 1 == bounds[chunk + 1]:
                    gather_chunk(chunk)
                    chunk += 1
            n_probed = len(probes)
        meta_recv = None
        if sharded_meta:
            # the images stay where they were rendered; ONE small gather: hits per scan (8 B per scan) to rank 0
            for st in streams:
                st.synchronize()
            meta = (range_all > 0).sum(dim=1)
            meta_recv = torch.empty((world, K), dtype=meta.dtype, device=dev) if rank == 0 else None
            works.extend(gather_to_root(meta, meta_recv, dst=0, copy_self=True))
        for wk in works:
            wk.wait()
        for st in streams:
            st.synchronize()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if meta_recv is not None:
            assert world == 1 or bool((meta_recv[1:] > 0).any()), "rank 0 did not receive the peers' metadata"
        if dist_on:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in probes[:n_probed]])) if probes else float("nan")
        hits = int((range_all[K - 1] > 0).sum().item()) if keep else -1
        if recv is not None:  # rank 0 really holds every rank's images
            # (its own images stay where they are: range_all / color_all)
            assert world == 1 or bool((recv[-1][0][1:, -1] > 0).any()), "rank 0 did not receive the peers' images"
        verify = verify_timed_scans(range_all, color_all, K) if keep else None
        return dt, kern_ms, hits, verify

    def verify_timed_scans(range_all, color_all, K):
        """The timed region really rendered: the range + label images three timed scans left behind (first, middle, last)
        are compared BIT FOR BIT with a fresh single-scan render of the same mesh (lt_scene_render_dev, one scan per
        launch, outside the clock), and -- when this rank's scene 0 is the golden scene (C2, seed 0, origin 0: fixture F5
        of tests/golden, made by the real reference) -- the SHA-256 of timed scan 0's images with the reference's."""
        import hashlib
        res = {"scans_compared": [], "ok": True}
        lab = scratch[0]["endcolors"].reshape(-1)[:R]
        for sl in sorted({0, K // 2, K - 1}):
            workers[0].set_mesh(*scenes[sl % len(scenes)])
            o = dict(scratch[0])
            o["endcolors"] = lab
            workers[0].render(raysets[0], origin, out=o, label_image=True)
            torch.cuda.synchronize()
            same = bool(torch.equal(range_all[sl].view(torch.int32), o["range"].view(torch.int32))) and \
                bool(torch.equal(color_all[sl], lab))
            res["scans_compared"].append(sl)
            res["ok"] = res["ok"] and same
        gpath = os.path.join(ROOT, "tests", "golden", "f5_c2_1m_64x2048.npz")
        if args.workload == "C2" and not args.target and rank == 0 and os.path.exists(gpath):
            g = np.load(gpath)
            if int(g["seed"]) == 0 and int(g["n_faces"]) == int(scenes[0][1].shape[0]) and int(g["H"]) == H and int(g["W"]) == W:
                rs_ = hashlib.sha256(range_all[0].cpu().numpy().tobytes()).digest()
                ls_ = hashlib.sha256(color_all[0].cpu().numpy().astype(np.int32).tobytes()).digest()
                gold = rs_ == bytes(g["range_sha256"].tobytes()) and ls_ == bytes(g["label_sha256"].tobytes())
                res["golden_sha256_of_timed_scan_0"] = bool(gold)
                res["ok"] = res["ok"] and bool(gold)
        res["what"] = ("range + label images of timed scans bit-identical to a fresh single-scan render of the same mesh"
                       + ("; timed scan 0 equals the real reference's images (golden F5, SHA-256)"
                          if "golden_sha256_of_timed_scan_0" in res else ""))
        return res

    # ---- counting passes (outside the clock): work per scan for the roofline ------------------------------
    cnt = {"scatter": [], "lbvh": []}
    hit_ray_counts = []
    for i in range(len(scenes)):
        workers[0].set_mesh(*scenes[i])
        o = workers[0].render(raysets[0], origin, out=scratch[0], count=True)
        cnt["scatter"].append((o["stats"]["nodes_visited"], o["stats"]["tris_tested"], o["stats"]["n_hits"]))
        hit_ray_counts.append(int((o["range"] > 0).sum().item()))
        if (args.strategy == "lbvh" or not args.no_other) and (args.strategy == "lbvh" or i < 12):  # (the side leg: 12 scenes say enough)
            workers[0].build()
            o = workers[0].trace(rays, origin, H, out=scratch[0], count=True)
            cnt["lbvh"].append((o["stats"]["nodes_visited"], o["stats"]["tris_tested"], o["stats"]["n_hits"]))
    phase = workers[0].build(stats=True) if (args.strategy == "lbvh" or not args.no_other) else {}
    torch.cuda.synchronize()

    KERNEL_SOURCES = {"scatter": ["lt_scatter.hip", "lt_internal.h", "lt_normalize.h"],
                      "lbvh": ["lt_trace.hip", "lt_build.hip", "lt_internal.h", "lt_normalize.h"]}

    def kernel_source_hash(strategy):
        import hashlib
        h = hashlib.sha256()
        for name in KERNEL_SOURCES[strategy]:
            with open(os.path.join(ROOT, "lidar_transfer_amd", "csrc", name), "rb") as fh:
                h.update(name.encode() + fh.read())
        return h.hexdigest()[:16]

    def measured_traffic(strategy, spl, kernel=None):
        """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/rNN/pmc.json
        (tools/pmc_to_json.py: 2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section).  An entry counts only for
        the workload, launch shape AND kernel sources it was collected on -- otherwise null, never a stale constant."""
        import glob
        want = kernel_source_hash(strategy)
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc*.json")), reverse=True):
            try:
                doc = json.load(open(path))
            except (OSError, ValueError):
                continue
            for e in doc.get("entries", []):
                if (e.get("workload") == args.workload and e.get("strategy") == strategy and not args.target
                        and e.get("scans_per_launch") == spl and e.get("kernel_source_hash") == want
                        and e.get("kernel") == (kernel or {"scatter": "k_sc_tris", "lbvh": "k_trace4"}[strategy])):
                    return float(e["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT), e
        return None, None, None

    def measured_ea(kernel, spl):
        """TCC_EA0 request counts per launch (tools/ea_to_json.py), valid for this workload, launch shape and kernel sources"""
        import glob
        want = kernel_source_hash("scatter")
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "ea_requests.json")), reverse=True):
            try:
                doc = json.load(open(path))
            except (OSError, ValueError):
                continue
            for e in doc.get("entries", []):
                if (e.get("workload") == args.workload and not args.target and e.get("scans_per_launch") == spl
                        and e.get("kernel_source_hash") == want and e.get("kernel") == kernel
                        and "tcc_ea0_rdreq_per_launch" in e and "tcc_ea0_atomic_per_launch" in e):
                    return dict(e, _path=os.path.relpath(path, ROOT))
        return None

    def roofline(strategy, serial_ms, insitu_ms):
        spl = args.batch if strategy == "scatter" else 1  # scans per launch of the dominant kernel
        c = np.mean(np.array(cnt[strategy], dtype=np.float64), axis=0)
        if strategy == "scatter":
            alg = n_faces * SC_B_TRI + n_verts * SC_B_VERT + c[1] * SC_B_TEST + c[2] * SC_B_HIT
            extra = {"kernel": "k_sc_tris", "bound": "hbm", "mt_tests_per_ray": round(c[1] / R, 2),
                     "candidate_bins_per_triangle": round(c[0] / n_faces, 3)}
        else:
            # k_trace4 walks an L2-resident tree: its HBM traffic is ~12 MB per launch against ~0.5 GB of algorithmic
            # bytes, so "hbm" only says which peak the contract figure is priced against -- what bounds the kernel is
            # the dependent chain of node fetches (`latency` below)
            alg = c[0] * LB_B_NODE + c[1] * LB_B_TRI + R * LB_B_RAY
            extra = {"kernel": "k_trace4", "bound": "hbm", "nodes_per_ray": round(c[0] / R, 2),
                     "tris_per_ray": round(c[1] / R, 2)}
        alg_scan = alg
        alg = alg * spl
        ach = alg / (serial_ms * 1e-3) / 1e9
        traffic, traffic_src, te = measured_traffic(strategy, spl)
        d = {"bound": extra.pop("bound"), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
             "avg_kernel_ms": round(serial_ms, 5),
             "algorithmic_bytes_per_launch": int(alg), "scans_per_launch": spl,
             "algorithmic_bytes_per_scan": int(alg_scan),
             "probe": "HIP events on the launch stream around the dominant kernel, launches of the timed region's shape "
                      "issued back to back on ONE stream right after the timed region (exclusive durations: nothing "
                      "runs beside the kernel; two overlapped batches of the timed region fill each other's tails, so "
                      "launches x avg_kernel_ms may exceed ms_per_step); reproduced under rocprofv3 by `bench.py "
                      "--probe-only` -> profiles/rNN/serial_probe_kernel_stats.csv"}
        if traffic:
            d["traffic_frac_of_peak"] = round(traffic / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            d["traffic_calibration"] = {"hbm_bytes": f"{FETCH_SIZE_FACTOR:g} x FETCH_SIZE + WRITE_SIZE", "factor": FETCH_SIZE_FACTOR,
                                        "source": "profiles/r05/fetch_calib.txt (tools/fetch_calib.hip: 12-byte coalesced triples, "
                                                  "12-byte windowed and random gathers -- this kernel's patterns -- all 2.0)"}
        if strategy == "scatter":
            # What ANY implementation must pull from HBM per scan: the index triples, every vertex once, one 8-byte atomic per
            # accepted hit.  The 16-byte grid entries of the Moller-Trumbore tests (SURVEY.md section 8d charges them as
            # "36 x n_tris") are reads of a ~2 MB bin grid that stays in L2: they are work, not HBM bytes.  `frac` is priced on
            # the compulsory bytes; the figure of rounds 1-4 (all algorithmic bytes against the HBM peak) stays beside it.
            comp_scan = n_faces * SC_B_TRI + n_verts * SC_B_VERT + c[2] * SC_B_HIT
            l2_scan = c[1] * SC_B_TEST
            comp = comp_scan * spl
            d["frac_incl_l2_bytes"] = d["frac"]
            d["achieved_incl_l2_bytes"] = d["achieved"]
            d["achieved"] = round(comp / (serial_ms * 1e-3) / 1e9, 1)
            d["frac"] = round(comp / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            d["bytes_per_scan"] = {"hbm_compulsory": int(comp_scan), "l2_resident": int(l2_scan),
                                   "hbm_compulsory_parts": {"faces_12B": int(n_faces * SC_B_TRI), "vertices_12B": int(n_verts * SC_B_VERT),
                                                            "hit_atomics_8B": int(c[2] * SC_B_HIT)}}
            d["hbm_compulsory_bytes_per_launch"] = int(comp)
            if traffic:
                d["traffic_over_compulsory"] = round(traffic / comp, 3)
            d["frac_of_achievable_copy"] = round(comp / (serial_ms * 1e-3) / 1e9 / 6300.0, 4)
            d["bound_note"] = ("hbm names the peak the contract prices against; at 8 waves/SIMD the kernel is LATENCY-bound: 0.16-0.17 "
                               "of HBM on compulsory bytes, ~0.4 of vector issue, ~0.4 of the random-request / atomic ceilings "
                               "(valu_issue, memory_side below) -- no single roof is near")
        valu = None
        if te and te.get("valu_active_quad_cycles_per_launch"):
            # the issue-bound view (same PMC passes): cycles a SIMD's vector ALU was busy = SQ_ACTIVE_INST_VALU (summed over
            # the chip's 1024 SIMDs) x SQ_CYCLES_PER_COUNT / 1024, against the launch's duration at the peak clock
            busy = te["valu_active_quad_cycles_per_launch"] * SQ_CYCLES_PER_COUNT / N_SIMD
            valu = {"wave_insts_per_launch": te["valu_wave_insts_per_launch"],
                    "busy_cycles_per_simd": int(busy), "busy_ms_at_peak_clock": round(busy / SHADER_GHZ / 1e6, 5),
                    "frac_of_kernel_time": round(busy / SHADER_GHZ / 1e6 / serial_ms, 4),
                    "note": "SQ_INSTS_VALU of the same launches (profiles pmc.json) x "
                            f"{SQ_CYCLES_PER_COUNT:g} cycles per wave64 instruction (tools/valu_calib.hip, "
                            "profiles/r03/valu_calib.txt: SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU, it is a count) / 1024 "
                            f"SIMDs at {SHADER_GHZ} GHz (the chip sustains 2.0-2.2 under load): the share of the kernel's "
                            "time a SIMD needs to ISSUE its vector work at the peak rate"}
            d["valu_issue"] = valu
        if strategy == "scatter":
            # the REQUEST view of the memory side (last block): TCC_EA0 counters of the same launches
            # (profiles/rNN/ea_requests.json, keyed like pmc.json) against the two ceilings the probes measured
            ea = measured_ea("k_sc_tris", spl)
            if ea:
                rd, at = ea["tcc_ea0_rdreq_per_launch"], ea["tcc_ea0_atomic_per_launch"]
                sec = serial_ms * 1e-3
                d["memory_side"] = {
                    "ea_read_requests_per_launch": rd, "ea_atomics_per_launch": at,
                    "read_requests_G_per_s": round(rd / sec / 1e9, 2), "atomics_G_per_s": round(at / sec / 1e9, 2),
                    "random_read_ceiling_G_per_s": RANDOM_REQ_CEILING_G, "atomic_ceiling_G_per_s": ATOMIC_CEILING_G,
                    "frac_of_random_read_ceiling": round(rd / sec / 1e9 / RANDOM_REQ_CEILING_G, 4),
                    "frac_of_atomic_ceiling": round(at / sec / 1e9 / ATOMIC_CEILING_G, 4),
                    "note": "every device-scope atomic is executed at the memory side (private L2s per XCD): one per accepted "
                            "hit; ceilings: tools/tlb_probe.hip (52 G random 64-byte requests/s with the chip full of "
                            "chains, any parallelism per lane), tools/atomic_probe.hip (27 G non-returning atomicMin/s on "
                            "an image-sized region) -> profiles/r03/tlb_probe.txt, atomic_probe.txt",
                    "source": ea["_path"]}
        if strategy == "lbvh":
            # The contract figure above prices the algorithmic bytes against HBM, but the tree is L2-resident (counter
            # traffic ~ 1/40 of the algorithmic bytes): the HBM view is kept as a sub-record and the block's bound / frac
            # name what the kernel is really up against -- vector issue when the counters are at hand, else L2 bandwidth.
            d["hbm"] = {"algorithmic_frac_of_hbm_peak": d["frac"], "traffic": traffic,
                        "traffic_achieved_GBs": round(traffic / (serial_ms * 1e-3) / 1e9, 1) if traffic else None,
                        "traffic_frac_of_peak": d.get("traffic_frac_of_peak"),
                        "note": "algorithmic bytes are served by L1 / L2; the HBM counters see only the cold misses"}
            d["l2"] = {"achieved": round(ach, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(ach / L2_PEAK_GBS, 4)}
            if valu:
                d.update(bound="valu", achieved=round(valu["wave_insts_per_launch"] / (serial_ms * 1e-3) / 1e9, 2),
                         peak=round(N_SIMD * SHADER_GHZ / VALU_CYCLES_PER_WAVE_INST, 1), unit="G wave-instructions/s",
                         frac=valu["frac_of_kernel_time"])
            else:
                d.update(bound="l2", achieved=d["l2"]["achieved"], peak=L2_PEAK_GBS, frac=d["l2"]["frac"])
        if insitu_ms == insitu_ms:
            base = d.get("hbm_compulsory_bytes_per_launch", alg)   # (scatter: the compulsory bytes, like `frac`)
            d["in_situ"] = {"avg_kernel_ms": round(insitu_ms, 5),
                            "achieved": round(base / (insitu_ms * 1e-3) / 1e9, 1),
                            "frac": round(base / (insitu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "note": f"the same launches inside the timed region (every {PROBE_EVERY}th sampled), where "
                                    "several launches overlap: the duration is NOT exclusive"}
        if strategy == "lbvh":
            # latency roofline of the traversal: a quad's visits are a dependent chain, one L2 round trip each
            visits = c[0] / R + c[1] / R / 4.0  # node steps + leaf steps (a leaf step tests up to 4 triangles)
            l2_ns = 200 / 2.4  # ~200 cycles L2 hit (MI355X_MICROARCH.md) at 2.4 GHz
            floor_ms = visits * l2_ns * 1e-6
            d["latency"] = {"dependent_steps_per_ray": round(visits, 1), "l2_hit_ns": round(l2_ns, 1),
                            "chain_floor_ms": round(floor_ms, 5), "frac": round(floor_ms / serial_ms, 4),
                            "note": "dependent node / leaf fetches x L2 hit latency = the shortest a ray's walk can "
                                    "be; with enough rays resident the launch could approach it"}
        d.update(extra)
        return d

    def serial_probe_ms(strategy, n=24):
        """Launches of the timed region's shape (scatter: one lt_scene_render_batch_dev of --batch scans; lbvh: build +
        trace of one scan), back to back on ONE stream, events around the dominant kernel: exclusive durations."""
        import ctypes as C
        from lidar_transfer_amd import _lib
        lib = _lib.load()
        vp = C.c_void_p
        evs = []
        B = args.batch if strategy == "scatter" else 1
        st = streams[0]
        org_b = (C.c_float * (3 * B))(*(list(origin) * B))
        with torch.cuda.stream(st):
            for i in range(n + 4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if strategy == "lbvh":
                    w = workers[0]
                    w.set_mesh(*scenes[i % len(scenes)])
                    w.build()
                    w.set_probe(e0, e1)
                    w.trace(rays, origin, H, out=scratch[0])
                else:
                    for j in range(B):
                        workers[j].set_mesh(*scenes[(i * B + j) % len(scenes)])
                    workers[0].set_probe(e0, e1)
                    arr = lambda vals: (vp * B)(*vals)  # noqa: E731
                    outs = {k: arr([scratch[j][k].data_ptr() for j in range(B)]) for k in
                            ("endpoints", "endcolors", "range", "endrem", "tri")}
                    _lib.check(lib.lt_scene_render_batch_dev(B, arr([workers[j]._h for j in range(B)]),
                                                             arr([raysets[j]._h for j in range(B)]), org_b,
                                                             outs["endpoints"], outs["endcolors"], outs["range"],
                                                             outs["endrem"], outs["tri"],
                                                             _lib.LT_TRACE_WRITE_MISSES, vp(st.cuda_stream)),
                               "serial probe")
                evs.append((e0, e1))
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs[4:]]))

    def isolated_kernel_ms(strategy, n=24):
        """Outside the clock: the dominant kernel alone on an otherwise idle GPU, ONE scan per launch."""
        # back to back on ONE stream (launches of one stream do not overlap), one synchronisation at the end: a
        # host round trip between launches lets the GPU drop its clocks and measures that instead
        w, evs = workers[0], []
        with torch.cuda.stream(streams[0]):
            for i in range(n + 8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w.set_mesh(*scenes[i % len(scenes)])
                if strategy == "lbvh":
                    w.build()
                    w.set_probe(e0, e1)
                    w.trace(rays, origin, H, out=scratch[0])
                else:
                    w.set_probe(e0, e1)
                    w.render(raysets[0], origin, out=scratch[0])
                evs.append((e0, e1))
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs[8:]]))

    def e2e_host_call(n_calls=10):
        """The PCIe-inclusive clock (SURVEY.md section 8d "end-to-end"): the drop-in call exactly as the reference's
        throw_rays_at_mesh issues it (fusion_lidar.py:434-451) -- mesh, rays and pre-zeroed images in pageable HOST
        numpy arrays, C_Trace uploads, renders, downloads; one scan per call, nothing overlapped."""
        from lidar_transfer_amd.raytracer import C_Trace
        v, f, c, r = [np.ascontiguousarray(x.cpu().numpy()).reshape(-1) for x in scenes[0]]
        hr = np.ascontiguousarray(rays.cpu().numpy()).reshape(-1)
        org = np.asarray(origin, np.float32)
        ts = []
        for i in range(n_calls + 2):
            ep = np.zeros(3 * R, np.float32); ec = np.zeros(3 * R, np.int32)
            rg = np.zeros(R, np.float32); rm = np.zeros(R, np.float32)
            t = time.perf_counter()
            C_Trace(hr, org, v, f, c, r, ep, ec, rg, rm, H, W)
            ts.append(time.perf_counter() - t)
        t = float(np.median(ts[2:]))
        h2d = v.nbytes + f.nbytes + c.nbytes + r.nbytes + hr.nbytes + ep.nbytes + ec.nbytes + rg.nbytes + rm.nbytes
        d2h = ep.nbytes + ec.nbytes + rg.nbytes + rm.nbytes
        return {"what": "lt_ctrace drop-in call: host mesh + rays + pre-zeroed images in, images out, one scan per call, "
                        "pageable memory, no overlap (fusion_lidar.py:434-451)",
                "ms_per_scan": round(t * 1e3, 4), "value": round(R / t / 1e6, 2), "unit": "Mrays/s",
                "scans_per_s": round(1.0 / t, 1), "h2d_bytes": int(h2d), "d2h_bytes": int(d2h),
                "pcie": {"bound": "pcie gen5 x16", "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                         "achieved": round(max(h2d, d2h) / t / 1e9, 2),
                         "frac": round(max(h2d, d2h) / t / 1e9 / PCIE_PEAK_GBS, 4), "measured_wire_GBs": PCIE_WIRE_GBS,
                         "note": "the larger direction's bytes / call time (the link is full duplex)"},
                "hits": int((rg > 0).sum())}

    def chain_pmc(kernel):
        """PMC record of a fusion-chain kernel from profiles/rNN/pmc_chain.json (tools/pmc_chain_to_json.py: FETCH_SIZE /
        WRITE_SIZE / SQ passes + the kernel trace of tools/prof_chain.py on the default volume); only when the kernel
        sources are the ones it was collected on -- otherwise None."""
        import glob
        import hashlib
        h = hashlib.sha256()
        for name in ("lt_tsdf.hip", "lt_mc.hip", "lt_internal.h"):
            with open(os.path.join(ROOT, "lidar_transfer_amd", "csrc", name), "rb") as fh:
                h.update(name.encode() + fh.read())
        want = h.hexdigest()[:16]
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_chain*.json")), reverse=True):
            try:
                doc = json.load(open(path))
            except (OSError, ValueError):
                continue
            for e in doc.get("entries", []):
                if e.get("kernel") == kernel and e.get("kernel_source_hash") == want:
                    e = dict(e)
                    e["source"] = os.path.relpath(path, ROOT)
                    return e
        return None

    def chain_roofline(kernels, compulsory, phase_ms, what):
        """Roofline record of one phase of the fusion chain: compulsory bytes (what ANY implementation must move: the voxels
        / mesh elements written, the fields read for them, the images) against the HBM peak over the phase's measured time,
        plus -- from the committed PMC passes -- counter traffic and the vector-issue share of the phase's dominant kernel."""
        d = {"kernels": kernels, "bound": "valu", "compulsory_bytes": int(compulsory), "what_is_counted": what,
             "phase_ms": round(phase_ms, 4),
             "hbm": {"achieved": round(compulsory / (phase_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(compulsory / (phase_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        e = chain_pmc(kernels[0])
        if e:
            busy = e["valu_active_quad_cycles_per_launch"] * SQ_CYCLES_PER_COUNT / N_SIMD
            kms = e["avg_kernel_ns"] * 1e-6
            d.update(achieved=round(e["valu_wave_insts_per_launch"] / (kms * 1e-3) / 1e9, 2),
                     peak=round(N_SIMD * SHADER_GHZ / VALU_CYCLES_PER_WAVE_INST, 1), unit="G wave-instructions/s",
                     frac=round(busy / SHADER_GHZ / 1e6 / kms, 4), traffic=e["hbm_bytes_per_launch"],
                     dominant_kernel={"kernel": kernels[0], "avg_kernel_ms": round(kms, 5),
                                      "valu_wave_insts_per_launch": e["valu_wave_insts_per_launch"],
                                      "traffic_over_compulsory": round(e["hbm_bytes_per_launch"] / max(compulsory, 1), 2),
                                      "source": e["source"]})
        else:
            d.update(achieved=None, peak=None, unit="G wave-instructions/s", frac=None, traffic=None,
                     note="no PMC record for the current kernel sources under profiles/ (tools/r03_profile.sh)")
        return d

    def fusion_chain(n=6, nscans=1):
        """Upstream + hot path without the mesh ever leaving HBM (SURVEY.md section 8f-1/2 + 8a): per output scan
        reset the TSDF volume, integrate `nscans` observations (fusion_lidar.py:252-287; the reference's `mesh` adaption
        fuses `number_of_scans` range images, all re-projected into the primary pose, into ONE volume --
        laserscan.py:874-903), marching cubes on the device (:403-424), render the target sensor's image from the mesh
        where it was written.  Volume = the reference's default voxel_bounds at 5 cm (config/lidar_transfer.yaml:
        2000 x 2000 x 200 voxels, 4 x 3.2 GB)."""
        import ctypes as C
        from lidar_transfer_amd import _lib
        from lidar_transfer_amd.fusion import DeviceMesh, TSDFVolume
        lib = _lib.load()
        if torch.cuda.get_device_properties(dev).total_memory < 60 * 2**30:
            return None
        vp = C.c_void_p
        w = workers[0]
        w.set_mesh(*scenes[0])
        o = w.render(raysets[0], origin)   # the observation: this very sensor looking at scene 0
        torch.cuda.synchronize()
        lab = o["endcolors"][:, 2].reshape(H, W).float().contiguous()
        folded0 = (lab * 65536.0).contiguous()   # label in channel 0 (laserscan.py:893-895), folded as fusion_lidar.py:261-264
        depth0 = o["range"].reshape(H, W).contiguous()
        remi = o["endrem"].reshape(H, W).contiguous()
        # observations 1 .. nscans - 1: the neighbouring scans of the reference are re-projected into the primary pose
        # (laserscan.py:876-879), i.e. nearly the same range image with centimetre noise and holes where the other pose
        # did not see the surface; the labels occasionally differ (the class-aware branch's "other class" path)
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234)
        obs = [(folded0, depth0, remi)]
        for k in range(1, nscans):
            noise = (torch.rand((H, W), device=dev, generator=gen) - 0.5) * 0.04
            hole = torch.rand((H, W), device=dev, generator=gen) < 0.05
            d_k = torch.where(hole | (depth0 == 0), torch.zeros_like(depth0), depth0 + noise).contiguous()
            flip = torch.rand((H, W), device=dev, generator=gen) < 0.02
            f_k = torch.where(flip, torch.full_like(folded0, 50.0 * 65536.0), folded0).contiguous()
            obs.append((f_k, d_k, remi))
        vol = TSDFVolume(np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]]), 0.05, wl["fov_up"], wl["fov_down"])
        mesh = DeviceMesh(local_rank)
        st = torch.cuda.current_stream(dev)
        sp = vp(st.cuda_stream)
        org = (C.c_float * 3)(*origin)
        out = scratch[0]
        obs_c = (vp * len(obs))(*[o_[0].data_ptr() for o_ in obs])
        obs_d = (vp * len(obs))(*[o_[1].data_ptr() for o_ in obs])
        obs_r = (vp * len(obs))(*[o_[2].data_ptr() for o_ in obs])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ms = np.zeros((n, 4))
        t_wall = []
        for i in range(n + 1):
            t0 = time.perf_counter()
            ev[0].record()
            _lib.check(lib.lt_tsdf_reset(vol._h, sp), "reset")
            ev[1].record()
            # all observations of the fresh volume in ONE call (lt_tsdf_integrate_multi_dev: one fused pass, bit-identical to
            # one lt_tsdf_integrate_dev per observation -- tests/test_tsdf_gpu.py)
            _lib.check(lib.lt_tsdf_integrate_multi_dev(vol._h, len(obs), obs_c, obs_d, obs_r, H, W, 1.0, _lib.LT_TSDF_MERGE, sp),
                       "integrate")
            ev[2].record()
            _lib.check(lib.lt_tsdf_extract_mesh_dev(vol._h, mesh._h, sp, None), "marching cubes")
            ev[3].record()
            _lib.check(lib.lt_scene_set_mesh(w._h, mesh._h), "set mesh")
            _lib.check(lib.lt_scene_render_dev(w._h, raysets[0]._h, org, out["endpoints"].data_ptr(),
                                               out["endcolors"].data_ptr(), out["range"].data_ptr(),
                                               out["endrem"].data_ptr(), out["tri"].data_ptr(),
                                               _lib.LT_TRACE_WRITE_MISSES, sp, None), "render")
            ev[4].record()
            torch.cuda.synchronize()
            if i > 0:
                t_wall.append(time.perf_counter() - t0)
                ms[i - 1] = [ev[k].elapsed_time(ev[k + 1]) for k in range(4)]
        hits_c = int((out["range"] > 0).sum().item())
        nv, nf = mesh.n_verts, mesh.n_faces
        t = float(np.median(t_wall))
        m = np.median(ms, axis=0)
        nvox = int(np.prod(vol._vol_dim))
        # voxels the fusion wrote (outside the clock): tsdf left its initial 1 or the weight its initial 0
        tv, wv, _, _ = vol.get_volume_tensors()
        n_written = 0
        for x0 in range(0, tv.shape[0], 250):   # (in slabs: the masks of the whole volume would be 1.6 GB)
            n_written += int(((tv[x0:x0 + 250] != 1) | (wv[x0:x0 + 250] != 0)).sum().item())
        # how many of this very volume's active cells are one of Lewiner's AMBIGUOUS cases (3, 4, 6, 7, 10, 12, 13: the cell's
        # eight values, not its signs, pick the tiling -- lt_mc.hip, lw_select); outside the clock.  The device's case index
        # -> Lewiner's case: LT_LWC_CASE of the generated table header.
        mc_cases = None
        try:
            import re
            hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lidar_transfer_amd", "csrc",
                                    "lt_mc_lewiner_table.h")).read()
            lw_case = np.array([int(x) for x in re.search(r"LT_LWC_CASE\[256\] = \{([^}]*)\}", hdr).group(1).split(",")])
            hist = np.zeros(256, np.int64)
            X = tv.shape[0]
            for x0 in range(0, X - 1, 100):
                ins = ~(tv[x0:min(x0 + 101, X)] > 0)     # the sign bit: NOT above the level
                if not bool(ins.any()):
                    continue
                sx, sy, sz = ins.shape
                idx = torch.zeros((sx - 1, sy - 1, sz - 1), dtype=torch.int32, device=dev)
                for c_ in range(8):
                    dx, dy, dz = c_ & 1, (c_ >> 1) & 1, (c_ >> 2) & 1
                    idx += ins[dx:sx - 1 + dx, dy:sy - 1 + dy, dz:sz - 1 + dz].to(torch.int32) << c_
                hist += torch.bincount(idx.reshape(-1), minlength=256).cpu().numpy()
                del ins, idx
            act = int(hist[1:255].sum())
            amb = int(sum(int(hist[c_]) for c_ in range(1, 255) if lw_case[c_] in (3, 4, 6, 7, 10, 12, 13)))
            mc_cases = {"active_cells": act, "ambiguous_cells": amb, "ambiguous_share": round(amb / max(act, 1), 5),
                        "by_lewiner_case": {str(k): int(hist[lw_case == k].sum()) for k in range(1, 15)},
                        "note": "the mesh is scikit-image 0.18.3's (Lewiner): vertices and face stream equal to the reference's "
                                "get_mesh on golden F10, render bit-identical (tests/test_pin_f10_f11_gpu.py)"}
        except Exception as e:  # noqa: BLE001
            mc_cases = {"error": repr(e)[:200]}
        del tv, wv
        mesh.close()
        vol.close()
        # compulsory bytes: integrate -- every written voxel's four fields out (and in again for the observations after the
        # first), plus the three images per observation; marching cubes -- the mesh out (verts 12 + colors 12 + rem 4 B per
        # vertex, 12 B per face), two tsdf samples + colour + remission in per vertex, one sign bit per voxel of the written
        # columns (~ the written voxels' words, 1/8 B each -- negligible)
        comp_int = n_written * 16 * (2 * nscans - 1) + nscans * 3 * R * 4
        comp_mc = nv * (28 + 16) + nf * 12
        rec = {"what": f"per output scan: reset 2000x2000x200 TSDF volume -> integrate {nscans} 64x2048 observation"
                       f"{'s' if nscans > 1 else ''} -> marching "
                       "cubes on the device -> render the target image; the mesh never leaves HBM (no PCIe between "
                       "fusion and range image)",
               "observations": nscans,
               "parity": {"integrate": "all four volumes bit-identical to the reference's own CUDA kernel source compiled by hipcc "
                                       "for gfx950 and run beside it (tests/test_tsdf_ref_kernel_gpu.py, tests/stress_tsdf_ref.py)",
                          "marching_cubes": "the arrays of the reference's get_mesh run with the real scikit-image 0.18.3 (goldens F10 "
                                            "/ F10b; tests/test_pin_f10_f11_gpu.py, tests/stress_mc.py, full size: "
                                            "tests/stress_mc_full.py)",
                          "render": "bit-identical to the reference raytracer's image of that mesh (F10; full size: 131 066 of "
                                    "131 072 pixels, profiles/r04/mc_full_size.txt)"},
               "ms_per_scan": round(t * 1e3, 3), "scans_per_s": round(1.0 / t, 1), "value": round(R / t / 1e6, 2),
               "unit": "Mrays/s", "voxels": nvox, "voxels_written": n_written, "mesh_verts": nv, "mesh_faces": nf,
               "hit_fraction": round(hits_c / R, 4), "marching_cubes_cases": mc_cases,
               "phase_ms": {"reset": round(float(m[0]), 3), "integrate": round(float(m[1]), 3),
                            "marching_cubes": round(float(m[2]), 3), "render": round(float(m[3]), 3)},
               "roofline": {
                   "integrate": chain_roofline(["k_tsdf_integrate_pix", "k_tsdf_integrate_written", "k_tsdf_integrate_quirk",
                                                "k_tsdf_dct"], comp_int,
                                               float(m[1]), "written voxels x 16 B out (+ in again after the first "
                                               "observation) + 3 images per observation"),
                   "marching_cubes": chain_roofline(["k_mc_emit_batch", "k_mc_words", "k_mc_amb", "k_mc_compact", "k_mc_clear",
                                                     "k_mc_scan1", "k_mc_scan2"], comp_mc, float(m[2]),
                                                    "mesh out (28 B per vertex, 12 B per face) + 16 B of field samples "
                                                    "in per vertex")}}
        return rec

    def deform_from_points(nscans=5, n=8):
        """The reference's REAL loop body from point clouds, composed and timed (laserscan.py:863-918 + :1121-1178): per output
        scan `nscans` source clouds (~120 k points each, float64 as after apply_pose) -> do_range_projection_new +
        do_label_projection_new per cloud (ONE lt_range_projection_batch_dev call) -> fresh 2000x2000x200 volume, integrate
        x nscans -> marching cubes -> ray cast of the target sensor -> write(): filter + pack the .bin / .label bytes
        (lt_pack_scan_dev).  Nothing leaves HBM but the mesh sizes and the number of packed points.  `verified`: the source
        images equal the single-cloud call's (lt_range_projection_dev, pinned to the reference's goldens), and the target
        images + packed bytes equal the step-by-step API run from those images."""
        import ctypes as C
        from lidar_transfer_amd import _lib
        from lidar_transfer_amd.deform import DeviceDeform
        if torch.cuda.get_device_properties(dev).total_memory < 60 * 2**30 or args.target:
            return None
        lib = _lib.load()
        vp = C.c_void_p
        w = workers[0]
        w.set_mesh(*scenes[0])
        o = w.render(raysets[0], origin)
        torch.cuda.synchronize()
        hit = o["tri"] >= 0
        p0 = o["endpoints"][hit].double()
        l0 = o["endcolors"][hit][:, 2].contiguous().to(torch.int32)
        r0 = o["endrem"][hit].contiguous()
        gen = torch.Generator(device=dev)
        gen.manual_seed(4321)
        clouds = []
        for k in range(nscans):   # the neighbouring scans, re-projected into the primary pose: the same surfaces, centimetre noise, holes
            if k == 0:
                clouds.append((p0.contiguous(), r0, l0))
                continue
            keep = torch.rand(p0.shape[0], device=dev, generator=gen) > 0.05
            scale = 1.0 + (torch.rand((int(keep.sum().item()), 1), device=dev, generator=gen, dtype=torch.float64) - 0.5) * 0.001
            lk = l0[keep].clone()
            flip = torch.rand(lk.shape[0], device=dev, generator=gen) < 0.02
            lk[flip] = 50
            clouds.append(((p0[keep] * scale).contiguous(), r0[keep].contiguous(), lk.contiguous()))
        n_pts = [int(c[0].shape[0]) for c in clouds]
        bnds = np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]])
        dd = DeviceDeform((H, W, wl["fov_up"], wl["fov_down"]), (H, W, wl["fov_up"], wl["fov_down"]), bnds, 0.05,
                          device=local_rank)
        st = torch.cuda.current_stream(dev)
        sp = vp(st.cuda_stream)
        org = (C.c_float * 3)(*origin)
        FLG = _lib.LT_TRACE_WRITE_MISSES | _lib.LT_TRACE_LABEL_IMAGE
        out = dd.scene.alloc_outputs(R, label_image=True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        ms = np.zeros((n, 6))
        t_wall = []
        packed = None
        src_keep = None  # (the source images are allocated once: a production caller keeps its buffers too)
        for i in range(n + 2):
            t0 = time.perf_counter()
            ev[0].record()
            src = dd.projector.project(clouds, dd.fov_up, dd.fov_down, H, W, new=True, remove=True, out=src_keep,
                                       outputs=("range", "rem", "label_folded"), stream=st)
            src_keep = src
            ev[1].record()
            _lib.check(lib.lt_tsdf_reset(dd.vol._h, sp), "reset")
            ev[2].record()
            oc = (vp * nscans)(*[s_["label_folded"].data_ptr() for s_ in src])
            od = (vp * nscans)(*[s_["range"].data_ptr() for s_ in src])
            orr = (vp * nscans)(*[s_["rem"].data_ptr() for s_ in src])
            _lib.check(lib.lt_tsdf_integrate_multi_dev(dd.vol._h, nscans, oc, od, orr, H, W, 1.0, _lib.LT_TSDF_MERGE, sp),
                       "integrate")
            ev[3].record()
            _lib.check(lib.lt_tsdf_extract_mesh_dev(dd.vol._h, dd.mesh_obj._h, sp, None), "marching cubes")
            ev[4].record()
            _lib.check(lib.lt_scene_set_mesh(dd.scene._h, dd.mesh_obj._h), "set mesh")
            _lib.check(lib.lt_scene_render_dev(dd.scene._h, dd.rayset._h, org, out["endpoints"].data_ptr(),
                                               out["endcolors"].data_ptr(), out["range"].data_ptr(),
                                               out["endrem"].data_ptr(), out["tri"].data_ptr(), FLG, sp, None), "render")
            ev[5].record()
            packed = dd._pack(out["endpoints"], False, out["endrem"], out["endcolors"], None, R, st)
            ev[6].record()
            torch.cuda.synchronize()
            if i > 1:
                t_wall.append(time.perf_counter() - t0)
                ms[i - 2] = [ev[k].elapsed_time(ev[k + 1]) for k in range(6)]
        # the same chain as ONE DeviceDeform.mesh() call (lt_range_projection_batch_dev -> lt_fusion_scan_dev -> lt_pack_scan_dev)
        t_one = []
        for i in range(n + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = dd.mesh(clouds, origin)
            torch.cuda.synchronize()
            if i:
                t_one.append(time.perf_counter() - t0)
        # ---- verification against the step-by-step API -----------------------------------------------------------------
        ok = True
        from lidar_transfer_amd.laserscan import SemLaserScan
        for k, (pk, rk, lk) in enumerate(clouds):
            s = SemLaserScan(H, W, 300, {})
            s.points, s.remissions, s.label = pk.cpu().numpy(), rk.cpu().numpy(), lk.cpu().numpy().astype(np.uint32)
            s.do_range_projection_new(dd.fov_up, dd.fov_down, remove=True)   # the single-cloud call (host arrays in and out)
            ok = ok and np.array_equal(src[k]["range"].cpu().numpy().view(np.int32), s.range_image.view(np.int32))
            ok = ok and np.array_equal(src[k]["rem"].cpu().numpy().view(np.int32), s.proj_remissions.view(np.int32))
            ok = ok and np.array_equal(src[k]["label_folded"].cpu().numpy(),
                                       np.floor(s.label_image[:, :, 0].astype(np.float32) * 256 * 256))
        same_call = bool(torch.equal(got["range"].reshape(-1).view(torch.int32), out["range"].view(torch.int32))) and \
            bool(torch.equal(got["label"].reshape(-1), out["endcolors"])) and bool(torch.equal(got["bin"], packed[0])) and \
            bool(torch.equal(got["label_file"], packed[1]))
        ok = ok and same_call
        hits_c = int((out["range"] > 0).sum().item())
        nv, nf = dd.mesh_obj.n_verts, dd.mesh_obj.n_faces
        n_packed = int(packed[0].shape[0])
        # ---- projection alone: `reps` batch calls back to back (events on the launch stream) ------------------------------
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        outs_keep = [dict(s_) for s_ in src]
        for _ in range(3):
            dd.projector.project(clouds, dd.fov_up, dd.fov_down, H, W, new=True, remove=True, out=outs_keep,
                                 outputs=("range", "rem", "label_folded"), stream=st)
        e0.record()
        for _ in range(reps):
            dd.projector.project(clouds, dd.fov_up, dd.fov_down, H, W, new=True, remove=True, out=outs_keep,
                                 outputs=("range", "rem", "label_folded"), stream=st)
        e1.record()
        torch.cuda.synchronize()
        proj_ms = e0.elapsed_time(e1) / reps
        # the single-cloud device call for comparison (four kernels + a host synchronisation per cloud)
        kept = C.c_int(0)
        t_single = []
        for rep_ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for (pk, rk, lk), s_ in zip(clouds, outs_keep):
                _lib.check(lib.lt_range_projection_dev(pk.data_ptr(), 1, rk.data_ptr(), lk.data_ptr(), int(pk.shape[0]),
                                                       float(dd.fov_up), float(dd.fov_down), H, W, None, 0,
                                                       _lib.LT_PROJ_NEW | _lib.LT_PROJ_REMOVE, None, 0, None, None, None,
                                                       None, None, None, None, None, None, s_["range"].data_ptr(), None,
                                                       s_["rem"].data_ptr(), None, None, None, 0.0, -1.0, 0.0,
                                                       C.byref(kept), sp), "lt_range_projection_dev")
            torch.cuda.synchronize()
            if rep_ > 1:
                t_single.append(time.perf_counter() - t0)
        filled = sum(int((s_["range"] > 0).sum().item()) for s_ in outs_keep)
        tot_pts = sum(n_pts)
        alg = tot_pts * (24 + 8) + nscans * R * (16 + 12) + filled * (24 + 8)
        dd.close()
        # ... and with three output scans in flight (FusionScanPipeline.submit_clouds: a projector, volume, mesh, scene, stream
        # and host thread per chain; projection + fusion chain per scan, no write())
        pipelined = None
        try:
            import gc
            from lidar_transfer_amd.pipeline import FusionScanPipeline
            if torch.cuda.get_device_properties(dev).total_memory >= 100 * 2**30:
                with FusionScanPipeline(bnds, 0.05, wl["fov_up"], wl["fov_down"], rays, H, chains=3, device=local_rank,
                                        label_image=True, source_hw=(H, W)) as pipe:
                    for tk_ in [pipe.submit_clouds(clouds, inputs_ready=True) for _ in range(12)]:
                        pipe.wait(tk_)
                    bufs = [pipe._chains[0]["scene"].alloc_outputs(R, label_image=True) for _ in range(36)]
                    torch.cuda.synchronize()
                    gc.collect()
                    gc.disable()
                    try:
                        tp0 = time.perf_counter()
                        tks = [pipe.submit_clouds(clouds, out=b_, inputs_ready=True) for b_ in bufs]
                        outs_p = [pipe.wait(tk_) for tk_ in tks]
                        dtp = time.perf_counter() - tp0
                    finally:
                        gc.enable()
                    okp = all(bool(torch.equal(o_["range"].view(torch.int32), out["range"].view(torch.int32))) and
                              bool(torch.equal(o_["endcolors"], out["endcolors"])) for o_ in outs_p)
                    pipelined = {"chains_in_flight": 3, "output_scans": len(bufs), "ms_per_output_scan": round(dtp / len(bufs) * 1e3, 4),
                                 "output_scans_per_s": round(len(bufs) / dtp, 1), "verified": bool(okp),
                                 "api": "lidar_transfer_amd.pipeline.FusionScanPipeline.submit_clouds (projection + fusion "
                                        "chain per scan; no write())"}
        except Exception as e:  # noqa: BLE001
            pipelined = {"error": repr(e)[:200]}
        m = np.median(ms, axis=0)
        t = float(np.median(t_wall))
        return {"what": f"deform('mesh') + write() per output scan from {nscans} float64 point clouds of {n_pts[0]}..{min(n_pts)} points "
                        f"(laserscan.py:863-918, :1121-1178): batched z-min projection -> reset 2000x2000x200 volume -> integrate "
                        f"x{nscans} -> marching cubes -> render {H}x{W} -> pack .bin/.label bytes; all in HBM",
                "observations": nscans, "points_per_scan": n_pts, "ms_per_output_scan": round(t * 1e3, 3),
                "ms_per_output_scan_one_call": round(float(np.median(t_one)) * 1e3, 3),
                "output_scans_per_s": round(1.0 / t, 1),
                "phase_ms": {"projection": round(float(m[0]), 4), "reset": round(float(m[1]), 4),
                             "integrate": round(float(m[2]), 4), "marching_cubes": round(float(m[3]), 4),
                             "render": round(float(m[4]), 4), "pack": round(float(m[5]), 4)},
                "mesh_verts": nv, "mesh_faces": nf, "hit_fraction": round(hits_c / R, 4), "points_written": n_packed,
                "verified": bool(ok), "pipelined": pipelined,
                "projection": {"ms": round(proj_ms, 4), "clouds_per_call": nscans, "us_per_cloud": round(proj_ms * 1e3 / nscans, 2),
                               "Mpoints_per_s": round(tot_pts / proj_ms / 1e3, 1), "dtype": "f64",
                               "algorithmic_bytes_per_call": int(alg),
                               "achieved": round(alg / (proj_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(alg / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "atomics_G_per_s": round(tot_pts / (proj_ms * 1e-3) / 1e9, 2),
                               "atomic_ceiling_G_per_s": ATOMIC_LANE_CEILING_ORDERED_G,
                               "frac_of_atomic_ceiling": round(tot_pts / (proj_ms * 1e-3) / 1e9 / ATOMIC_LANE_CEILING_ORDERED_G, 4),
                               "atomic_ceiling_note": "LANE atomics/s with neighbouring lanes on neighbouring cells (runs >= 8 per 64-byte "
                                                      "line: a scan's firing order), tools/atomic_probe.hip -> profiles/r05/atomic_probe.txt; "
                                                      f"random cells reach {ATOMIC_CEILING_G:g} G/s (one request per lane)",
                               "single_cloud_call_ms_per_cloud": round(float(np.median(t_single)) * 1e3 / nscans, 4),
                               "bytes": "per point 24 B in + one 8-B memory-side atomicMin; per cell 16 B key read + re-arm, 12 B "
                                        "of images out (range, remission, folded label); per filled cell 24 + 8 B gathered",
                               "kernels": ["k_pb_project", "k_pb_resolve"]}}

    def mergemesh_from_points(n=8):
        """`deform('mergemesh')` + `write()` -- the adaption the reference's shipped config selects (config/lidar_transfer.yaml:3,
        `number_of_scans: 1`; laserscan.py:921-1012, :1121-1178) -- from ONE 120 k-point source cloud at the reference's default
        volume parameters (voxel_bounds +-50 / +-50 / +-5 m given as the YAML's ints, voxel 0.05 m): target-FOV projection onto
        the source image, the kept points' bounds read back (48 bytes), `vol_bnds` clipped in place, a volume of that geometry,
        one class-aware integrate, marching cubes, ray cast, pack.  Parity of the chain: goldens F14 / F14b (pytest -m gpu);
        here: wall clock per output scan, and that a second DeviceDeform gives the same bytes."""
        from lidar_transfer_amd.deform import DeviceDeform
        if torch.cuda.get_device_properties(dev).total_memory < 60 * 2**30 or args.target:
            return None
        w = workers[0]
        w.set_mesh(*scenes[0])
        o = w.render(raysets[0], origin)
        torch.cuda.synchronize()
        hit = o["tri"] >= 0
        cloud = [(o["endpoints"][hit].double().contiguous(), o["endrem"][hit].contiguous(),
                  o["endcolors"][hit][:, 2].contiguous().to(torch.int32))]
        sensor = (H, W, wl["fov_up"], wl["fov_down"])
        res = []
        for rep in range(2):
            bnds = np.array([-50, 50, -50, 50, -5, 5]).reshape(3, 2)
            dd = DeviceDeform(sensor, sensor, bnds, 0.05, device=local_rank, mesh_volume=False)
            for _ in range(3):
                got = dd.mergemesh(cloud)
            torch.cuda.synchronize()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                got = dd.mergemesh(cloud)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res.append((float(np.median(ts)), got["bin"].clone(), got["label_file"].clone(), got["range"].clone(), got["vol_dim"],
                        bnds.tolist(), got["n_faces"]))
            dd.close()
        same = bool(torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and
                    torch.equal(res[0][3].view(torch.int32), res[1][3].view(torch.int32)))
        return {"what": "DeviceDeform.mergemesh: one 120 k-point cloud -> projection (target FOV) -> bounds read-back -> volume of the "
                        "clipped geometry -> integrate -> marching cubes -> ray cast -> packed .bin / .label bytes",
                "ms_per_output_scan": round(min(r_[0] for r_ in res) * 1e3, 4), "points_in": int(cloud[0][0].shape[0]),
                "vol_dim": list(res[0][4]), "vol_bnds_after": res[0][5], "mesh_faces": int(res[0][6]),
                "points_written": int(res[0][1].shape[0]), "hit_fraction": round(float((res[0][3] > 0).float().mean().item()), 4),
                "verified": same, "parity": "goldens F14 / F14b (tests/test_deform_gpu.py): the reference's own deform('mergemesh') + write()"}

    def e2e_pipelined(n_scans=200, depth=4):
        """The same host-buffer work for a SEQUENCE of scans (the reference's loop over output scans): lt_hostpipe keeps
        `depth` scans in flight -- uploads of scans i+1, i+2 (two uploader threads) | render of scan i | download of scan
        i-1 on separate HIP streams, pageable numpy arrays, colours as the uint8 [V,3] get_mesh returns, all five images downloaded.  Measured by
        tools/hostpipe_rate.py in a numpy-only subprocess (LIDARHIP_NO_TORCH=1: the system ROCm runtime) and -- `in_torch_process`
        -- with torch imported first, as the reference's caller has it (laserscan.py:6).  Both reach the same steady rate;
        the HIP 7.0 runtime bundled with the torch wheel stalls ONCE for 36-54 ms at the 81st scan of a process (round 4's
        "35 % slower" was that stall inside a 200-scan measurement): the tool warms up over 100 scans and reports the stall it
        saw there (profiles/r05/hostpipe_torch.txt)."""
        if args.workload != "C2" or args.target:
            return None
        res = {}
        for name, extra in (("numpy_only", {"LIDARHIP_NO_TORCH": "1"}), ("in_torch_process", {})):
            env = dict(os.environ)
            env.update(extra)
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hostpipe_rate.py"), str(depth), str(n_scans),
                                    "6"], capture_output=True, text=True, timeout=300, env=env)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                res[name] = json.loads(line[-1]) if (r.returncode == 0 and line) else None
            except (subprocess.TimeoutExpired, ValueError):
                res[name] = None
        m = res.get("numpy_only")
        if not m:
            return None
        t = m["ms_per_scan"] * 1e-3
        h2d = m["h2d_bytes"]
        out = {"what": f"lt_hostpipe: {n_scans} scans, {depth} in flight (uploads i+1, i+2 on two threads | render i | download i-1), host "
                       f"meshes in pageable numpy arrays, colours uint8 [V,3] as get_mesh returns them, all five images "
                       f"downloaded; numpy-only process", "ms_per_scan": round(t * 1e3, 4),
               "value": round(R / t / 1e6, 2), "unit": "Mrays/s", "scans_per_s": round(1.0 / t, 1), "h2d_bytes": int(h2d),
               "d2h_bytes": int(m["d2h_bytes"]),
               "pcie": {"bound": "pcie gen5 x16, one direction", "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                        "achieved": round(h2d / t / 1e9, 2), "frac": round(h2d / t / 1e9 / PCIE_PEAK_GBS, 4),
                        "measured_wire_GBs": PCIE_WIRE_GBS, "frac_of_wire": round(h2d / t / 1e9 / PCIE_WIRE_GBS, 4),
                        "note": "upload bytes per scan / time per scan; the link is full duplex and the downloads "
                                "run under the uploads"}, "hits": m["hits"],
               "uploader_thread_ms_per_scan": m.get("worker_upload_ms"),
               "single_call_ms_in_this_process": m.get("single_call_ms")}
        if res.get("in_torch_process"):
            it = res["in_torch_process"]
            out["in_torch_process"] = {"ms_per_scan": it["ms_per_scan"], "GBs": it["GBs"],
                                       "frac_of_wire": round(it["GBs"] / PCIE_WIRE_GBS, 4),
                                       "one_time_stall_in_warmup_ms": it.get("warmup_longest_gap_ms"),
                                       "one_time_stall_at_scan": it.get("warmup_longest_gap_at_scan"),
                                       "note": "same loop, torch imported first (its bundled HIP 7.0 runtime): the same steady "
                                               "rate; that runtime stalls once per process (reported here, inside the 100-scan "
                                               "warm-up)"}
            out["numpy_only_warmup_longest_gap_ms"] = m.get("warmup_longest_gap_ms")
        return out

    failures = {}

    def guarded(name, fn, *a, **k):
        """An optional leg of the bench must never cost the headline line: exceptions are reported, not raised."""
        try:
            return fn(*a, **k)
        except BaseException as e:  # noqa: BLE001  (SystemExit from a helper included)
            if isinstance(e, KeyboardInterrupt):
                raise
            failures[name] = repr(e)[:300]
            sys.stderr.write(f"bench.py: optional leg '{name}' failed: {e!r}\n")
            return None

    if args.probe_only:
        # warm-up + the serial probe only: every k_sc_tris<false, *> (or k_trace4) launch of this process is a launch of
        # the timed region's shape, exclusive on its stream -- the population profiles/rNN/serial_probe_kernel_stats.csv holds
        ser_ms = serial_probe_ms(args.strategy, n=64)
        if rank == 0:
            rl = roofline(args.strategy, ser_ms, float("nan"))
            os.write(real_stdout, (json.dumps({"probe_only": True, "strategy": args.strategy, "kernel": rl["kernel"],
                                               "avg_kernel_ms": rl["avg_kernel_ms"], "launches": 64,
                                               "scans_per_launch": rl["scans_per_launch"],
                                               "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"],
                                               "achieved": rl["achieved"], "frac": rl["frac"], "bound": rl["bound"]}) + "\n").encode())
        shared_rays.close()
        for wk in workers:
            wk.close()
        return

    dt, kern_ms, hits, verify = run(args.strategy, K, Wm, keep=True)
    ser_ms = guarded("serial_probe", serial_probe_ms, args.strategy)
    if ser_ms is None:
        ser_ms = float("nan")
    other = None
    if not args.no_other:
        def other_leg():
            oname = "lbvh" if args.strategy == "scatter" else "scatter"
            Ko = max(20, K // 16) if oname == "lbvh" else max(K, 400)  # a scatter scan is ~15x shorter than an LBVH scan
            odt, okern, _, _ = run(oname, Ko, max(4, Wm // 16), keep=False)
            return {"strategy": oname, "value": round(world * Ko * R / odt / 1e6, 3), "unit": "Mrays/s",
                    "ms_per_scan": round(odt / Ko * 1e3, 4), "scans": Ko,
                    "roofline": roofline(oname, serial_probe_ms(oname, n=12), okern)}
        other = guarded("other_strategy", other_leg)

    iso_ms = guarded("isolated_kernel", isolated_kernel_ms, args.strategy)

    def one_batch_in_flight():
        """The timed region once more with ONE batch in flight (one stream, launches strictly one after the other), every
        k_sc_tris launch bracketed by HIP events: here the kernel durations are exclusive AND inside a wall-clocked region,
        so launches x kernel time must FIT in the region's time -- the check the default region (two overlapped batches,
        which fill each other's tails) cannot offer."""
        Ko = max(args.batch * 8, min(K, 1024))
        odt, okern, _, _ = run("scatter", Ko, max(args.batch * 2, min(Wm, 64)), keep=False, groups=1, probe_every=1)
        n_launch = (Ko + args.batch - 1) // args.batch
        return {"batches_in_flight": 1, "scans": Ko, "value": round(world * Ko * R / odt / 1e6, 3), "unit": "Mrays/s",
                "region_ms": round(odt * 1e3, 4), "k_sc_tris_launches": n_launch,
                "k_sc_tris_avg_ms": round(okern, 5), "launches_x_kernel_ms": round(n_launch * okern, 4),
                "kernel_share_of_region": round(n_launch * okern / (odt * 1e3), 4),
                "fits_in_region": bool(n_launch * okern <= odt * 1e3),
                "note": "one stream, one batch of scans in flight: k_sc_tris -> k_sc_rest -> k_sc_resolve strictly in turn; "
                        "HIP events around EVERY k_sc_tris launch of the region"}

    one_batch = guarded("one_batch_in_flight", one_batch_in_flight) if (args.strategy == "scatter" and args.batch > 1) else None
    # the PCIe-inclusive clocks and the fusion chain are single-GPU records (like cpu_baseline): rank 0 at N = 1 only
    e2e = guarded("e2e_single_call", e2e_host_call) if (rank == 0 and world == 1 and not args.no_e2e) else None
    if e2e:
        e2e = {"single_call": e2e, "pipelined": guarded("e2e_pipelined", e2e_pipelined)}
    def fusion_chain_pipelined(chains=3):
        """The same chain with `chains` output scans in flight (lidar_transfer_amd.pipeline.FusionScanPipeline: own volume,
        mesh, scene, HIP stream and host thread each -- output scans are independent, lidar_deform.py:393-462): the chain's
        sparse sweeps leave the chip half empty, scans in flight fill each other's gaps.  tools/chain_pipeline.py; one and
        five observations per scan on the same pipeline; every timed scan's images are compared bit for bit with the
        single chain's (`verified`).  Runs BEFORE the single-chain legs: its 38 GB of volumes should be the process's
        first big allocation."""
        if torch.cuda.get_device_properties(dev).total_memory < 100 * 2**30:
            return None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import chain_pipeline
        return chain_pipeline.run_cases(chains, ((12, 1), (12, 5)), local_rank, args.workload)

    pipelined = guarded("fusion_chain_pipelined", fusion_chain_pipelined) if (rank == 0 and not args.no_chain and world == 1) else None
    chain = guarded("fusion_chain", fusion_chain) if (rank == 0 and not args.no_chain and world == 1) else None
    chain5 = guarded("fusion_chain_nscans5", fusion_chain, 4, 5) if (chain and rank == 0 and world == 1) else None

    from_points = guarded("deform_from_points", deform_from_points) if (chain and rank == 0 and world == 1) else None
    mergemesh_leg = guarded("mergemesh_from_points", mergemesh_from_points) if (chain and rank == 0 and world == 1) else None
    if chain:
        chain["pipelined"] = (pipelined or [None, None])[0]
    if chain5:
        chain5["pipelined"] = (pipelined or [None, None])[1]
    if rank == 0:
        value = world * K * R / dt / 1e6
        rl = roofline(args.strategy, ser_ms, kern_ms)
        if iso_ms is not None:
            iso_b = rl["bytes_per_scan"]["hbm_compulsory"] if "bytes_per_scan" in rl else rl["algorithmic_bytes_per_scan"]
            rl["isolated"] = {"avg_kernel_ms": round(iso_ms, 5),
                              "achieved": round(iso_b / (iso_ms * 1e-3) / 1e9, 1),
                              "frac": round(iso_b / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "same kernel, launches of ONE scan each, back to back on one stream (nothing beside "
                                      "them), after the timed region"}
        if args.strategy == "scatter":
            # path_frac: the figure that is bounded by the DRIVER's clock -- algorithmic bytes of all three kernels per step
            # over ms_per_step against the HBM peak.  k_sc_tris: as above; k_sc_resolve: per ray the 8-B z-min cell read and
            # re-armed (16 B) + the five images written (range 4, label 4, remission 4, end point 12, triangle 4 = 28 B), per
            # hit ray the winner's face (12 B), three remissions (12 B) and one label (4 B); k_sc_rest redoes deferred work
            # of k_sc_tris: no algorithmic bytes of its own.
            cs = np.mean(np.array(cnt["scatter"], dtype=np.float64), axis=0)
            hit_rays = float(np.mean(hit_ray_counts)) if hit_ray_counts else float(cs[2])
            b_tris = rl["algorithmic_bytes_per_scan"]
            b_res = R * (16 + 28) + hit_rays * 28
            step_s = dt / args.steps
            rl["path_frac"] = round((b_tris + b_res) * SPS / step_s / 1e9 / HBM_PEAK_GBS, 4)
            rl["path"] = {"algorithmic_bytes_per_scan": {"k_sc_tris": int(b_tris), "k_sc_resolve": int(b_res), "k_sc_rest": 0},
                          "achieved": round((b_tris + b_res) * SPS / step_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": rl["path_frac"],
                          "frac_k_sc_tris_bytes_only": round(b_tris * SPS / step_s / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "algorithmic bytes of the whole three-kernel path per step / ms_per_step (the driver-timed "
                                  "number) / 8 TB/s; needs no kernel-exclusivity argument"}
            # Does the figure named `frac` fit the driver's clock?  `avg_kernel_ms` is an EXCLUSIVE duration (one stream, nothing
            # beside the kernel); in the timed region the batches of three streams overlap and fill each other's tails, so
            # launches x avg_kernel_ms may exceed ms_per_step.  `frac_on_step_clock` charges the WHOLE step to k_sc_tris
            # (compulsory bytes of the step's launches / ms_per_step): it fits by construction and bounds the kernel from below.
            launches = SPS / args.batch
            comp_scan = rl["bytes_per_scan"]["hbm_compulsory"]
            rl["step_clock"] = {"launches_per_step": launches, "avg_kernel_ms_exclusive": rl["avg_kernel_ms"],
                                "launches_x_avg_kernel_ms": round(launches * rl["avg_kernel_ms"], 4),
                                "ms_per_step": round(step_s * 1e3, 4),
                                "fits": bool(launches * rl["avg_kernel_ms"] <= step_s * 1e3),
                                "achieved_on_step_clock": round(comp_scan * SPS / step_s / 1e9, 1),
                                "frac_on_step_clock": round(comp_scan * SPS / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                "note": "exclusive kernel durations overlap in the timed region (3 batches in flight); "
                                        "frac_on_step_clock = compulsory HBM bytes of the step / ms_per_step / 8 TB/s -- the whole "
                                        "step charged to this kernel; one_batch_in_flight below is the configuration in which "
                                        "launches x kernel time does fit its own region"}
            if one_batch:
                rl["one_batch_in_flight"] = one_batch
            # all three kernels of a scan together at the measured scan rate: the HBM bandwidth the whole path
            # sustains over the timed region (PMC traffic per launch from profiles/rNN/pmc.json, null when stale)
            parts = [measured_traffic("scatter", args.batch, k)[0] for k in ("k_sc_tris", "k_sc_rest", "k_sc_resolve")]
            if all(p is not None for p in parts):
                per_scan = sum(parts) / args.batch
                rl["whole_path"] = {"hbm_bytes_per_scan": int(per_scan),
                                    "sustained_GBs": round(K / dt * per_scan / 1e9, 1),
                                    "frac_of_peak": round(K / dt * per_scan / 1e9 / HBM_PEAK_GBS, 4),
                                    "note": "PMC traffic of k_sc_tris + k_sc_rest + k_sc_resolve per scan x scans/s of "
                                            "this rank; 6290 GB/s is what a float4 copy reaches on this chip"}
                ents = [measured_traffic("scatter", args.batch, k)[2] for k in ("k_sc_tris", "k_sc_rest", "k_sc_resolve")]
                if all(e and e.get("valu_active_quad_cycles_per_launch") for e in ents):
                    # the three kernels' vector-issue cycles per SIMD per scan against the wall time of a scan in the timed
                    # region (world == 1 figure of this rank): how full the chip's vector units are over the whole region
                    busy = sum(e["valu_active_quad_cycles_per_launch"] for e in ents) * SQ_CYCLES_PER_COUNT / N_SIMD / args.batch
                    rl["whole_path"]["valu_busy_frac_of_timed_region"] = round(busy / SHADER_GHZ / 1e9 / (dt / K), 4)
            else:
                rl["whole_path"] = None
        out = {
            "metric": "Mrays/sec, one new ~1M-triangle mesh per scan -> 64x2048 range/label image",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {H}x{W} rays vs {n_faces}-triangle synthetic scene "
                                   f"(fov {wl['fov_up']}/{wl['fov_down']}); 1 step = one batch of {SPS} scans, each "
                                   f"with a new mesh; {len(scenes)} distinct scenes cycled",
                       "scans_per_step": SPS, "ms_per_scan": round(dt / K * 1e3, 5),
                       "strategy": args.strategy,
                       "parallelism": f"scan-parallel x{world}" + (
                           (f", range f32 + label {str(label_dtype)[6:]} images gathered to rank 0 over RCCL (in 11 pieces inside the timed region, overlapped with the rendering)"
                            if gather_info.get("mode") != "sharded" else
                            ", images stay sharded on the ranks that rendered them (a rank's image stream exceeds the headroom of its xGMI link into one root), per-scan metadata gathered to rank 0 over RCCL")
                           if dist.is_initialized() else ""),
                       "gather": gather_info or None,
                       "streams_per_gpu": S, "scans_per_call": args.batch if args.strategy == "scatter" else 1},
            "scans_per_s": round(world * K / dt, 2),
            "hit_fraction": round(hits / R, 4),
            "verified": bool(verify and verify["ok"]), "verification": verify,
            "roofline": rl,
        }
        if failures:
            out["failed_legs"] = failures
        if phase:
            out["lbvh_phase_ms"] = {k: round(v, 4) for k, v in phase.items() if k.startswith("ms_") and k != "ms_trace"}
        if other:
            out["other_strategy"] = other
        if e2e:
            out["e2e"] = e2e
        if chain:
            out["fusion_chain"] = chain
        if chain5:
            out["fusion_chain_nscans5"] = chain5
        if mergemesh_leg:
            out["mergemesh_from_points"] = mergemesh_leg
        if from_points:
            out["deform_from_points"] = from_points
            out["projection"] = from_points["projection"]
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            cb = guarded("cpu_baseline", cpu_baseline, wl, 0, args.cpu_reps or 12)
            out["cpu_baseline"] = cb
            if cb:
                # matching clocks, all against the REAL reference on this box's host cores:
                #   device_resident vs the reference's e2e call   (what a scan costs when the mesh is where the renderer is)
                #   device_resident vs the reference's TRACE-ONLY loop (its BVH already built: the most favourable CPU clock)
                #   e2e single call / pipelined (PCIe inclusive, host buffers in and out) vs the reference's e2e call
                def _r(a, b):
                    return round(a / b, 1) if (a and b) else None
                ck = cb.get("clocks", {})
                tr_all = (ck.get("all_threads") or {}).get("trace_only_Mrays_s")
                tr_one = (ck.get("one_thread") or {}).get("trace_only_Mrays_s")
                e2e_one = (ck.get("one_thread") or {}).get("e2e_Mrays_s")
                sc_v = e2e["single_call"]["value"] if (e2e and e2e.get("single_call")) else None
                pp_v = e2e["pipelined"]["value"] if (e2e and e2e.get("pipelined")) else None
                out["speedup_vs_cpu_baseline"] = {"device_resident": _r(value / world, cb["value"]),
                                                  "device_resident_vs_trace_only_all_threads": _r(value / world, tr_all),
                                                  "device_resident_vs_trace_only_one_thread": _r(value / world, tr_one),
                                                  "e2e_single_call": _r(sc_v, cb["value"]),
                                                  "e2e_single_call_vs_one_thread": _r(sc_v, e2e_one),
                                                  "e2e_pipelined": _r(pp_v, cb["value"])}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())  # the ONE line on stdout
    shared_rays.close()
    for wk in workers:
        wk.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
