#!/usr/bin/env python3
"""bench.py -- virtual-LiDAR synthesis throughput on MI355X (one process per GPU): the HEADLINE line only.

A *step* is one pass of the hot path over one batch of input: 64 scans (--calls-per-step 8 x --batch 8), each with a NEW
triangle mesh (the reference rebuilds its BVH per call, RayTracer.cpp:54) -> closest hit of one ray per (beam, azimuth)
cell -> range / label / remission / end point / triangle images; meshes, rays and images resident in HBM.  Workload at
N=1: BASELINE.json configs[1] ("C2": ~1 M-triangle scene, 64x2048 HDL-64E target, fov +3/-25), synthetic.

    python bench.py [--gpus N --steps K --warmup W]            # N=1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Scans shard over ranks (weak scaling), no data-path collective; the range + label images are gathered to rank 0 over
RCCL inside the timed region (one logical gather issued in pieces that overlap the rendering) unless the links cannot
carry them (LT_BENCH_GATHER=auto, the default: decided from a measured link rate before the clock; config.gather says
which).  Rank 0 prints ONE JSON line (< 8 KB) with a compact `roofline` (dominant kernel k_sc_tris, HIP events on its
launch stream) and `cpu_baseline` (the real reference raytracer, oracle/_ref, on this box's host cores).  Everything
else -- the other strategy, the fusion / deform / mergemesh chains, PCIe-inclusive clocks -- is tools/bench_chains.py,
which writes profiles/rNN/bench_extras.json.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

PROBE_EVERY = int(os.environ.get("LT_BENCH_PROBE_EVERY", "8"))  # HIP-event pair around every n-th dominant launch
RAMP_MS = float(os.environ.get("LT_BENCH_RAMP_MS", "80"))  # least GPU work before the clock (the chip's ramp to its sustained rate)
MAX_LINE_BYTES = 8000   # the driver parses ONE stdout line; round 5's 21.6 KB line was not parsed


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step = --calls-per-step x --batch = 64 scans, each with a new mesh, ~0.77 ms; default 64 steps = 4096 scans
    ap.add_argument("--steps", type=int, default=64, help="timed steps; a step = --calls-per-step batches of --batch scans")
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps before the clock")
    ap.add_argument("--calls-per-step", type=int, default=int(os.environ.get("LT_BENCH_CALLS_PER_STEP", "8")))
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--target", default="", help="target sensor YAML (lidar_deform.py --target); overrides the workload's sensor")
    ap.add_argument("--strategy", default=os.environ.get("LT_BENCH_STRATEGY", "scatter"), choices=["scatter", "lbvh"])
    # faces + vertices of all scenes (18.3 MB each on C2) must exceed twice the 256 MiB Infinity Cache
    # (profiles/r04/scenes_sweep.jsonl): 48 x 18.3 MB = 878 MB
    ap.add_argument("--scenes", type=int, default=int(os.environ.get("LT_BENCH_SCENES", "48")))
    # 24 = three batch calls of 8 scans in flight (profiles/r04/streams_sweep.txt)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("LT_BENCH_STREAMS", "24")))
    ap.add_argument("--batch", type=int, default=int(os.environ.get("LT_BENCH_BATCH", "8")),
                    help="scans per lt_scene_render_batch_dev call (scatter strategy, at most 8)")
    ap.add_argument("--tris", type=int, default=0, help="override the workload's triangle count (reduced test configurations)")
    ap.add_argument("--job", default="", help="'c5/DIV': render the reference's multi-sequence job (SemanticKITTI 00-07 scan counts / DIV, "
                    "lidar_deform.py:385-390) block-partitioned over the ranks and gathered in job order; scaling 'strong'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=0, help="reference runs for the CPU baseline (0 = auto)")
    for flag in ("--no-e2e", "--no-chain", "--no-other"):  # legs that moved to tools/bench_chains.py: accepted, ignored
        ap.add_argument(flag, action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe-only", action="store_true",
                    help="only the serial probe of the dominant kernel (the command profiled under rocprofv3 "
                         "--kernel-trace --stats so that roofline.avg_kernel_ms can be recomputed from a CSV)")
    return ap.parse_args(argv)


def launch_command(n_gpus: int, argv, port: int):
    """The one-process-per-GPU launch of this script on one node (what the driver itself would type)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def relaunch_multi_gpu(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become N ranks under torch.distributed.run."""
    import socket
    import torch
    if torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but this node shows {torch.cuda.device_count()} GPU(s)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's peer-to-peer transport needs it here
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    os.execve(sys.executable, launch_command(args.gpus, sys.argv[1:], port), env)


class Harness:
    """Synthetic inputs resident in HBM + the timed region.  tools/bench_chains.py reuses it for the side records."""

    def __init__(self, args, backend="nccl"):
        import torch
        import torch.distributed as dist
        from lidar_transfer_amd.laserscan import create_rays
        from lidar_transfer_amd.raytracer import RaySet, Scene
        from lidar_transfer_amd.synth import WORKLOADS, synth_scene
        self.args, self.torch, self.dist = args, torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        # LT_BENCH_SHARE_GPU=1: every rank on cuda:0 (tests: the N-rank code path on a one-GPU box, gloo transport)
        self.local_rank = 0 if os.environ.get("LT_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(self.local_rank)
        self.dev = dev = torch.device("cuda", self.local_rank)
        if (self.world > 1 or "RANK" in os.environ) and not dist.is_initialized():
            import datetime  # (a rank that never arrives ends the run after 3 minutes, not after the default 10)
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=180),
                                    **({"device_id": dev} if backend == "nccl" else {}))
        wl = dict(WORKLOADS[args.workload])
        if args.target:  # the target scanner as the reference reads it (lidar_deform.py:302-315)
            from lidar_transfer_amd.config import load_sensor
            sensor = load_sensor(args.target)
            wl.update(H=sensor.H, W=sensor.W, fov_up=float(sensor.fov_up), fov_down=float(sensor.fov_down))
        if args.tris:
            wl["tris"] = args.tris
        self.wl, self.H, self.W = wl, wl["H"], wl["W"]
        self.R = R = self.H * self.W
        args.batch = min(max(1, args.batch), 8)  # lt_scene_render_batch_dev takes at most 8 scans
        S = max(1, args.streams)
        self.S = S = (S + args.batch - 1) // args.batch * args.batch
        self.SPS = args.batch * max(1, args.calls_per_step)
        self.K, self.Wm = args.steps * self.SPS, args.warmup * self.SPS
        self.job, self.first = None, 0
        if args.job:  # the job's scan list, block-partitioned: this rank renders scans [first, first + K) of it
            from lidar_transfer_amd.dist import C5_SEQUENCES, job_scan_list, partition
            div = int(args.job.split("/")[1]) if "/" in args.job else 1
            self.job = job_scan_list([(n, -(-c // div)) for n, c in C5_SEQUENCES])
            blocks = [partition(range(len(self.job)), self.world, r) for r in range(self.world)]
            self.K, self.first = len(blocks[self.rank]), (blocks[self.rank][0] if blocks[self.rank] else 0)
        need = self.K * R * 8 + (self.K * R * 6 * (self.world - 1) if self.rank == 0 else 0)
        if need > 0.7 * torch.cuda.get_device_properties(dev).total_memory:
            raise SystemExit(f"bench.py: --steps {args.steps} ({self.K} scans) keeps {need / 2**30:.0f} GiB of images; use fewer steps")
        self.scenes = []
        for i in range(args.scenes):
            v, f, c, r = synth_scene(i if self.job else 1000 * self.rank + i, wl["tris"])  # (a job's scenes belong to its scans)
            self.scenes.append(tuple(torch.from_numpy(x).to(dev) for x in (v, f, c, r)))
        self.n_faces = int(self.scenes[0][1].shape[0])
        self.n_verts = int(np.mean([int(s[0].shape[0]) for s in self.scenes]))
        # labels travel as int16 when every label fits (SemanticKITTI labels are < 260): 6 instead of 8 bytes per ray
        lo = min(int(s[2][:, 2].min()) for s in self.scenes)
        hi = max(int(s[2][:, 2].max()) for s in self.scenes)
        if dist.is_initialized():  # every rank must pick the same width
            lohi = torch.tensor([-lo, hi], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(lohi, op=dist.ReduceOp.MAX)
            lo, hi = -int(lohi[0].item()), int(lohi[1].item())
        self.label_dtype = torch.int16 if (-32768 <= lo and hi <= 32767) else torch.int32
        self.rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], self.H, self.W)).to(dev)
        self.origin = (0.0, 0.0, 0.0)
        self.streams = [torch.cuda.Stream(dev) for _ in range(S)]
        self.workers = [Scene(self.local_rank) for _ in range(S)]
        # ONE ray set for all workers (one target sensor model, laserscan.py:1092-1119): read-only during renders
        self.shared_rays = RaySet(self.rays, self.H)
        torch.cuda.synchronize()
        self.raysets = [self.shared_rays] * S
        self.scratch = [self.workers[0].alloc_outputs(R) for _ in range(S)]
        self.gather_info, self.ramp_info = {}, None
        self.cnt = None

    def close(self):
        self.shared_rays.close()
        for w in self.workers:
            w.close()
        if self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()

    def _ctl(self, t):
        """a control scalar on the device the process group's transport moves (gloo: host memory)"""
        return t.to(self.dev if self.dist.get_backend() == "nccl" else "cpu")

    # ------------------------------------------------------------------------------------------ the timed region
    def run(self, strategy, K, Wm, keep, groups=None, probe_every=PROBE_EVERY):
        """Timed region for one strategy: (seconds, mean in-situ dominant-kernel ms, hits of the last scan, verification).
        `groups`: batches in flight (default all --streams / --batch); HIP events around every `probe_every`-th launch."""
        import ctypes as C
        from lidar_transfer_amd import _lib
        from lidar_transfer_amd.dist import XGMI_LINK_GBS, choose_gather, gather_to_root, measure_link_gbs
        torch, dist, args, dev, R, S = self.torch, self.dist, self.args, self.dev, self.R, self.S
        world, rank, streams, scenes = self.world, self.rank, self.streams, self.scenes
        probe_every = max(1, probe_every)
        dist_on = dist.is_initialized()
        range_all = torch.zeros((K, R), dtype=torch.float32, device=dev) if keep else None
        # the kept colour output is the semantic-label image (LT_TRACE_LABEL_IMAGE = deform's unpack, laserscan.py:912)
        color_all = torch.zeros((K, R), dtype=torch.int32, device=dev) if keep else None
        probes = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range((K + probe_every - 1) // probe_every)]
        for e in probes:  # materialised before the clock; recorded by the library on the launch stream
            e[0].record(), e[1].record()
        # the gather (range f32 + label = 6-8 B per ray) is ONE logical collective issued in pieces that overlap the rendering
        # of the following scans; the last eighth goes in quarters so that 1/32 of the images is exposed after the last scan
        n_chunks = int(os.environ.get("LT_BENCH_GATHER_CHUNKS", "8")) if (dist_on and keep) else 1
        do_gather = dist_on and keep and n_chunks > 0
        n_chunks = max(n_chunks, 1)

        n_base = n_chunks

        def chunk_bounds(k_):  # (the same number of pieces on every rank, whatever its block size: empty pieces are skipped)
            b = [k_ * c // n_base for c in range(n_base + 1)]
            if dist_on and keep and n_base > 1:
                b = b[:-1] + [b[-2] + (b[-1] - b[-2]) * q // 4 for q in (1, 2, 3)] + [b[-1]]
            return b
        Ks = [K] * world  # every rank's block size: equal (weak scaling) unless a --job is partitioned
        if do_gather and self.job:
            dist.all_gather_object(Ks, K)
        bounds = chunk_bounds(K)
        n_chunks = len(bounds) - 1
        off = self.first if self.job else 0  # scan i of this rank is scan off + i of the job
        works, sharded_meta, recv = [], False, None
        lib = _lib.load()
        vp = C.c_void_p
        org = (C.c_float * 3)(*self.origin)
        sh = [vp(st.cuda_stream) for st in streams]
        wh = [w._h for w in self.workers]
        rh = [r._h for r in self.raysets]
        mesh_args = [(vp(v.data_ptr()), vp(f.data_ptr()), vp(c.data_ptr()), vp(r.data_ptr()), v.numel() // 3, f.numel() // 3)
                     for v, f, c, r in scenes]
        pr = [(vp(a.cuda_event), vp(b.cuda_event)) for a, b in probes]
        sp = [{k: vp(t.data_ptr()) for k, t in self.scratch[s].items()} for s in range(S)]
        rng_p = [vp(range_all[k].data_ptr()) for k in range(K)] if keep else None
        col_p = [vp(color_all[k].data_ptr()) for k in range(K)] if keep else None
        rays_p = vp(self.rays.data_ptr())
        FL = _lib.LT_TRACE_WRITE_MISSES | _lib.LT_TRACE_LABEL_IMAGE

        def step(i, slot=None, timed=False):  # one scan per call (lbvh: build + trace; scatter with --batch 1)
            s = i % S
            h, o = wh[s], sp[s]
            p_rng = rng_p[slot] if (keep and slot is not None) else o["range"]
            p_col = col_p[slot] if (keep and slot is not None) else o["endcolors"]
            rc = lib.lt_scene_set_mesh_dev(h, *mesh_args[(off + i) % len(scenes)])
            if timed and slot % probe_every == 0:
                rc |= lib.lt_scene_set_probe(h, *pr[slot // probe_every])
            if strategy == "lbvh":
                rc |= lib.lt_scene_build(h, sh[s], None)
                rc |= lib.lt_scene_trace_dev(h, rays_p, org, R, self.H, o["endpoints"], p_col, p_rng, o["endrem"], o["tri"],
                                             FL, sh[s], None)
            else:
                rc |= lib.lt_scene_render_dev(h, rh[s], org, o["endpoints"], p_col, p_rng, o["endrem"], o["tri"], FL, sh[s], None)
            _lib.check(rc, "bench step") if rc else None

        # scatter, batched: BATCH consecutive scans (own mesh, worker, images) = ONE call = three launches for all of them
        BATCH = args.batch if strategy == "scatter" else 1
        n_groups = S
        if BATCH > 1:
            n_groups = min(groups, S // BATCH) if groups else S // BATCH
            arr = lambda vals: (vp * BATCH)(*vals)  # noqa: E731
            grp_scenes = [arr([wh[g * BATCH + j] for j in range(BATCH)]) for g in range(n_groups)]
            grp_rays = [arr([rh[g * BATCH + j] for j in range(BATCH)]) for g in range(n_groups)]
            grp_out = [{k: arr([sp[g * BATCH + j][k] for j in range(BATCH)]) for k in
                        ("endpoints", "endrem", "tri", "range", "endcolors")} for g in range(n_groups)]
            org_b = (C.c_float * (3 * BATCH))(*(list(self.origin) * BATCH))
            if keep:
                nb_all = (K + BATCH - 1) // BATCH
                slot_rng = [arr([rng_p[min(b * BATCH + j, K - 1)] for j in range(BATCH)]) for b in range(nb_all)]
                slot_col = [arr([col_p[min(b * BATCH + j, K - 1)] for j in range(BATCH)]) for b in range(nb_all)]

        def step_batch(i0, nb, slot0=None, timed=False):
            g = (i0 // BATCH) % n_groups
            rc = 0
            for j in range(nb):
                rc |= lib.lt_scene_set_mesh_dev(wh[g * BATCH + j], *mesh_args[(off + i0 + j) % len(scenes)])
            b = i0 // BATCH
            if timed and b % probe_every == 0:
                rc |= lib.lt_scene_set_probe(wh[g * BATCH], *pr[b // probe_every])
            o = grp_out[g]
            use_slots = keep and slot0 is not None
            rc |= lib.lt_scene_render_batch_dev(nb, grp_scenes[g], grp_rays[g], org_b, o["endpoints"],
                                                slot_col[b] if use_slots else o["endcolors"],
                                                slot_rng[b] if use_slots else o["range"], o["endrem"], o["tri"], FL, sh[g])
            _lib.check(rc, "bench step (batch)") if rc else None

        def gather_chunk(c):
            c0, c1 = bounds[c], bounds[c + 1]
            cur = torch.cuda.current_stream(dev)
            for st in (streams[:n_groups] if BATCH > 1 else streams):  # starts when this chunk's scans are done
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
            label_chunk = color_all[c0:c1].to(self.label_dtype)
            # grouped send/recv, 7 peers -> root over 7 separate xGMI links; the root's own part stays where it is
            for k, src in enumerate((range_all[c0:c1], label_chunk)):
                works.extend(gather_to_root(src, recv[c][k] if rank == 0 else None, dst=0, copy_self=False))

        torch.cuda.synchronize()
        t0w = t_w = time.perf_counter()
        for i in range(0, Wm, BATCH):
            step_batch(i, min(BATCH, Wm - i)) if BATCH > 1 else step(i)
        torch.cuda.synchronize()
        t_w = time.perf_counter() - t_w
        # the chip needs ~30 ms under load before it runs at its sustained rate (profiles/r06/steps_sweep2.txt: the same
        # 20 timed steps read 10.3 Grays/s behind 4 ms of warm-up, 11.5 behind 30 ms): the W warm-up steps asked for are
        # extended to RAMP_MS of GPU work -- untimed like them, counted in config.warmup_ramp
        n_ramp = 0
        if Wm > 0 and RAMP_MS > 0 and t_w * 1e3 < RAMP_MS:
            n_ramp = int(np.ceil((RAMP_MS * 1e-3 - t_w) * Wm / max(t_w, 1e-6) / BATCH)) * BATCH
            for i in range(Wm, Wm + n_ramp, BATCH):
                step_batch(i, BATCH) if BATCH > 1 else step(i)
            torch.cuda.synchronize()
        self.ramp_info = {"min_ms": RAMP_MS, "scans": n_ramp, "ms": round((time.perf_counter() - t0w) * 1e3, 1)}
        if do_gather:
            # what the links sustain (MEASURED with all peers sending at once, before the clock) against what the ranks produce
            # (the slowest rank's warm-up rate): LT_BENCH_GATHER=auto (default) keeps the images on the ranks that rendered
            # them when they would exceed 0.9 of the link rate (only per-scan metadata is gathered); =root / =sharded force
            tw = self._ctl(torch.tensor([t_w], dtype=torch.float64))
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            rate = Wm / max(float(tw.item()), 1e-9)
            per_scan = R * (4 + (2 if self.label_dtype == torch.int16 else 4))
            try:
                link = measure_link_gbs(dev) if world > 1 else None
            except RuntimeError as e:  # (the decision then rests on the nominal link rate)
                print(f"[bench] link measurement failed: {e!r}", file=sys.stderr, flush=True)
                link = None
            mode = os.environ.get("LT_BENCH_GATHER", "auto")
            if mode not in ("root", "sharded"):
                mode = choose_gather(world, per_scan, rate, link_gbs=link)
            link_rate = (link or XGMI_LINK_GBS) * 1e9 / per_scan
            bound_ = bool(world > 1 and mode == "root" and rate > link_rate)
            self.gather_info = {"mode": mode, "requested": os.environ.get("LT_BENCH_GATHER", "auto"),
                                "bytes_per_scan": per_scan, "scans_per_s_per_rank_warmup": round(rate, 1),
                                "per_link_GBs_needed": round(per_scan * rate / 1e9, 2),
                                "per_link_GBs_measured": round(link, 2) if link else None,
                                "link_bound_scans_per_s_per_rank": round(link_rate, 1), "link_bound": bound_,
                                "predicted_job_scans_per_s": round(world * (min(rate, link_rate) if (world > 1 and mode == "root") else rate), 1)}
            if rank == 0:  # before the clock, on stderr: a first N-GPU run explains itself
                print(f"[bench] gather over {world} ranks: {json.dumps(self.gather_info)}", file=sys.stderr, flush=True)
            if mode == "sharded":
                do_gather, sharded_meta = False, True
        if do_gather:
            if rank == 0:  # recv[c][k][r]: piece c of rank r's range (k = 0) / label (k = 1) images; the root's own stay where they are
                pb = [chunk_bounds(k_) for k_ in Ks]
                recv = [[[torch.empty((pb[r][c + 1] - pb[r][c] if r else 0, R), dtype=dt_, device=dev) for r in range(world)]
                         for dt_ in (torch.float32, self.label_dtype)] for c in range(n_chunks)]
            # warm-up of the collective (RCCL sets its channels up lazily) and of the exact torch ops gather_chunk uses
            w_lab = color_all[0:2].to(self.label_dtype)
            torch.empty_like(range_all[0:2]).copy_(range_all[0:2], non_blocking=True)
            torch.empty_like(w_lab).copy_(w_lab, non_blocking=True)
            wbuf = torch.zeros((4, R), dtype=torch.float32, device=dev)
            wl_ = [torch.empty_like(wbuf) for _ in range(world)] if rank == 0 else None
            for _ in range(2):
                for wk in gather_to_root(wbuf, wl_, dst=0, copy_self=False):
                    wk.wait()
            torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        chunk = 0
        for i in range(0, K, BATCH):
            nb = min(BATCH, K - i)
            step_batch(i, nb, slot0=i, timed=True) if BATCH > 1 else step(i, slot=i, timed=True)
            while do_gather and chunk < n_chunks and i + nb >= bounds[chunk + 1]:
                gather_chunk(chunk)
                chunk += 1
        while do_gather and chunk < n_chunks:  # (a rank with an empty block still takes part in every piece)
            gather_chunk(chunk)
            chunk += 1
        n_probed = ((K + BATCH - 1) // BATCH + probe_every - 1) // probe_every
        meta_recv = None
        if sharded_meta:  # the images stay where they were rendered; ONE small gather: hits per scan (8 B per scan)
            for st in streams:
                st.synchronize()
            meta = (range_all > 0).sum(dim=1)
            meta_recv = [torch.empty((k_,), dtype=meta.dtype, device=dev) for k_ in Ks] if rank == 0 else None
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
        assert meta_recv is None or world == 1 or bool((torch.cat(meta_recv[1:]) > 0).any()), "rank 0 did not receive the peers' metadata"
        if dist_on:
            tmax = self._ctl(torch.tensor([dt], dtype=torch.float64))
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in probes[:n_probed]])) if probes else float("nan")
        hits = int((range_all[K - 1] > 0).sum().item()) if (keep and K) else -1
        assert recv is None or world == 1 or bool((recv[-1][0][1] > 0).any()), "rank 0 did not receive the peers' images"
        self.last_images = (range_all, color_all, recv)
        import bench_lib as bl
        return dt, kern_ms, hits, (bl.verify_timed_scans(self, range_all, color_all, K, off) if (keep and K) else None)

    def gathered(self):
        """rank 0 after a gathering run: the (range [N, R] f32, label [N, R]) images of ALL ranks' scans, in job order"""
        torch = self.torch
        rng, col, recv = self.last_images
        parts = ([rng], [col.to(self.label_dtype)])
        for r in range(1, self.world if recv else 1):
            for k in (0, 1):
                parts[k].extend(recv[c][k][r] for c in range(len(recv)))
        return torch.cat(parts[0]), torch.cat(parts[1])


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_multi_gpu(args)  # does not return
    # stdout carries exactly one JSON line: everything else written to fd 1 (RCCL's banner ...) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import bench_lib as bl
    hz = Harness(args, backend=os.environ.get("LT_BENCH_BACKEND", "nccl"))
    world, rank, R = hz.world, hz.rank, hz.R

    def emit(obj):
        os.write(real_stdout, (bl.fit_line(obj, MAX_LINE_BYTES) + "\n").encode())

    if args.probe_only:
        ser_ms = bl.serial_probe_ms(hz, args.strategy, n=64)
        if rank == 0:
            bl.count_work(hz, (args.strategy,))
            rl = bl.roofline_compact(hz, args.strategy, ser_ms, float("nan"), None)
            emit({"probe_only": True, "strategy": args.strategy, "launches": 64, **rl})
        hz.close()
        return
    dt, kern_ms, hits, verify = hz.run(args.strategy, hz.K, hz.Wm, keep=True)
    if os.environ.get("LT_BENCH_DUMP") and rank == 0:  # tests: what rank 0 holds after the gather, in job order
        rg, lb = hz.gathered()
        np.savez(os.environ["LT_BENCH_DUMP"], range=rg.cpu().numpy(), label=lb.cpu().numpy().astype(np.int32))
    try:
        ser_ms = bl.serial_probe_ms(hz, args.strategy)
    except Exception as e:  # noqa: BLE001  (never lose the headline line)
        sys.stderr.write(f"bench.py: serial probe failed: {e!r}\n")
        ser_ms = float("nan")
    if rank == 0:
        bl.count_work(hz, (args.strategy,))
        n_scans = len(hz.job) if hz.job else world * hz.K
        xport = "RCCL" if (hz.dist.is_initialized() and hz.dist.get_backend() == "nccl") else "gloo (host transport)"
        value = n_scans * R / dt / 1e6
        steps = 1 if hz.job else args.steps   # (a --job is one pass over its scan list)
        step_s = dt / steps
        out = {
            "metric": "Mrays/sec, one new ~1M-triangle mesh per scan -> 64x2048 range/label image",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "strong" if hz.job else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {hz.H}x{hz.W} rays vs {hz.n_faces}-triangle synthetic scene (fov "
                                   f"{hz.wl['fov_up']}/{hz.wl['fov_down']}); 1 step = {hz.SPS} scans, each a new mesh; "
                                   f"{len(hz.scenes)} scenes cycled",
                       "scans_per_step": n_scans if hz.job else hz.SPS, "ms_per_scan": round(dt / max(n_scans, 1) * world * 1e3, 5), "strategy": args.strategy,
                       "parallelism": f"scan-parallel x{world}" + ("" if not hz.dist.is_initialized() else (
                           f", images stay sharded on the ranks that rendered them, per-scan metadata gathered to rank 0 over {xport}"
                           if hz.gather_info.get("mode") == "sharded" else
                           f", range f32 + label images gathered to rank 0 over {xport} inside the timed region")),
                       "gather": hz.gather_info or None, "warmup_ramp": hz.ramp_info,
                       "job": args.job or None, "scans_in_flight": hz.S, "scans_per_call": args.batch if args.strategy == "scatter" else 1},
            "scans_per_s": round(n_scans / dt, 2), "hit_fraction": round(hits / R, 4),
            "verified": bool(verify and verify["ok"]), "verification": verify,
            "roofline": bl.roofline_compact(hz, args.strategy, ser_ms, kern_ms, None if hz.job else step_s),
        }
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            try:
                out["cpu_baseline"] = bl.cpu_baseline_compact(hz.wl, 0, args.cpu_reps or 12)
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(f"bench.py: cpu_baseline failed: {e!r}\n")
                out["cpu_baseline"] = None
            cb = out["cpu_baseline"]
            if cb:
                out["speedup_vs_cpu_baseline"] = {"e2e_call_all_threads": round(value / cb["value"], 1),
                                                  "trace_only_all_threads": round(value / cb["trace_only_Mrays_s"], 1)
                                                  if cb.get("trace_only_Mrays_s") else None}
        sys.stdout.flush()
        emit(out)
    hz.close()


if __name__ == "__main__":
    main()
