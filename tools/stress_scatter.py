#!/usr/bin/env python3
"""Randomised GPU stress: scatter strategy vs LBVH strategy (both on the GPU, bit for bit) over many random
sensor grids, origins and triangle soups.  Not part of the test suite; run it after touching lt_scatter.hip:
    gpurun -- python tools/stress_scatter.py --cases 300"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import synth_scene
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_trace_gpu import _adversarial_soup

def make_case(rng):
    """One random case (same draws, same order as ever: seeds reproduce)."""
    H = int(rng.choice([1, 2, 5, 16, 64, 128])); W = int(rng.choice([1, 3, 64, 301, 1024, 2048, 4000]))
    if rng.random() < 0.03: H, W = 5000, int(rng.choice([1, 3]))     # more rows than the bin grid has (4096)
    elif rng.random() < 0.03: H, W = int(rng.choice([1, 2])), 10000  # more columns than the bin grid has (8192)
    up = float(rng.uniform(-5, 60)); down = float(up - rng.uniform(0.5, 80))
    kind = rng.integers(0, 3)
    if kind == 0:
        v, f, c, r = _adversarial_soup(rng, int(rng.integers(10, 6000)))
    elif kind == 1:
        v, f, c, r = synth_scene(int(rng.integers(0, 1 << 30)), int(rng.integers(2000, 200000)))
    else:  # low-poly: a handful of huge triangles (every one of them goes through the big-triangle queue)
        v, f, c, r = _adversarial_soup(rng, int(rng.integers(1, 40)))
        v = (v * 30).astype(np.float32)
    if rng.random() < 0.12:  # broken mesh: non-finite / huge vertices, degenerate faces
        v = v.copy(); f = f.copy()
        k = max(1, v.shape[0] // 200)
        v[rng.integers(0, v.shape[0], k), rng.integers(0, 3, k)] = rng.choice([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30], k)
        fd = rng.integers(0, f.shape[0], max(1, f.shape[0] // 100))
        f[fd, 1] = f[fd, 0]
    origin = tuple(float(x) for x in rng.normal(size=3) * rng.choice([0.0, 0.1, 2.0, 2.0, 1e3, 1e5]))
    rays = create_rays(up, down, H, W)
    rk = rng.random()
    if rk < 0.15:    # jittered grid: an irregular ray set
        rays = (rays + rng.normal(size=rays.shape).astype(np.float32) * 1e-3).astype(np.float32)
    elif rk < 0.25:  # azimuth grid WITHOUT the duplicated seam column (W columns over [-pi, pi))
        az = (-np.pi + 2 * np.pi * (np.arange(W) + rng.random()) / W)
        el = np.deg2rad(np.linspace(up, down, H))
        rays = np.stack([np.cos(el)[:, None] * np.cos(az)[None], np.cos(el)[:, None] * np.sin(az)[None],
                         np.sin(el)[:, None] * np.ones(W)[None]], -1).reshape(-1, 3).astype(np.float32)
    elif rk < 0.35:  # two beam blocks with different spacing (HDL-64 style): rows are not equidistant
        el = np.deg2rad(np.concatenate([np.linspace(up, (up + down) / 2, H - H // 2, endpoint=False),
                                        np.linspace((up + down) / 2, down, H // 2) ** 1.0]))[:H]
        az = np.linspace(np.pi, -np.pi, W)
        rays = np.stack([np.cos(el)[:, None] * np.cos(az)[None], np.cos(el)[:, None] * np.sin(az)[None],
                         np.sin(el)[:, None] * np.ones(W)[None]], -1).reshape(-1, 3).astype(np.float32)
    elif rk < 0.42:  # many rays share a direction: bins holding several rays (the grid's slot-range entries)
        idx = rng.integers(0, rays.shape[0], size=rays.shape[0] // 3)
        rays[idx] = rays[rng.integers(0, rays.shape[0], size=idx.size)] * rng.uniform(0.5, 3.0, (idx.size, 1)).astype(np.float32)
    elif rk < 0.46:  # unnormalised, zero and non-finite rays
        rays = (rays * rng.uniform(0.1, 50.0, (rays.shape[0], 1))).astype(np.float32)
        broken = rng.integers(0, rays.shape[0], size=max(1, rays.shape[0] // 50))
        rays[broken[: broken.size // 2]] = 0.0
        rays[broken[broken.size // 2:], rng.integers(0, 3)] = np.nan
    return H, W, up, down, kind, v, f, c, r, origin, rays, rk


ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=100); ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--batch", type=int, default=0, help="also render groups of up to this many cases with one lt_scene_render_batch_dev call")
ap.add_argument("--first", type=int, default=0, help="skip the GPU work of the cases before this one (same random draws)")
ap.add_argument("--oracle", action="store_true", help="also compare with the brute-force CPU oracle where tris x rays < 3e7")
a = ap.parse_args()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(a.seed)
bad = 0; tot_rays = tot_hits = tot_tris = 0; n_or = 0; pending = []; n_batched = 0


def flush(pending):
    """the scans rendered one by one above, once more as ONE batch call"""
    global bad
    outs = Scene.render_batch([p[0] for p in pending], [p[1] for p in pending], [p[2] for p in pending])
    torch.cuda.synchronize()
    for (sc_, rs_, org_, A_, case_), o in zip(pending, outs):
        if not all(torch.equal(A_[k].view(torch.int32), o[k].view(torch.int32)) for k in ("tri", "range", "endpoints", "endcolors", "endrem")):
            bad += 1
            print(f"BATCH MISMATCH case {case_}")
        rs_.close(); sc_.close()

for case in range(a.cases):
    H, W, up, down, kind, v, f, c, r, origin, rays, rk = make_case(rng)
    if case < a.first:
        continue
    if os.environ.get('LT_STRESS_VERBOSE'): print(f'case {case}: H={H} W={W} kind={kind} rk={rk:.3f} tris={f.shape[0]} origin={origin}', flush=True)
    sc = Scene(0); t = [torch.from_numpy(x).to(dev) for x in (v, f, c, r)]
    sc.set_mesh(*t); rt = torch.from_numpy(rays).to(dev); rs = RaySet(rt, H)
    A = sc.render(rs, origin); sc.build(); B = sc.trace(rt, origin, H)
    same = all(torch.equal(A[k].view(torch.int32), B[k].view(torch.int32)) for k in ("tri", "range", "endpoints", "endcolors", "endrem"))
    tot_rays += H * W; tot_hits += int((A['tri'] >= 0).sum()); tot_tris += int(f.shape[0])
    if a.oracle and f.shape[0] * H * W < 3e7:
        from oracle import binding as ob
        ref = ob.oracle_trace(rays, np.asarray(origin, np.float32), v, f, c, r, H, mode=ob.MODE_BRUTE, norm=ob.NORM_SSE_TABLE)
        n_or += 1
        for k in ("tri", "range", "endpoints", "endcolors", "endrem"):
            if not np.array_equal(A[k].cpu().numpy().reshape(-1).view(np.int32), np.ascontiguousarray(ref[k]).reshape(-1).view(np.int32)):
                same = False
                print(f"ORACLE MISMATCH case {case}: {k}")
    if not same:
        bad += 1
        nd = int((A["tri"] != B["tri"]).sum())
        print(f"MISMATCH case {case}: H={H} W={W} fov=({up:.2f},{down:.2f}) kind={kind} tris={f.shape[0]} origin={origin} differing rays={nd}")
        if os.environ.get("LT_STRESS_DUMP"):  # which rays, and the triangle each path found (numpy dump for a CPU post-mortem)
            os.makedirs(os.environ["LT_STRESS_DUMP"], exist_ok=True)
            ids = torch.nonzero(A["tri"] != B["tri"]).reshape(-1)
            np.savez(os.path.join(os.environ["LT_STRESS_DUMP"], f"case{case}.npz"), rays=ids.cpu().numpy(),
                     tri_scatter=A["tri"][ids].cpu().numpy(), tri_lbvh=B["tri"][ids].cpu().numpy(),
                     t_scatter=A["range"][ids].cpu().numpy(), t_lbvh=B["range"][ids].cpu().numpy())
    if a.batch > 1:
        pending.append((sc, rs, origin, {k: t_.clone() for k, t_ in A.items() if hasattr(t_, 'clone')}, case)); n_batched += 1
        if len(pending) == a.batch or case % 5 == 0:  # groups of 1 .. batch scans
            flush(pending); pending = []
    else:
        rs.close(); sc.close()
if pending:
    flush(pending)
print(f"{a.cases} cases, {bad} mismatches; {tot_tris} triangles, {tot_rays} rays, {tot_hits} hits; {n_or} cases also against the brute-force oracle, {n_batched} also in batch calls")
sys.exit(1 if bad else 0)
