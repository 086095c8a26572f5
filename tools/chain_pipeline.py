#!/usr/bin/env python3
"""The fusion chain (reset -> integrate -> marching cubes -> render, mesh never leaves HBM) with SEVERAL output scans in
flight: lidar_transfer_amd.pipeline.FusionScanPipeline -- every output scan of the reference's loop has its own volume,
mesh and image (lidar_deform.py:393-462), so `chains` host threads each run the chain for their own scans on their own HIP
stream, volume, mesh and scene.  The kernels of the chain are mostly sparse sweeps that leave the chip half empty
(DESIGN.md section 7c); chains in flight fill each other's gaps.
    python tools/chain_pipeline.py [chains [scans per chain [observations]]]     -> one JSON line
Every scan's range / label image is compared bit for bit with the single chain's."""
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.pipeline import FusionScanPipeline
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene


def run_cases(chains=3, cases=((12, 1),), device=0, workload="C2", warm=8, voxel=0.05):
    """cases: (scans per chain, observations per scan) pairs, all timed on ONE multi-chain pipeline (created first: where the
    driver places its volumes matters) and then on one single-chain pipeline for the reference."""
    wl = WORKLOADS[workload]; H, W = wl["H"], wl["W"]; dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    mesh0 = [torch.from_numpy(x).to(dev) for x in synth_scene(0, wl["tris"])]
    rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], H, W)).to(dev)
    rs = RaySet(rays, H)
    sc0 = Scene(device); sc0.set_mesh(*mesh0)
    o = sc0.render(rs, (0, 0, 0)); torch.cuda.synchronize()
    # the observation: this very sensor looking at scene 0; label in channel 0 (laserscan.py:893-895), folded
    folded = (o["endcolors"][:, 2].reshape(H, W).float() * 65536.0).contiguous()
    depth = o["range"].reshape(H, W).clone(); remi = o["endrem"].reshape(H, W).clone()
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    obs_all = [(folded, depth, remi)]
    for k in range(1, max(c[1] for c in cases)):  # the neighbouring scans re-projected into the primary pose: noise, holes, a few other labels
        noise = (torch.rand((H, W), device=dev, generator=gen) - 0.5) * 0.04
        hole = torch.rand((H, W), device=dev, generator=gen) < 0.05
        flip = torch.rand((H, W), device=dev, generator=gen) < 0.02
        obs_all.append((torch.where(flip, torch.full_like(folded, 50.0 * 65536.0), folded).contiguous(),
                        torch.where(hole | (depth == 0), torch.zeros_like(depth), depth + noise).contiguous(), remi))
    torch.cuda.synchronize()
    sc0.close(); rs.close()
    bnds = np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]])

    def timed(pipe, n, obs):
        """wall time of n scans per chain, all submitted at once; the images of every scan"""
        n_chains = len(pipe._chains)
        for t in [pipe.submit(obs, inputs_ready=True) for _ in range(warm * n_chains)]:
            pipe.wait(t)
        # (the images of every timed scan are kept for the comparison below: allocated BEFORE the clock -- fresh device
        # memory inside the timed region would be hipMalloc calls, not the caching allocator's recycling)
        bufs = [pipe._chains[0]["scene"].alloc_outputs(pipe.n_rays) for _ in range(n * n_chains)]
        torch.cuda.synchronize()
        # (no collector pass inside the clock: with torch's object graph a full collection is a 30-60 ms pause that
        # holds the interpreter lock -- it landed in the first scans of a burst and looked like a GPU stall)
        gc.collect()
        gc.disable()
        try:
            t0 = time.perf_counter()
            tickets = [pipe.submit(obs, out=b, inputs_ready=True) for b in bufs]  # (synchronised above)
            outs = [pipe.wait(t) for t in tickets]
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        return dt, outs

    multi = {}
    with FusionScanPipeline(bnds, voxel, wl["fov_up"], wl["fov_down"], rays, H, chains=chains, device=device) as pipe:
        nvol = int(np.prod(pipe._chains[0]["vol"]._vol_dim))
        for n, n_obs in cases:
            multi[(n, n_obs)] = timed(pipe, n, obs_all[:n_obs])
    recs = []
    R = H * W
    with FusionScanPipeline(bnds, voxel, wl["fov_up"], wl["fov_down"], rays, H, chains=1, device=device) as pipe:
        for n, n_obs in cases:
            t1, outs1 = timed(pipe, n, obs_all[:n_obs])
            ref_range, ref_label, faces = outs1[-1]["range"], outs1[-1]["endcolors"], outs1[-1]["n_faces"]
            tn, outs = multi.pop((n, n_obs))
            same = all(torch.equal(q["range"], ref_range) and torch.equal(q["endcolors"], ref_label) and q["n_faces"] == faces
                       for q in outs)
            recs.append({"chains_in_flight": chains, "scans_per_chain": n, "observations": n_obs,
                         "one_chain_ms_per_scan": round(t1 / n * 1e3, 4),
                         "ms_per_scan": round(tn / (n * chains) * 1e3, 4), "scans_per_s": round(n * chains / tn, 1),
                         "value": round(R * n * chains / tn / 1e6, 2), "unit": "Mrays/s",
                         "gain_over_one_chain": round((t1 / n) / (tn / (n * chains)), 3),
                         "verified": bool(same), "scans_verified": len(outs), "mesh_faces": int(faces),
                         "hbm_resident_GB": round(chains * 4 * nvol * 4 / 2**30, 1),
                         "api": "lidar_transfer_amd.pipeline.FusionScanPipeline"})
            del outs, outs1
    return recs


def run(chains=3, n=12, n_obs=1, device=0, workload="C2", warm=8, voxel=0.05):
    return run_cases(chains, ((n, n_obs),), device, workload, warm, voxel)[0]


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:] if x.isdigit()]
    print(json.dumps(run(*a)))
