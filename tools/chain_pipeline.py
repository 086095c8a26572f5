#!/usr/bin/env python3
"""The fusion chain (reset -> integrate -> marching cubes -> render, mesh never leaves HBM) with SEVERAL output scans in
flight: every output scan of the reference's loop has its own volume, mesh and image (lidar_deform.py:393-462), so `chains`
host threads each run the chain for their own scans on their own HIP stream, volume, mesh and scene.  The kernels of the chain
are mostly sparse sweeps that leave the chip half empty; chains in flight fill each other's gaps.
    python tools/chain_pipeline.py [chains [scans per chain [observations]]]     -> one JSON line
Every chain's last range / label image is compared bit for bit with the single chain's."""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
from lidar_transfer_amd.fusion import DeviceMesh, TSDFVolume
from lidar_transfer_amd.laserscan import create_rays
from lidar_transfer_amd.raytracer import RaySet, Scene
from lidar_transfer_amd.synth import WORKLOADS, synth_scene


def run(chains=3, n=12, n_obs=1, device=0, workload="C2", warm=2, voxel=0.05):
    wl = WORKLOADS[workload]; H, W = wl["H"], wl["W"]; dev = torch.device("cuda", device)
    lib = _lib.load()
    mesh0 = [torch.from_numpy(x).to(dev) for x in synth_scene(0, wl["tris"])]
    rays = torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], H, W)).to(dev)
    rs = RaySet(rays, H)
    sc0 = Scene(device); sc0.set_mesh(*mesh0)
    o = sc0.render(rs, (0, 0, 0)); torch.cuda.synchronize()
    folded = (o["endcolors"][:, 2].reshape(H, W).float() * 65536.0).contiguous()
    depth = o["range"].reshape(H, W).clone(); remi = o["endrem"].reshape(H, W).clone()
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    obs = [(folded, depth)]
    for k in range(1, n_obs):
        noise = (torch.rand((H, W), device=dev, generator=gen) - 0.5) * 0.04
        hole = torch.rand((H, W), device=dev, generator=gen) < 0.05
        flip = torch.rand((H, W), device=dev, generator=gen) < 0.02
        obs.append((torch.where(flip, torch.full_like(folded, 50.0 * 65536.0), folded).contiguous(),
                    torch.where(hole | (depth == 0), torch.zeros_like(depth), depth + noise).contiguous()))
    torch.cuda.synchronize()
    org = (C.c_float * 3)(0, 0, 0)
    bnds = np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]])

    class Chain:
        def __init__(self):
            self.vol = TSDFVolume(bnds, voxel, wl["fov_up"], wl["fov_down"])
            self.mesh = DeviceMesh(device)
            self.sc = Scene(device)
            self.out = self.sc.alloc_outputs(H * W)
            self.stream = torch.cuda.Stream(dev)
            self.sp = C.c_void_p(self.stream.cuda_stream)

        def scan(self):
            _lib.check(lib.lt_tsdf_reset(self.vol._h, self.sp), "reset")
            for f_k, d_k in obs:
                _lib.check(lib.lt_tsdf_integrate_dev(self.vol._h, f_k.data_ptr(), d_k.data_ptr(), remi.data_ptr(), H, W, 1.0,
                                                     _lib.LT_TSDF_MERGE, self.sp), "integrate")
            _lib.check(lib.lt_tsdf_extract_mesh_dev(self.vol._h, self.mesh._h, self.sp, None), "marching cubes")
            _lib.check(lib.lt_scene_set_mesh(self.sc._h, self.mesh._h), "set mesh")
            o = self.out
            _lib.check(lib.lt_scene_render_dev(self.sc._h, rs._h, org, o["endpoints"].data_ptr(), o["endcolors"].data_ptr(),
                                               o["range"].data_ptr(), o["endrem"].data_ptr(), o["tri"].data_ptr(),
                                               _lib.LT_TRACE_WRITE_MISSES, self.sp, None), "render")

        def close(self):
            self.mesh.close(); self.vol.close()

    def timed(cs, n_each):
        """wall time of n_each scans on every chain of cs, all in flight together"""
        bar = threading.Barrier(len(cs) + 1)
        errs = []

        def work(c):
            try:
                torch.cuda.set_device(dev)
                for _ in range(warm):
                    c.scan()
                c.stream.synchronize()
                bar.wait()  # (BrokenBarrierError if another chain failed: ends this one too)
                for _ in range(n_each):
                    c.scan()
                c.stream.synchronize()
            except threading.BrokenBarrierError:
                pass
            except BaseException as e:  # noqa: BLE001
                errs.append(repr(e))
                bar.abort()
        th = [threading.Thread(target=work, args=(c,)) for c in cs]
        for t in th:
            t.start()
        try:
            bar.wait()
        except threading.BrokenBarrierError:
            pass  # (a chain failed in its warm-up: reported below)
        t0 = time.perf_counter()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        if errs:
            raise RuntimeError(errs[0])
        return dt

    cs = [Chain() for _ in range(chains)]
    t1 = timed(cs[:1], n)
    ref_range = cs[0].out["range"].clone(); ref_label = cs[0].out["endcolors"].clone()
    tn = timed(cs, n)
    torch.cuda.synchronize()
    same = all(torch.equal(c.out["range"], ref_range) and torch.equal(c.out["endcolors"], ref_label) for c in cs)
    R = H * W
    rec = {"chains_in_flight": chains, "scans_per_chain": n, "observations": n_obs,
           "one_chain_ms_per_scan": round(t1 / n * 1e3, 4),
           "ms_per_scan": round(tn / (n * chains) * 1e3, 4), "scans_per_s": round(n * chains / tn, 1),
           "value": round(R * n * chains / tn / 1e6, 2), "unit": "Mrays/s",
           "gain_over_one_chain": round((t1 / n) / (tn / (n * chains)), 3),
           "verified": bool(same), "mesh_faces": cs[0].mesh.n_faces,
           "hbm_resident_GB": round(chains * 4 * np.prod(cs[0].vol._vol_dim) * 4 / 2**30, 1)}
    for c in cs:
        c.close()
    return rec


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:] if x.isdigit()]
    print(json.dumps(run(*a)))
