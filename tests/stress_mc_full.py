#!/usr/bin/env python3
"""tests/stress_mc_full.py -- marching cubes at BASELINE's FULL size against the checker: the default 2000 x 2000 x 200 volume
(800 M voxels) with the fused C2 street scene of the bench's `fusion_chain` (one 64 x 2048 observation rendered from the ~1 M
triangle scene), extracted on the device and by the CPU oracle (oracle/lt_mc_oracle.c = scikit-image 0.18.3's arrays, goldens
F10 / F10b) from the downloaded fields.  The same vertices, the same face stream, and after lt_mesh_renumber_dev the same
arrays element for element.  One-off (the oracle walks 800 M cells single-threaded and needs ~25 GB of host memory): run by
hand / by tools/r04_final.sh, not by pytest."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from mesh_canon import assert_same_mesh
    from lidar_transfer_amd.fusion import TSDFVolume
    from lidar_transfer_amd.laserscan import create_rays
    from lidar_transfer_amd.raytracer import RaySet, Scene
    from lidar_transfer_amd.synth import WORKLOADS, synth_scene
    from oracle import binding as ob
    wl = WORKLOADS["C2"]
    H, W = wl["H"], wl["W"]
    dev = torch.device("cuda", 0)
    sc = Scene(0)
    sc.set_mesh(*[torch.from_numpy(x).to(dev) for x in synth_scene(0, wl["tris"])])
    rs = RaySet(torch.from_numpy(create_rays(wl["fov_up"], wl["fov_down"], H, W)).to(dev), H)
    o = sc.render(rs, (0, 0, 0))
    torch.cuda.synchronize()
    lab = o["endcolors"][:, 2].reshape(H, W).float()
    label3 = torch.stack([lab, torch.zeros_like(lab), torch.zeros_like(lab)], 2)
    vol = TSDFVolume(np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]]), 0.05, wl["fov_up"], wl["fov_down"])
    vol.integrate(label3, o["range"].reshape(H, W).clone(), o["endrem"].reshape(H, W).clone(), np.eye(4))
    mesh = vol.extract_mesh()
    got = [t.cpu().numpy() for t in mesh.tensors()]
    tsdf, _, color, rem = [t.cpu().numpy() for t in vol.get_volume_tensors()]
    t0 = time.time()
    want = ob.marching_cubes(tsdf, color, rem, np.float32(0.05), vol._vol_origin)
    t1 = time.time()
    assert_same_mesh(got, want, "default volume: ")
    v, f, c, r = [t.cpu().numpy() for t in mesh.renumber().tensors()]
    same = (np.array_equal(v.view(np.int32), want[0].view(np.int32)) and np.array_equal(f, want[1]) and
            np.array_equal(c, np.asarray(want[2])) and np.array_equal(r.view(np.int32), want[3].view(np.int32)))
    print(f"default volume {tsdf.shape}: {want[0].shape[0]} vertices, {want[1].shape[0]} faces -- device mesh = oracle mesh (vertices, face "
          f"stream); after renumbering the arrays are {'EQUAL element for element' if same else 'DIFFERENT'}; oracle {t1 - t0:.0f} s on the host")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
