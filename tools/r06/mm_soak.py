#!/usr/bin/env python3
"""deform('mergemesh') with scans in flight, for a long time: random sequences whose bounds shrink at random scans (the
speculative launch on the previous scan's geometry is then wrong and the scan is run again), 2 / 3 / 4 chains, every output
scan compared with the one-scan-at-a-time DeviceDeform.mergemesh run -- range bits, labels, remissions, volume dimensions,
the bounds left behind.   python tools/r06/mm_soak.py [sequences] [scans per sequence] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_deform_gpu as T
from lidar_transfer_amd.laserscan import create_rays_device
from lidar_transfer_amd.pipeline import FusionScanPipeline

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_scan = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
g = T._gold("f14_deform_mergemesh.npz")
pts, rem, lab = T._gold_clouds(g, "a0", 1)[0]
H, W, fu, fd = 32, 512, 3.0, -25.0
cfg = np.array([[-7, 7], [-7, 7], [-2, 3]])
rays = create_rays_device(fu, fd, H, W, device=0)
bad = scans = 0
tot = {"scans": 0, "waited": 0, "rerun": 0}
for s in range(n_seq):
    lim, seq = 6.4, []
    for k in range(n_scan):
        if rng.random() < 0.15:
            lim = max(2.4, lim - float(rng.choice([0.5, 1.0])))   # the bounds move at this scan
        keep = (pts[:, 0] < lim) & (pts.norm(dim=1) > 0) & (torch.from_numpy(rng.random(len(pts)) < 0.98).to(pts.device))
        seq.append([(pts[keep].contiguous(), rem[keep].contiguous(), lab[keep].contiguous())])
    want, _ = T._mm_serial(seq, cfg.copy())
    chains = int(rng.choice([2, 3, 4]))
    bnds = cfg.copy()
    with FusionScanPipeline(bnds, 0.1, fu, fd, rays, H, chains=chains, device=0, label_image=True, source_hw=(H, W),
                            fixed_volume=False) as pipe:
        tickets = [pipe.submit_mergemesh(c, inputs_ready=True) for c in seq]
        for k, t in enumerate(tickets):
            got, w = pipe.wait(t), want[k]
            ok = (got["vol_dim"] == w["vol_dim"] and got["vol_bnds_after"] == w["after"] and
                  torch.equal(got["range"].view(-1).view(torch.int32), w["range"].view(-1).view(torch.int32)) and
                  torch.equal(got["endcolors"].view(-1), w["label"].view(-1)) and torch.equal(got["endrem"].view(-1), w["rem"].view(-1)))
            bad += 0 if ok else 1
            scans += 1
        bad += 0 if np.array_equal(bnds, want[-1]["bnds"]) else 1
        for key in tot:
            tot[key] += pipe._mm_state.stats[key]
print(f"{n_seq} sequences x {n_scan} scans, 2-4 chains: {scans} output scans compared with the serial run, {bad} mismatches; "
      f"geometry records: {tot}")
