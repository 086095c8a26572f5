import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from lidar_transfer_amd.laserscan import Projector
from lidar_transfer_amd.synth import synth_cloud
dev = torch.device("cuda", 0)
p, rm, lb = synth_cloud(3, 130000, dtype=np.float64)
cl = [(torch.from_numpy(p).to(dev), torch.from_numpy(rm).to(dev), torch.from_numpy(lb.astype(np.int32)).to(dev))]
pj = Projector(0)
for outs in (("range", "rem", "label_folded"), ("range", "rem", "label_folded", "bnds")):
    keep = None
    for _ in range(5):
        keep = pj.project(cl, 3.0, -25.0, 64, 2048, new=True, remove=True, outputs=outs, out=keep)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        pj.project(cl, 3.0, -25.0, 64, 2048, new=True, remove=True, outputs=outs, out=keep)
    e1.record(); torch.cuda.synchronize()
    print(outs[-1], round(e0.elapsed_time(e1) / 50 * 1e3, 2), "us per call")
    if "bnds" in outs:
        b = keep[0]["bnds"].cpu().numpy(); q = p[np.linalg.norm(p, axis=1) > 0]
        print(b.tolist())
