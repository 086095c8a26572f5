#!/usr/bin/env python3
"""lt_compare_dev on a 64 x 2048 image pair with a street scene's label mix (a handful of label pairs): us per call."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_transfer_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
n, NL = 64 * 2048, 512
rng = np.random.default_rng(0)
lab = rng.choice([40, 50, 10, 80, 0, 70, 48], size=n, p=[0.55, 0.25, 0.06, 0.02, 0.04, 0.05, 0.03]).astype(np.int32)
lab2 = np.where(rng.random(n) < 0.05, 50, lab).astype(np.int32)
t = lambda a: torch.from_numpy(a).to(dev)
sl, tl = t(lab), t(lab2)
sc = t(np.where(lab[:, None] == 0, 0, 0.5).astype(np.float32).repeat(3, 1).copy())
sr, tr = t(rng.random(n).astype(np.float32) * 60), t(rng.random(n).astype(np.float32) * 60)
conf = torch.empty((NL, NL), dtype=torch.int64, device=dev)
rd, md = torch.empty(n, device=dev), torch.empty(n, device=dev)
slm, tlm = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
sq = torch.zeros(1, dtype=torch.float64, device=dev)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def call():
    assert lib.lt_compare_dev(sl.data_ptr(), sc.data_ptr(), tl.data_ptr(), sr.data_ptr(), tr.data_ptr(), sr.data_ptr(), tr.data_ptr(), n, NL,
                              conf.data_ptr(), rd.data_ptr(), md.data_ptr(), slm.data_ptr(), tlm.data_ptr(), sq.data_ptr(), st) == 0
for _ in range(5): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): call()
e1.record(); torch.cuda.synchronize()
c = conf.cpu().numpy()
want = np.zeros((NL, NL), np.int64); m = lab != 0
np.add.at(want, (np.where(m, lab2, 0), np.where(m, lab, 0)), 1)
print("lt_compare_dev: %.1f us per call (incl. two memsets); confusion matrix equal to numpy: %s; %d distinct pairs" % (
    e0.elapsed_time(e1) / 50 * 1e3, bool(np.array_equal(c, want)), int((c > 0).sum())))
