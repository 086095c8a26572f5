#!/usr/bin/env python3
"""Per-wave wall-clock stamps of k_tsdf_integrate_cols on the default volume (LIDARHIP_DEBUG_TSDF=1): how long the waves
live, how many are alive at a time -- what the launch's duration is made of."""
import ctypes as C, os, sys
os.environ["LIDARHIP_DEBUG_TSDF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import runpy
sys.argv = [sys.argv[0], "2"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "prof_chain.py"), run_name="__main__")
from lidar_transfer_amd import _lib
lib = _lib.load()
nw = 250000
buf = np.zeros(2 * nw, np.uint64)
lib.lt_debug_tsdf_wave_times.argtypes = [C.c_void_p, C.c_int]
assert lib.lt_debug_tsdf_wave_times(buf.ctypes.data_as(C.c_void_p), nw) == 0
t = buf.reshape(nw, 2).astype(np.int64)
start = (t[:, 0] - t[:, 0].min()) / 100.0
dur = t[:, 1] / 100.0
end = start + dur
print("k_tsdf_integrate_cols: %d waves, span %.1f us, sum of wave lives %.0f us = %.1f us on 8192 slots" % (nw, end.max(), dur.sum(), dur.sum() / 8192))
print("  (a workgroup = one chunk of 64 columns, its four waves share the live columns)")
for name, a in (("start", start), ("duration", dur), ("end", end)):
    print("  %-9s mean %8.1f p50 %8.1f p90 %8.1f p99 %8.1f max %8.1f" % (name, a.mean(), *[np.percentile(a, p) for p in (50, 90, 99)], a.max()))
edges = np.linspace(0, end.max(), 17)
alive = [int(((start <= x) & (end > x)).sum()) for x in edges]
print("  waves alive at 16 instants:", alive)
heavy = dur > 5.0
print("  waves living > 5 us: %d (%.1f %%), their lives sum to %.0f us" % (heavy.sum(), 100.0 * heavy.mean(), dur[heavy].sum()))
