#!/usr/bin/env python3
"""The two hierarchy builders node for node: subprocess per builder (LIDARHIP_HIER is read once), nodes4 dumped, compared."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from lidar_transfer_amd import _lib
    from lidar_transfer_amd.raytracer import Scene
    from lidar_transfer_amd.synth import synth_scene
    lib = _lib.load()
    out = {}
    for seed, tris in ((0, 300), (1, 5000), (0, 20000), (2, 200000), (0, 1000000)):
        v, f, c, r = synth_scene(seed, tris)
        sc = Scene(0)
        sc.set_mesh(*[torch.from_numpy(x).cuda() for x in (v, f, c, r)])
        sc.build()
        torch.cuda.synchronize()
        assert lib.lt_debug_nodes4_fill(sc._h, 0xAB) == 0
        sc.build()
        torch.cuda.synchronize()
        n = f.shape[0]
        a = np.empty((n, 32), np.uint32)
        assert lib.lt_debug_nodes4_get(sc._h, a.ctypes.data_as(C.c_void_p), n) == 0
        out[f"s{seed}_{tris}"] = a
        try:
            sc.status()
        except RuntimeError as e:
            print("status:", e)
        sc.close()
    np.savez(sys.argv[2], **out)
    sys.exit(0)
res = {}
for mode in ("seg", "agg"):
    env = dict(os.environ, LIDARHIP_HIER=mode)
    path = f"/tmp/hier_{mode}.npz"
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], check=True, env=env, timeout=600)
    res[mode] = np.load(path)
for k in res["seg"].files:
    a, b = res["seg"][k], res["agg"][k]
    wa, wb = (a != 0xABABABAB).any(1), (b != 0xABABABAB).any(1)
    same_set = np.array_equal(wa, wb)
    both = wa & wb
    diff = np.flatnonzero((a[both] != b[both]).any(1))
    idx = np.flatnonzero(both)[diff]
    print(k, "nodes", a.shape[0], "written seg", int(wa.sum()), "agg", int(wb.sum()), "same set", same_set, "differing among both", len(diff))
    only_a, only_b = np.flatnonzero(wa & ~wb), np.flatnonzero(wb & ~wa)
    print("   only seg:", only_a[:10].tolist(), "only agg:", only_b[:10].tolist())
    for i in idx[:3]:
        print("   node", int(i)); print("     seg", a[i].view(np.float32)[:16], a[i].view(np.int32)[[6, 14, 22, 30]]); print("     agg", b[i].view(np.float32)[:16], b[i].view(np.int32)[[6, 14, 22, 30]])
    if a.shape[0] <= 400:
        print("   root seg", a[0].view(np.int32)[[6, 14, 22, 30]], "agg", b[0].view(np.int32)[[6, 14, 22, 30]])
