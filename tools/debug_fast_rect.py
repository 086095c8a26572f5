#!/usr/bin/env python3
"""CPU-side check of the per-vertex bin rectangles of lt_scatter.hip (sc_vertex_record / sc_fast_rect) on one case of
tools/stress_scatter.py: brute-force Moller-Trumbore (float64, generous) over all (triangle, ray) pairs of the triangles
that take the fast path, and every accepted pair whose ray lies OUTSIDE the triangle's rectangle is reported.

    python tools/debug_fast_rect.py <case number> [seed]
Only make_case() of the stress tool is used; no GPU."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import types
# make_case needs lidar_transfer_amd.raytracer only by name
spec = importlib.util.spec_from_file_location("stress", os.path.join(ROOT, "tools", "stress_scatter.py"))


def load_make_case():
    src = open(os.path.join(ROOT, "tools", "stress_scatter.py")).read()
    head = src[:src.index("ap = argparse.ArgumentParser()")]
    head = head.replace("import numpy as np, torch", "import numpy as np").replace(
        "from lidar_transfer_amd.raytracer import RaySet, Scene\n", "")
    head = head[:head.index("if __name__ != \"__main__\":")] if "if __name__ != \"__main__\":" in head else head
    ns = {"__name__": "stress_cpu", "__file__": os.path.join(ROOT, "tools", "stress_scatter.py")}
    sys.modules.setdefault("torch", types.ModuleType("torch"))
    exec(compile(head, "stress_scatter.py", "exec"), ns)
    return ns["make_case"]


def main():
    case_no = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    make_case = load_make_case()
    rng = np.random.default_rng(seed)
    for k in range(case_no + 1):
        H, W, up, down, kind, v, f, c, r, origin, rays, rk = make_case(rng)
    print(f"case {case_no}: H={H} W={W} fov=({up:.2f},{down:.2f}) kind={kind} rk={rk:.3f} tris={f.shape[0]} origin={origin}")
    o = np.asarray(origin, np.float32)
    d = rays.astype(np.float64); d /= np.linalg.norm(d, axis=1, keepdims=True)
    phi = np.arctan2(d[:, 1], d[:, 0]); th = np.arctan2(d[:, 2], np.hypot(d[:, 0], d[:, 1]))
    # the bin grid (k_rs_fit / k_rs_keys)
    def az_off(phi0, sc):
        x0 = (phi0 + np.pi) * sc
        return x0 - np.floor(x0 + 0.5)
    devs = []
    for K in (W, W - 1):
        if K < 1: devs.append(np.inf); continue
        sc = K / (2 * np.pi); off = az_off(phi[0], sc)
        x = (phi + np.pi) * sc - off
        devs.append(np.abs(x - np.floor(x + 0.5)).max())
    nb_az = min(max(W, 1), 8192)
    if nb_az >= 5 and devs[1] + 0.01 < devs[0]: nb_az -= 1
    nb_el = min(H, 4096)
    az_scale = nb_az / (2 * np.pi); off = az_off(phi[0], az_scale)
    el_lo, el_hi = th.min(), th.max()
    el_scale = (nb_el - 1) / (el_hi - el_lo) if (nb_el > 1 and el_hi > el_lo) else 0.0
    x = (phi + np.pi) * az_scale - off; y = (th - el_lo) * el_scale
    cx = np.floor(x + 0.5); cy = np.clip(np.floor(y + 0.5), 0, nb_el - 1)
    dev_az, dev_el = np.abs(x - cx).max(), np.abs(y - cy).max()
    ia = cx.astype(int) % nb_az
    print(f"grid nb_az={nb_az} nb_el={nb_el} dev_az={dev_az:.4f} dev_el={dev_el:.4f} el_scale={el_scale:.4f} az_scale={az_scale:.4f}")
    SL = 4e-3
    P = dict(az_mid=np.pi * az_scale - off, el_mid=-el_lo * el_scale, lim_x=0.1 * az_scale, lim_y=0.1 * el_scale,
             kx=0.3 * el_scale / az_scale ** 2, ky=0.3 / el_scale if el_scale else 0, pda=dev_az + SL, pde=dev_el + SL)
    # vertex records
    p = v.astype(np.float32) - o[None]
    p = p.astype(np.float64)
    q = p[:, 0] ** 2 + p[:, 1] ** 2; d2 = q + p[:, 2] ** 2; rho = np.sqrt(q)
    ok = (q >= 0.0025) & (d2 < 1e30) & (np.abs(p[:, 2]) <= 1.96 * rho)
    with np.errstate(all="ignore"):
        gx = np.arctan2(p[:, 1], p[:, 0]) * az_scale + P["az_mid"]; gy = np.arctan2(p[:, 2], rho) * el_scale + P["el_mid"]
        pa = (3e-4 + 2e-4 / np.sqrt(0.9 * q)) * az_scale; pe = (3e-4 + 2e-4 / np.sqrt(0.9 * d2)) * el_scale
    gx[~ok] = 0.0; gy[~ok] = 0.0; pa[~ok] = np.inf; pe[~ok] = 0.0      # (marked by an infinite padding, as the kernel does)
    A, B, C = f[:, 0], f[:, 1], f[:, 2]
    with np.errstate(all="ignore"):
        d1 = gx[B] - gx[A]; d2_ = gx[C] - gx[A]
        d1 -= nb_az * np.rint(d1 / nb_az); d2_ -= nb_az * np.rint(d2_ / nb_az)
        lo = np.minimum(0, np.minimum(d1, d2_)); hi = np.maximum(0, np.maximum(d1, d2_))
        ylo = np.minimum(gy[A], np.minimum(gy[B], gy[C])); yhi = np.maximum(gy[A], np.maximum(gy[B], gy[C]))
        wx, wy = hi - lo, yhi - ylo
        small = (wx <= P["lim_x"]) & (wy <= P["lim_y"])
        PA = np.maximum(pa[A], np.maximum(pa[B], pa[C])) + P["pda"]
        PE = np.maximum(pe[A], np.maximum(pe[B], pe[C])) + P["pde"] + P["kx"] * wx ** 2 + P["ky"] * wy ** 2
        e0 = np.maximum(np.ceil(ylo - PE), 0); e1 = np.minimum(np.floor(yhi + PE), nb_el - 1)
        fa0 = np.ceil(gx[A] + lo - PA); fa1 = np.floor(gx[A] + hi + PA)
    small &= np.maximum(pa[A], np.maximum(pa[B], pa[C])) < 1e30
    print(f"fast-path triangles: {small.sum()} of {f.shape[0]}")
    # brute force over fast triangles x rays (chunked), float64 MT with tolerance
    tri = v[f].astype(np.float64) - o.astype(np.float64)[None, None]
    idx = np.nonzero(small)[0]
    bad = 0
    ray_col = cx; ray_row = cy
    for s0 in range(0, idx.size, 2000):
        ii = idx[s0:s0 + 2000]
        T = tri[ii]
        e1v = T[:, 1] - T[:, 0]; e2v = T[:, 2] - T[:, 0]
        h = np.cross(d[None, :, :], e2v[:, None, :])
        a = np.einsum("tk,trk->tr", e1v, h)
        with np.errstate(all="ignore"):
            inv = 1.0 / a
            svec = -T[:, 0]
            u = np.einsum("tk,trk->tr", svec, h) * inv
            qv = np.cross(svec, e1v)
            vv = np.einsum("rk,tk->tr", d, qv) * inv
            t = np.einsum("tk,tk->t", e2v, qv)[:, None] * inv
        tol = 1e-5
        acc = (np.abs(a) > 1e-7) & (u >= -tol) & (u <= 1 + tol) & (vv >= -tol) & (u + vv <= 1 + tol) & (t > 1e-6)
        ti, ri = np.nonzero(acc)
        for tt, rr in zip(ti, ri):
            g = ii[tt]
            row_ok = e0[g] <= ray_row[rr] <= e1[g]
            c_ = ray_col[rr]
            # column inside [fa0, fa1] modulo nb_az
            col_ok = any(fa0[g] <= c_ + k * nb_az <= fa1[g] for k in (-1, 0, 1)) or (fa1[g] - fa0[g] + 1 >= nb_az)
            if not (row_ok and col_ok):
                bad += 1
                if bad <= 12:
                    print(f"  MISSED tri {g} ray {rr}: row {ray_row[rr]} in [{e0[g]},{e1[g]}]={row_ok}; col {c_} in [{fa0[g]},{fa1[g]}]={col_ok}; "
                          f"ylo {ylo[g]:.4f} yhi {yhi[g]:.4f} PE {PE[g]:.4f} ray y {y[rr]:.4f}; gx {gx[A[g]]+lo[g]:.3f}..{gx[A[g]]+hi[g]:.3f} PA {PA[g]:.4f} ray x {x[rr]:.4f}; "
                          f"d {np.sqrt(d2[A[g]]):.2f} th_deg {np.degrees(np.arctan2(p[A[g],2], rho[A[g]])):.2f}")
    print("missed (triangle, ray) pairs:", bad)


if __name__ == "__main__":
    main()
