"""Helpers of bench.py / tools/bench_chains.py: the compact `roofline` and `cpu_baseline` records of the headline line,
the full ones of the extras file, the PMC look-ups under profiles/, and the size guard of the one stdout line."""
from __future__ import annotations

import glob
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth, 8 XCDs x 4 MiB (MI355X_MICROARCH.md, L2 section)
RANDOM_REQ_CEILING_G = 52.0   # G random 64-byte read requests/s (tools/tlb_probe.hip, profiles/r03)
ATOMIC_CEILING_G = 27.0       # G memory-side atomic requests/s, random cells (tools/atomic_probe.hip, profiles/r05)
ATOMIC_LANE_CEILING_ORDERED_G = 200.0  # lane atomics/s, runs of >= 8 neighbouring lanes per line (profiles/r05/atomic_probe.txt)
# FETCH_SIZE counts requests x 64 B while a request moves a 128-byte line for every pattern of this path
# (tools/fetch_calib.hip -> profiles/r05/fetch_calib.txt): HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE
FETCH_SIZE_FACTOR = 2.0
N_SIMD, SHADER_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; peak engine clock
# SQ_INSTS_VALU counts issued wave64 instructions; a SIMD issues one every ~2 cycles (tools/valu_calib.hip, profiles/r03)
SQ_CYCLES_PER_COUNT = 2.0
VALU_CYCLES_PER_WAVE_INST = 2.0
PCIE_PEAK_GBS = 63.0   # PCIe Gen5 x16 per direction
PCIE_WIRE_GBS = 56.3   # pinned hipMemcpyAsync of 26 MB on this box (tools/pcie_probe.hip)

# ---- algorithmic bytes per unit ----------------------------------------------------
# scatter, k_sc_tris: per triangle 3 indices (12 B); per VERTEX 12 B once; per Moller-Trumbore test one 16-B grid entry
# (L2-resident); per accepted hit one 8-B atomic
SC_B_TRI, SC_B_VERT, SC_B_TEST, SC_B_HIT = 12, 12, 16, 8
# lbvh, k_trace4: one 4-wide node 128 B, one triangle record 48 B; per ray 12 B direction + 44 B out + 40 B hit gather
LB_B_NODE, LB_B_TRI, LB_B_RAY = 128, 48, 12 + 44 + 40

KERNEL_SOURCES = {"scatter": ["lt_scatter.hip", "lt_internal.h", "lt_normalize.h"],
                  "lbvh": ["lt_trace.hip", "lt_build.hip", "lt_internal.h", "lt_normalize.h"],
                  "chain": ["lt_tsdf.hip", "lt_mc.hip", "lt_internal.h"]}
DOMINANT = {"scatter": "k_sc_tris", "lbvh": "k_trace4"}


def fit_line(obj: dict, limit: int) -> str:
    """json.dumps(obj) guaranteed to stay under `limit` bytes: optional keys are dropped, least important first (the
    driver parses ONE stdout line; a line it cannot parse leaves the round unmeasured)."""
    line = json.dumps(obj, separators=(",", ":"))
    for key in ("verification", "speedup_vs_cpu_baseline", "hit_fraction", "scans_per_s"):
        if len(line.encode()) < limit:
            break
        obj = {k: v for k, v in obj.items() if k != key}
        line = json.dumps(obj, separators=(",", ":"))
    if len(line.encode()) >= limit:  # last resort: the contract's keys only
        keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
        obj = {k: obj[k] for k in keep if k in obj}
        if isinstance(obj.get("config"), dict):
            obj["config"] = {"workload": str(obj["config"].get("workload", ""))[:300]}
        for k in ("roofline", "cpu_baseline"):
            if isinstance(obj.get(k), dict):
                obj[k] = {kk: vv for kk, vv in obj[k].items() if not isinstance(vv, (dict, list, str)) or kk in ("bound", "unit", "kind", "kernel")}
        line = json.dumps(obj, separators=(",", ":"))
    return line


def kernel_source_hash(which: str) -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[which]:
        with open(os.path.join(ROOT, "lidar_transfer_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + fh.read())
    return h.hexdigest()[:16]


def _entries(pattern):
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", pattern)), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        for e in doc.get("entries", []):
            yield os.path.relpath(path, ROOT), e


def measured_traffic(args, strategy, spl, kernel=None):
    """HBM bytes per launch of a kernel from the PMC passes committed under profiles/rNN/pmc*.json (tools/pmc_to_json.py:
    2 x FETCH_SIZE + WRITE_SIZE).  An entry counts only for the workload, launch shape AND kernel sources it was collected
    on -- otherwise null, never a stale constant."""
    want = kernel_source_hash(strategy)
    for path, e in _entries("pmc*.json"):
        if (e.get("workload") == args.workload and e.get("strategy") == strategy and not args.target
                and e.get("scans_per_launch") == spl and e.get("kernel_source_hash") == want
                and e.get("kernel") == (kernel or DOMINANT[strategy])):
            return float(e["hbm_bytes_per_launch"]), path, e
    return None, None, None


def measured_ea(args, kernel, spl):
    """TCC_EA0 request counts per launch (tools/ea_to_json.py), valid for this workload, launch shape and kernel sources"""
    want = kernel_source_hash("scatter")
    for path, e in _entries("ea_requests.json"):
        if (e.get("workload") == args.workload and not args.target and e.get("scans_per_launch") == spl
                and e.get("kernel_source_hash") == want and e.get("kernel") == kernel
                and "tcc_ea0_rdreq_per_launch" in e and "tcc_ea0_atomic_per_launch" in e):
            return dict(e, _path=path)
    return None


def chain_pmc(kernel):
    """PMC record of a fusion-chain kernel from profiles/rNN/pmc_chain*.json; only for the current kernel sources"""
    want = kernel_source_hash("chain")
    for path, e in _entries("pmc_chain*.json"):
        if e.get("kernel") == kernel and e.get("kernel_source_hash") == want:
            return dict(e, source=path)
    return None


def roofline_compact(hz, strategy, serial_ms, insitu_ms, step_s):
    """The headline line's roofline record for the dominant kernel.

    scatter / k_sc_tris: `achieved` = HBM-COMPULSORY bytes per launch (what any implementation must pull from HBM: index
    triples 12 B x F, every vertex once 12 B x V, one 8-byte atomic per accepted hit; x scans per launch) / the kernel's
    exclusive duration (HIP events on the launch stream, launches of the timed region's shape back to back on one stream);
    `frac` against 8 TB/s.  `frac_incl_l2_bytes` adds SURVEY 8d's per-test bytes (16 B grid entry per Moller-Trumbore test),
    which are L2-resident reads -- the figure rounds 1-4 called `frac`.  `traffic` = 2 x FETCH_SIZE + WRITE_SIZE per launch
    from the separate --pmc passes under profiles/ (null when the kernel sources changed since they were collected).
    `step_clock.frac_on_step_clock` charges the WHOLE step to this kernel (compulsory bytes of the step / ms_per_step): it
    fits the driver's clock by construction."""
    args = hz.args
    spl = args.batch if strategy == "scatter" else 1
    c = np.mean(np.array(hz.cnt[strategy], dtype=np.float64), axis=0)
    R = hz.R
    if strategy == "scatter":
        comp_scan = hz.n_faces * SC_B_TRI + hz.n_verts * SC_B_VERT + c[2] * SC_B_HIT
        alg_scan = comp_scan + c[1] * SC_B_TEST
    else:
        alg_scan = comp_scan = c[0] * LB_B_NODE + c[1] * LB_B_TRI + R * LB_B_RAY
    comp = comp_scan * spl
    sec = serial_ms * 1e-3
    traffic, src, _ = measured_traffic(args, strategy, spl)
    d = {"kernel": DOMINANT[strategy], "bound": "hbm", "achieved": round(comp / sec / 1e9, 1), "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(comp / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
         "traffic_over_compulsory": round(traffic / comp, 3) if traffic else None, "traffic_source": src,
         "avg_kernel_ms": round(serial_ms, 5), "scans_per_launch": spl,
         "hbm_compulsory_bytes_per_launch": int(comp), "algorithmic_bytes_per_launch": int(alg_scan * spl),
         "frac_incl_l2_bytes": round(alg_scan * spl / sec / 1e9 / HBM_PEAK_GBS, 4),
         "tests_per_ray": round(c[1] / R, 2)}
    if insitu_ms == insitu_ms:
        d["in_situ_avg_kernel_ms"] = round(insitu_ms, 5)   # same launches inside the timed region: NOT exclusive
    if step_s:
        launches = hz.SPS / spl
        d["step_clock"] = {"launches_per_step": launches, "launches_x_avg_kernel_ms": round(launches * serial_ms, 4),
                           "ms_per_step": round(step_s * 1e3, 4),
                           "frac_on_step_clock": round(comp_scan * hz.SPS / step_s / 1e9 / HBM_PEAK_GBS, 4)}
    d["note"] = ("latency-bound at 8 waves/SIMD: no single roof is near (vector issue / random-request / atomic ceilings "
                 "~0.4 each, profiles/rNN/bench_extras.json); exclusive durations of overlapped batches do not add up to "
                 "ms_per_step")
    return d


_CPU_CODE = r"""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, %r)
import numpy as np
from oracle import binding as ob
from lidar_transfer_amd.synth import synth_scene
from lidar_transfer_amd.laserscan import create_rays
wl = json.loads(sys.argv[1]); seed = int(sys.argv[2]); reps = int(sys.argv[3]); kind = sys.argv[4]; budget = float(sys.argv[5])
v, f, c, r = synth_scene(seed, wl["tris"])
rays = create_rays(wl["fov_up"], wl["fov_down"], wl["H"], wl["W"]); org = np.zeros(3, np.float32)
libc = C.CDLL(None)
libc.setvbuf(C.c_void_p.in_dll(libc, "stdout"), None, 2, 0)          # _IONBF: a printf is written when it is made
r_fd, w_fd = os.pipe()
saved = os.dup(1); os.dup2(w_fd, 1); os.close(w_fd)
stamps = []
def reader():
    buf = b""
    while True:
        chunk = os.read(r_fd, 65536)
        if not chunk:
            break
        now = time.perf_counter()
        buf += chunk
        while b"\n" in buf:
            line, buf = buf.split(b"\n", 1)
            stamps.append((now, line.decode(errors="replace")))
th = threading.Thread(target=reader, daemon=True); th.start()
runs = []
t_all = time.perf_counter()
for i in range(reps):
    del stamps[:]
    t0 = time.perf_counter()
    if kind == "port":
        o = ob.oracle_trace(rays, org, v, f, c, r, wl["H"], mode=ob.MODE_REF_BVH, norm=ob.NORM_SSE)
        t1 = time.perf_counter()
        st = o["stats"]
        runs.append({"e2e_s": t1 - t0, "setup_build_s": (st["t_setup_ms"] + st["t_build_ms"]) * 1e-3,
                     "build_ms_printed": st["t_build_ms"], "trace_only_s": st["t_trace_ms"] * 1e-3})
    else:
        ob.ref_trace(rays, org, v, f, c, r, wl["H"], kind=kind)
        t1 = time.perf_counter()
        time.sleep(0.002)                                           # let the reader drain the pipe
        built = [(t, l) for t, l in stamps if "Built BVH" in l]
        rend = [(t, l) for t, l in stamps if "Rendering image" in l]
        run = {"e2e_s": t1 - t0}
        if built:
            try:
                run["build_ms_printed"] = float(built[-1][1].rsplit(" in ", 1)[1].split()[0])
            except (IndexError, ValueError):
                pass
        if rend or built:
            t_split = (rend or built)[-1][0]                       # the trace loop starts right after this line
            run["setup_build_s"] = t_split - t0
            run["trace_only_s"] = t1 - t_split
        runs.append(run)
    if time.perf_counter() - t_all > budget:
        break
os.dup2(saved, 1)
sys.stderr.write("LTBASE " + json.dumps({"runs": runs, "threads": ob.num_threads(), "faces": int(f.shape[0])}) + "\n")
""" % ROOT


def cpu_baseline(workload: dict, seed: int, reps: int):
    """Time the real reference (oracle/_ref, prebuilt from /root/reference) in subprocesses on this host; bounded sample.
    TWO clocks at TWO thread counts: end-to-end `ctrace` (triangle set-up RayTracer.cpp:32-51 + BVH
    build BVH.cpp:143-243 + trace RayTracer.cpp:62-92) and trace-only, at OMP_NUM_THREADS = nproc and = 1.  The split comes
    from the reference ITSELF: it prints "[Statistic] Built BVH ..." (BVH.cpp:125) and "Rendering image ..."
    (RayTracer.cpp:60) on C stdout between its phases; the subprocess timestamps every line on arrival."""
    n_rays = workload["H"] * workload["W"]
    nproc = os.cpu_count() or 1
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        model = "unknown"

    def one(kind, threads, reps_, budget):
        env = dict(os.environ)
        env["OMP_NUM_THREADS"] = str(threads)
        try:
            res = subprocess.run([sys.executable, "-c", _CPU_CODE, json.dumps(workload), str(seed), str(reps_), kind, str(budget)],
                                 capture_output=True, text=True, timeout=300, env=env)
        except subprocess.TimeoutExpired:
            return None
        line = [l for l in res.stderr.splitlines() if l.startswith("LTBASE ")]
        if res.returncode != 0 or not line:
            return None
        info = json.loads(line[0][7:])
        runs = info["runs"]
        best = min(runs, key=lambda r: r["e2e_s"])
        d = {"threads": info["threads"], "runs": len(runs), "faces": info["faces"], "e2e_s": round(best["e2e_s"], 4),
             "e2e_Mrays_s": round(n_rays / best["e2e_s"] / 1e6, 4)}
        tr = [r["trace_only_s"] for r in runs if "trace_only_s" in r]
        if tr:
            d.update(trace_only_s=round(min(tr), 5), trace_only_Mrays_s=round(n_rays / min(tr) / 1e6, 3),
                     setup_build_s=round(min(r["setup_build_s"] for r in runs if "setup_build_s" in r), 4))
        bm = [r["build_ms_printed"] for r in runs if "build_ms_printed" in r]
        if bm:
            d["bvh_build_ms_printed_by_reference"] = min(bm)
        return d

    for kind in ("fast", "strict", "port"):
        if kind != "port" and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", f"libref_{kind}.so")):
            continue
        many = one(kind, nproc, reps, 15.0)
        if not many:
            continue
        single = one(kind, 1, max(2, reps // 3), 12.0)
        lib = f"oracle/_ref/libref_{kind}.so" if kind != "port" else "oracle restatement"
        return {"value": many["e2e_Mrays_s"], "unit": "Mrays/s", "cores": many["threads"],
                "kind": "port" if kind == "port" else "reference",
                "sample": f"{many['runs']} end-to-end ctrace calls (triangle set-up + BVH build + trace) on one "
                          f"{workload['H']}x{workload['W']} scan vs {many['faces']} triangles, min of runs, {lib}, "
                          f"OpenMP threads={many['threads']}; and {single['runs'] if single else 0} calls at 1 thread",
                "s_per_scan": many["e2e_s"], "scans_per_s": round(1.0 / many["e2e_s"], 3), "cpu_model": model,
                "nproc": nproc, "clocks": {"all_threads": many, "one_thread": single}}
    return None


def cpu_baseline_compact(workload, seed, reps):
    """The headline line's record: value / unit / cores / kind / sample + the two clocks in one number each."""
    cb = cpu_baseline(workload, seed, reps)
    if not cb:
        return None
    ck = cb["clocks"]
    one = ck.get("one_thread") or {}
    return {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"],
            "s_per_scan": cb["s_per_scan"], "trace_only_Mrays_s": ck["all_threads"].get("trace_only_Mrays_s"),
            "bvh_build_ms": ck["all_threads"].get("bvh_build_ms_printed_by_reference"),
            "one_thread": {"e2e_Mrays_s": one.get("e2e_Mrays_s"), "trace_only_Mrays_s": one.get("trace_only_Mrays_s")},
            "cpu_model": cb["cpu_model"], "nproc": cb["nproc"]}


def verify_timed_scans(hz, range_all, color_all, K, first=0):
    """Timed scans (first, middle, last) compared BIT FOR BIT with a fresh single-scan render of the same mesh outside the
    clock; timed scan 0 of rank 0 on C2 also against golden F5 (SHA-256 of the real reference's images)."""
    import hashlib
    torch, R = hz.torch, hz.R
    res = {"scans_compared": [], "ok": True}
    lab = hz.scratch[0]["endcolors"].reshape(-1)[:R]
    for sl in sorted({0, K // 2, K - 1}):
        hz.workers[0].set_mesh(*hz.scenes[(first + sl) % len(hz.scenes)])
        o = dict(hz.scratch[0])
        o["endcolors"] = lab
        hz.workers[0].render(hz.raysets[0], hz.origin, out=o, label_image=True)
        torch.cuda.synchronize()
        same = bool(torch.equal(range_all[sl].view(torch.int32), o["range"].view(torch.int32))) and \
            bool(torch.equal(color_all[sl], lab))
        res["scans_compared"].append(sl)
        res["ok"] &= same
    gpath = os.path.join(ROOT, "tests", "golden", "f5_c2_1m_64x2048.npz")
    if hz.args.workload == "C2" and not hz.args.target and hz.rank == 0 and not hz.job and os.path.exists(gpath):
        g = np.load(gpath)
        if int(g["seed"]) == 0 and int(g["n_faces"]) == hz.n_faces and int(g["H"]) == hz.H and int(g["W"]) == hz.W:
            rs_ = hashlib.sha256(range_all[0].cpu().numpy().tobytes()).digest()
            ls_ = hashlib.sha256(color_all[0].cpu().numpy().astype(np.int32).tobytes()).digest()
            gold = rs_ == bytes(g["range_sha256"].tobytes()) and ls_ == bytes(g["label_sha256"].tobytes())
            res["golden_f5_sha256"] = bool(gold)
            res["ok"] = res["ok"] and bool(gold)
    return res

# ------------------------------------------------------------------------------------------ roofline inputs
def count_work(hz, strategies=("scatter",), n_lbvh=12):
    """counting passes outside the clock: candidate bins / nodes, Moller-Trumbore tests and accepted hits per scan"""
    cnt = {"scatter": [], "lbvh": [], "hit_rays": []}
    w = hz.workers[0]
    for i, sc in enumerate(hz.scenes):
        w.set_mesh(*sc)
        o = w.render(hz.raysets[0], hz.origin, out=hz.scratch[0], count=True)
        cnt["scatter"].append((o["stats"]["nodes_visited"], o["stats"]["tris_tested"], o["stats"]["n_hits"]))
        cnt["hit_rays"].append(int((o["range"] > 0).sum().item()))
        if "lbvh" in strategies and (hz.args.strategy == "lbvh" or i < n_lbvh):
            w.build()
            o = w.trace(hz.rays, hz.origin, hz.H, out=hz.scratch[0], count=True)
            cnt["lbvh"].append((o["stats"]["nodes_visited"], o["stats"]["tris_tested"], o["stats"]["n_hits"]))
    hz.torch.cuda.synchronize()
    hz.cnt = cnt
    return cnt

def serial_probe_ms(hz, strategy, n=24):
    """Launches of the timed region's shape (scatter: one lt_scene_render_batch_dev of --batch scans; lbvh: build + trace
    of one scan) back to back on ONE stream, HIP events around the dominant kernel: exclusive durations."""
    import ctypes as C
    from lidar_transfer_amd import _lib
    torch = hz.torch
    lib, vp, evs = _lib.load(), C.c_void_p, []
    B = hz.args.batch if strategy == "scatter" else 1
    st = hz.streams[0]
    org_b = (C.c_float * (3 * B))(*(list(hz.origin) * B))
    arr = lambda vals: (vp * B)(*vals)  # noqa: E731
    with torch.cuda.stream(st):
        for i in range(n + 4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if strategy == "lbvh":
                w = hz.workers[0]
                w.set_mesh(*hz.scenes[i % len(hz.scenes)])
                w.build()
                w.set_probe(e0, e1)
                w.trace(hz.rays, hz.origin, hz.H, out=hz.scratch[0])
            else:
                for j in range(B):
                    hz.workers[j].set_mesh(*hz.scenes[(i * B + j) % len(hz.scenes)])
                hz.workers[0].set_probe(e0, e1)
                outs = {k: arr([hz.scratch[j][k].data_ptr() for j in range(B)]) for k in
                        ("endpoints", "endcolors", "range", "endrem", "tri")}
                _lib.check(lib.lt_scene_render_batch_dev(B, arr([hz.workers[j]._h for j in range(B)]),
                                                         arr([hz.raysets[j]._h for j in range(B)]), org_b,
                                                         outs["endpoints"], outs["endcolors"], outs["range"], outs["endrem"],
                                                         outs["tri"], _lib.LT_TRACE_WRITE_MISSES, vp(st.cuda_stream)),
                           "serial probe")
            evs.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs[4:]]))
