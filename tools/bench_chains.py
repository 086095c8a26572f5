#!/usr/bin/env python3
"""tools/bench_chains.py -- the side records of the bench (everything that is NOT the headline line of bench.py):

  roofline_full        the dominant kernel's full record (valu_issue, memory_side, step_clock, one_batch_in_flight, path, whole_path)
  other_strategy       a short run of the LBVH strategy with its own roofline + build phases
  e2e                  PCIe-inclusive clocks: one lt_ctrace call per scan; lt_hostpipe with scans in flight
  fusion_chain[_nscans5]   reset -> integrate -> marching cubes -> render, mesh never leaves HBM; pipelined (3 chains)
  deform_from_points   deform('mesh') + write() from five point clouds; projection record
  mergemesh_from_points    deform('mergemesh') + write() (the reference's default adaption), serial and pipelined
  cpu_baseline         the full record (both clocks, both thread counts)

    python tools/bench_chains.py [--out profiles/r06/bench_extras.json] [bench.py's flags]

Writes ONE JSON document to --out (default gpurun_out/bench_extras.json) and a two-line summary to stdout.  Uses bench.py's
Harness (same inputs, same timed region)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import bench_lib as bl  # noqa: E402
from bench_lib import (ATOMIC_CEILING_G, ATOMIC_LANE_CEILING_ORDERED_G, FETCH_SIZE_FACTOR, HBM_PEAK_GBS, L2_PEAK_GBS,  # noqa: E402
                       LB_B_NODE, LB_B_RAY, LB_B_TRI, N_SIMD, PCIE_PEAK_GBS, PCIE_WIRE_GBS, RANDOM_REQ_CEILING_G, SC_B_HIT,
                       SC_B_TEST, SC_B_TRI, SC_B_VERT, SHADER_GHZ, SQ_CYCLES_PER_COUNT, VALU_CYCLES_PER_WAVE_INST)

PROBE_EVERY = bench.PROBE_EVERY


def main():
    argv = sys.argv[1:]
    out_path = os.path.join(ROOT, "gpurun_out", "bench_extras.json")
    flags = {"--no-e2e": False, "--no-chain": False, "--no-other": False}
    if "--out" in argv:
        i = argv.index("--out")
        out_path = os.path.abspath(argv[i + 1])
        del argv[i:i + 2]
    for f in list(flags):
        if f in argv:
            flags[f] = True
            argv.remove(f)
    args = bench.parse(argv)
    args.no_e2e, args.no_chain, args.no_other = flags["--no-e2e"], flags["--no-chain"], flags["--no-other"]
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    hz = bench.Harness(args)
    import torch
    torch_ = torch  # noqa: F841
    dev, H, W, R, wl, S = hz.dev, hz.H, hz.W, hz.R, hz.wl, hz.S
    world, rank, local_rank = hz.world, hz.rank, hz.local_rank
    workers, raysets, scratch, scenes, streams = hz.workers, hz.raysets, hz.scratch, hz.scenes, hz.streams
    origin, rays, n_faces, n_verts = hz.origin, hz.rays, hz.n_faces, hz.n_verts
    K, Wm, SPS = hz.K, hz.Wm, hz.SPS
    run = hz.run
    def serial_probe_ms(strategy, n=24):
        return bl.serial_probe_ms(hz, strategy, n)

    cnt = bl.count_work(hz, ("scatter",) if args.no_other and args.strategy == "scatter" else ("scatter", "lbvh"))
    hit_ray_counts = cnt["hit_rays"]
    phase = workers[0].build(stats=True) if (args.strategy == "lbvh" or not args.no_other) else {}
    torch.cuda.synchronize()

    def measured_traffic(strategy, spl, kernel=None):
        return bl.measured_traffic(args, strategy, spl, kernel)

    def measured_ea(kernel, spl):
        return bl.measured_ea(args, kernel, spl)

    chain_pmc = bl.chain_pmc

    def roofline(strategy, serial_ms, insitu_ms):
        spl = args.batch if strategy == "scatter" else 1  # scans per launch of the dominant kernel
        c = np.mean(np.array(cnt[strategy], dtype=np.float64), axis=0)
        if strategy == "scatter":
            alg = n_faces * SC_B_TRI + n_verts * SC_B_VERT + c[1] * SC_B_TEST + c[2] * SC_B_HIT
            extra = {"kernel": "k_sc_tris", "bound": "hbm", "mt_tests_per_ray": round(c[1] / R, 2),
                     "candidate_bins_per_triangle": round(c[0] / n_faces, 3)}
        else:
            # k_trace4 walks an L2-resident tree: its HBM traffic is ~12 MB per launch against ~0.5 GB of algorithmic
            # bytes, so "hbm" only says which peak the contract figure is priced against -- what bounds the kernel is
            # the dependent chain of node fetches (`latency` below)
            alg = c[0] * LB_B_NODE + c[1] * LB_B_TRI + R * LB_B_RAY
            extra = {"kernel": "k_trace4", "bound": "hbm", "nodes_per_ray": round(c[0] / R, 2),
                     "tris_per_ray": round(c[1] / R, 2)}
        alg_scan = alg
        alg = alg * spl
        ach = alg / (serial_ms * 1e-3) / 1e9
        traffic, traffic_src, te = measured_traffic(strategy, spl)
        d = {"bound": extra.pop("bound"), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
             "avg_kernel_ms": round(serial_ms, 5),
             "algorithmic_bytes_per_launch": int(alg), "scans_per_launch": spl,
             "algorithmic_bytes_per_scan": int(alg_scan),
             "probe": "HIP events on the launch stream around the dominant kernel, launches of the timed region's shape "
                      "issued back to back on ONE stream right after the timed region (exclusive durations: nothing "
                      "runs beside the kernel; two overlapped batches of the timed region fill each other's tails, so "
                      "launches x avg_kernel_ms may exceed ms_per_step); reproduced under rocprofv3 by `bench.py "
                      "--probe-only` -> profiles/rNN/serial_probe_kernel_stats.csv"}
        if traffic:
            d["traffic_frac_of_peak"] = round(traffic / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            d["traffic_calibration"] = {"hbm_bytes": f"{FETCH_SIZE_FACTOR:g} x FETCH_SIZE + WRITE_SIZE", "factor": FETCH_SIZE_FACTOR,
                                        "source": "profiles/r05/fetch_calib.txt (tools/fetch_calib.hip: 12-byte coalesced triples, "
                                                  "12-byte windowed and random gathers -- this kernel's patterns -- all 2.0)"}
        if strategy == "scatter":
            # What ANY implementation must pull from HBM per scan: the index triples, every vertex once, one 8-byte atomic per
            # accepted hit.  The 16-byte grid entries of the Moller-Trumbore tests (SURVEY.md section 8d charges them as
            # "36 x n_tris") are reads of a ~2 MB bin grid that stays in L2: they are work, not HBM bytes.  `frac` is priced on
            # the compulsory bytes; the figure of rounds 1-4 (all algorithmic bytes against the HBM peak) stays beside it.
            comp_scan = n_faces * SC_B_TRI + n_verts * SC_B_VERT + c[2] * SC_B_HIT
            l2_scan = c[1] * SC_B_TEST
            comp = comp_scan * spl
            d["frac_incl_l2_bytes"] = d["frac"]
            d["achieved_incl_l2_bytes"] = d["achieved"]
            d["achieved"] = round(comp / (serial_ms * 1e-3) / 1e9, 1)
            d["frac"] = round(comp / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            d["bytes_per_scan"] = {"hbm_compulsory": int(comp_scan), "l2_resident": int(l2_scan),
                                   "hbm_compulsory_parts": {"faces_12B": int(n_faces * SC_B_TRI), "vertices_12B": int(n_verts * SC_B_VERT),
                                                            "hit_atomics_8B": int(c[2] * SC_B_HIT)}}
            d["hbm_compulsory_bytes_per_launch"] = int(comp)
            if traffic:
                d["traffic_over_compulsory"] = round(traffic / comp, 3)
            d["frac_of_achievable_copy"] = round(comp / (serial_ms * 1e-3) / 1e9 / 6300.0, 4)
            d["bound_note"] = ("hbm names the peak the contract prices against; at 8 waves/SIMD the kernel is LATENCY-bound: 0.16-0.17 "
                               "of HBM on compulsory bytes, ~0.4 of vector issue, ~0.4 of the random-request / atomic ceilings "
                               "(valu_issue, memory_side below) -- no single roof is near")
        valu = None
        if te and te.get("valu_active_quad_cycles_per_launch"):
            # the issue-bound view (same PMC passes): cycles a SIMD's vector ALU was busy = SQ_ACTIVE_INST_VALU (summed over
            # the chip's 1024 SIMDs) x SQ_CYCLES_PER_COUNT / 1024, against the launch's duration at the peak clock
            busy = te["valu_active_quad_cycles_per_launch"] * SQ_CYCLES_PER_COUNT / N_SIMD
            valu = {"wave_insts_per_launch": te["valu_wave_insts_per_launch"],
                    "busy_cycles_per_simd": int(busy), "busy_ms_at_peak_clock": round(busy / SHADER_GHZ / 1e6, 5),
                    "frac_of_kernel_time": round(busy / SHADER_GHZ / 1e6 / serial_ms, 4),
                    "note": "SQ_INSTS_VALU of the same launches (profiles pmc.json) x "
                            f"{SQ_CYCLES_PER_COUNT:g} cycles per wave64 instruction (tools/valu_calib.hip, "
                            "profiles/r03/valu_calib.txt: SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU, it is a count) / 1024 "
                            f"SIMDs at {SHADER_GHZ} GHz (the chip sustains 2.0-2.2 under load): the share of the kernel's "
                            "time a SIMD needs to ISSUE its vector work at the peak rate"}
            d["valu_issue"] = valu
        if strategy == "scatter":
            # the REQUEST view of the memory side (last block): TCC_EA0 counters of the same launches
            # (profiles/rNN/ea_requests.json, keyed like pmc.json) against the two ceilings the probes measured
            ea = measured_ea("k_sc_tris", spl)
            if ea:
                rd, at = ea["tcc_ea0_rdreq_per_launch"], ea["tcc_ea0_atomic_per_launch"]
                sec = serial_ms * 1e-3
                d["memory_side"] = {
                    "ea_read_requests_per_launch": rd, "ea_atomics_per_launch": at,
                    "read_requests_G_per_s": round(rd / sec / 1e9, 2), "atomics_G_per_s": round(at / sec / 1e9, 2),
                    "random_read_ceiling_G_per_s": RANDOM_REQ_CEILING_G, "atomic_ceiling_G_per_s": ATOMIC_CEILING_G,
                    "frac_of_random_read_ceiling": round(rd / sec / 1e9 / RANDOM_REQ_CEILING_G, 4),
                    "frac_of_atomic_ceiling": round(at / sec / 1e9 / ATOMIC_CEILING_G, 4),
                    "note": "every device-scope atomic is executed at the memory side (private L2s per XCD): one per accepted "
                            "hit; ceilings: tools/tlb_probe.hip (52 G random 64-byte requests/s with the chip full of "
                            "chains, any parallelism per lane), tools/atomic_probe.hip (27 G non-returning atomicMin/s on "
                            "an image-sized region) -> profiles/r03/tlb_probe.txt, atomic_probe.txt",
                    "source": ea["_path"]}
        if strategy == "lbvh":
            # The contract figure above prices the algorithmic bytes against HBM, but the tree is L2-resident (counter
            # traffic ~ 1/40 of the algorithmic bytes): the HBM view is kept as a sub-record and the block's bound / frac
            # name what the kernel is really up against -- vector issue when the counters are at hand, else L2 bandwidth.
            d["hbm"] = {"algorithmic_frac_of_hbm_peak": d["frac"], "traffic": traffic,
                        "traffic_achieved_GBs": round(traffic / (serial_ms * 1e-3) / 1e9, 1) if traffic else None,
                        "traffic_frac_of_peak": d.get("traffic_frac_of_peak"),
                        "note": "algorithmic bytes are served by L1 / L2; the HBM counters see only the cold misses"}
            d["l2"] = {"achieved": round(ach, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(ach / L2_PEAK_GBS, 4)}
            if valu:
                d.update(bound="valu", achieved=round(valu["wave_insts_per_launch"] / (serial_ms * 1e-3) / 1e9, 2),
                         peak=round(N_SIMD * SHADER_GHZ / VALU_CYCLES_PER_WAVE_INST, 1), unit="G wave-instructions/s",
                         frac=valu["frac_of_kernel_time"])
            else:
                d.update(bound="l2", achieved=d["l2"]["achieved"], peak=L2_PEAK_GBS, frac=d["l2"]["frac"])
        if insitu_ms == insitu_ms:
            base = d.get("hbm_compulsory_bytes_per_launch", alg)   # (scatter: the compulsory bytes, like `frac`)
            d["in_situ"] = {"avg_kernel_ms": round(insitu_ms, 5),
                            "achieved": round(base / (insitu_ms * 1e-3) / 1e9, 1),
                            "frac": round(base / (insitu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "note": f"the same launches inside the timed region (every {PROBE_EVERY}th sampled), where "
                                    "several launches overlap: the duration is NOT exclusive"}
        if strategy == "lbvh":
            # latency roofline of the traversal: a quad's visits are a dependent chain, one L2 round trip each
            visits = c[0] / R + c[1] / R / 4.0  # node steps + leaf steps (a leaf step tests up to 4 triangles)
            l2_ns = 200 / 2.4  # ~200 cycles L2 hit (MI355X_MICROARCH.md) at 2.4 GHz
            floor_ms = visits * l2_ns * 1e-6
            d["latency"] = {"dependent_steps_per_ray": round(visits, 1), "l2_hit_ns": round(l2_ns, 1),
                            "chain_floor_ms": round(floor_ms, 5), "frac": round(floor_ms / serial_ms, 4),
                            "note": "dependent node / leaf fetches x L2 hit latency = the shortest a ray's walk can "
                                    "be; with enough rays resident the launch could approach it"}
        d.update(extra)
        return d

    def isolated_kernel_ms(strategy, n=24):
        """Outside the clock: the dominant kernel alone on an otherwise idle GPU, ONE scan per launch."""
        # back to back on ONE stream (launches of one stream do not overlap), one synchronisation at the end: a
        # host round trip between launches lets the GPU drop its clocks and measures that instead
        w, evs = workers[0], []
        with torch.cuda.stream(streams[0]):
            for i in range(n + 8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w.set_mesh(*scenes[i % len(scenes)])
                if strategy == "lbvh":
                    w.build()
                    w.set_probe(e0, e1)
                    w.trace(rays, origin, H, out=scratch[0])
                else:
                    w.set_probe(e0, e1)
                    w.render(raysets[0], origin, out=scratch[0])
                evs.append((e0, e1))
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs[8:]]))

    def e2e_host_call(n_calls=10):
        """The PCIe-inclusive clock (SURVEY.md section 8d "end-to-end"): the drop-in call exactly as the reference's
        throw_rays_at_mesh issues it (fusion_lidar.py:434-451) -- mesh, rays and pre-zeroed images in pageable HOST
        numpy arrays, C_Trace uploads, renders, downloads; one scan per call, nothing overlapped."""
        from lidar_transfer_amd.raytracer import C_Trace
        v, f, c, r = [np.ascontiguousarray(x.cpu().numpy()).reshape(-1) for x in scenes[0]]
        hr = np.ascontiguousarray(rays.cpu().numpy()).reshape(-1)
        org = np.asarray(origin, np.float32)
        ts = []
        for i in range(n_calls + 2):
            ep = np.zeros(3 * R, np.float32); ec = np.zeros(3 * R, np.int32)
            rg = np.zeros(R, np.float32); rm = np.zeros(R, np.float32)
            t = time.perf_counter()
            C_Trace(hr, org, v, f, c, r, ep, ec, rg, rm, H, W)
            ts.append(time.perf_counter() - t)
        t = float(np.median(ts[2:]))
        h2d = v.nbytes + f.nbytes + c.nbytes + r.nbytes + hr.nbytes + ep.nbytes + ec.nbytes + rg.nbytes + rm.nbytes
        d2h = ep.nbytes + ec.nbytes + rg.nbytes + rm.nbytes
        return {"what": "lt_ctrace drop-in call: host mesh + rays + pre-zeroed images in, images out, one scan per call, "
                        "pageable memory, no overlap (fusion_lidar.py:434-451)",
                "ms_per_scan": round(t * 1e3, 4), "value": round(R / t / 1e6, 2), "unit": "Mrays/s",
                "scans_per_s": round(1.0 / t, 1), "h2d_bytes": int(h2d), "d2h_bytes": int(d2h),
                "pcie": {"bound": "pcie gen5 x16", "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                         "achieved": round(max(h2d, d2h) / t / 1e9, 2),
                         "frac": round(max(h2d, d2h) / t / 1e9 / PCIE_PEAK_GBS, 4), "measured_wire_GBs": PCIE_WIRE_GBS,
                         "note": "the larger direction's bytes / call time (the link is full duplex)"},
                "hits": int((rg > 0).sum())}

    def chain_roofline(kernels, compulsory, phase_ms, what):
        """Roofline record of one phase of the fusion chain: compulsory bytes (what ANY implementation must move: the voxels
        / mesh elements written, the fields read for them, the images) against the HBM peak over the phase's measured time,
        plus -- from the committed PMC passes -- counter traffic and the vector-issue share of the phase's dominant kernel."""
        d = {"kernels": kernels, "bound": "valu", "compulsory_bytes": int(compulsory), "what_is_counted": what,
             "phase_ms": round(phase_ms, 4),
             "hbm": {"achieved": round(compulsory / (phase_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(compulsory / (phase_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        e = chain_pmc(kernels[0])
        if e:
            busy = e["valu_active_quad_cycles_per_launch"] * SQ_CYCLES_PER_COUNT / N_SIMD
            kms = e["avg_kernel_ns"] * 1e-6
            d.update(achieved=round(e["valu_wave_insts_per_launch"] / (kms * 1e-3) / 1e9, 2),
                     peak=round(N_SIMD * SHADER_GHZ / VALU_CYCLES_PER_WAVE_INST, 1), unit="G wave-instructions/s",
                     frac=round(busy / SHADER_GHZ / 1e6 / kms, 4), traffic=e["hbm_bytes_per_launch"],
                     dominant_kernel={"kernel": kernels[0], "avg_kernel_ms": round(kms, 5),
                                      "valu_wave_insts_per_launch": e["valu_wave_insts_per_launch"],
                                      "traffic_over_compulsory": round(e["hbm_bytes_per_launch"] / max(compulsory, 1), 2),
                                      "source": e["source"]})
        else:
            d.update(achieved=None, peak=None, unit="G wave-instructions/s", frac=None, traffic=None,
                     note="no PMC record for the current kernel sources under profiles/ (tools/r03_profile.sh)")
        return d

    def fusion_chain(n=6, nscans=1):
        """Upstream + hot path without the mesh ever leaving HBM (SURVEY.md section 8f-1/2 + 8a): per output scan
        reset the TSDF volume, integrate `nscans` observations (fusion_lidar.py:252-287; the reference's `mesh` adaption
        fuses `number_of_scans` range images, all re-projected into the primary pose, into ONE volume --
        laserscan.py:874-903), marching cubes on the device (:403-424), render the target sensor's image from the mesh
        where it was written.  Volume = the reference's default voxel_bounds at 5 cm (config/lidar_transfer.yaml:
        2000 x 2000 x 200 voxels, 4 x 3.2 GB)."""
        import ctypes as C
        from lidar_transfer_amd import _lib
        from lidar_transfer_amd.fusion import DeviceMesh, TSDFVolume
        lib = _lib.load()
        if torch.cuda.get_device_properties(dev).total_memory < 60 * 2**30:
            return None
        vp = C.c_void_p
        w = workers[0]
        w.set_mesh(*scenes[0])
        o = w.render(raysets[0], origin)   # the observation: this very sensor looking at scene 0
        torch.cuda.synchronize()
        lab = o["endcolors"][:, 2].reshape(H, W).float().contiguous()
        folded0 = (lab * 65536.0).contiguous()   # label in channel 0 (laserscan.py:893-895), folded as fusion_lidar.py:261-264
        depth0 = o["range"].reshape(H, W).contiguous()
        remi = o["endrem"].reshape(H, W).contiguous()
        # observations 1 .. nscans - 1: the neighbouring scans of the reference are re-projected into the primary pose
        # (laserscan.py:876-879), i.e. nearly the same range image with centimetre noise and holes where the other pose
        # did not see the surface; the labels occasionally differ (the class-aware branch's "other class" path)
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234)
        obs = [(folded0, depth0, remi)]
        for k in range(1, nscans):
            noise = (torch.rand((H, W), device=dev, generator=gen) - 0.5) * 0.04
            hole = torch.rand((H, W), device=dev, generator=gen) < 0.05
            d_k = torch.where(hole | (depth0 == 0), torch.zeros_like(depth0), depth0 + noise).contiguous()
            flip = torch.rand((H, W), device=dev, generator=gen) < 0.02
            f_k = torch.where(flip, torch.full_like(folded0, 50.0 * 65536.0), folded0).contiguous()
            obs.append((f_k, d_k, remi))
        vol = TSDFVolume(np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]]), 0.05, wl["fov_up"], wl["fov_down"])
        mesh = DeviceMesh(local_rank)
        st = torch.cuda.current_stream(dev)
        sp = vp(st.cuda_stream)
        org = (C.c_float * 3)(*origin)
        out = scratch[0]
        obs_c = (vp * len(obs))(*[o_[0].data_ptr() for o_ in obs])
        obs_d = (vp * len(obs))(*[o_[1].data_ptr() for o_ in obs])
        obs_r = (vp * len(obs))(*[o_[2].data_ptr() for o_ in obs])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ms = np.zeros((n, 4))
        t_wall = []
        for i in range(n + 1):
            t0 = time.perf_counter()
            ev[0].record()
            _lib.check(lib.lt_tsdf_reset(vol._h, sp), "reset")
            ev[1].record()
            # all observations of the fresh volume in ONE call (lt_tsdf_integrate_multi_dev: one fused pass, bit-identical to
            # one lt_tsdf_integrate_dev per observation -- tests/test_tsdf_gpu.py)
            _lib.check(lib.lt_tsdf_integrate_multi_dev(vol._h, len(obs), obs_c, obs_d, obs_r, H, W, 1.0, _lib.LT_TSDF_MERGE, sp),
                       "integrate")
            ev[2].record()
            _lib.check(lib.lt_tsdf_extract_mesh_dev(vol._h, mesh._h, sp, None), "marching cubes")
            ev[3].record()
            _lib.check(lib.lt_scene_set_mesh(w._h, mesh._h), "set mesh")
            _lib.check(lib.lt_scene_render_dev(w._h, raysets[0]._h, org, out["endpoints"].data_ptr(),
                                               out["endcolors"].data_ptr(), out["range"].data_ptr(),
                                               out["endrem"].data_ptr(), out["tri"].data_ptr(),
                                               _lib.LT_TRACE_WRITE_MISSES, sp, None), "render")
            ev[4].record()
            torch.cuda.synchronize()
            if i > 0:
                t_wall.append(time.perf_counter() - t0)
                ms[i - 1] = [ev[k].elapsed_time(ev[k + 1]) for k in range(4)]
        hits_c = int((out["range"] > 0).sum().item())
        nv, nf = mesh.n_verts, mesh.n_faces
        t = float(np.median(t_wall))
        m = np.median(ms, axis=0)
        nvox = int(np.prod(vol._vol_dim))
        # voxels the fusion wrote (outside the clock): tsdf left its initial 1 or the weight its initial 0
        tv, wv, _, _ = vol.get_volume_tensors()
        n_written = 0
        for x0 in range(0, tv.shape[0], 250):   # (in slabs: the masks of the whole volume would be 1.6 GB)
            n_written += int(((tv[x0:x0 + 250] != 1) | (wv[x0:x0 + 250] != 0)).sum().item())
        # how many of this very volume's active cells are one of Lewiner's AMBIGUOUS cases (3, 4, 6, 7, 10, 12, 13: the cell's
        # eight values, not its signs, pick the tiling -- lt_mc.hip, lw_select); outside the clock.  The device's case index
        # -> Lewiner's case: LT_LWC_CASE of the generated table header.
        mc_cases = None
        try:
            import re
            hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lidar_transfer_amd", "csrc",
                                    "lt_mc_lewiner_table.h")).read()
            lw_case = np.array([int(x) for x in re.search(r"LT_LWC_CASE\[256\] = \{([^}]*)\}", hdr).group(1).split(",")])
            hist = np.zeros(256, np.int64)
            X = tv.shape[0]
            for x0 in range(0, X - 1, 100):
                ins = ~(tv[x0:min(x0 + 101, X)] > 0)     # the sign bit: NOT above the level
                if not bool(ins.any()):
                    continue
                sx, sy, sz = ins.shape
                idx = torch.zeros((sx - 1, sy - 1, sz - 1), dtype=torch.int32, device=dev)
                for c_ in range(8):
                    dx, dy, dz = c_ & 1, (c_ >> 1) & 1, (c_ >> 2) & 1
                    idx += ins[dx:sx - 1 + dx, dy:sy - 1 + dy, dz:sz - 1 + dz].to(torch.int32) << c_
                hist += torch.bincount(idx.reshape(-1), minlength=256).cpu().numpy()
                del ins, idx
            act = int(hist[1:255].sum())
            amb = int(sum(int(hist[c_]) for c_ in range(1, 255) if lw_case[c_] in (3, 4, 6, 7, 10, 12, 13)))
            mc_cases = {"active_cells": act, "ambiguous_cells": amb, "ambiguous_share": round(amb / max(act, 1), 5),
                        "by_lewiner_case": {str(k): int(hist[lw_case == k].sum()) for k in range(1, 15)},
                        "note": "the mesh is scikit-image 0.18.3's (Lewiner): vertices and face stream equal to the reference's "
                                "get_mesh on golden F10, render bit-identical (tests/test_pin_f10_f11_gpu.py)"}
        except Exception as e:  # noqa: BLE001
            mc_cases = {"error": repr(e)[:200]}
        del tv, wv
        mesh.close()
        vol.close()
        # compulsory bytes: integrate -- every written voxel's four fields out (and in again for the observations after the
        # first), plus the three images per observation; marching cubes -- the mesh out (verts 12 + colors 12 + rem 4 B per
        # vertex, 12 B per face), two tsdf samples + colour + remission in per vertex, one sign bit per voxel of the written
        # columns (~ the written voxels' words, 1/8 B each -- negligible)
        comp_int = n_written * 16 * (2 * nscans - 1) + nscans * 3 * R * 4
        comp_mc = nv * (28 + 16) + nf * 12
        rec = {"what": f"per output scan: reset 2000x2000x200 TSDF volume -> integrate {nscans} 64x2048 observation"
                       f"{'s' if nscans > 1 else ''} -> marching "
                       "cubes on the device -> render the target image; the mesh never leaves HBM (no PCIe between "
                       "fusion and range image)",
               "observations": nscans,
               "parity": {"integrate": "numpy branch: volumes equal to the reference's own integrate() on goldens F8 / F13 / F14; CUDA branch: "
                                       "bit-identical to the reference's kernel TEXT compiled by hipcc for gfx950 (a stand-in toolchain: "
                                       "pins the text, not a CUDA run; tests/test_tsdf_ref_kernel_gpu.py, tests/stress_tsdf_ref.py)",
                          "marching_cubes": "the arrays of the reference's get_mesh run with the real scikit-image 0.18.3 (goldens F10 "
                                            "/ F10b; tests/test_pin_f10_f11_gpu.py, tests/stress_mc.py, full size: "
                                            "tests/stress_mc_full.py)",
                          "render": "bit-identical to the reference raytracer's image of that mesh (F10; full size: 131 066 of "
                                    "131 072 pixels, profiles/r04/mc_full_size.txt)"},
               "ms_per_scan": round(t * 1e3, 3), "scans_per_s": round(1.0 / t, 1), "value": round(R / t / 1e6, 2),
               "unit": "Mrays/s", "voxels": nvox, "voxels_written": n_written, "mesh_verts": nv, "mesh_faces": nf,
               "hit_fraction": round(hits_c / R, 4), "marching_cubes_cases": mc_cases,
               "phase_ms": {"reset": round(float(m[0]), 3), "integrate": round(float(m[1]), 3),
                            "marching_cubes": round(float(m[2]), 3), "render": round(float(m[3]), 3)},
               "roofline": {
                   "integrate": chain_roofline(["k_tsdf_integrate_pix", "k_tsdf_integrate_written", "k_tsdf_integrate_quirk",
                                                "k_tsdf_dct"], comp_int,
                                               float(m[1]), "written voxels x 16 B out (+ in again after the first "
                                               "observation) + 3 images per observation"),
                   "marching_cubes": chain_roofline(["k_mc_emit_batch", "k_mc_words", "k_mc_amb", "k_mc_compact", "k_mc_clear",
                                                     "k_mc_scan1", "k_mc_scan2"], comp_mc, float(m[2]),
                                                    "mesh out (28 B per vertex, 12 B per face) + 16 B of field samples "
                                                    "in per vertex")}}
        return rec

    def deform_from_points(nscans=5, n=8):
        """The reference's REAL loop body from point clouds, composed and timed (laserscan.py:863-918 + :1121-1178): per output
        scan `nscans` source clouds (~120 k points each, float64 as after apply_pose) -> do_range_projection_new +
        do_label_projection_new per cloud (ONE lt_range_projection_batch_dev call) -> fresh 2000x2000x200 volume, integrate
        x nscans -> marching cubes -> ray cast of the target sensor -> write(): filter + pack the .bin / .label bytes
        (lt_pack_scan_dev).  Nothing leaves HBM but the mesh sizes and the number of packed points.  `verified`: the source
        images equal the single-cloud call's (lt_range_projection_dev, pinned to the reference's goldens), and the target
        images + packed bytes equal the step-by-step API run from those images."""
        import ctypes as C
        from lidar_transfer_amd import _lib
        from lidar_transfer_amd.deform import DeviceDeform
        if torch.cuda.get_device_properties(dev).total_memory < 60 * 2**30 or args.target:
            return None
        lib = _lib.load()
        vp = C.c_void_p
        w = workers[0]
        w.set_mesh(*scenes[0])
        o = w.render(raysets[0], origin)
        torch.cuda.synchronize()
        hit = o["tri"] >= 0
        p0 = o["endpoints"][hit].double()
        l0 = o["endcolors"][hit][:, 2].contiguous().to(torch.int32)
        r0 = o["endrem"][hit].contiguous()
        gen = torch.Generator(device=dev)
        gen.manual_seed(4321)
        clouds = []
        for k in range(nscans):   # the neighbouring scans, re-projected into the primary pose: the same surfaces, centimetre noise, holes
            if k == 0:
                clouds.append((p0.contiguous(), r0, l0))
                continue
            keep = torch.rand(p0.shape[0], device=dev, generator=gen) > 0.05
            scale = 1.0 + (torch.rand((int(keep.sum().item()), 1), device=dev, generator=gen, dtype=torch.float64) - 0.5) * 0.001
            lk = l0[keep].clone()
            flip = torch.rand(lk.shape[0], device=dev, generator=gen) < 0.02
            lk[flip] = 50
            clouds.append(((p0[keep] * scale).contiguous(), r0[keep].contiguous(), lk.contiguous()))
        n_pts = [int(c[0].shape[0]) for c in clouds]
        bnds = np.array([[-50.0, 50.0], [-50.0, 50.0], [-5.0, 5.0]])
        dd = DeviceDeform((H, W, wl["fov_up"], wl["fov_down"]), (H, W, wl["fov_up"], wl["fov_down"]), bnds, 0.05,
                          device=local_rank)
        st = torch.cuda.current_stream(dev)
        sp = vp(st.cuda_stream)
        org = (C.c_float * 3)(*origin)
        FLG = _lib.LT_TRACE_WRITE_MISSES | _lib.LT_TRACE_LABEL_IMAGE
        out = dd.scene.alloc_outputs(R, label_image=True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        ms = np.zeros((n, 6))
        t_wall = []
        packed = None
        src_keep = None  # (the source images are allocated once: a production caller keeps its buffers too)
        for i in range(n + 2):
            t0 = time.perf_counter()
            ev[0].record()
            src = dd.projector.project(clouds, dd.fov_up, dd.fov_down, H, W, new=True, remove=True, out=src_keep,
                                       outputs=("range", "rem", "label_folded"), stream=st)
            src_keep = src
            ev[1].record()
            _lib.check(lib.lt_tsdf_reset(dd.vol._h, sp), "reset")
            ev[2].record()
            oc = (vp * nscans)(*[s_["label_folded"].data_ptr() for s_ in src])
            od = (vp * nscans)(*[s_["range"].data_ptr() for s_ in src])
            orr = (vp * nscans)(*[s_["rem"].data_ptr() for s_ in src])
            _lib.check(lib.lt_tsdf_integrate_multi_dev(dd.vol._h, nscans, oc, od, orr, H, W, 1.0, _lib.LT_TSDF_MERGE, sp),
                       "integrate")
            ev[3].record()
            _lib.check(lib.lt_tsdf_extract_mesh_dev(dd.vol._h, dd.mesh_obj._h, sp, None), "marching cubes")
            ev[4].record()
            _lib.check(lib.lt_scene_set_mesh(dd.scene._h, dd.mesh_obj._h), "set mesh")
            _lib.check(lib.lt_scene_render_dev(dd.scene._h, dd.rayset._h, org, out["endpoints"].data_ptr(),
                                               out["endcolors"].data_ptr(), out["range"].data_ptr(),
                                               out["endrem"].data_ptr(), out["tri"].data_ptr(), FLG, sp, None), "render")
            ev[5].record()
            packed = dd._pack(out["endpoints"], False, out["endrem"], out["endcolors"], None, R, st)
            ev[6].record()
            torch.cuda.synchronize()
            if i > 1:
                t_wall.append(time.perf_counter() - t0)
                ms[i - 2] = [ev[k].elapsed_time(ev[k + 1]) for k in range(6)]
        # the same chain as ONE DeviceDeform.mesh() call (lt_range_projection_batch_dev -> lt_fusion_scan_dev -> lt_pack_scan_dev)
        t_one = []
        for i in range(n + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = dd.mesh(clouds, origin)
            torch.cuda.synchronize()
            if i:
                t_one.append(time.perf_counter() - t0)
        # ---- verification against the step-by-step API -----------------------------------------------------------------
        ok = True
        from lidar_transfer_amd.laserscan import SemLaserScan
        for k, (pk, rk, lk) in enumerate(clouds):
            s = SemLaserScan(H, W, 300, {})
            s.points, s.remissions, s.label = pk.cpu().numpy(), rk.cpu().numpy(), lk.cpu().numpy().astype(np.uint32)
            s.do_range_projection_new(dd.fov_up, dd.fov_down, remove=True)   # the single-cloud call (host arrays in and out)
            ok = ok and np.array_equal(src[k]["range"].cpu().numpy().view(np.int32), s.range_image.view(np.int32))
            ok = ok and np.array_equal(src[k]["rem"].cpu().numpy().view(np.int32), s.proj_remissions.view(np.int32))
            ok = ok and np.array_equal(src[k]["label_folded"].cpu().numpy(),
                                       np.floor(s.label_image[:, :, 0].astype(np.float32) * 256 * 256))
        same_call = bool(torch.equal(got["range"].reshape(-1).view(torch.int32), out["range"].view(torch.int32))) and \
            bool(torch.equal(got["label"].reshape(-1), out["endcolors"])) and bool(torch.equal(got["bin"], packed[0])) and \
            bool(torch.equal(got["label_file"], packed[1]))
        ok = ok and same_call
        hits_c = int((out["range"] > 0).sum().item())
        nv, nf = dd.mesh_obj.n_verts, dd.mesh_obj.n_faces
        n_packed = int(packed[0].shape[0])
        # ---- projection alone: `reps` batch calls back to back (events on the launch stream) ------------------------------
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        outs_keep = [dict(s_) for s_ in src]
        for _ in range(3):
            dd.projector.project(clouds, dd.fov_up, dd.fov_down, H, W, new=True, remove=True, out=outs_keep,
                                 outputs=("range", "rem", "label_folded"), stream=st)
        e0.record()
        for _ in range(reps):
            dd.projector.project(clouds, dd.fov_up, dd.fov_down, H, W, new=True, remove=True, out=outs_keep,
                                 outputs=("range", "rem", "label_folded"), stream=st)
        e1.record()
        torch.cuda.synchronize()
        proj_ms = e0.elapsed_time(e1) / reps
        # the single-cloud device call for comparison (four kernels + a host synchronisation per cloud)
        kept = C.c_int(0)
        t_single = []
        for rep_ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for (pk, rk, lk), s_ in zip(clouds, outs_keep):
                _lib.check(lib.lt_range_projection_dev(pk.data_ptr(), 1, rk.data_ptr(), lk.data_ptr(), int(pk.shape[0]),
                                                       float(dd.fov_up), float(dd.fov_down), H, W, None, 0,
                                                       _lib.LT_PROJ_NEW | _lib.LT_PROJ_REMOVE, None, 0, None, None, None,
                                                       None, None, None, None, None, None, s_["range"].data_ptr(), None,
                                                       s_["rem"].data_ptr(), None, None, None, 0.0, -1.0, 0.0,
                                                       C.byref(kept), sp), "lt_range_projection_dev")
            torch.cuda.synchronize()
            if rep_ > 1:
                t_single.append(time.perf_counter() - t0)
        filled = sum(int((s_["range"] > 0).sum().item()) for s_ in outs_keep)
        tot_pts = sum(n_pts)
        alg = tot_pts * (24 + 8) + nscans * R * (16 + 12) + filled * (24 + 8)
        dd.close()
        # ... and with three output scans in flight (FusionScanPipeline.submit_clouds: a projector, volume, mesh, scene, stream
        # and host thread per chain; projection + fusion chain per scan, no write())
        pipelined = None
        try:
            import gc
            from lidar_transfer_amd.pipeline import FusionScanPipeline
            if torch.cuda.get_device_properties(dev).total_memory >= 100 * 2**30:
                with FusionScanPipeline(bnds, 0.05, wl["fov_up"], wl["fov_down"], rays, H, chains=3, device=local_rank,
                                        label_image=True, source_hw=(H, W)) as pipe:
                    for tk_ in [pipe.submit_clouds(clouds, inputs_ready=True) for _ in range(12)]:
                        pipe.wait(tk_)
                    bufs = [pipe._chains[0]["scene"].alloc_outputs(R, label_image=True) for _ in range(36)]
                    torch.cuda.synchronize()
                    gc.collect()
                    gc.disable()
                    try:
                        tp0 = time.perf_counter()
                        tks = [pipe.submit_clouds(clouds, out=b_, inputs_ready=True) for b_ in bufs]
                        outs_p = [pipe.wait(tk_) for tk_ in tks]
                        dtp = time.perf_counter() - tp0
                    finally:
                        gc.enable()
                    okp = all(bool(torch.equal(o_["range"].view(torch.int32), out["range"].view(torch.int32))) and
                              bool(torch.equal(o_["endcolors"], out["endcolors"])) for o_ in outs_p)
                    pipelined = {"chains_in_flight": 3, "output_scans": len(bufs), "ms_per_output_scan": round(dtp / len(bufs) * 1e3, 4),
                                 "output_scans_per_s": round(len(bufs) / dtp, 1), "verified": bool(okp),
                                 "api": "lidar_transfer_amd.pipeline.FusionScanPipeline.submit_clouds (projection + fusion "
                                        "chain per scan; no write())"}
        except Exception as e:  # noqa: BLE001
            pipelined = {"error": repr(e)[:200]}
        m = np.median(ms, axis=0)
        t = float(np.median(t_wall))
        return {"what": f"deform('mesh') + write() per output scan from {nscans} float64 point clouds of {n_pts[0]}..{min(n_pts)} points "
                        f"(laserscan.py:863-918, :1121-1178): batched z-min projection -> reset 2000x2000x200 volume -> integrate "
                        f"x{nscans} -> marching cubes -> render {H}x{W} -> pack .bin/.label bytes; all in HBM",
                "observations": nscans, "points_per_scan": n_pts, "ms_per_output_scan": round(t * 1e3, 3),
                "ms_per_output_scan_one_call": round(float(np.median(t_one)) * 1e3, 3),
                "output_scans_per_s": round(1.0 / t, 1),
                "phase_ms": {"projection": round(float(m[0]), 4), "reset": round(float(m[1]), 4),
                             "integrate": round(float(m[2]), 4), "marching_cubes": round(float(m[3]), 4),
                             "render": round(float(m[4]), 4), "pack": round(float(m[5]), 4)},
                "mesh_verts": nv, "mesh_faces": nf, "hit_fraction": round(hits_c / R, 4), "points_written": n_packed,
                "verified": bool(ok), "pipelined": pipelined,
                "projection": {"ms": round(proj_ms, 4), "clouds_per_call": nscans, "us_per_cloud": round(proj_ms * 1e3 / nscans, 2),
                               "Mpoints_per_s": round(tot_pts / proj_ms / 1e3, 1), "dtype": "f64",
                               "algorithmic_bytes_per_call": int(alg),
                               "achieved": round(alg / (proj_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(alg / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "atomics_G_per_s": round(tot_pts / (proj_ms * 1e-3) / 1e9, 2),
                               "atomic_ceiling_G_per_s": ATOMIC_LANE_CEILING_ORDERED_G,
                               "frac_of_atomic_ceiling": round(tot_pts / (proj_ms * 1e-3) / 1e9 / ATOMIC_LANE_CEILING_ORDERED_G, 4),
                               "atomic_ceiling_note": "LANE atomics/s with neighbouring lanes on neighbouring cells (runs >= 8 per 64-byte "
                                                      "line: a scan's firing order), tools/atomic_probe.hip -> profiles/r05/atomic_probe.txt; "
                                                      f"random cells reach {ATOMIC_CEILING_G:g} G/s (one request per lane)",
                               "single_cloud_call_ms_per_cloud": round(float(np.median(t_single)) * 1e3 / nscans, 4),
                               "bytes": "per point 24 B in + one 8-B memory-side atomicMin; per cell 16 B key read + re-arm, 12 B "
                                        "of images out (range, remission, folded label); per filled cell 24 + 8 B gathered",
                               "kernels": ["k_pb_project", "k_pb_resolve"]}}

    def mergemesh_from_points(n=8):
        """`deform('mergemesh')` + `write()` -- the adaption the reference's shipped config selects (config/lidar_transfer.yaml:3,
        `number_of_scans: 1`; laserscan.py:921-1012, :1121-1178) -- from ONE 120 k-point source cloud at the reference's default
        volume parameters (voxel_bounds +-50 / +-50 / +-5 m given as the YAML's ints, voxel 0.05 m): target-FOV projection onto
        the source image, the kept points' bounds read back (48 bytes), `vol_bnds` clipped in place, a volume of that geometry,
        one class-aware integrate, marching cubes, ray cast, pack.  Parity of the chain: goldens F14 / F14b (pytest -m gpu);
        here: wall clock per output scan, and that a second DeviceDeform gives the same bytes."""
        from lidar_transfer_amd.deform import DeviceDeform
        if torch.cuda.get_device_properties(dev).total_memory < 60 * 2**30 or args.target:
            return None
        w = workers[0]
        w.set_mesh(*scenes[0])
        o = w.render(raysets[0], origin)
        torch.cuda.synchronize()
        hit = o["tri"] >= 0
        cloud = [(o["endpoints"][hit].double().contiguous(), o["endrem"][hit].contiguous(),
                  o["endcolors"][hit][:, 2].contiguous().to(torch.int32))]
        sensor = (H, W, wl["fov_up"], wl["fov_down"])
        res = []
        for rep in range(2):
            bnds = np.array([-50, 50, -50, 50, -5, 5]).reshape(3, 2)
            dd = DeviceDeform(sensor, sensor, bnds, 0.05, device=local_rank, mesh_volume=False)
            for _ in range(3):
                got = dd.mergemesh(cloud)
            torch.cuda.synchronize()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                got = dd.mergemesh(cloud)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res.append((float(np.median(ts)), got["bin"].clone(), got["label_file"].clone(), got["range"].clone(), got["vol_dim"],
                        bnds.tolist(), got["n_faces"]))
            mm_stats = dict(dd._mm_state.stats)
            dd.close()
        same = bool(torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and
                    torch.equal(res[0][3].view(torch.int32), res[1][3].view(torch.int32)))
        # ... and with three output scans in flight (FusionScanPipeline.submit_mergemesh: the bounds statements run on the device in
        # submission order, every chain is launched on the previous scan's geometry and verified afterwards)
        pipelined = None
        try:
            import gc
            from lidar_transfer_amd.pipeline import FusionScanPipeline
            bnds_p = np.array([-50, 50, -50, 50, -5, 5]).reshape(3, 2)
            with FusionScanPipeline(bnds_p, 0.05, wl["fov_up"], wl["fov_down"], rays, H, chains=3, device=local_rank,
                                    label_image=True, source_hw=(H, W), fixed_volume=False) as pipe:
                for tk_ in [pipe.submit_mergemesh(cloud, inputs_ready=True) for _ in range(12)]:
                    pipe.wait(tk_)
                bufs = [pipe._chains[0]["scene"].alloc_outputs(R, label_image=True) for _ in range(144)]
                torch.cuda.synchronize()
                gc.collect()
                gc.disable()
                try:
                    reps_p = []
                    for _ in range(3):   # (three bursts of 144 scans; the median: one burst may contain a runtime hiccup of ~10 ms)
                        tp0 = time.perf_counter()
                        tks = [pipe.submit_mergemesh(cloud, out=b_, inputs_ready=True) for b_ in bufs]
                        outs_p = [pipe.wait(tk_) for tk_ in tks]
                        reps_p.append(time.perf_counter() - tp0)
                    dtp = float(np.median(reps_p))
                finally:
                    gc.enable()
                okp = all(bool(torch.equal(o_["range"].view(-1).view(torch.int32), res[0][3].view(-1).view(torch.int32))) for o_ in outs_p)
                pipelined = {"chains_in_flight": 3, "output_scans": len(bufs), "ms_per_output_scan": round(dtp / len(bufs) * 1e3, 4),
                             "ms_per_output_scan_bursts": [round(x / len(bufs) * 1e3, 4) for x in reps_p],
                             "output_scans_per_s": round(len(bufs) / dtp, 1), "verified": bool(okp),
                             "geometry_stats": dict(pipe._mm_state.stats),
                             "api": "lidar_transfer_amd.pipeline.FusionScanPipeline.submit_mergemesh (no write())"}
        except Exception as e:  # noqa: BLE001
            pipelined = {"error": repr(e)[:300]}
        return {"what": "DeviceDeform.mergemesh: one 120 k-point cloud -> projection (target FOV) -> bounds statements on the device "
                        "(no read-back before the fusion) -> volume of the clipped geometry -> integrate -> marching cubes -> ray "
                        "cast -> packed .bin / .label bytes",
                "ms_per_output_scan": round(min(r_[0] for r_ in res) * 1e3, 4), "points_in": int(cloud[0][0].shape[0]),
                "vol_dim": list(res[0][4]), "vol_bnds_after": res[0][5], "mesh_faces": int(res[0][6]),
                "points_written": int(res[0][1].shape[0]), "hit_fraction": round(float((res[0][3] > 0).float().mean().item()), 4),
                "verified": same, "pipelined": pipelined, "geometry_stats": mm_stats,
                "parity": "goldens F14 / F14b (tests/test_deform_gpu.py): the reference's own deform('mergemesh') + write()"}

    def e2e_pipelined(n_scans=200, depth=4):
        """The same host-buffer work for a SEQUENCE of scans (the reference's loop over output scans): lt_hostpipe keeps
        `depth` scans in flight -- uploads of scans i+1, i+2 (two uploader threads) | render of scan i | download of scan
        i-1 on separate HIP streams, pageable numpy arrays, colours as the uint8 [V,3] get_mesh returns, all five images downloaded.  Measured by
        tools/hostpipe_rate.py in a numpy-only subprocess (LIDARHIP_NO_TORCH=1: the system ROCm runtime) and -- `in_torch_process`
        -- with torch imported first, as the reference's caller has it (laserscan.py:6).  Both reach the same steady rate;
        the HIP 7.0 runtime bundled with the torch wheel stalls ONCE for 36-54 ms at the 81st scan of a process (round 4's
        "35 % slower" was that stall inside a 200-scan measurement): the tool warms up over 100 scans and reports the stall it
        saw there (profiles/r05/hostpipe_torch.txt)."""
        if args.workload != "C2" or args.target:
            return None
        res = {}
        for name, extra in (("numpy_only", {"LIDARHIP_NO_TORCH": "1"}), ("in_torch_process", {})):
            env = dict(os.environ)
            env.update(extra)
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hostpipe_rate.py"), str(depth), str(n_scans),
                                    "6"], capture_output=True, text=True, timeout=300, env=env)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                res[name] = json.loads(line[-1]) if (r.returncode == 0 and line) else None
            except (subprocess.TimeoutExpired, ValueError):
                res[name] = None
        m = res.get("numpy_only")
        if not m:
            return None
        t = m["ms_per_scan"] * 1e-3
        h2d = m["h2d_bytes"]
        out = {"what": f"lt_hostpipe: {n_scans} scans, {depth} in flight (uploads i+1, i+2 on two threads | render i | download i-1), host "
                       f"meshes in pageable numpy arrays, colours uint8 [V,3] as get_mesh returns them, all five images "
                       f"downloaded; numpy-only process", "ms_per_scan": round(t * 1e3, 4),
               "value": round(R / t / 1e6, 2), "unit": "Mrays/s", "scans_per_s": round(1.0 / t, 1), "h2d_bytes": int(h2d),
               "d2h_bytes": int(m["d2h_bytes"]),
               "pcie": {"bound": "pcie gen5 x16, one direction", "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                        "achieved": round(h2d / t / 1e9, 2), "frac": round(h2d / t / 1e9 / PCIE_PEAK_GBS, 4),
                        "measured_wire_GBs": PCIE_WIRE_GBS, "frac_of_wire": round(h2d / t / 1e9 / PCIE_WIRE_GBS, 4),
                        "note": "upload bytes per scan / time per scan; the link is full duplex and the downloads "
                                "run under the uploads"}, "hits": m["hits"],
               "uploader_thread_ms_per_scan": m.get("worker_upload_ms"),
               "single_call_ms_in_this_process": m.get("single_call_ms")}
        if res.get("in_torch_process"):
            it = res["in_torch_process"]
            out["in_torch_process"] = {"ms_per_scan": it["ms_per_scan"], "GBs": it["GBs"],
                                       "frac_of_wire": round(it["GBs"] / PCIE_WIRE_GBS, 4),
                                       "one_time_stall_in_warmup_ms": it.get("warmup_longest_gap_ms"),
                                       "one_time_stall_at_scan": it.get("warmup_longest_gap_at_scan"),
                                       "note": "same loop, torch imported first (its bundled HIP 7.0 runtime): the same steady "
                                               "rate; that runtime stalls once per process (reported here, inside the 100-scan "
                                               "warm-up)"}
            out["numpy_only_warmup_longest_gap_ms"] = m.get("warmup_longest_gap_ms")
        return out

    failures = {}

    def guarded(name, fn, *a, **k):
        """An optional leg of the bench must never cost the headline line: exceptions are reported, not raised."""
        try:
            return fn(*a, **k)
        except BaseException as e:  # noqa: BLE001  (SystemExit from a helper included)
            if isinstance(e, KeyboardInterrupt):
                raise
            failures[name] = repr(e)[:300]
            sys.stderr.write(f"bench_chains.py: optional leg '{name}' failed: {e!r}\n")
            return None

    dt, kern_ms, hits, verify = run(args.strategy, K, Wm, keep=True)
    ser_ms = guarded("serial_probe", serial_probe_ms, args.strategy)
    if ser_ms is None:
        ser_ms = float("nan")
    other = None
    if not args.no_other:
        def other_leg():
            oname = "lbvh" if args.strategy == "scatter" else "scatter"
            Ko = max(20, K // 16) if oname == "lbvh" else max(K, 400)  # a scatter scan is ~15x shorter than an LBVH scan
            odt, okern, _, _ = run(oname, Ko, max(4, Wm // 16), keep=False)
            return {"strategy": oname, "value": round(world * Ko * R / odt / 1e6, 3), "unit": "Mrays/s",
                    "ms_per_scan": round(odt / Ko * 1e3, 4), "scans": Ko,
                    "roofline": roofline(oname, serial_probe_ms(oname, n=12), okern)}
        other = guarded("other_strategy", other_leg)

    iso_ms = guarded("isolated_kernel", isolated_kernel_ms, args.strategy)

    def one_batch_in_flight():
        """The timed region once more with ONE batch in flight (one stream, launches strictly one after the other), every
        k_sc_tris launch bracketed by HIP events: here the kernel durations are exclusive AND inside a wall-clocked region,
        so launches x kernel time must FIT in the region's time -- the check the default region (two overlapped batches,
        which fill each other's tails) cannot offer."""
        Ko = max(args.batch * 8, min(K, 1024))
        odt, okern, _, _ = run("scatter", Ko, max(args.batch * 2, min(Wm, 64)), keep=False, groups=1, probe_every=1)
        n_launch = (Ko + args.batch - 1) // args.batch
        return {"batches_in_flight": 1, "scans": Ko, "value": round(world * Ko * R / odt / 1e6, 3), "unit": "Mrays/s",
                "region_ms": round(odt * 1e3, 4), "k_sc_tris_launches": n_launch,
                "k_sc_tris_avg_ms": round(okern, 5), "launches_x_kernel_ms": round(n_launch * okern, 4),
                "kernel_share_of_region": round(n_launch * okern / (odt * 1e3), 4),
                "fits_in_region": bool(n_launch * okern <= odt * 1e3),
                "note": "one stream, one batch of scans in flight: k_sc_tris -> k_sc_rest -> k_sc_resolve strictly in turn; "
                        "HIP events around EVERY k_sc_tris launch of the region"}

    one_batch = guarded("one_batch_in_flight", one_batch_in_flight) if (args.strategy == "scatter" and args.batch > 1) else None
    # the PCIe-inclusive clocks and the fusion chain are single-GPU records (like cpu_baseline): rank 0 at N = 1 only
    e2e = guarded("e2e_single_call", e2e_host_call) if (rank == 0 and world == 1 and not args.no_e2e) else None
    if e2e:
        e2e = {"single_call": e2e, "pipelined": guarded("e2e_pipelined", e2e_pipelined)}
    def fusion_chain_pipelined(chains=3):
        """The same chain with `chains` output scans in flight (lidar_transfer_amd.pipeline.FusionScanPipeline: own volume,
        mesh, scene, HIP stream and host thread each -- output scans are independent, lidar_deform.py:393-462): the chain's
        sparse sweeps leave the chip half empty, scans in flight fill each other's gaps.  tools/chain_pipeline.py; one and
        five observations per scan on the same pipeline; every timed scan's images are compared bit for bit with the
        single chain's (`verified`).  Runs BEFORE the single-chain legs: its 38 GB of volumes should be the process's
        first big allocation."""
        if torch.cuda.get_device_properties(dev).total_memory < 100 * 2**30:
            return None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import chain_pipeline
        return chain_pipeline.run_cases(chains, ((12, 1), (12, 5)), local_rank, args.workload)

    pipelined = guarded("fusion_chain_pipelined", fusion_chain_pipelined) if (rank == 0 and not args.no_chain and world == 1) else None
    chain = guarded("fusion_chain", fusion_chain) if (rank == 0 and not args.no_chain and world == 1) else None
    chain5 = guarded("fusion_chain_nscans5", fusion_chain, 4, 5) if (chain and rank == 0 and world == 1) else None

    from_points = guarded("deform_from_points", deform_from_points) if (chain and rank == 0 and world == 1) else None
    mergemesh_leg = guarded("mergemesh_from_points", mergemesh_from_points) if (chain and rank == 0 and world == 1) else None
    if chain:
        chain["pipelined"] = (pipelined or [None, None])[0]
    if chain5:
        chain5["pipelined"] = (pipelined or [None, None])[1]
    if rank == 0:
        value = world * K * R / dt / 1e6
        rl = roofline(args.strategy, ser_ms, kern_ms)
        if iso_ms is not None:
            iso_b = rl["bytes_per_scan"]["hbm_compulsory"] if "bytes_per_scan" in rl else rl["algorithmic_bytes_per_scan"]
            rl["isolated"] = {"avg_kernel_ms": round(iso_ms, 5),
                              "achieved": round(iso_b / (iso_ms * 1e-3) / 1e9, 1),
                              "frac": round(iso_b / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "same kernel, launches of ONE scan each, back to back on one stream (nothing beside "
                                      "them), after the timed region"}
        if args.strategy == "scatter":
            # path_frac: the figure that is bounded by the DRIVER's clock -- algorithmic bytes of all three kernels per step
            # over ms_per_step against the HBM peak.  k_sc_tris: as above; k_sc_resolve: per ray the 8-B z-min cell read and
            # re-armed (16 B) + the five images written (range 4, label 4, remission 4, end point 12, triangle 4 = 28 B), per
            # hit ray the winner's face (12 B), three remissions (12 B) and one label (4 B); k_sc_rest redoes deferred work
            # of k_sc_tris: no algorithmic bytes of its own.
            cs = np.mean(np.array(cnt["scatter"], dtype=np.float64), axis=0)
            hit_rays = float(np.mean(hit_ray_counts)) if hit_ray_counts else float(cs[2])
            b_tris = rl["algorithmic_bytes_per_scan"]
            b_res = R * (16 + 28) + hit_rays * 28
            step_s = dt / args.steps
            rl["path_frac"] = round((b_tris + b_res) * SPS / step_s / 1e9 / HBM_PEAK_GBS, 4)
            rl["path"] = {"algorithmic_bytes_per_scan": {"k_sc_tris": int(b_tris), "k_sc_resolve": int(b_res), "k_sc_rest": 0},
                          "achieved": round((b_tris + b_res) * SPS / step_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": rl["path_frac"],
                          "frac_k_sc_tris_bytes_only": round(b_tris * SPS / step_s / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "algorithmic bytes of the whole three-kernel path per step / ms_per_step (the driver-timed "
                                  "number) / 8 TB/s; needs no kernel-exclusivity argument"}
            # Does the figure named `frac` fit the driver's clock?  `avg_kernel_ms` is an EXCLUSIVE duration (one stream, nothing
            # beside the kernel); in the timed region the batches of three streams overlap and fill each other's tails, so
            # launches x avg_kernel_ms may exceed ms_per_step.  `frac_on_step_clock` charges the WHOLE step to k_sc_tris
            # (compulsory bytes of the step's launches / ms_per_step): it fits by construction and bounds the kernel from below.
            launches = SPS / args.batch
            comp_scan = rl["bytes_per_scan"]["hbm_compulsory"]
            rl["step_clock"] = {"launches_per_step": launches, "avg_kernel_ms_exclusive": rl["avg_kernel_ms"],
                                "launches_x_avg_kernel_ms": round(launches * rl["avg_kernel_ms"], 4),
                                "ms_per_step": round(step_s * 1e3, 4),
                                "fits": bool(launches * rl["avg_kernel_ms"] <= step_s * 1e3),
                                "achieved_on_step_clock": round(comp_scan * SPS / step_s / 1e9, 1),
                                "frac_on_step_clock": round(comp_scan * SPS / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                "note": "exclusive kernel durations overlap in the timed region (3 batches in flight); "
                                        "frac_on_step_clock = compulsory HBM bytes of the step / ms_per_step / 8 TB/s -- the whole "
                                        "step charged to this kernel; one_batch_in_flight below is the configuration in which "
                                        "launches x kernel time does fit its own region"}
            if one_batch:
                rl["one_batch_in_flight"] = one_batch
            # all three kernels of a scan together at the measured scan rate: the HBM bandwidth the whole path
            # sustains over the timed region (PMC traffic per launch from profiles/rNN/pmc.json, null when stale)
            parts = [measured_traffic("scatter", args.batch, k)[0] for k in ("k_sc_tris", "k_sc_rest", "k_sc_resolve")]
            if all(p is not None for p in parts):
                per_scan = sum(parts) / args.batch
                rl["whole_path"] = {"hbm_bytes_per_scan": int(per_scan),
                                    "sustained_GBs": round(K / dt * per_scan / 1e9, 1),
                                    "frac_of_peak": round(K / dt * per_scan / 1e9 / HBM_PEAK_GBS, 4),
                                    "note": "PMC traffic of k_sc_tris + k_sc_rest + k_sc_resolve per scan x scans/s of "
                                            "this rank; 6290 GB/s is what a float4 copy reaches on this chip"}
                ents = [measured_traffic("scatter", args.batch, k)[2] for k in ("k_sc_tris", "k_sc_rest", "k_sc_resolve")]
                if all(e and e.get("valu_active_quad_cycles_per_launch") for e in ents):
                    # the three kernels' vector-issue cycles per SIMD per scan against the wall time of a scan in the timed
                    # region (world == 1 figure of this rank): how full the chip's vector units are over the whole region
                    busy = sum(e["valu_active_quad_cycles_per_launch"] for e in ents) * SQ_CYCLES_PER_COUNT / N_SIMD / args.batch
                    rl["whole_path"]["valu_busy_frac_of_timed_region"] = round(busy / SHADER_GHZ / 1e9 / (dt / K), 4)
            else:
                rl["whole_path"] = None
        out = {
            "metric": "Mrays/sec, one new ~1M-triangle mesh per scan -> 64x2048 range/label image",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {H}x{W} rays vs {n_faces}-triangle synthetic scene "
                                   f"(fov {wl['fov_up']}/{wl['fov_down']}); 1 step = one batch of {SPS} scans, each "
                                   f"with a new mesh; {len(scenes)} distinct scenes cycled",
                       "scans_per_step": SPS, "ms_per_scan": round(dt / K * 1e3, 5),
                       "strategy": args.strategy,
                       "parallelism": f"scan-parallel x{world}",
                       "gather": hz.gather_info or None,
                       "streams_per_gpu": S, "scans_per_call": args.batch if args.strategy == "scatter" else 1},
            "scans_per_s": round(world * K / dt, 2),
            "hit_fraction": round(hits / R, 4),
            "verified": bool(verify and verify["ok"]), "verification": verify,
            "roofline": rl,
        }
        if failures:
            out["failed_legs"] = failures
        if phase:
            out["lbvh_phase_ms"] = {k: round(v, 4) for k, v in phase.items() if k.startswith("ms_") and k != "ms_trace"}
        if other:
            out["other_strategy"] = other
        if e2e:
            out["e2e"] = e2e
        if chain:
            out["fusion_chain"] = chain
        if chain5:
            out["fusion_chain_nscans5"] = chain5
        if mergemesh_leg:
            out["mergemesh_from_points"] = mergemesh_leg
        if from_points:
            out["deform_from_points"] = from_points
            out["projection"] = from_points["projection"]
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            cb = guarded("cpu_baseline", bl.cpu_baseline, wl, 0, args.cpu_reps or 12)
            out["cpu_baseline"] = cb
            if cb:
                # matching clocks, all against the REAL reference on this box's host cores:
                #   device_resident vs the reference's e2e call   (what a scan costs when the mesh is where the renderer is)
                #   device_resident vs the reference's TRACE-ONLY loop (its BVH already built: the most favourable CPU clock)
                #   e2e single call / pipelined (PCIe inclusive, host buffers in and out) vs the reference's e2e call
                def _r(a, b):
                    return round(a / b, 1) if (a and b) else None
                ck = cb.get("clocks", {})
                tr_all = (ck.get("all_threads") or {}).get("trace_only_Mrays_s")
                tr_one = (ck.get("one_thread") or {}).get("trace_only_Mrays_s")
                e2e_one = (ck.get("one_thread") or {}).get("e2e_Mrays_s")
                sc_v = e2e["single_call"]["value"] if (e2e and e2e.get("single_call")) else None
                pp_v = e2e["pipelined"]["value"] if (e2e and e2e.get("pipelined")) else None
                out["speedup_vs_cpu_baseline"] = {"device_resident": _r(value / world, cb["value"]),
                                                  "device_resident_vs_trace_only_all_threads": _r(value / world, tr_all),
                                                  "device_resident_vs_trace_only_one_thread": _r(value / world, tr_one),
                                                  "e2e_single_call": _r(sc_v, cb["value"]),
                                                  "e2e_single_call_vs_one_thread": _r(sc_v, e2e_one),
                                                  "e2e_pipelined": _r(pp_v, cb["value"])}
        os.makedirs(os.path.dirname(out_path), exist_ok=True)
        with open(out_path, "w") as fh:
            json.dump(out, fh, indent=1)
            fh.write("\n")
        summary = {"extras": os.path.relpath(out_path, ROOT), "value": out["value"], "failed_legs": failures or None,
                   "fusion_chain_ms": (chain or {}).get("ms_per_scan"), "mergemesh_ms": (mergemesh_leg or {}).get("ms_per_output_scan"),
                   "deform_ms": (from_points or {}).get("ms_per_output_scan")}
        os.write(real_stdout, (json.dumps(summary) + "\n").encode())
    hz.close()


if __name__ == "__main__":
    main()
