"""The ONE stdout line of bench.py must stay parseable by the driver: < 8 KB, json.loads-able, contract keys present
(round 5's 21.6 KB line left the round unmeasured)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _line(extra_note=""):
    """a headline record with every field at a realistic worst-case width (8 ranks, gather info, all roofline keys)"""
    rl = {"kernel": "k_sc_tris", "bound": "hbm", "achieved": 1357.4, "peak": 8000.0, "unit": "GB/s", "frac": 0.1697,
          "traffic": 339312345.0, "traffic_over_compulsory": 2.196, "traffic_source": "profiles/r06/pmc.json",
          "avg_kernel_ms": 0.11355, "scans_per_launch": 8, "hbm_compulsory_bytes_per_launch": 154512345,
          "algorithmic_bytes_per_launch": 305812345, "frac_incl_l2_bytes": 0.3366, "tests_per_ray": 9.01,
          "in_situ_avg_kernel_ms": 0.21234,
          "step_clock": {"launches_per_step": 8.0, "launches_x_avg_kernel_ms": 0.9084, "ms_per_step": 0.7655,
                         "frac_on_step_clock": 0.2018}, "note": "n" * 300 + extra_note}
    cb = {"value": 0.3412, "unit": "Mrays/s", "cores": 256, "kind": "reference", "sample": "s" * 260, "s_per_scan": 0.3841,
          "trace_only_Mrays_s": 1.634, "bvh_build_ms": 251.0, "one_thread": {"e2e_Mrays_s": 0.2412, "trace_only_Mrays_s": 0.912},
          "cpu_model": "AMD EPYC 9575F 64-Core Processor", "nproc": 256}
    gi = {"mode": "sharded", "requested": "auto", "bytes_per_scan": 786432, "scans_per_s_per_rank_warmup": 81234.5,
          "per_link_GBs_needed": 63.88, "per_link_GBs_measured": 48.12, "link_bound_scans_per_s_per_rank": 61188.1,
          "link_bound": False, "predicted_job_scans_per_s": 649876.0}
    return {"metric": "Mrays/sec, one new ~1M-triangle mesh per scan -> 64x2048 range/label image", "value": 10959.123,
            "unit": "Mrays/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 0.7655, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w" * 200, "scans_per_step": 64, "ms_per_scan": 0.01196, "strategy": "scatter",
                       "parallelism": "p" * 160, "gather": gi, "scans_in_flight": 24, "scans_per_call": 8},
            "scans_per_s": 83611.2, "hit_fraction": 0.9831, "verified": True,
            "verification": {"scans_compared": [0, 2048, 4095], "ok": True, "golden_f5_sha256": True},
            "roofline": rl, "cpu_baseline": cb,
            "speedup_vs_cpu_baseline": {"e2e_call_all_threads": 32119.4, "trace_only_all_threads": 6706.9}}


def test_headline_line_is_small_and_parseable():
    import bench
    import bench_lib as bl
    line = bl.fit_line(_line(), bench.MAX_LINE_BYTES)
    assert len(line.encode()) < 8000 < 10000 and "\n" not in line
    out = json.loads(line)
    for k in CONTRACT:
        assert k in out, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_compulsory", "avg_kernel_ms", "kernel",
              "hbm_compulsory_bytes_per_launch", "scans_per_launch", "step_clock"):
        assert k in out["roofline"], k
    assert "frac_on_step_clock" in out["roofline"]["step_clock"]
    for k in ("value", "unit", "cores", "kind", "sample", "s_per_scan", "trace_only_Mrays_s"):
        assert k in out["cpu_baseline"], k
    assert "workload" in out["config"] and "model" not in out["config"]


def test_an_oversized_record_is_pruned_not_printed():
    """whatever a future edit adds: the guard drops optional keys, then strings, and the contract keys survive"""
    import bench
    import bench_lib as bl
    rec = _line(extra_note="x" * 20000)
    rec["verification"]["blob"] = "y" * 5000
    line = bl.fit_line(rec, bench.MAX_LINE_BYTES)
    assert len(line.encode()) < 8000
    out = json.loads(line)
    for k in CONTRACT:
        assert k in out, k
    assert out["roofline"]["frac"] == 0.1697 and out["cpu_baseline"]["value"] == 0.3412 and "workload" in out["config"]


def test_bench_py_is_the_headline_only():
    """bench.py stays a readable headline (< 500 lines); the side records live in tools/bench_chains.py"""
    n = sum(1 for _ in open(os.path.join(ROOT, "bench.py")))
    assert n < 500, n
    assert os.path.exists(os.path.join(ROOT, "tools", "bench_chains.py"))
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "fusion_chain" not in src and "mergemesh_from_points" not in src
