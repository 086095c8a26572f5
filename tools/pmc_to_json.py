#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_bench.sh (run on bench.py itself) into profiles/rNN/pmc.json,
the file bench.py reads `roofline.traffic` from.  Every entry is keyed by workload, strategy, scans per launch and a
hash of the kernel's sources, so that bench.py reports null instead of a stale constant once a kernel changes.

    python tools/pmc_to_json.py gpurun_out/pmcb profiles/r02/pmc.json [--workload C2 --batch 8]

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: FETCH_SIZE under-reports reads by 2x on gfx950
(MI355X_MICROARCH.md, HBM section); the two counters come from separate --pmc passes (--kernel-trace only)."""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = {"scatter": ["lt_scatter.hip", "lt_internal.h", "lt_normalize.h"],
                  "lbvh": ["lt_trace.hip", "lt_build.hip", "lt_internal.h", "lt_normalize.h"]}
STRATEGY_OF = {"k_sc_tris": "scatter", "k_sc_rest": "scatter", "k_sc_resolve": "scatter", "k_trace4": "lbvh",
               "k_hierarchy4": "lbvh"}


def source_hash(strategy):
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[strategy]:
        with open(os.path.join(ROOT, "lidar_transfer_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + fh.read())
    return h.hexdigest()[:16]


def per_kernel_mean(root, counter, subdir=None):
    """mean over launches of the per-dispatch counter sum (a counter is reported once per XCD / instance), split by
    grid size so that batch launches and single-scan launches of one kernel are told apart"""
    acc = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, subdir or counter, "**", "*counter_collection.csv"), recursive=True)):
        per = collections.defaultdict(float)
        meta = {}
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            meta[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip(),
                                      int(r.get("Grid_Size", 0) or 0))
        for d, v in per.items():
            acc[meta[d]].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("out")
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--command", default="python bench.py --no-cpu-baseline --no-other --no-e2e")
    a = ap.parse_args()
    fetch, write = per_kernel_mean(a.root, "FETCH_SIZE"), per_kernel_mean(a.root, "WRITE_SIZE")
    # optional third pass (tools/r02_profile.sh): SQ_INSTS_VALU (wave instructions) and SQ_ACTIVE_INST_VALU (quad-cycles
    # the VALUs were issuing, summed over the SIMDs) -- the issue-bound view of a kernel that moves few bytes per flop
    valu_n = per_kernel_mean(a.root, "SQ_INSTS_VALU", "SQ_INSTS_VALU_group")
    valu_c = per_kernel_mean(a.root, "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU_group")
    # The grid of a launch grows with the scans it holds (and varies a little with the scene's face count): launches
    # of one kernel are clustered by round(grid / smallest grid) -- 1 = a single-scan launch (counting / isolated
    # passes), --batch = the batch call of the timed region -- and averaged (weighted by launches) per cluster.
    raw = collections.defaultdict(list)
    for (kern, grid), (f, n) in sorted(fetch.items()):
        if kern in STRATEGY_OF and (kern, grid) in write:
            raw[kern].append((grid, n, f, write[(kern, grid)][0], valu_n.get((kern, grid), (None,))[0],
                              valu_c.get((kern, grid), (None,))[0]))
    entries = []
    for kern, rows in raw.items():
        gmin = min(r[0] for r in rows)
        clusters = collections.defaultdict(list)
        for r in rows:
            clusters[max(1, int(round(r[0] / gmin)))].append(r)
        for ratio, rs in sorted(clusters.items()):
            n = sum(r[1] for r in rs)
            f = sum(r[1] * r[2] for r in rs) / n
            w = sum(r[1] * r[3] for r in rs) / n
            spl = ratio if STRATEGY_OF[kern] == "scatter" else 1
            # k_sc_rest's grid is not proportional to the scans it holds (512 workgroups for a single-scan call, 128 per scan
            # of a batch call since round 4): every launch above the smallest grid is a batch launch
            if kern == "k_sc_rest" and ratio > 1:
                spl = a.batch
            e = {"kernel": kern, "scans_per_launch": spl, "launches": n,
                 "grid_size_mean": int(sum(r[0] * r[1] for r in rs) / n), "fetch_kib": round(f, 1),
                 "write_kib": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                 "strategy": STRATEGY_OF[kern], "workload": a.workload,
                 "kernel_source_hash": source_hash(STRATEGY_OF[kern])}
            if all(r[4] is not None and r[5] is not None for r in rs):
                e["valu_wave_insts_per_launch"] = int(sum(r[1] * r[4] for r in rs) / n)
                e["valu_active_quad_cycles_per_launch"] = int(sum(r[1] * r[5] for r in rs) / n)
            entries.append(e)
    doc = {"what": "HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, "
                   "--kernel-trace only) of: " + a.command,
           "formula": "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)",
           "entries": entries}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for e in entries:
        print(e["kernel"], "scans/launch", e["scans_per_launch"], "launches", e["launches"],
              round(e["hbm_bytes_per_launch"] / 1e6, 2), "MB")


if __name__ == "__main__":
    main()
