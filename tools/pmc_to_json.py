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


def per_kernel_mean(root, counter):
    """mean over launches of the per-dispatch counter sum (a counter is reported once per XCD / instance), split by
    grid size so that batch launches and single-scan launches of one kernel are told apart"""
    acc = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, counter, "**", "*counter_collection.csv"), recursive=True)):
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
    # The grid of a launch grows with the scans it holds (and varies a little with the scene's face count): launches
    # of one kernel are clustered by round(grid / smallest grid) -- 1 = a single-scan launch (counting / isolated
    # passes), --batch = the batch call of the timed region -- and averaged (weighted by launches) per cluster.
    raw = collections.defaultdict(list)
    for (kern, grid), (f, n) in sorted(fetch.items()):
        if kern in STRATEGY_OF and (kern, grid) in write:
            raw[kern].append((grid, n, f, write[(kern, grid)][0]))
    entries = []
    for kern, rows in raw.items():
        gmin = min(g for g, _, _, _ in rows)
        clusters = collections.defaultdict(list)
        for g, n, f, w in rows:
            clusters[max(1, int(round(g / gmin)))].append((g, n, f, w))
        for ratio, rs in sorted(clusters.items()):
            n = sum(r[1] for r in rs)
            f = sum(r[1] * r[2] for r in rs) / n
            w = sum(r[1] * r[3] for r in rs) / n
            spl = ratio if STRATEGY_OF[kern] == "scatter" else 1
            entries.append({"kernel": kern, "scans_per_launch": spl, "launches": n,
                            "grid_size_mean": int(sum(r[0] * r[1] for r in rs) / n), "fetch_kib": round(f, 1),
                            "write_kib": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                            "strategy": STRATEGY_OF[kern], "workload": a.workload,
                            "kernel_source_hash": source_hash(STRATEGY_OF[kern])})
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
