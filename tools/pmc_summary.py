#!/usr/bin/env python3
"""Summarise the counter passes of tools/pmc_render.sh / pmc_lbvh.sh (gpurun_out/<dir>/<group>/p_counter_collection.csv)
into the text table kept under profiles/: mean counter value per launch and kernel, plus the HBM traffic
2 x FETCH_SIZE + WRITE_SIZE (KiB; FETCH_SIZE under-reports reads by 2x on gfx950, MI355X_MICROARCH.md).

  python tools/pmc_summary.py gpurun_out/pmc2 > profiles/r01/e_pmc_scatter.txt
"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc2"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, "*", "p_counter_collection.csv"))):
    per_dispatch = collections.defaultdict(float)  # a counter is reported once per XCD / instance: sum them
    names = {}
    for r in csv.DictReader(open(path)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
    for (disp, ctr), v in per_dispatch.items():
        acc[names[disp]][ctr].append(v)
print("# rocprofv3 --pmc passes (one counter group per run, --kernel-trace only), mean over the launches of a kernel.")
print("# SQ_* wave / instruction-cycle counters are in quad-cycles (MI355X_MICROARCH.md).  FETCH_SIZE / WRITE_SIZE in KiB;")
print("# FETCH_SIZE under-reports reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section).")
for kern in sorted(acc):
    print()
    print(kern)
    c = acc[kern]
    for ctr in sorted(c):
        print("    %-28s %16.1f" % (ctr, sum(c[ctr]) / len(c[ctr])))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        print("    => HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE = %.1f MB" % ((2 * f + w) * 1024 / 1e6))
