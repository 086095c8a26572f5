#!/usr/bin/env python3
"""TCC_EA0 passes on `bench.py --probe-only` (tools/r03_profile.sh: gpurun_out/r03/ea/<counters>/...) -> ea_requests.json:
memory-side read requests and atomics per batch launch of the scatter kernels, keyed like pmc.json (workload, launch shape,
hash of the kernel sources) -- bench.py prices them against the measured request ceilings (roofline.memory_side).
    python tools/ea_to_json.py <dir with the passes> <out.json>"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_to_json import source_hash  # noqa: E402


def main(src, out):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in sorted(glob.glob(os.path.join(src, "*", "p_counter_collection.csv")) +
                    glob.glob(os.path.join(src, "*", "*", "p_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if k.startswith("k_sc_") and "<true" not in r["Kernel_Name"].split("(")[0]:
                acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    entries = []
    for k in sorted(acc):
        n = min(len(d) for d in acc[k].values())
        if n < 20:   # (the serial probe's batch launches; the handful of single-scan launches of the set-up do not count)
            continue
        e = {"kernel": k, "scans_per_launch": 8, "launches": n, "strategy": "scatter", "workload": "C2",
             "kernel_source_hash": source_hash("scatter")}
        for c, d in sorted(acc[k].items()):
            e[c.replace("_sum", "").lower() + "_per_launch"] = round(sum(d.values()) / len(d))
        entries.append(e)
    json.dump({"what": "memory-side (EA) requests per batch launch, separate two-counter passes of rocprofv3 --kernel-trace "
                       "--pmc on: python bench.py --probe-only", "entries": entries}, open(out, "w"), indent=1)
    print(json.dumps(entries)[:600])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
