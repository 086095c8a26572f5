#!/usr/bin/env python3
"""PMC passes + kernel trace of the fusion chain (tools/prof_chain.py under rocprofv3, tools/r03_profile.sh) ->
profiles/rNN/pmc_chain.json, the file bench.py reads the per-kernel rooflines of `fusion_chain` from.

    python tools/pmc_chain_to_json.py gpurun_out/r03/pmc_chain gpurun_out/r03/chain/s_kernel_stats.csv profiles/r03/pmc_chain.json

Entries are keyed by a hash of lt_tsdf.hip / lt_mc.hip / lt_internal.h: a changed kernel gives `null` rooflines, never a
stale constant.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 correction, MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_tsdf_integrate_pix", "k_tsdf_integrate_pix_multi", "k_tsdf_integrate_quirk_multi", "k_tsdf_dct4", "k_tsdf_integrate_written", "k_tsdf_integrate_quirk", "k_tsdf_dct",
           "k_tsdf_integrate_cols", "k_tsdf_columns", "k_tsdf_colmax", "k_tsdf_reset_cols", "k_mc_words", "k_mc_amb", "k_mc_compact",
           "k_mc_emit_batch", "k_mc_clear", "k_mc_scan1", "k_mc_scan2")


def source_hash():
    h = hashlib.sha256()
    for name in ("lt_tsdf.hip", "lt_mc.hip", "lt_internal.h"):
        with open(os.path.join(ROOT, "lidar_transfer_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + fh.read())
    return h.hexdigest()[:16]


def short(name):
    return name.split("(")[0].split("<")[0].replace("void ", "").strip()


def counter_means(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        per = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(path)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (d, c), v in per.items():
            acc[names[d]][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    pmc_root, stats_csv, out = sys.argv[1:4]
    cm = counter_means(pmc_root)
    avg = {}
    for r in csv.DictReader(open(stats_csv)):
        avg[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]))
    entries = []
    for k in KERNELS:
        c = cm.get(k)
        if not c or k not in avg or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        e = {"kernel": k, "avg_kernel_ns": round(avg[k][0], 1), "calls_in_trace": avg[k][1],
             "fetch_kib": round(c["FETCH_SIZE"], 1), "write_kib": round(c["WRITE_SIZE"], 1),
             "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
             "kernel_source_hash": source_hash()}
        if "SQ_INSTS_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            e["valu_wave_insts_per_launch"] = int(c["SQ_INSTS_VALU"])
            e["valu_active_quad_cycles_per_launch"] = int(c["SQ_ACTIVE_INST_VALU"])
        for extra in ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_BUSY_CYCLES"):
            if extra in c:
                e[extra.lower()] = int(c[extra])
        entries.append(e)
    doc = {"what": "fusion chain on the default 2000 x 2000 x 200 volume, one C2 observation per output scan "
                   "(tools/prof_chain.py): rocprofv3 --pmc passes (separate runs, --kernel-trace only) + kernel trace",
           "formula": "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)",
           "entries": entries}
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(doc, open(out, "w"), indent=1)
    for e in entries:
        print(e["kernel"], round(e["avg_kernel_ns"] / 1e3, 1), "us", round(e["hbm_bytes_per_launch"] / 1e6, 1), "MB",
              e.get("valu_wave_insts_per_launch"))


if __name__ == "__main__":
    main()
