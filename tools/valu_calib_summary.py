#!/usr/bin/env python3
"""Relate the s_memtime figures of tools/valu_calib.bin to the SQ counters of the same launches.

    python tools/valu_calib_summary.py gpurun_out/r03 > profiles/r03/valu_calib.txt

Per launch (regime x waves per SIMD; the LAST of the three repetitions of each shape is the one the stand-alone run
reports): SQ_INSTS_VALU must equal waves x 4096 + the handful of set-up instructions; SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
is the counter's unit per wave64 instruction; x 4 (quad-cycles, MI355X_MICROARCH.md) it should equal the s_memtime cycles per
instruction a SIMD spends ISSUING (2 or 4?) -- that is the number this tool exists to fix."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03"
print("== stand-alone run (s_memtime per wave, 100 MHz wall clock, HIP events) ==")
run = os.path.join(root, "valu_calib_run.txt")
print(open(run).read() if os.path.exists(run) else "(missing)")
rows = collections.OrderedDict()
for path in sorted(glob.glob(os.path.join(root, "valu_calib_pmc", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        k = int(r["Dispatch_Id"])
        d = rows.setdefault(k, {"kernel": r["Kernel_Name"].split("(")[0], "grid": int(r.get("Grid_Size", 0) or 0)})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("== the same launches under rocprofv3 --pmc (sum over XCDs / SEs) ==")
print("dispatch kernel grid waves | SQ_INSTS_VALU  per-wave | SQ_ACTIVE_INST_VALU  per VALU inst | SQ_WAVE_CYCLES per wave per inst | "
      "SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE")
for k, d in rows.items():
    waves = d.get("SQ_WAVES", 0) or (d["grid"] // 64)
    iv, av = d.get("SQ_INSTS_VALU", 0), d.get("SQ_ACTIVE_INST_VALU", 0)
    wc = d.get("SQ_WAVE_CYCLES", 0)
    print("%3d %-22s %8d %6d | %12.0f %8.1f | %14.0f %6.3f | %14.0f %6.3f | %12.0f | %10.0f" % (
        k, d["kernel"][:22], d["grid"], waves, iv, iv / max(waves, 1), av, av / max(iv, 1), wc,
        wc / max(waves, 1) / 4096.0, d.get("SQ_BUSY_CYCLES", 0), d.get("GRBM_GUI_ACTIVE", 0)))
print()
print("reading: column 'per VALU inst' x 4 = shader cycles a SIMD's VALU is ACTIVE per wave64 instruction if the counter is in")
print("quad-cycles; compare with the stand-alone '=> SIMD-cycles per wave-instruction' at 8 waves per SIMD (ind8).")
