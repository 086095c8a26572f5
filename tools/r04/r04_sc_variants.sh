# VERDICT r03 item 3 exit criterion: every k_sc_tris variant of round 4 with its SQ_BUSY_CYCLES, SQ_INSTS_VALU and the
# duration of the batch launch (serial probe), on ONE box -> gpurun_out/r04/sc_variants.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; : > $O/sc_variants.txt
run() {  # $1 = label, LIDARHIP_EXTRA_FLAGS / env already set
  (cd $R && python -c "from lidar_transfer_amd import build; build.build_lib()" > /dev/null 2>&1)
  rm -rf $O/scv_tmp
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/scv_tmp -o p -- python $R/bench.py --probe-only > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/scv_tmp/stats -o s -- python $R/bench.py --probe-only > $O/scv_tmp/probe.json 2>/dev/null
  (cd $R && python bench.py --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null > $O/scv_tmp/bench.json)
  python - "$1" $O/scv_tmp >> $O/sc_variants.txt <<'P'
import csv, glob, json, sys, collections
label, root = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_sc_tris" not in k: continue
        acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
big = [d for d in acc.values() if d.get("SQ_WAVES", 0) > 30000]   # the batch launches (8 scans)
m = lambda c: sum(d.get(c, 0) for d in big) / max(len(big), 1)
dur = None
for path in glob.glob(root + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_sc_tris" in r["Name"]: dur = float(r["AverageNs"])
b = json.load(open(root + "/bench.json"))
print(json.dumps({"variant": label, "batch_launches": len(big), "SQ_BUSY_CYCLES": int(m("SQ_BUSY_CYCLES")), "SQ_INSTS_VALU": int(m("SQ_INSTS_VALU")),
                  "SQ_WAVES": int(m("SQ_WAVES")), "k_sc_tris_avg_ns_rocprof": dur, "bench_Mrays_s": b["value"], "ms_per_step": b["ms_per_step"],
                  "serial_probe_ms": b["roofline"]["avg_kernel_ms"]}))
P
}
unset LIDARHIP_EXTRA_FLAGS; run "round 4 default: LT_SC_T=448, two triangles per lane, flat job lookup"
export LIDARHIP_EXTRA_FLAGS="-DLT_SC_T=256"; run "LT_SC_T=256 (one triangle per lane: round 3's shape)"
export LIDARHIP_EXTRA_FLAGS="-DLT_SC_T=384"; run "LT_SC_T=384"
export LIDARHIP_EXTRA_FLAGS="-DLT_SC_T=512"; run "LT_SC_T=512 (22.5 KB LDS: 7 workgroups per CU)"
export LIDARHIP_EXTRA_FLAGS="-DLT_SC_PRIO=1"; run "LT_SC_T=448 + s_setprio 3 while the gathers are issued"
unset LIDARHIP_EXTRA_FLAGS
(cd $R && python -c "from lidar_transfer_amd import build; build.build_lib()" > /dev/null 2>&1)
rm -rf $O/scv_tmp
cat $O/sc_variants.txt
