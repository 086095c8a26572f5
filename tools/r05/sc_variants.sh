# the vertex-window variants of k_sc_tris next to the default with its SQ_BUSY_CYCLES, SQ_INSTS_VALU and the
# duration of the batch launch (serial probe), on ONE box -> gpurun_out/r05/sc_variants.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; : > $O/sc_variants.txt
run() {  # $1 = label, LIDARHIP_EXTRA_FLAGS / env already set
  (cd $R && python -c "from lidar_transfer_amd import build; build.build_lib()" > /dev/null 2>&1)
  rm -rf $O/scv_tmp
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/scv_tmp -o p -- python $R/bench.py --probe-only > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/scv_tmp/stats -o s -- python $R/bench.py --probe-only > $O/scv_tmp/probe.json 2>/dev/null
  (cd $R && python bench.py --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null > $O/scv_tmp/bench.json)
  (cd $R && timeout 900 python -m pytest tests/test_trace_gpu.py tests/test_properties_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -1 > $O/scv_tmp/parity.txt)
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
                  "serial_probe_ms": b["roofline"]["avg_kernel_ms"], "verified": b.get("verified"),
                  "parity_suite": open(root + "/parity.txt").read().strip()}))
P
}
unset LIDARHIP_EXTRA_FLAGS; run "round 5 default (= round 4: LT_SC_T=448, index triples then six global vertex gathers)"
export LIDARHIP_EXTRA_FLAGS="-DLT_SC_WIN=1024"; run "vertex window 1024 in LDS (12 KB: 5 workgroups per CU) + k_sc_win pre-pass, T=448"
export LIDARHIP_EXTRA_FLAGS="-DLT_SC_WIN=512"; run "vertex window 512 in LDS (6 KB: 6 workgroups per CU) + k_sc_win pre-pass, T=448"
export LIDARHIP_EXTRA_FLAGS="-DLT_SC_WIN=640 -DLT_SC_T=256"; run "vertex window 640 + LT_SC_T=256 (10 + 7.5 KB LDS: 8 workgroups per CU) + k_sc_win pre-pass"
unset LIDARHIP_EXTRA_FLAGS
(cd $R && python -c "from lidar_transfer_amd import build; build.build_lib()" > /dev/null 2>&1)
rm -rf $O/scv_tmp
