# vector instructions of k_sc_tris by section: rebuilds with -DLT_SC_STOP=n (1: triangle loads, 2: + angular bounds and LDS
# record, 3: + prefix sums and cap, unset: everything) and counts SQ_INSTS_VALU of an isolated C2 render
cd $GRAFT_REPO_ROOT
for st in 1 2 3 0; do
  if [ $st = 0 ]; then export LIDARHIP_EXTRA_FLAGS=""; else export LIDARHIP_EXTRA_FLAGS="-DLT_SC_STOP=$st"; fi
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/scs$st -o p -- python $GRAFT_REPO_ROOT/tools/prof_render.py --reps 4 > /dev/null 2>&1)
  echo "stop $st"; python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open("gpurun_out/scs$st/p_counter_collection.csv")):
    if "k_sc_tris<false" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in sorted(acc.items()): print("   ", c, round(sum(d.values())/len(d)))
PY
done
