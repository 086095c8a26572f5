# vector instructions of k_mc_emit_batch by section: rebuilds with -DLT_MC_STOP=n (1: staging of records / sign words / neighbour
# records, 2: + vertex list, 3: + vertex pass, 4: + cell list and scan, unset: everything) and counts SQ_INSTS_VALU on the chain
cd $GRAFT_REPO_ROOT
for st in 1 2 3 4 0; do
  if [ $st = 0 ]; then export LIDARHIP_EXTRA_FLAGS=""; else export LIDARHIP_EXTRA_FLAGS="-DLT_MC_STOP=$st"; fi
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mcs$st -o p -- python $GRAFT_REPO_ROOT/tools/prof_chain.py 3 > /dev/null 2>&1)
  echo "stop $st"; python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/mcs$st/p_counter_collection.csv")):
    if "k_mc_emit_batch" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in sorted(acc.items()): print("   ", c, round(sum(d.values())/len(d)))
PY
done
