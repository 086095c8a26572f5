cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmci -o p -- python $R/tools/prof_chain.py 3 > /dev/null 2>&1
cd $R; python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/pmci/p_counter_collection.csv")):
    if "k_tsdf_integrate_cols" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in sorted(acc.items()): print("   ", c, round(sum(d.values())/len(d)))
PY
