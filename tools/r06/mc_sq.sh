#!/bin/bash
# SQ counters of the marching-cubes kernels on the fusion chain (separate --pmc passes with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/mcsq; mkdir -p $R/gpurun_out/mcsq
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/mcsq/$tag -o p -- python $R/tools/prof_chain.py 4 > $R/gpurun_out/mcsq_$tag.log 2>&1 || echo "FAILED $tag"
done
cd $R; python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/mcsq/*/**/p_counter_collection.csv", recursive=True) + glob.glob("gpurun_out/mcsq/*/p_counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"][:28], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, d), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
for k in sorted(acc):
    if "k_mc_" not in k and "k_tsdf_integrate" not in k: continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-26s %14.0f" % (c, sum(v) / len(v)))
PY
