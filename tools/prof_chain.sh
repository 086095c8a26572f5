# rocprofv3 --kernel-trace --stats of the fusion chain on the default 2000 x 2000 x 200 volume -> gpurun_out/chain/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chain -o s -- python $R/tools/prof_chain.py 12 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/chain/s_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if float(r["TotalDurationNs"]) > 2e5: print(n[:44].ljust(44), r["Calls"].rjust(4), "avg us", round(float(r["AverageNs"])/1e3,2))
PY
