cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 1024 2048 4096 8192 100000; do
LIDARHIP_SC_CAP=$c rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chaincap -o s -- python $R/tools/prof_chain.py 8 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/chaincap/s_kernel_stats.csv")))
print("cap $c", " ".join("%s %.1f" % (r["Name"].split("<")[0].replace("void ",""), float(r["AverageNs"])/1e3) for r in rows if "k_sc_" in r["Name"]))
PY
done
