cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in ${CAPS:-40 48}; do
LIDARHIP_STEP_CAP=$c rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lb$c -o s -- python $R/tools/prof_scan.py --reps 40 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/lb$c/s_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if n.startswith("k_") or "k_trace" in n or "fill" in n.lower():
        print("cap $c", n[:40], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,2))
PY
done
