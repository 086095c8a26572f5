# per-kernel durations of the LBVH path, one C2 scan at a time (rocprofv3 --kernel-trace --stats of tools/prof_scan.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/lbx; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lbx -o s -- python $R/tools/prof_scan.py --reps 40 > /dev/null 2>&1
f=$(ls /tmp/lbx/*/s_kernel_stats.csv /tmp/lbx/s_kernel_stats.csv 2>/dev/null | head -1)
python -c "import csv; [print(r['Name'][:26], r['Calls'], round(float(r['AverageNs'])/1e3,1)) for r in csv.DictReader(open('$f')) if float(r['Percentage']) > 0.5]"
