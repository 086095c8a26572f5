# per-kernel durations of the marching-cubes kernels on the default volume (rocprofv3 --kernel-trace --stats of tools/prof_chain.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/mcx; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mcx -o s -- python $R/tools/prof_chain.py 12 > /dev/null 2>&1
f=$(ls /tmp/mcx/*/s_kernel_stats.csv /tmp/mcx/s_kernel_stats.csv 2>/dev/null | head -1)
python -c "import csv; [print(r['Name'][:22], r['Calls'], round(float(r['AverageNs'])/1e3,1)) for r in csv.DictReader(open('$f')) if 'k_mc_' in r['Name'][:12]]"
python $R/tools/prof_chain.py 24 --phases 2>/dev/null | tail -1 | cut -c1-200
