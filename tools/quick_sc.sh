# parity of the scatter path + isolated kernel times of one C2 render (rocprofv3 --stats)
cd $GRAFT_REPO_ROOT && python -m pytest tests/test_trace_gpu.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/qs -o q -- python $GRAFT_REPO_ROOT/tools/prof_render.py --reps 20 > /dev/null 2>&1
python - <<'PY'
import csv, os
for r in csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/qs/q_kernel_stats.csv')):
    if 'k_sc_' in r['Name']: print(r['Name'].split('(')[0], r['Calls'], r['AverageNs'], r['MinNs'])
PY
