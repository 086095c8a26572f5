# k_sc_tris A/B on the GPU box: rebuild liblidarhip.so with each flag set, parity (trace tests + a short stress run),
# the bench's device-resident figure on C2 and C3, isolated kernel times (rocprofv3 --stats of one scan at a time)
#   (default)            per-vertex records, positions gathered by the lanes that have candidates
#   -DLT_SC_EARLY_POS    ... positions gathered by every lane at once
#   -DLT_SC_NO_FAST      round 2's kernel: tri_bins for every triangle
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sc_ab
for fl in "" "-DLT_SC_EARLY_POS" "-DLT_SC_NO_FAST"; do
  export LIDARHIP_EXTRA_FLAGS="$fl"
  tag=$(echo "x$fl" | tr -c 'A-Za-z0-9_' '_')
  echo "=== flags: [$fl]"
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -2
  python -m pytest tests/test_trace_gpu.py -x -q -m gpu 2>&1 | tail -1
  python tools/stress_scatter.py --cases 150 2>&1 | tail -2
  for wlk in C2 C3; do
    python bench.py --workload $wlk --no-cpu-baseline --no-e2e --no-chain --no-other 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$wlk', 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'serial', r['avg_kernel_ms'], 'isolated', r['isolated']['avg_kernel_ms'], 'verified', d['verified'], 'tests/ray', r['mt_tests_per_ray'], 'cand/tri', r['candidate_bins_per_triangle'])"
  done
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sc_ab/$tag -o q -- python $GRAFT_REPO_ROOT/tools/prof_render.py --reps 20 > /dev/null 2>&1)
  python - <<PY
import csv, glob
for p in glob.glob('gpurun_out/sc_ab/$tag/**/q_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'k_sc_' in r['Name']: print('   ', r['Name'].split('(')[0], r['Calls'], r['AverageNs'], r['MinNs'])
PY
done
