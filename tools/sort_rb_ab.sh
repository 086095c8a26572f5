cd $GRAFT_REPO_ROOT
for v in "-DLT_RB=8" "-DLT_RB=8 -DLT_SORT_TILE=2048" "-DLT_RB=10"; do
  export LIDARHIP_EXTRA_FLAGS="$v"
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -1
  echo "== $v"; python -m pytest tests/test_trace_gpu.py -x -q -k "lbvh or scene_api or baseline_sizes" 2>&1 | grep -E "passed|failed" | tail -1
  CAPS=40 bash tools/prof_lbvh_caps.sh 2>&1 | grep "k_hist\|k_scan\|k_scatter"
  python tools/prof_scan.py --reps 10 2>/dev/null | head -1
done
unset LIDARHIP_EXTRA_FLAGS; python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null
