# radix sort tile size A/B on the GPU box (rebuilds with -DLT_SORT_TILE=n): LBVH build kernels of an isolated C2 scan
cd $GRAFT_REPO_ROOT
for t in 4096 2048 1024; do
  export LIDARHIP_EXTRA_FLAGS="-DLT_SORT_TILE=$t"
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -1
  echo "tile $t"; python -m pytest tests/test_trace_gpu.py -x -q -k "lbvh or scene_api or baseline_sizes" 2>&1 | grep -E "passed|failed" | tail -1
  CAPS=40 bash tools/prof_lbvh_caps.sh 2>&1 | grep "k_hist\|k_scan\|k_scatter"
done
