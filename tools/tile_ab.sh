# k_trace4 tile shape A/B on the GPU box (rebuilds with -DLT_TILE4_H/-DLT_TILE4_W)
cd $GRAFT_REPO_ROOT
for hw in "2 8" "1 16" "4 4" "8 2"; do
  set -- $hw
  export LIDARHIP_EXTRA_FLAGS="-DLT_TILE4_H=$1 -DLT_TILE4_W=$2"
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -1
  echo "tile $1 x $2"; CAPS=40 bash tools/prof_lbvh_caps.sh 2>&1 | grep trace4
done
