cd $GRAFT_REPO_ROOT
for w in 6 8; do
  export LIDARHIP_EXTRA_FLAGS="-DLT_TSDF_WAVES_ATTR=__attribute__((amdgpu_waves_per_eu($w,$w)))"
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -1
  echo "waves_per_eu $w"; bash tools/prof_chain.sh 2>&1 | grep "k_tsdf_int"
done
