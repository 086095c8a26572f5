# where k_tsdf_integrate_pix spends its time: rebuild with parts switched off (results are wrong then) and time the kernel
cd $GRAFT_REPO_ROOT
for fl in "" "-DLT_PIX_NO_BITS" "-DLT_PIX_NO_EVAL" "-DLT_PIX_NO_MARK" "-DLT_PIX_NO_STAGE" "-DLT_PIX_NO_EVAL -DLT_PIX_NO_MARK" "-DLT_PIX_NO_EVAL -DLT_PIX_NO_MARK -DLT_PIX_NO_STAGE"; do
  export LIDARHIP_EXTRA_FLAGS="$fl"
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
  rm -rf gpurun_out/chain2
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/chain2 -o s -- python $GRAFT_REPO_ROOT/tools/prof_chain.py 12 > /dev/null 2>&1)
  echo "[$fl] $(grep integrate_pix gpurun_out/chain2/s_kernel_stats.csv | sed 's/.*)",//' | cut -d, -f1-3)"
done
