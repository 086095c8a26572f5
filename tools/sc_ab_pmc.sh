# instruction counts + durations of the scatter kernels for the per-vertex path vs round 2's kernel (-DLT_SC_NO_FAST):
# SQ_INSTS_VALU / LDS / VMEM per isolated C2 render (tools/prof_render.py), and three repetitions of the C2 bench each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sc_ab2
for fl in "" "-DLT_SC_NO_FAST" "" "-DLT_SC_NO_FAST"; do
  export LIDARHIP_EXTRA_FLAGS="$fl"
  tag=$(echo "x$fl" | tr -c 'A-Za-z0-9_' '_')
  echo "=== flags: [$fl]"
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -1
  for rep in 1 2; do
    python bench.py --workload C2 --no-cpu-baseline --no-e2e --no-chain --no-other 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C2 Mrays/s', d['value'], 'serial', r['avg_kernel_ms'], 'isolated', r['isolated']['avg_kernel_ms'])"
  done
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sc_ab2/$tag/SQ -o p -- python $GRAFT_REPO_ROOT/tools/prof_render.py --reps 6 > /dev/null 2>&1)
  python tools/pmc_summary.py gpurun_out/sc_ab2/$tag | grep -A9 "k_sc_tris<false\|k_sc_verts\|k_sc_rest<false"
done
