# A/B of k_sc_tris variants on ONE box: triangles per workgroup (LT_SC_T) and extra flags; parity first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
O=gpurun_out/r04/sc_ab.txt; : > $O
bline() { python bench.py --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench value', d['value'], 'ms/step', d['ms_per_step'], 'serial', r['avg_kernel_ms'], 'iso', r['isolated']['avg_kernel_ms'], 'one_batch', r.get('one_batch_in_flight',{}).get('value'), 'path_frac', r.get('path_frac'), 'verified', d['verified'])"; }
echo "== default build (LT_SC_T=448)" >> $O
timeout 900 python -m pytest tests/test_trace_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1 >> $O
timeout 900 python tools/stress_scatter.py --cases ${STRESS_CASES:-150} --oracle --batch 8 2>&1 | tail -2 >> $O
bline >> $O; bline >> $O
for V in "$@"; do
  case "$V" in
    env:*) unset LIDARHIP_EXTRA_FLAGS
           python -c "from lidar_transfer_amd import build; build.build_lib()" > /dev/null 2>&1
           echo "== $V" >> $O
           ( export "${V#env:}"; bline >> $O; bline >> $O ) ;;
    *) export LIDARHIP_EXTRA_FLAGS="$V"
       python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
       echo "== $V" >> $O
       bline >> $O; bline >> $O ;;
  esac
done
unset LIDARHIP_EXTRA_FLAGS
python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
echo "== default again" >> $O
bline >> $O
cat $O
