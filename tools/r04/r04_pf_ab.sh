cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04; O=gpurun_out/r04/pf_ab.txt; : > $O
bline() { python bench.py --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench value', d['value'], 'serial', r['avg_kernel_ms'], 'iso', r['isolated']['avg_kernel_ms'], 'one_batch', r.get('one_batch_in_flight',{}).get('value'), 'verified', d['verified'])"; }
for V in "$@"; do
  if [ "$V" = default ]; then unset LIDARHIP_EXTRA_FLAGS; else export LIDARHIP_EXTRA_FLAGS="$V"; fi
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
  echo "== $V" >> $O
  timeout 600 python -m pytest tests/test_trace_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1 >> $O
  bline >> $O; bline >> $O
done
unset LIDARHIP_EXTRA_FLAGS
python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
cat $O
