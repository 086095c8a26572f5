cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
export LT_RATE_GAPS=1
for v in torch numpy torch numpy; do
  echo "== $v"
  if [ $v = numpy ]; then export LIDARHIP_NO_TORCH=1; else unset LIDARHIP_NO_TORCH; fi
  timeout 120 python tools/hostpipe_rate.py 4 400 2>&1 | grep "gaps\|ms_per_scan" | cut -c1-600
done | tee $O/hostpipe_ab5.txt
