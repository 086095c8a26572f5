cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
for v in torch numpy; do
  echo "== $v"
  if [ $v = numpy ]; then export LIDARHIP_NO_TORCH=1; else unset LIDARHIP_NO_TORCH; fi
  timeout 120 python tools/hostpipe_rate.py 4 160 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-400
done | tee $O/hostpipe_ab4.txt
