# compiler-flag A/B on the GPU box: rebuilds liblidarhip.so with each LIDARHIP_EXTRA_FLAGS set, times one isolated C2
# render and the bench's device-resident figure
cd $GRAFT_REPO_ROOT
for fl in "" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy" "-mllvm -amdgpu-schedule-metric-bias=0" "-O2" "-mllvm -amdgpu-load-store-vectorizer=0"; do
  export LIDARHIP_EXTRA_FLAGS="$fl"
  echo "=== flags: [$fl]"
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -2
  python bench.py --no-cpu-baseline --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['isolated']['avg_kernel_ms'], d['other_strategy']['value'])"
done
