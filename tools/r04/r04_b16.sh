cd $GRAFT_REPO_ROOT
export LIDARHIP_EXTRA_FLAGS="-DLT_SC_MAX_BATCH=16"
python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -2
for cfg in "8 24" "16 32" "16 48" "12 36"; do set -- $cfg
  LT_BENCH_MAX_BATCH=16 python bench.py --batch $1 --streams $2 --calls-per-step $((64/$1)) --no-cpu-baseline --no-other --no-e2e --no-chain 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $1 streams $2 value', d['value'], d['config']['scans_per_step'], d['verified'])"
done
unset LIDARHIP_EXTRA_FLAGS
python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
