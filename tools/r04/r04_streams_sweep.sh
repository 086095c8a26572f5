cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for st in 16 24; do
  python bench.py --streams $st --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st value', d['value'], 'ms/step', d['ms_per_step'], 'verified', d['verified'])"
  python bench.py --streams $st --steps 20 --warmup 5 --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st steps 20 value', d['value'], 'ms/step', d['ms_per_step'])"
done
done
GPU_MAX_HW_QUEUES=6 python bench.py --streams 24 --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq6 streams 24 value', d['value'])"
GPU_MAX_HW_QUEUES=3 python bench.py --streams 24 --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq3 streams 24 value', d['value'])"
