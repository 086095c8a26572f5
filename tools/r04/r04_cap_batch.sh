cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cap in 8192 4096 3072 2048; do
  LIDARHIP_SC_CAP=$cap python bench.py --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('batch cap $cap value', d['value'], 'steps20?', d['steps'], 'serial', r['avg_kernel_ms'], 'frac', r['frac'], 'one_batch', r['one_batch_in_flight']['value'])"
done; done
for cap in 8192 4096; do
  LIDARHIP_SC_CAP=$cap python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps 20 cap $cap value', d['value'])"
done
