cd $GRAFT_REPO_ROOT
for rep in 1 2; do for rb in 512 128 64 32; do
  LIDARHIP_SC_REST_BLOCKS=$rb python bench.py --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('rest blocks $rb value', d['value'], 'verified', d['verified'], 'one_batch', r['one_batch_in_flight']['value'])"
done; done
