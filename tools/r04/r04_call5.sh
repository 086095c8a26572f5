cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_trace_gpu.py tests/test_host_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
timeout 600 python tools/stress_scatter.py --cases 300 --oracle --batch 8 2>&1 | tail -1
CAPS="1024 1536 8192" bash tools/r04/r04_cap_iso.sh "" | grep -v "^$"
python bench.py --no-cpu-baseline --no-other --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['verified'], 'e2e single', d['e2e']['single_call']['ms_per_scan'], 'pipelined', (d['e2e'].get('pipelined') or {}).get('ms_per_scan'))"
