cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_deform_gpu.py tests/test_mc_gpu.py -m gpu -q -x -o faulthandler_timeout=120 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-other > gpurun_out/r04/call6_bench.json 2> gpurun_out/r04/call6_bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r04/call6_bench.json'))
print('value',d['value'],'failed',d.get('failed_legs'))
for k in ('fusion_chain','fusion_chain_nscans5'):
    c=d.get(k); 
    if c: print(k, c['ms_per_scan'], c['phase_ms'], (c.get('pipelined') or {}).get('ms_per_scan'))
fp=d.get('deform_from_points'); 
if fp: print({k:fp[k] for k in ('ms_per_output_scan','ms_per_output_scan_one_call','phase_ms','verified','pipelined')})
P
