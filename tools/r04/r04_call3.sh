cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_mc_gpu.py tests/test_multigpu_gpu.py tests/test_pin_f10_f11_gpu.py -m gpu -q -x -s 2>&1 | grep -v "amdgpu.ids\|Constructing\|Built BVH\|Rendering image" | tail -25 > gpurun_out/r04/call3_tests.txt
cat gpurun_out/r04/call3_tests.txt
( time python bench.py --no-cpu-baseline --no-e2e > gpurun_out/r04/call3_bench.json 2> gpurun_out/r04/call3_bench.err ) 2>&1 | tail -4
( time python bench.py --no-cpu-baseline --no-e2e --no-chain --no-other > /dev/null 2>&1 ) 2>&1 | tail -4
python - <<'P'
import json
d=json.load(open('gpurun_out/r04/call3_bench.json'))
print('value',d['value'],'failed',d.get('failed_legs'))
print(json.dumps(d['fusion_chain'].get('marching_cubes_cases')))
fp=d.get('deform_from_points'); 
if fp: print({k:fp[k] for k in ('ms_per_output_scan','ms_per_output_scan_one_call','phase_ms','verified')})
P
