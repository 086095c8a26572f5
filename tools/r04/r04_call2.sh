cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_projection_gpu.py tests/test_deform_gpu.py tests/test_host_gpu.py tests/test_tsdf_gpu.py tests/test_mc_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r04/call2_tests.txt
cat gpurun_out/r04/call2_tests.txt
timeout 900 python bench.py --no-cpu-baseline --no-e2e > gpurun_out/r04/call2_bench.json 2> gpurun_out/r04/call2_bench.err
tail -5 gpurun_out/r04/call2_bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r04/call2_bench.json'))
print('value',d['value'],'ms_per_step',d['ms_per_step'],'verified',d['verified'],'failed',d.get('failed_legs'))
r=d['roofline']; print('frac',r['frac'],'path_frac',r.get('path_frac'),'one_batch',r.get('one_batch_in_flight'))
print('proj',json.dumps(d.get('projection')))
fp=d.get('deform_from_points'); 
if fp: print({k:fp[k] for k in ('ms_per_output_scan','ms_per_output_scan_one_call','phase_ms','verified','points_written')})
P
