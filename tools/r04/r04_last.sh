cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for i in 1 2 3; do python tools/chain_pipeline.py 3 8 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('chain_pipeline 3 8 5:', d['one_chain_ms_per_scan'], d['ms_per_scan'], d['verified'])"; done
python tools/chain_pipeline.py 3 16 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('chain_pipeline 3 16 5:', d['one_chain_ms_per_scan'], d['ms_per_scan'], d['verified'])"
python bench.py > gpurun_out/r04/bench.json 2> gpurun_out/r04/bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_shape.json 2>/dev/null
python -c "
import json
for f in ('bench.json','bench_driver_shape.json'):
    d=json.load(open('gpurun_out/r04/'+f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], d['verified'], r['frac'], r['path_frac'], r.get('whole_path',{}) and r['whole_path'].get('frac_of_peak'), d.get('failed_legs'))"
