# VERDICT r03 "what's weak" 3(b): is the headline Infinity-Cache assisted?  k_sc_tris reads faces + vertices = 18.3 MB per
# C2 scene; 12 scenes = 220 MB < 256 MiB.  Sweep the number of distinct scenes cycled (timing runs, then FETCH_SIZE /
# WRITE_SIZE passes per count) -> gpurun_out/r04/scenes_sweep.jsonl
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
: > $O/scenes_sweep.jsonl
for n in 4 12 24 32 48 64; do
  for rep in 1 2; do
    python bench.py --scenes $n --no-cpu-baseline --no-other --no-e2e --no-chain 2>> $O/scenes_sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'scenes':$n,'rep':$rep,'value':d['value'],'ms_per_step':d['ms_per_step'],'verified':d['verified'],'serial_kernel_ms':r['avg_kernel_ms'],'frac':r['frac'],'in_situ_ms':r.get('in_situ',{}).get('avg_kernel_ms')}))" >> $O/scenes_sweep.jsonl
  done
done
cd /tmp && export TMPDIR=/tmp
for n in 12 32 64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_scenes$n/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --scenes $n --no-cpu-baseline --no-other --no-e2e --no-chain --steps 16 --warmup 2 > $O/pmc_scenes${n}_$c.log 2>&1 || echo "FAILED pmc scenes $n $c"
  done
  (cd $GRAFT_REPO_ROOT && python tools/pmc_to_json.py gpurun_out/r04/pmc_scenes$n gpurun_out/r04/pmc_scenes$n.json --command "python bench.py --scenes $n --no-cpu-baseline --no-other --no-e2e --no-chain --steps 16 --warmup 2" > $O/pmc_scenes$n.txt 2>&1)
done
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +4M -delete
cat $O/scenes_sweep.jsonl; cat $O/pmc_scenes*.txt
