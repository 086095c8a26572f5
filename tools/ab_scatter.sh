cd $GRAFT_REPO_ROOT
python -m pytest tests/test_trace_gpu.py -x -q -m gpu 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['isolated']['avg_kernel_ms'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcf/$grp -o p -- python $R/bench.py --no-cpu-baseline --no-other --no-e2e --no-chain --steps 16 --warmup 2 > /dev/null 2>&1
done
cd $R; python tools/pmc_to_json.py gpurun_out/pmcf gpurun_out/pmcf.json | grep " 8 "
