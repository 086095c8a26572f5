# Round-end evidence: (1) scatter PMC passes, (2) rocprofv3 --kernel-trace --stats of the default bench command,
# (3) the full default bench line (with cpu_baseline).  Outputs under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
bash $R/tools/pmc_render.sh
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats -o s -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/bench_profiled.json 2> $R/gpurun_out/final/bench_profiled.err
cd $R && python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 600 gpurun_out/final/bench.json
