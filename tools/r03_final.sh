# the round's closing call on the GPU box: GPU suite, smoke(), stress cases, then tools/r03_profile.sh and the chain's side files
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{ timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout 900 python tools/stress_scatter.py --cases 600 2>&1 | tail -2; } > gpurun_out/r03/gpu_suite.txt 2>&1
bash tools/r03_profile.sh > gpurun_out/r03/profile.log 2>&1
{ LIDARHIP_DEBUG_TSDF=1 python tools/prof_chain.py 3 --ranges 2>&1 | grep -v amdgpu.ids | tail -6
  python tools/prof_chain.py 2 5 --ranges 2>&1 | grep -v amdgpu.ids | tail -2; } > gpurun_out/r03/pix_counts.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/chain5 -o s -- python $GRAFT_REPO_ROOT/tools/prof_chain.py 6 5 > /dev/null 2>&1)
cp $(ls gpurun_out/r03/chain5/*/s_kernel_stats.csv gpurun_out/r03/chain5/s_kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r03/chain5_kernel_stats.csv
find gpurun_out/r03 -name "*kernel_trace.csv" -size +2M -delete
bash tools/tlb_probe.sh > /dev/null 2>&1
bash tools/atomic_probe.sh > /dev/null 2>&1
bash tools/occ_probe.sh > /dev/null 2>&1
for c in 2 3 4; do python tools/chain_pipeline.py $c 16 1 2>&1 | tail -1; done > gpurun_out/r03/chain_pipeline.txt
python tools/chain_pipeline.py 3 8 5 2>&1 | tail -1 >> gpurun_out/r03/chain_pipeline.txt
for st in 1 2 3; do
  export LIDARHIP_EXTRA_FLAGS="-DLT_MC_STAMP=$st"   # (exported: the library rebuilds itself when its flags differ from the caller's)
  python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
  echo "== LT_MC_STAMP=$st ($( [ $st = 1 ] && echo k_mc_words || ( [ $st = 2 ] && echo k_mc_compact || echo k_mc_emit_batch ) ))"; python tools/mc_wave_times.py 2>&1 | grep -v amdgpu.ids | tail -12
done > gpurun_out/r03/mc_wave_times.txt
unset LIDARHIP_EXTRA_FLAGS
python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
cat gpurun_out/r03/gpu_suite.txt; tail -c 400 gpurun_out/r03/bench.json
