# the round's closing call on the GPU box: GPU suite, smoke(), stress cases, then tools/r04_profile.sh and the chain's side files
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
{ timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -4
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout 1500 python tools/stress_scatter.py --cases 1000 --oracle --batch 8 2>&1 | tail -2
  timeout 900 python tests/stress_mc.py 300 1 2>&1 | tail -2          # marching cubes vs the oracle (= scikit-image's arrays)
  timeout 900 python tests/stress_tsdf_ref.py 300 7 2>&1 | tail -2; } > gpurun_out/r04/gpu_suite.txt 2>&1   # integrate vs the reference's kernel build
# row f2 at BASELINE's full size: the device's marching cubes on the default 800 M-voxel volume vs the oracle (= scikit-image's arrays)
(echo "# python tests/stress_mc_full.py"; timeout 1200 python tests/stress_mc_full.py 2>&1 | grep "default volume") > gpurun_out/r04/mc_full_size_rerun.txt  # (profiles/r04/mc_full_size.txt: the verbose run, kept by hand)
# row f1's pin: the reference's own kernel source (compiled for gfx950, oracle/_ref) next to the product -- what ran, what it cost
timeout 900 python -m pytest tests/test_tsdf_ref_kernel_gpu.py -m gpu -q -s 2>&1 | grep -E "reference kernel|default volume|passed|failed" > gpurun_out/r04/tsdf_ref_kernel.txt
# yardstick for the LBVH build's sort (ms_sort in the bench line): the ROCm library's device radix sort on the same job
(/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/sort_probe tools/sort_probe.hip 2>/dev/null && /tmp/sort_probe && /tmp/sort_probe 2500000) > gpurun_out/r04/sort_probe.txt 2>&1
bash tools/r04/r04_profile.sh > gpurun_out/r04/profile.log 2>&1
{ LIDARHIP_DEBUG_TSDF=1 python tools/prof_chain.py 3 --ranges 2>&1 | grep -v amdgpu.ids | tail -6
  python tools/prof_chain.py 2 5 --ranges 2>&1 | grep -v amdgpu.ids | tail -2; } > gpurun_out/r04/pix_counts.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/chain5 -o s -- python $GRAFT_REPO_ROOT/tools/prof_chain.py 6 5 > /dev/null 2>&1)
cp $(ls gpurun_out/r04/chain5/*/s_kernel_stats.csv gpurun_out/r04/chain5/s_kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r04/chain5_kernel_stats.csv
find gpurun_out/r04 -name "*kernel_trace.csv" -size +2M -delete
for c in 2 3 4; do python tools/chain_pipeline.py $c 16 1 2>&1 | tail -1; done > gpurun_out/r04/chain_pipeline.txt
python tools/chain_pipeline.py 3 8 5 2>&1 | tail -1 >> gpurun_out/r04/chain_pipeline.txt
# the gather modes of the timed region at world size 1 under torchrun (RCCL communicator, gather path executed)
for g in root sharded; do
  LT_BENCH_GATHER=$g HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | tail -1
done > gpurun_out/r04/bench_gather_modes.jsonl
# the other BASELINE configurations through the same bench (parity-test cases; for the record)
for w in C1 C3 C4; do python bench.py --workload $w --scenes 24 --no-cpu-baseline --no-e2e --no-chain 2>/dev/null | tail -1; done > gpurun_out/r04/bench_configs.jsonl
# k_sc_tris variants on this box: SQ_BUSY_CYCLES + duration of the batch launch
bash tools/r04/r04_sc_variants.sh > gpurun_out/r04/sc_variants.log 2>&1
cat gpurun_out/r04/gpu_suite.txt; tail -c 400 gpurun_out/r04/bench.json
