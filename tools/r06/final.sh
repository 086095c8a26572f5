# the round's closing call on the GPU box: GPU suite, smoke(), stress cases, then tools/r06/profile.sh and the side files
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{ LT_TEST_ARTIFACTS=$GRAFT_REPO_ROOT/gpurun_out/r06 timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -4
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout 1500 python tools/stress_scatter.py --cases 1000 --oracle --batch 8 2>&1 | tail -2
  timeout 900 python tests/stress_mc.py 300 1 2>&1 | tail -2          # marching cubes vs the oracle (= scikit-image's arrays)
  timeout 900 python tests/stress_tsdf_ref.py 300 7 2>&1 | tail -2; } > gpurun_out/r06/gpu_suite.txt 2>&1   # integrate vs the reference's kernel build
timeout 900 python -m pytest tests/test_tsdf_ref_kernel_gpu.py -m gpu -q -s 2>&1 | grep -E "reference kernel|default volume|passed|failed" > gpurun_out/r06/tsdf_ref_kernel.txt
bash tools/r06/profile.sh > gpurun_out/r06/profile.log 2>&1
for c in 2 3 4; do python tools/chain_pipeline.py $c 16 1 2>&1 | tail -1; done > gpurun_out/r06/chain_pipeline.txt
python tools/chain_pipeline.py 3 8 5 2>&1 | tail -1 >> gpurun_out/r06/chain_pipeline.txt
python tools/mm_pipeline_probe.py 240 2>&1 | grep chains > gpurun_out/r06/mergemesh_pipeline.txt
for g in auto root sharded; do
  LT_BENCH_GATHER=$g HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline 2>gpurun_out/r06/bench_gather_$g.err | tail -1
done > gpurun_out/r06/bench_gather_modes.jsonl
for w in C1 C3 C4; do python bench.py --workload $w --scenes 24 --no-cpu-baseline 2>/dev/null | tail -1; done > gpurun_out/r06/bench_configs.jsonl
for w in C1 C2 C3 C4; do python bench.py --workload $w --scenes 12 --steps 8 --strategy lbvh --no-cpu-baseline 2>/dev/null | tail -1; done > gpurun_out/r06/bench_configs_lbvh.jsonl
# C5 reduced (the real sequence-length ratios / 100) through bench.py's chunked gather: 8 ranks sharing this GPU, gloo transport
LT_BENCH_SHARE_GPU=1 LT_BENCH_BACKEND=gloo LT_BENCH_GATHER=root OMP_NUM_THREADS=4 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --job c5/100 --tris 20000 --scenes 8 --streams 8 --warmup 1 --no-cpu-baseline 2> gpurun_out/r06/bench_c5_job8.err | tail -1 > gpurun_out/r06/bench_c5_job8.json
cat gpurun_out/r06/gpu_suite.txt; tail -c 300 gpurun_out/r06/bench.json
