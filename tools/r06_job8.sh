cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
LT_BENCH_DUMP=/tmp/eight.npz LT_BENCH_SHARE_GPU=1 LT_BENCH_BACKEND=gloo LT_BENCH_GATHER=root OMP_NUM_THREADS=4 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --job c5/100 --tris 20000 --scenes 8 --streams 8 --warmup 1 --no-cpu-baseline > gpurun_out/r06/job8.out 2> gpurun_out/r06/job8.err
echo rc=$?; grep -v "^W0\|^\[W" gpurun_out/r06/job8.err | grep -i -B2 -A12 "error\|abort\|assert\|Traceback" | head -80
