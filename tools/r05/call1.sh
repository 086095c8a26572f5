cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -q -x -o faulthandler_timeout=300 2>&1 | grep -v "amdgpu.ids\|Constructing\|Built BVH\|Rendering image" | tail -15 > gpurun_out/r05/call1_tests.txt
cat gpurun_out/r05/call1_tests.txt
timeout 900 python bench.py > gpurun_out/r05/call1_bench.json 2> gpurun_out/r05/call1_bench.err
tail -c 1500 gpurun_out/r05/call1_bench.json
