cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_pin_f10_f11_gpu.py tests/test_tsdf_gpu.py -m gpu -q -s -k "f15 or numpy_mode" -o faulthandler_timeout=300 2>&1 | grep -v "amdgpu.ids\|Constructing\|Built BVH\|Rendering image" | grep -v "^  \|^    \|^$" | tail -40 > gpurun_out/r05/call4_tests.txt
cat gpurun_out/r05/call4_tests.txt
