cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_deform_gpu.py tests/test_projection_gpu.py tests/test_tsdf_gpu.py tests/test_pin_f10_f11_gpu.py -m gpu -q -o faulthandler_timeout=300 2>&1 | grep -v "amdgpu.ids\|Constructing\|Built BVH\|Rendering image" | grep -v "^  \|^    \|^$" | tail -60 > gpurun_out/r05/call2_tests.txt
cat gpurun_out/r05/call2_tests.txt
