cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 120 python -c "
import __graft_entry__ as g
g.smoke()
" 2>&1 | grep -v amdgpu.ids | tail -5
timeout 900 python -m pytest tests/test_trace_gpu.py -m gpu -q -x -o faulthandler_timeout=200 2>&1 | grep -v "amdgpu.ids\|Constructing\|Built BVH\|Rendering image" | grep -v "^  \|^    \|^$" | tail -30 > gpurun_out/r05/call5_tests.txt
cat gpurun_out/r05/call5_tests.txt
