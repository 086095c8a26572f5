cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r04/gpu_suite_start.txt
bash tools/r04/r04_scenes_sweep.sh > gpurun_out/r04/scenes_sweep.log 2>&1
cat gpurun_out/r04/gpu_suite_start.txt; tail -30 gpurun_out/r04/scenes_sweep.log
