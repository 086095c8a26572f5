#!/bin/bash
# marching cubes after a kernel change: the pins, the stress run, kernel durations, the emission's sections
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_pin_f10_f11_gpu.py tests/test_mc_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tests/stress_mc.py 150 7 2>&1 | tail -2
bash tools/mc_kernels.sh 2>&1 | tail -12
export LIDARHIP_EXTRA_FLAGS=-DLT_MC_STAMP=3
python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" 2>&1 | tail -1
timeout 200 python tools/mc_wave_times.py
