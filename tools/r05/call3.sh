cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
python tools/r05/dbg_vol.py 2>&1 | grep -v amdgpu.ids | tail -40
