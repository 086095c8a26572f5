# VALU calibration (tools/valu_calib.hip): s_memtime cycles per wave64 v_fma_f32, then the same launches under the SQ
# counters -> gpurun_out/r03/valu_calib.txt (summary by tools/valu_calib_summary.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
[ -x $R/tools/valu_calib.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $R/tools/valu_calib.bin $R/tools/valu_calib.hip 2> /dev/null
$R/tools/valu_calib.bin > $O/valu_calib_run.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/valu_calib_pmc -o p -- $R/tools/valu_calib.bin > $O/valu_calib_pmc_run.txt 2>&1 || echo "FAILED valu_calib pmc"
cd $R
python tools/valu_calib_summary.py $O > $O/valu_calib.txt 2>&1
cat $O/valu_calib.txt
