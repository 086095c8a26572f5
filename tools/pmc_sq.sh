# SQ instruction-issue counters of one isolated C2 render (scatter strategy): python tools/pmc_summary.py gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc2/$tag -o p -- python $R/tools/prof_render.py --reps 6 > $R/gpurun_out/pmc2_$tag.log 2>&1 || echo "FAILED $tag"
done
cd $R; python tools/pmc_summary.py gpurun_out/pmc2 | grep -A16 "k_sc_tris<false"
