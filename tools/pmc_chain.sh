# SQ instruction-issue counters of the fusion chain's kernels: python tools/pmc_summary.py gpurun_out/pmc3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc3/$tag -o p -- python $R/tools/prof_chain.py 4 > $R/gpurun_out/pmc3_$tag.log 2>&1 || echo "FAILED $tag"
done
cd $R; python tools/pmc_summary.py gpurun_out/pmc3 | grep -A22 "k_tsdf_integrate_cols\|k_mc_words\|k_mc_emit_batch\|k_mc_compact"
