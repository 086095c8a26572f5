# HBM traffic of the dominant kernel measured on the bench command itself (batches of 8 scans, 12 scenes cycled, one
# shared ray set): separate FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only.  Outputs under gpurun_out/pmcb/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcb/$tag -o p -- python $R/bench.py --no-cpu-baseline --no-other --steps 100 --warmup 10 > $R/gpurun_out/pmcb_$tag.log 2>&1 || echo "FAILED $tag"
done
