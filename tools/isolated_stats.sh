# rocprofv3 --kernel-trace --stats of ONE scan at a time (no overlap): scatter strategy (tools/prof_render.py) and
# LBVH strategy (tools/prof_scan.py) on workload C2.  Outputs under gpurun_out/iso/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/iso/scatter -o s -- python $R/tools/prof_render.py --reps 40 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/iso/lbvh -o s -- python $R/tools/prof_scan.py --reps 40 > /dev/null 2>&1
ls $R/gpurun_out/iso/*/
