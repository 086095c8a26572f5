# Round-6 evidence, everything under gpurun_out/r06/ (summaries copied to profiles/r06/ afterwards):
#   serial_probe/         rocprofv3 --kernel-trace --stats of `bench.py --probe-only`: ONLY launches of the timed region's
#                         shape, exclusive on one stream -> roofline.avg_kernel_ms reproducible from a CSV
#   pmc_bench/            FETCH_SIZE / WRITE_SIZE / SQ passes on bench.py itself (-> pmc.json via tools/pmc_to_json.py)
#   ea/                   TCC_EA0 read-request / write-request / atomic counts of `bench.py --probe-only` (-> ea_requests.json)
#   pmc_lbvh/             ... on the LBVH path (tools/prof_scan.py) (-> pmc_lbvh.json)
#   pmc_chain/ + chain/   ... on the fusion chain (tools/prof_chain.py) (-> pmc_chain.json)
#   stats/, iso_*/        rocprofv3 --kernel-trace --stats of the default bench command / of one scan at a time
#   bench.json            the default bench line (run LAST, with the json files above already in profiles/r06/)
#   bench_extras.json     tools/bench_chains.py: the side records
# (--pmc passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
P=$R/profiles/r06
mkdir -p $O $P
rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_probe -o s -- python $R/bench.py --probe-only > $O/serial_probe.json 2> $O/serial_probe.err
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  if [ "$tag" = "SQ_INSTS_VALU" ]; then tag=SQ_INSTS_VALU_group; fi
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_bench/$tag -o p -- python $R/bench.py --no-cpu-baseline --steps 16 --warmup 2 > $O/pmc_bench_$tag.log 2>&1 || echo "FAILED pmc_bench $tag"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_lbvh/$tag -o p -- python $R/tools/prof_scan.py --reps 5 > $O/pmc_lbvh_$tag.log 2>&1 || echo "FAILED pmc_lbvh $tag"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_chain/$tag -o p -- python $R/tools/prof_chain.py 4 > $O/pmc_chain_$tag.log 2>&1 || echo "FAILED pmc_chain $tag"
done
# memory-side requests of the scatter kernels (two counters per pass: the TCC counters are per channel)
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_ATOMIC_sum TCC_REQ_sum"; do
  tag=$(echo $grp | tr ' ' '-')
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/ea/$tag -o p -- python $R/bench.py --probe-only > $O/ea_$tag.log 2>&1 || echo "FAILED ea $tag"
done
# the chain's write requests by size (integrate: 46 MB of writes for 14 MB of voxels -- short z runs in four field arrays)
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $grp | tr ' ' '-')
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/ea_chain/$tag -o p -- python $R/tools/prof_chain.py 4 > $O/ea_chain_$tag.log 2>&1 || echo "FAILED ea_chain $tag"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso_scatter -o s -- python $R/tools/prof_render.py --reps 40 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso_lbvh -o s -- python $R/tools/prof_scan.py --reps 40 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain -o s -- python $R/tools/prof_chain.py 12 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain5f -o s -- python $R/tools/prof_chain.py 8 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mergemesh -o s -- python $R/tools/mm_pipeline_probe.py 96 > $O/mergemesh_pipeline_profiled.txt 2>&1
cd $R
python tools/pmc_to_json.py gpurun_out/r06/pmc_bench gpurun_out/r06/pmc.json --command "python bench.py --no-cpu-baseline --steps 16 --warmup 2" > gpurun_out/r06/pmc_to_json.log 2>&1
python tools/pmc_to_json.py gpurun_out/r06/pmc_lbvh gpurun_out/r06/pmc_lbvh.json --command "python tools/prof_scan.py --reps 5" >> gpurun_out/r06/pmc_to_json.log 2>&1
python tools/pmc_chain_to_json.py gpurun_out/r06/pmc_chain $(ls gpurun_out/r06/chain/*/s_kernel_stats.csv gpurun_out/r06/chain/s_kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r06/pmc_chain.json >> gpurun_out/r06/pmc_to_json.log 2>&1
python tools/ea_to_json.py gpurun_out/r06/ea gpurun_out/r06/ea_requests.json >> gpurun_out/r06/pmc_to_json.log 2>&1
python tools/pmc_summary.py gpurun_out/r06/pmc_bench > gpurun_out/r06/pmc_bench.txt 2>&1
python tools/pmc_summary.py gpurun_out/r06/pmc_lbvh > gpurun_out/r06/pmc_lbvh.txt 2>&1
python tools/pmc_summary.py gpurun_out/r06/pmc_chain > gpurun_out/r06/pmc_chain.txt 2>&1
python tools/pmc_summary.py gpurun_out/r06/ea_chain > gpurun_out/r06/ea_chain.txt 2>&1
# bench.py reads pmc*.json from profiles/: put them there before the final line is measured
cp gpurun_out/r06/pmc.json gpurun_out/r06/pmc_lbvh.json gpurun_out/r06/pmc_chain.json gpurun_out/r06/ea_requests.json $P/ 2>/dev/null
for d in serial_probe stats iso_scatter iso_lbvh chain chain5f mergemesh; do
  f=$(ls $O/$d/*/s_kernel_stats.csv $O/$d/s_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
done
# the stats CSVs are what is kept; the per-dispatch traces are large
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +4M -delete
python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err
python tools/bench_chains.py --out gpurun_out/r06/bench_extras.json > gpurun_out/r06/bench_extras.summary 2> gpurun_out/r06/bench_extras.err
wc -c gpurun_out/r06/bench.json; tail -c 400 gpurun_out/r06/bench.json; cat gpurun_out/r06/bench_extras.summary
