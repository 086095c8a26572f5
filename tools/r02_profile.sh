# Round-2 evidence, everything under gpurun_out/r02/ (copied to profiles/r02/ afterwards):
#   pmc_bench/        FETCH_SIZE / WRITE_SIZE / TCC passes on bench.py itself (-> profiles/r02/pmc.json via tools/pmc_to_json.py)
#   pmc_lbvh/         FETCH_SIZE / WRITE_SIZE passes on the LBVH path (tools/prof_scan.py)
#   stats/            rocprofv3 --kernel-trace --stats of the default bench command
#   iso_{scatter,lbvh}/  ... of one scan at a time
#   chain/            ... of the fusion chain (reset -> integrate -> marching cubes -> render) on the default volume
#   bench.json        the default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  if [ "$tag" = "SQ_INSTS_VALU" ]; then tag=SQ_INSTS_VALU_group; fi
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_bench/$tag -o p -- python $R/bench.py --no-cpu-baseline --no-other --no-e2e --no-chain --steps 16 --warmup 2 > $O/pmc_bench_$tag.log 2>&1 || echo "FAILED pmc_bench $tag"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_lbvh/$tag -o p -- python $R/tools/prof_scan.py --reps 5 > $O/pmc_lbvh_$tag.log 2>&1 || echo "FAILED pmc_lbvh $tag"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu-baseline --no-e2e > $O/bench_profiled.json 2> $O/bench_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso_scatter -o s -- python $R/tools/prof_render.py --reps 40 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso_lbvh -o s -- python $R/tools/prof_scan.py --reps 40 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain -o s -- python $R/tools/prof_chain.py 12 > /dev/null 2>&1
cd $R
python tools/pmc_to_json.py gpurun_out/r02/pmc_bench gpurun_out/r02/pmc.json --command "python bench.py --no-cpu-baseline --no-other --no-e2e --no-chain --steps 16 --warmup 2" > gpurun_out/r02/pmc_to_json.log 2>&1
python tools/pmc_to_json.py gpurun_out/r02/pmc_lbvh gpurun_out/r02/pmc_lbvh.json --command "python tools/prof_scan.py --reps 5" >> gpurun_out/r02/pmc_to_json.log 2>&1
python tools/wave_times.py --quad > gpurun_out/r02/wave_times_quad.txt 2>&1
LIDARHIP_DEBUG_HIER=1 python tools/wave_times.py --hier 2>&1 | grep -A20 "^k_hierarchy4:" > gpurun_out/r02/wave_times_hier.txt
./tools/pcie_probe.bin > gpurun_out/r02/pcie_probe.txt 2>&1
./tools/hostpipe_probe.bin 4 > gpurun_out/r02/hostpipe_probe.txt 2>&1
python tools/bench_aux.py > gpurun_out/r02/bench_aux.jsonl 2> /dev/null
# the stats CSVs are what is kept; the per-dispatch traces are large
find $O -name "*kernel_trace.csv" -size +2M -delete
python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err
tail -c 400 gpurun_out/r02/bench.json
