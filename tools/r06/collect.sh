#!/bin/bash
# after tools/r06/final.sh: the summaries the judge reads, gpurun_out/r06/ (scratch) -> profiles/r06/ (tracked)
cd "$(dirname "$0")/../.."
for f in bench.json bench_c5_job8.json bench_configs.jsonl bench_configs_lbvh.jsonl bench_extras.json bench_gather_modes.jsonl \
         bench_profiled.json chain5f_kernel_stats.csv chain_kernel_stats.csv chain_pipeline.txt ea_chain.txt ea_requests.json \
         f13b_f14b_table.json gpu_suite.txt iso_lbvh_kernel_stats.csv iso_scatter_kernel_stats.csv mergemesh_kernel_stats.csv \
         mergemesh_pipeline.txt pmc.json pmc_bench.txt pmc_chain.json pmc_chain.txt pmc_lbvh.json pmc_lbvh.txt serial_probe.json \
         serial_probe_kernel_stats.csv tsdf_ref_kernel.txt; do
  [ -f gpurun_out/r06/$f ] && cp gpurun_out/r06/$f profiles/r06/$f
done
[ -f gpurun_out/r06/stats_kernel_stats.csv ] && cp gpurun_out/r06/stats_kernel_stats.csv profiles/r06/bench_kernel_stats.csv
ls -la profiles/r06 | wc -l
