# resident waves per CU vs LDS per workgroup (tools/occ_probe.hip) -> gpurun_out/r03/occ_probe.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/occ_probe tools/occ_probe.hip 2> /dev/null
timeout 300 /tmp/occ_probe > gpurun_out/r03/occ_probe.txt 2>&1
cat gpurun_out/r03/occ_probe.txt
