# address-translation reach (tools/tlb_probe.hip): ns per dependent hop over 1 .. 51 GB -> gpurun_out/r03/tlb_probe.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/tlb_probe tools/tlb_probe.hip 2> /dev/null
timeout 300 /tmp/tlb_probe > gpurun_out/r03/tlb_probe.txt 2>&1
cat gpurun_out/r03/tlb_probe.txt
