# memory-side atomic rate (tools/atomic_probe.hip) -> gpurun_out/r03/atomic_probe.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_probe tools/atomic_probe.hip 2> /dev/null
timeout 300 /tmp/atomic_probe > gpurun_out/r03/atomic_probe.txt 2>&1
cat gpurun_out/r03/atomic_probe.txt
