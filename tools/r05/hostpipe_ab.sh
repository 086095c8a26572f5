# hostpipe upload variants in a numpy-only process (system ROCm runtime) and in a torch process (the wheel's runtime)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
run() { # tag, env...
  tag=$1; shift
  line=$(env "$@" timeout 120 python tools/hostpipe_rate.py 4 120 2>/dev/null | tail -1)
  echo "$tag | $* | $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print({k:d[k] for k in d if k in ('ms_per_scan','GBs','single_call_ms','upload_GBs')})" "$line" 2>/dev/null || echo "$line" | cut -c1-200)"
}
{
run numpy LIDARHIP_NO_TORCH=1
run numpy_reg LIDARHIP_NO_TORCH=1 LIDARHIP_HOSTPIPE_REGISTER=1
run torch LT_NOP=1
run torch_reg LIDARHIP_HOSTPIPE_REGISTER=1
run torch_up3 LIDARHIP_HOSTPIPE_UPLOADERS=3
run torch_up4 LIDARHIP_HOSTPIPE_UPLOADERS=4
run torch_pinmin64k GPU_PINNED_MIN_XFER_SIZE=64
run torch_pinmin16m GPU_PINNED_MIN_XFER_SIZE=16384
run torch_pinxfer64 GPU_PINNED_XFER_SIZE=64
run torch_pinxfer256 GPU_PINNED_XFER_SIZE=256
run torch_stage32 GPU_STAGING_BUFFER_SIZE=32
run torch_nosdma HSA_ENABLE_SDMA=0
run numpy_nosdma LIDARHIP_NO_TORCH=1 HSA_ENABLE_SDMA=0
} | tee $O/hostpipe_ab.txt
python - <<'P'
import torch, ctypes
print("torch", torch.__version__, "hip", torch.version.hip)
import os
for l in open("/proc/self/maps"):
    if "libamdhip64" in l or "libhsa-runtime" in l: print(l.split()[-1]); 
P
