cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
run() { tag=$1; shift
  line=$(env "$@" timeout 120 python tools/hostpipe_rate.py 4 160 2>/dev/null | tail -1)
  echo "$tag | $* | $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print({k:d[k] for k in d if k in ('ms_per_scan','GBs','single_call_ms')})" "$line" 2>/dev/null || echo "$line" | cut -c1-200)"
}
{
run torch_a LT_NOP=1
run torch_q8 GPU_MAX_HW_QUEUES=8
run torch_q16 GPU_MAX_HW_QUEUES=16
run torch_q2 GPU_MAX_HW_QUEUES=2
run torch_up1 LIDARHIP_HOSTPIPE_UPLOADERS=1
run torch_dd0 AMD_DIRECT_DISPATCH=0
run torch_b LT_NOP=1
run numpy_a LIDARHIP_NO_TORCH=1
run numpy_q8 LIDARHIP_NO_TORCH=1 GPU_MAX_HW_QUEUES=8
run numpy_q2 LIDARHIP_NO_TORCH=1 GPU_MAX_HW_QUEUES=2
run numpy_up1 LIDARHIP_NO_TORCH=1 LIDARHIP_HOSTPIPE_UPLOADERS=1
} | tee $O/hostpipe_ab2.txt
