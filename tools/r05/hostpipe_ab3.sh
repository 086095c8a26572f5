cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
run() { tag=$1; shift
  line=$(env "$@" timeout 120 python tools/hostpipe_rate.py 4 160 2>/dev/null | tail -1)
  echo "$tag | $* | $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print({k:d[k] for k in d if k in ('ms_per_scan','GBs','single_call_ms')})" "$line" 2>/dev/null || echo "$line" | cut -c1-300)"
}
{
run torch_pinned LT_RATE_PINNED=1
run torch_pinned_up1 LT_RATE_PINNED=1 LIDARHIP_HOSTPIPE_UPLOADERS=1
run numpy_pinned LIDARHIP_NO_TORCH=1 LT_RATE_PINNED=1
run torch LT_NOP=1
} | tee $O/hostpipe_ab3.txt
