cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
export LT_RATE_GAPS=1
{
echo "== torch n=1500"; timeout 200 python tools/hostpipe_rate.py 4 1500 2>&1 | grep "gaps\|ms_per_scan" | cut -c1-700
echo "== torch n=1500 ROC_SIGNAL_POOL_SIZE=4096"; ROC_SIGNAL_POOL_SIZE=4096 timeout 200 python tools/hostpipe_rate.py 4 1500 2>&1 | grep "gaps\|ms_per_scan" | cut -c1-700
echo "== numpy n=1500"; LIDARHIP_NO_TORCH=1 timeout 200 python tools/hostpipe_rate.py 4 1500 2>&1 | grep "gaps\|ms_per_scan" | cut -c1-300
} | tee $O/hostpipe_ab6.txt
