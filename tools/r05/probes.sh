# r05 probes: atomic ceilings by access pattern; FETCH_SIZE calibration for 12-byte patterns (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_probe $R/tools/atomic_probe.hip 2> /dev/null
timeout 300 /tmp/atomic_probe > $O/atomic_probe.txt 2>&1
cat $O/atomic_probe.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/tools/fetch_calib.hip 2> /dev/null
timeout 300 /tmp/fetch_calib > $O/fetch_calib_run.txt 2>&1
cat $O/fetch_calib_run.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/fetch_calib_pmc/$tag -o p -- /tmp/fetch_calib > $O/fetch_calib_$tag.log 2>&1 || echo "FAILED $tag"
done
python3 - <<'P'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r05"
res=collections.defaultdict(dict)
for path in glob.glob(O+"/fetch_calib_pmc/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(float); meta={}
    for r in csv.DictReader(open(path)):
        key=(r["Dispatch_Id"], r["Counter_Name"]); per[key]+=float(r["Counter_Value"]); meta[r["Dispatch_Id"]]=r["Kernel_Name"].split("(")[0]
    acc=collections.defaultdict(list)
    for (d,c),v in per.items(): acc[(meta[d],c)].append(v)
    for (k,c),v in acc.items(): res[k][c]=sum(v[1:])/max(1,len(v)-1)
with open(O+"/fetch_calib_counters.txt","w") as f:
    for k in sorted(res):
        line=k+" "+" ".join(f"{c}={v:.1f}" for c,v in sorted(res[k].items()))
        print(line); f.write(line+"\n")
P
