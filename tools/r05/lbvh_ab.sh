# LBVH strategy, ONE scan at a time (tools/prof_scan.py): kernel times per variant.  bash tools/r05/lbvh_ab.sh "<tag>=<ENV=VAL>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
for v in "$@"; do
  tag=${v%%=*}; envs=${v#*=}; [ "$envs" = "-" ] && envs="LT_NOP=1"
  rm -rf /tmp/lb_$tag
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lb_$tag -o s -- python $R/tools/prof_scan.py --reps 40 > /tmp/lb_$tag.log 2>&1
  python3 - "$tag" "$envs" > $O/lbvh_ab_$tag.txt <<'P'
import csv, sys
tag, envs = sys.argv[1], sys.argv[2]
print(f"# variant {tag}: {envs}")
tot = 0; build = 0
for r in sorted(csv.DictReader(open(f"/tmp/lb_{tag}/s_kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"])):
    k = r["Name"].split("(")[0].replace("void ", "").strip(); calls = int(r["Calls"]); us = float(r["AverageNs"]) / 1e3
    if calls < 30 or k.startswith("__amd") or "at::native" in k: continue
    per_scan = us * calls / 40.0
    print(f"{k[:36]:36s} calls {calls:4d} avg {us:8.2f} us  per scan {per_scan:8.2f} us")
    tot += per_scan
    if not k.startswith("k_trace"): build += per_scan
print(f"kernels per scan: {tot:.1f} us (build {build:.1f}, trace {tot - build:.1f})")
P
  cat $O/lbvh_ab_$tag.txt
done
