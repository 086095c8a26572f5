# A/B of the fusion chain's kernels: for every variant (an env assignment, "-" = none) the kernel times (rocprofv3 --stats) and
# FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes).   bash tools/r05/chain_ab.sh "<tag>=<ENV=VAL>" ...  -> gpurun_out/r05/chain_ab_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
NOBS=${NOBS:-1}
for v in "$@"; do
  tag=${v%%=*}; envs=${v#*=}; [ "$envs" = "-" ] && envs="LT_NOP=1"
  rm -rf /tmp/ab_$tag
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$tag/stats -o s -- python $R/tools/prof_chain.py 12 $NOBS > /tmp/ab_$tag.log 2>&1
  if [ -z "$NO_PMC" ]; then
  for grp in FETCH_SIZE WRITE_SIZE; do
    env $envs rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/ab_$tag/$grp -o p -- python $R/tools/prof_chain.py 4 $NOBS > /dev/null 2>&1 || echo "FAILED $grp"
  done
  fi
  python3 - "$tag" "$envs" > $O/chain_ab_$tag.txt <<'P'
import csv, glob, collections, sys
tag, envs = sys.argv[1], sys.argv[2]
def short(n): return n.split("(")[0].replace("void ","").strip()
times = {}
for r in csv.DictReader(open(f"/tmp/ab_{tag}/stats/s_kernel_stats.csv")):
    times[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
pmc = collections.defaultdict(dict)
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(f"/tmp/ab_{tag}/{grp}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float); meta = {}
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != grp: continue
            per[r["Dispatch_Id"]] += float(r["Counter_Value"]); meta[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        acc = collections.defaultdict(list)
        for d, v in per.items(): acc[meta[d]].append(v)
        for k, v in acc.items(): pmc[k][grp] = sum(v) / len(v)
print(f"# variant {tag}: {envs}")
tot = 0
for k, (calls, us) in sorted(times.items(), key=lambda t: -t[1][0] * t[1][1]):
    if calls < 4 or k.startswith("__amd") or "at::native" in k: continue
    f, w = pmc.get(k, {}).get("FETCH_SIZE"), pmc.get(k, {}).get("WRITE_SIZE")
    hbm = f"{(2 * f + w) * 1024 / 1e6:8.1f} MB (2 x FETCH {f * 1024 / 1e6:.1f} + WRITE {w * 1024 / 1e6:.1f})" if f is not None and w is not None else ""
    print(f"{k[:40]:40s} calls {calls:4d}  avg {us:8.2f} us   {hbm}")
    if k.startswith("k_mc_"): tot += us
print(f"marching cubes kernels together: {tot:.1f} us")
P
  cat $O/chain_ab_$tag.txt
done
