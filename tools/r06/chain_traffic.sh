#!/bin/bash
# the fusion chain's kernels: durations and HBM traffic (2 x FETCH_SIZE + WRITE_SIZE per launch; separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ct; mkdir -p $R/gpurun_out/ct
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/ct/$grp -o p -- python $R/tools/prof_chain.py 4 > $R/gpurun_out/ct_$grp.log 2>&1 || echo "FAILED $grp"
done
cd $R
bash tools/mc_kernels.sh 2>&1 | tail -9
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ct/*/p_counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"][:30], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, d), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
for k in sorted(acc):
    if not any(s in k for s in ("k_mc_", "k_tsdf_")): continue
    f = acc[k].get("FETCH_SIZE", [0]); w = acc[k].get("WRITE_SIZE", [0])
    fm, wm = sum(f) / len(f), sum(w) / len(w)   # (KB units of the counters on gfx950: x 1024 / 1e6 -> MB; reads x 2)
    print("%-32s reads %8.1f MB  writes %8.1f MB  total %8.1f MB" % (k, 2 * fm * 1024 / 1e6, wm * 1024 / 1e6, (2 * fm + wm) * 1024 / 1e6))
PY
