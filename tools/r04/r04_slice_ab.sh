# k_sc_rest (one wave per slice): candidates per slice, single-scan launches (tools/prof_render.py) + the batch bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; : > $O/slice_ab.txt
for V in "$@"; do
  if [ "$V" = default ]; then unset LIDARHIP_EXTRA_FLAGS; else export LIDARHIP_EXTRA_FLAGS="$V"; fi
  (cd $R && python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1)
  echo "== $V" >> $O/slice_ab.txt
  (cd $R && timeout 600 python -m pytest tests/test_trace_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1 >> $O/slice_ab.txt)
  rm -rf $O/slab; rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab -o s -- python $R/tools/prof_render.py --reps 40 > /dev/null 2>&1
  python - $O/slab >> $O/slice_ab.txt <<'P'
import csv, glob, sys
for path in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_sc_" in r["Name"] and "true" not in r["Name"]: print("  iso", r["Name"][:34], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
P
  (cd $R && python bench.py --no-cpu-baseline --no-other --no-e2e --no-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  bench', d['value'], d['verified'])" >> $O/slice_ab.txt)
done
unset LIDARHIP_EXTRA_FLAGS
(cd $R && python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1)
rm -rf $O/slab
cat $O/slice_ab.txt
