cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; : > $O/cap_iso.txt
export LIDARHIP_EXTRA_FLAGS="$1"
(cd $R && python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1)
for cap in ${CAPS:-1024 2048 8192}; do
  echo "== $1 single-scan LIDARHIP_SC_CAP=$cap" >> $O/cap_iso.txt
  rm -rf $O/slab; LIDARHIP_SC_CAP=$cap rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab -o s -- python $R/tools/prof_render.py --reps 40 > /dev/null 2>&1
  python - $O/slab >> $O/cap_iso.txt <<'P'
import csv, glob, sys
tot = 0
for path in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_sc_" in r["Name"] and "true" not in r["Name"]:
            print("  iso", r["Name"][:34], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us"); tot += float(r["AverageNs"]) / 1e3
print("  sum", round(tot, 2))
P
done
unset LIDARHIP_EXTRA_FLAGS
(cd $R && python -c "from lidar_transfer_amd import build; build.build_lib(force=True)" > /dev/null 2>&1)
rm -rf $O/slab; cat $O/cap_iso.txt
