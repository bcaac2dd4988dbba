#!/bin/bash
# wall clock + kernel time of the correspondence search under HITL_KD_FAST = 0 / 1 / 2 on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for rep in 1 2; do
for f in ${KDF:-0 1 2}; do
  OUT=gpurun_out/kdfast_$f
  rm -rf $OUT; mkdir -p $OUT
  HITL_KD_FAST=$f REPS=4 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/dev/search_time.py > $OUT/out.txt 2> $OUT/err.txt
  echo "HITL_KD_FAST=$f: $(grep 'search ms' $OUT/out.txt)"
  python3 - <<PY
import csv, glob
for fn in glob.glob("$OUT/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(fn)):
        if "k_match_kd" in r["Name"]: print("   %-40s calls %s avg %.1f us" % (r["Name"].split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done
