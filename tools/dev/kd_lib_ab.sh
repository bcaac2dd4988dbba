#!/bin/bash
# same-box A/B of the correspondence search between builds of the library: usage kd_lib_ab.sh lib1.so lib2.so ... ("-" = the in-tree build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "$@"; do
  OUT=gpurun_out/kdlib
  rm -rf $OUT; mkdir -p $OUT
  if [ "$lib" = "-" ]; then unset HITL_LIBHITLGPU; else export HITL_LIBHITLGPU=$(realpath $lib); fi
  REPS=4 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/dev/search_time.py > $OUT/out.txt 2> $OUT/err.txt
  python3 - <<PY
import csv, glob
for fn in glob.glob("$OUT/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(fn)):
        if "k_match_kd_tgt" in r["Name"] or "k_compact" in r["Name"]: print("%-40s %-36s avg %.1f us   %s" % ("$lib", r["Name"].split("(")[0].replace("void hitl::","")[:36], float(r["AverageNs"]) / 1e3, open("$OUT/out.txt").read().strip().split("pairs")[0]))
PY
done
done
