#!/bin/bash
# same-box A/B of the linearisation kernel between builds of the library: usage stf_lib_ab.sh lib1.so lib2.so ... ("-" = in-tree)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "$@"; do
  OUT=gpurun_out/stflib
  rm -rf $OUT; mkdir -p $OUT
  if [ "$lib" = "-" ]; then unset HITL_LIBHITLGPU; else export HITL_LIBHITLGPU=$(realpath $lib); fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/stf_time.py > $OUT/out.txt 2> $OUT/err.txt
  python3 - <<PY
import csv, glob
for fn in glob.glob("$OUT/t/*/*kernel_trace.csv"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(fn)) if "k_linearize_stf" in r["Kernel_Name"]]
    w = d[2:]
    print("%-44s n=%d avg %.1f us min %.1f" % ("$lib", len(w), sum(w) / len(w), min(w)))
PY
done
done
