#!/bin/bash
# kernel-trace timing of the linearisation kernel for a list of "VARIANT:WAVES_PER_CU" settings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-sweep}
shift
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "$@"; do
  v=${cfg%%:*}; w=${cfg##*:}
  HITL_STF_VARIANT=$v HITL_STF_WAVES_PER_CU=$w timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c_${v}_$w -- python3 tools/stf_time.py > $OUT/c_${v}_$w.out 2> $OUT/c_${v}_$w.err
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/c_${v}_$w/*/*kernel_trace.csv"):
    d = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_linearize_stf" in n: d.setdefault(n.split("(")[0][:60] + " grid " + r["Grid_Size_X"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, x in d.items():
        y = x[2:] if len(x) > 4 else x
        print("variant $v waves/CU $w: %-70s n=%d avg %.1f us  min %.1f  max %.1f" % (n, len(y), sum(y) / len(y), min(y), max(y)))
PY
done 2>&1 | tee $OUT/sweep.txt
