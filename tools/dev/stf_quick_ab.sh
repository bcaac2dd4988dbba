#!/bin/bash
# kernel-trace A/B of HITL_STF_VARIANT values on ONE box (tools/stf_time.py, C4): usage stf_quick_ab.sh tag v1 v2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-stfq}
shift
mkdir -p $OUT
export TMPDIR=/tmp
for v in "$@" "$@"; do
  HITL_STF_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/v$v -- python3 tools/stf_time.py > $OUT/v$v.out 2> $OUT/v$v.err
  python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/v$v/*/*kernel_trace.csv"))[-1:]:
    d = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_linearize_stf" in n: d.setdefault(n.split("(")[0][:60], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, v in d.items():
        w = v[2:] if len(v) > 4 else v
        print("variant $v: %-50s n=%d avg %.1f us  min %.1f  max %.1f" % (n, len(w), sum(w) / len(w), min(w), max(w)))
PY
  rm -rf $OUT/v$v
done 2>&1 | tee $OUT/ab.txt
