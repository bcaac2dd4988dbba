#!/bin/bash
# kernel-trace A/B of k_linearize_stf variants: prints avg/min duration of the stf kernel per variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-stft}
mkdir -p $OUT
shift
for v in "$@"; do
  HITL_STF_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v$v -- python3 tools/stf_time.py > $OUT/v$v.out 2> $OUT/v$v.err
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/v$v/*/*kernel_trace.csv"):
    d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "k_linearize_stf<" in r["Kernel_Name"]]
    d = d[2:]
    if d: print("variant $v: n=%d avg %.1f us  min %.1f  max %.1f" % (len(d), sum(d) / len(d), min(d), max(d)))
PY
done
