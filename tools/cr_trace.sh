#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/${1:-crt}
mkdir -p $OUT
shift
for v in "$@"; do
  HITL_CR_ABL=$v timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/v$v -- python3 tools/cr_time.py > $OUT/v$v.out 2> $OUT/v$v.err < /dev/null
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/v$v/*/*kernel_trace.csv"):
    rows = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(f)) if "k_cr_chunk<" in r["Kernel_Name"])
    d = [x[1] for x in rows]
    ns = 3
    per = [d[i::ns] for i in range(ns)]
    print("abl $v: " + "  ".join("stage%d avg %.1f min %.1f" % (i, sum(x[2:]) / max(len(x[2:]), 1), min(x)) for i, x in enumerate(per)))
PY
done
