#!/bin/bash
# GPU box: kernel-trace durations of the tree build and the walk kernels of tools/kd_time.py
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/kdb; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/kd_time.py > $OUT/out.txt 2> $OUT/err.txt < /dev/null
tail -1 $OUT/out.txt
python3 - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)): d[r["Kernel_Name"].split("(")[0][:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]: print("%-50s %3d x %9.1f us" % (k, len(v), sum(v) / len(v)))
PY
