#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace of an arbitrary python script; prints per-kernel calls/total/avg/min/max.
# usage: tools/trace_cmd.sh <tag> <script.py> [args]
TAG=${1:-tc}; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/out.txt 2> $OUT/err.txt < /dev/null
tail -5 $OUT/out.txt
python3 - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
print("| kernel | calls | total us | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / max(tot, 1e-9)))
PY
