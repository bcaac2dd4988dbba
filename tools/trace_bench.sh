#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of a short bench.py run; prints per-kernel avg/min/max.
# usage: tools/trace_bench.sh <tag> [bench args]
TAG=${1:-tb}; shift
ARGS=${*:-"--steps 3 --warmup 1 --no-cpu-baseline"}
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err < /dev/null
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
python3 -c "
import json
try:
    r = json.load(open('$OUT/bench.json')); print('bench: ms_per_step', r['ms_per_step'], 'iters', r['lm_iterations_per_resolve'], 'stf_ms', r['roofline']['avg_launch_ms'])
except Exception as e: print('no bench json', e)
" < /dev/null
