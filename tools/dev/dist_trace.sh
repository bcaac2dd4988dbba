#!/bin/bash
# Runs on the GPU box: kernel trace of the C4 re-solve on the plain path and on the submap code path taken by ONE rank
# (HITL_FORCE_DIST=1; separator form and interface-column form): busy time per LM iteration and the kernels behind it.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
for mode in plain dist distcols; do
  OUT=gpurun_out/disttr_$mode
  rm -rf $OUT; mkdir -p $OUT
  unset HITL_FORCE_DIST HITL_DIST_COLUMNS
  if [ "$mode" != "plain" ]; then export HITL_FORCE_DIST=1; fi           # dist: the separator form (default since round 4)
  if [ "$mode" = "distcols" ]; then export HITL_DIST_COLUMNS=1; fi       # distcols: the interface-column form (rounds 2-3)
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/dev/solve_c4.py > $OUT/out.txt 2> $OUT/err.txt < /dev/null
  echo "== $mode: $(tail -1 $OUT/out.txt)"
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void hitl::", "").replace("hitl::", "")) for r in csv.DictReader(open(f)))
    stf = [i for i, r in enumerate(rows) if "k_linearize_stf" in r[2]]
    per = 72
    blk = rows[stf[-per]:]
    span = (blk[-1][1] - blk[0][0]) / 1e3
    busy = sum(e - s for s, e, k in blk) / 1e3
    d = collections.defaultdict(list)
    for s, e, k in blk: d[k[:40]].append((e - s) / 1e3)
    print("last solve: span %.1f us, busy %.1f us per iteration (%d stf launches)" % (span / per, busy / per, per))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])): print("   %-42s %5.2f per iteration x %8.2f us = %7.1f us per iteration" % (k, len(v) / per, sum(v) / len(v), sum(v) / per))
PY
done
