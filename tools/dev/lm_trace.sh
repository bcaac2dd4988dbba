#!/bin/bash
# Runs on the GPU box: kernel trace of the C4 re-solve with the host-driven and the device-driven LM loop; per kernel averages
# and, for the last solve of each run, span (first kernel start .. last kernel end), busy time and their difference.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
for mode in host device; do
  OUT=gpurun_out/lmtr_$mode
  rm -rf $OUT; mkdir -p $OUT
  if [ "$mode" = "device" ]; then export HITL_DEVICE_LM=1; else unset HITL_DEVICE_LM; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/dev/solve_c4.py > $OUT/out.txt 2> $OUT/err.txt < /dev/null
  echo "== $mode: $(tail -1 $OUT/out.txt)"
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void hitl::", "").replace("hitl::", "")) for r in csv.DictReader(open(f)))
    # last solve = kernels after the last k_pose_trig-like set-up marker: take the last 71-iteration block by k_linearize_stf launches
    stf = [i for i, r in enumerate(rows) if "k_linearize_stf" in r[2]]
    per = 72
    first = stf[-per]
    blk = rows[first:]
    span = (blk[-1][1] - blk[0][0]) / 1e3
    busy = sum(e - s for s, e, k in blk) / 1e3
    d = collections.defaultdict(list)
    for s, e, k in blk: d[k[:40]].append((e - s) / 1e3)
    print("last solve: span %.1f us, busy %.1f us, idle %.1f us over %d stf launches -> %.1f / %.1f / %.1f per iteration" % (span, busy, span - busy, per, span / per, busy / per, (span - busy) / per))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])): print("   %-42s %4d x %8.2f us = %9.1f" % (k, len(v), sum(v) / len(v), sum(v)))
PY
done
