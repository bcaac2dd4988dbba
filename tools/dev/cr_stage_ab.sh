#!/bin/bash
# Runs on the GPU box: per-stage averages of k_cr_chunk on the LM path of C4 for several settings of an environment switch.
# usage: tools/dev/cr_stage_ab.sh VAR v1 v2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
VAR=$1; shift
for v in "$@"; do
  OUT=gpurun_out/crab_$v
  mkdir -p $OUT
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  rm -rf $OUT/trace
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/dev/solve_c4.py > $OUT/out.txt 2> $OUT/err.txt < /dev/null
  tail -1 $OUT/out.txt
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/*/*kernel_trace.csv"):
    rows = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    groups, cur = [], None
    for t, d, k in rows:
        if "k_linearize_stf" in k or "k_assemble" in k: cur = None
        elif "k_cr_chunk<" in k or "k_cr_chunk_back" in k:
            if cur is None: cur = []; groups.append(cur)
            cur.append(d)
    groups = [g for g in groups if len(g) == 6][4:]
    if groups:
        avg = [sum(g[i] for g in groups) / len(groups) for i in range(6)]
        print("$VAR=$v: stages " + " ".join("%.1f" % a for a in avg) + "  total %.1f us over %d solves" % (sum(avg), len(groups)))
PY
done
