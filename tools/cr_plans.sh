#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/${1:-crp}
mkdir -p $OUT
shift
for v in "$@"; do
  tag=$(echo $v | tr ',' '_')
  HITL_CR_PLAN=$v timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/p$tag -- python3 tools/cr_time.py > $OUT/p$tag.out 2> $OUT/p$tag.err < /dev/null
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/p$tag/*/*kernel_trace.csv"):
    rows = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    # group chunk launches between consecutive k_build_system launches
    groups, cur = [], None
    for t, d, k in rows:
        if "k_build_system" in k: cur = []; groups.append(cur)
        elif "k_cr_chunk<" in k and cur is not None: cur.append(d)
    groups = [g for g in groups if g][2:]
    if groups:
        n = len(groups[0])
        avg = [sum(g[i] for g in groups) / len(groups) for i in range(n)]
        print("plan $v: stages " + " ".join("%.1f" % a for a in avg) + "  total %.1f us" % sum(avg))
PY
done
