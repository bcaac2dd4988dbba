#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
HITL_TIMING=1 REPS=3 python3 tools/dev/search_time.py 2>&1 | tail -40
rm -rf gpurun_out/searchtr; mkdir -p gpurun_out/searchtr
REPS=3 timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/searchtr/t -- python3 tools/dev/search_time.py > gpurun_out/searchtr/out.txt 2>&1
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/searchtr/t/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void hitl::","").replace("hitl::","")[:40]))
for f in glob.glob("gpurun_out/searchtr/t/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", r.get("Bytes", ""))))
rows.sort()
# last search: from the last k_match_kd_tgt backwards to the previous k_compact end
idx=[i for i,r in enumerate(rows) if "k_match_kd_tgt" in r[2]]
last=idx[-1]
start=last
while start>0 and rows[last][0]-rows[start-1][1] < 3_000_000 and "k_compact" not in rows[start-1][2]: start-=1
t0=rows[start][0]
for s,e,k in rows[start:]:
    print("%9.1f us  +%8.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, k))
PY
