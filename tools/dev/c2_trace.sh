#!/bin/bash
# kernel trace of C2-size re-solves (1 000 poses x 360 observations, mode C): where a small map's LM iteration goes
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/c2trace; rm -rf $OUT; mkdir -p $OUT
POSES=1000 OBS=360 AB=tail REPS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/dev/lm_ab.py > $OUT/out.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 20: print("   %-50s calls %5s avg %8.2f us" % (r["Name"].split("(")[0].replace("void hitl::","").replace("hitl::","")[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
f = glob.glob("$OUT/t/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void hitl::","").replace("hitl::","") for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_cr_tail<9")]
i0 = idx[len(idx) // 2]
prev = None
for i in range(i0, min(i0 + 10, len(rows))):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print("%-30s dur %7.2f us  gap before %6.2f us" % (names[i][:30], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0)); prev = e
PY
